"""Box size of the two-level (A A^T)^-1 (SHM_TL_PREF_ROWS): set-up wait and solve phase over the large-m workloads."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import shm_import
shm = shm_import.load()
from signed_heat_3d_amd.host_abi import HostSolver
for f, hc, prec in [("rocker.obj", 4.0, 64), ("rocker.obj", 5.0, 64), ("rocker.obj", 5.0, 32), ("knot.obj", 4.0, 64), ("knot.obj", 5.0, 64), ("SprayBottle.pc", 5.0, 64), ("SprayBottle.pc", 6.0, 32), ("chair.obj", 5.0, 64)]:
    pre = HostSolver(os.path.join(R, "data", f)).preprocess(hCoef=hc)
    s = shm.GridSolver(precision=prec, verbose=True)
    s.set_problem(pre["pos"], pre["wnormal"], pre["area"], pre["lam"], pre["n"], pre["bbox_min"], pre["cell"])
    s.solve(scrub=not f.endswith(".pc"))
    st = s.solve(scrub=not f.endswith(".pc"))
    print("%s n=%d fp%d m=%d: total %.1f conv %.1f wait %.2f setup %.1f pcg %.1f (%d its, %.3f ms/it)" % (f, pre["n"], prec, st.m, st.ms_total, st.ms_conv, st.ms_wait_setup, st.ms_setup, st.ms_pcg, st.iters, st.ms_pcg / max(1, st.iters)), flush=True)
    s.close()

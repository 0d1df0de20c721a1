"""Probe: does the lazily created Schur stream (stream3) still run beside the main stream in the second and third solver of a process?
rocker.obj at n = 362 (cg_form 3: G inverted on stream2, S assembled on stream3, both beside Step 1)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import shm_import
shm = shm_import.load()
from signed_heat_3d_amd.host_abi import HostSolver
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pre = HostSolver(os.path.join(ROOT, "data/rocker.obj")).preprocess(hCoef=4.5)
pre2 = HostSolver(os.path.join(ROOT, "data/rocker.obj")).preprocess(hCoef=4.0)
for rep in range(3):
    for p in (pre, pre2):
        s = shm.GridSolver()
        s.set_problem(p["pos"], p["wnormal"], p["area"], p["lam"], p["n"], p["bbox_min"], p["cell"])
        s.solve()
        st = s.solve()
        print("solver %d n=%d: cg_form %d conv %.1f wait_setup %.2f pcg %.1f total %.1f" % (rep, p["n"], st.cg_form, st.ms_conv, st.ms_wait_setup, st.ms_pcg, st.ms_total), flush=True)
        s.close()

"""Tiered fp64 Step 1 (csrc/shm_conv_tiered.hip.h): time, executed pairs per tier, and the error of Y and phi against the all-fp64 kernel
(SHM_CONV_EXACT=1) for a range of far thresholds G (SHM_CONV_TIER_LOG).   python tools/tier_sweep.py [file hCoef [G ...]]"""
import os, sys, numpy as np
import os as _os; _os.environ.setdefault("SHM_DEBUG_KNOBS", "1")   # this tool drives the library's experiment knobs (read only behind this gate)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import shm_import
shm = shm_import.load()
from signed_heat_3d_amd.host_abi import HostSolver
path = sys.argv[1] if len(sys.argv) > 1 else "data/bunny_small.obj"
hc = float(sys.argv[2]) if len(sys.argv) > 2 else 4.0
Gs = [a for a in sys.argv[3:]] or ["exact", "25", "12", "10", "8", "7", "6", "5", "4"]
pre = HostSolver(path).preprocess(hCoef=hc)
print("%s n=%d S=%d lambda*cell=%.3f" % (path, pre["n"], pre["S"], pre["lam"] * pre["cell"]), flush=True)
ref = None
for g in Gs:
    os.environ.pop("SHM_CONV_EXACT", None); os.environ.pop("SHM_CONV_TIER_LOG", None)
    if g == "exact": os.environ["SHM_CONV_EXACT"] = "1"
    else: os.environ["SHM_CONV_TIER_LOG"] = g
    s = shm.GridSolver()
    s.set_problem(pre["pos"], pre["wnormal"], pre["area"], pre["lam"], pre["n"], pre["bbox_min"], pre["cell"])
    st = s.solve(scrub=(not path.endswith(".pc"))); st = s.solve(scrub=(not path.endswith(".pc")))
    phi, _ = s.get_phi()
    s.run_conv()
    Y = np.stack([s.get_field(k) for k in (0, 1, 2)], axis=1)
    if ref is None: ref = (Y, phi)
    ok = np.isfinite(Y).all(1) & np.isfinite(ref[0]).all(1)
    nom = float(pre["n"]) ** 3 * pre["S"]
    print("G=%-5s conv %.2f ms total %.2f ms  pairs fp64 %.3f fp32 %.3f of N*S  max|dY| %.2e  max|dphi| %.2e" % (
        g, st.ms_conv, st.ms_total, st.pairs_fp64 / nom, st.pairs_fp32 / nom, np.abs(Y[ok] - ref[0][ok]).max(), np.abs(phi - ref[1]).max()), flush=True)
    s.close()

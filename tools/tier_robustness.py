"""Every shipped data file at five grid sizes: the tiered fp64 Step 1 (shipped) against the all-fp64 kernel (SHM_CONV_EXACT=1) -- max|dY| over the nodes
where both are finite, whether the non-finite sets agree, max|dphi|, and the share of pairs each tier took.     python tools/tier_robustness.py"""
import os, sys, numpy as np
import os as _os; _os.environ.setdefault("SHM_DEBUG_KNOBS", "1")   # this tool drives the library's experiment knobs (read only behind this gate)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import shm_import
shm = shm_import.load()
from signed_heat_3d_amd.host_abi import HostSolver
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
files = ["bunny_small.obj", "polygon-bear.obj", "rocker.obj", "chair.obj", "knot.obj", "bunny.pc", "rocker.pc", "chair.pc", "knot.pc", "SprayBottle.pc"]
worst = 0.0
for f in files:
    for hc in (0.0, 1.0, 2.0, 3.0, 4.0):
        pre = HostSolver(os.path.join(ROOT, "data", f)).preprocess(hCoef=hc)
        out = {}
        for exact in (True, False):
            os.environ.pop("SHM_CONV_EXACT", None)
            if exact: os.environ["SHM_CONV_EXACT"] = "1"
            s = shm.GridSolver()
            s.set_problem(pre["pos"], pre["wnormal"], pre["area"], pre["lam"], pre["n"], pre["bbox_min"], pre["cell"])
            st = s.solve(scrub=not f.endswith(".pc"), allow_noconv=True)
            phi, _ = s.get_phi()
            s.run_conv()
            out[exact] = (np.stack([s.get_field(k) for k in (0, 1, 2)], 1), phi, st)
            s.close()
        (Ye, pe, ste), (Yt, pt, stt) = out[True], out[False]
        fe, ft = np.isfinite(Ye).all(1), np.isfinite(Yt).all(1)
        ok = fe & ft
        dY = float(np.abs(Yt[ok] - Ye[ok]).max()) if ok.any() else float("nan")
        nom = float(pre["n"]) ** 3 * pre["S"]
        fin = np.isfinite(pe) & np.isfinite(pt)
        dphi = float(np.abs(pt[fin] - pe[fin]).max()) if fin.any() else float("nan")
        worst = max(worst, dY)
        print("%-16s n=%3d S=%5d  pairs fp64 %.3f fp32 %.3f dropped %.3f  max|dY| %.2e  max|dphi| %.2e (max|phi| %.2f)  non-finite Y nodes %d / %d%s" % (
            f, pre["n"], pre["S"], stt.pairs_fp64 / nom, stt.pairs_fp32 / nom, max(0.0, 1.0 - (stt.pairs_fp64 + stt.pairs_fp32) / nom), dY, dphi,
            float(np.abs(pe[fin]).max()) if fin.any() else float("nan"), int((~ft).sum()), int((~fe).sum()), "" if (fe == ft).all() else "  (sets differ)"), flush=True)
print("worst max|dY| over all files and sizes: %.2e" % worst)

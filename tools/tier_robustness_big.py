"""The tiered fp64 Step 1 (shm_opts.step1_arith = AUTO) against the all-fp64 kernel (EXACT_F64) at the FULL sizes of BASELINE.json's configs and beyond:
max|dY| and max|dphi| over sampled z-planes (every `stride`-th plane plus the two bbox planes; the fields are 3-26 GB each at these sizes), the share of
pairs each tier took, and whether the non-finite sets agree.  Continues tools/tier_robustness.py (all files at 16^3 ... 256^3).
    python tools/tier_robustness_big.py [--skip-1024]"""
import os
import os as _os; _os.environ.setdefault("SHM_DEBUG_KNOBS", "1")   # this tool drives the library's experiment knobs (read only behind this gate)
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import shm_import

shm = shm_import.load()
from signed_heat_3d_amd.host_abi import HostSolver

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = [("bunny_small.obj", 4.0), ("knot.obj", 4.0), ("bunny_small.obj", 5.0), ("bunny.pc", 5.0), ("knot.obj", 5.0), ("rocker.obj", 5.0), ("chair.obj", 5.0), ("polygon-bear.obj", 5.0), ("knot.pc", 5.0),
         ("SprayBottle.pc", 5.0), ("SprayBottle.pc", 6.0), ("knot.obj", 6.0)]
if "--skip-1024" in sys.argv:
    CASES = [c for c in CASES if c[1] < 6.0]
if "--cases" in sys.argv:   # --cases file hCoef file hCoef ...
    a = sys.argv[sys.argv.index("--cases") + 1:]
    CASES = [(a[i], float(a[i + 1])) for i in range(0, len(a) - 1, 2)]
worst = 0.0
for f, hc in CASES:
    pre = HostSolver(os.path.join(ROOT, "data", f)).preprocess(hCoef=hc)
    n = pre["n"]
    stride = max(1, n // 32)
    ks = sorted(set(list(range(0, n, stride)) + [n - 1]))
    out = {}
    for arith in ("exact_f64", "auto"):
        s = shm.GridSolver()
        s.set_problem(pre["pos"], pre["wnormal"], pre["area"], pre["lam"], n, pre["bbox_min"], pre["cell"])
        t0 = time.time()
        st = s.solve(scrub=not f.endswith(".pc"), allow_noconv=True, step1=arith)
        dt = time.time() - t0
        phi = np.stack([s.get_field_planes(s.FIELD_PHI, k, k + 1) for k in ks])
        s.run_conv(step1=arith)
        Y = np.stack([np.stack([s.get_field_planes(c, k, k + 1) for c in (0, 1, 2)], axis=1) for k in ks])
        out[arith] = (Y, phi, st, dt)
        s.close()
    (Ye, pe, ste, dte), (Yt, pt, stt, dtt) = out["exact_f64"], out["auto"]
    fe, ft = np.isfinite(Ye).all(-1), np.isfinite(Yt).all(-1)
    ok = fe & ft
    dY = float(np.abs(Yt[ok] - Ye[ok]).max()) if ok.any() else float("nan")
    nom = float(n) ** 3 * pre["S"]
    fin = np.isfinite(pe) & np.isfinite(pt)
    dphi = float(np.abs(pt[fin] - pe[fin]).max()) if fin.any() else float("nan")
    worst = max(worst, dY)
    if ok.any():   # the planes that carry the largest differences (round 6: candidates for the oracle planes of tests/test_gpu_parity.py::STEP1_WORST_PLANES)
        per_plane = [float(np.abs(np.where(ok[a][:, None], Yt[a] - Ye[a], 0.0)).max()) for a in range(len(ks))]
        top = sorted(range(len(ks)), key=lambda a: -per_plane[a])[:5]
        print("    worst planes: " + ", ".join("k=%d %.2e" % (ks[a], per_plane[a]) for a in top), flush=True)
    print("%-16s n=%4d S=%5d  pairs fp64 %.3f fp32 %.3f redone %.4f dropped %.3f  Step 1 %.0f ms (all-fp64 %.0f)  max|dY| %.2e  max|dphi| %.2e (max|phi| %.2f)  "
          "non-finite Y nodes %d / %d%s  [%d planes]" % (
              f, n, pre["S"], stt.pairs_fp64 / nom, stt.pairs_fp32 / nom, stt.pairs_redone / nom, max(0.0, 1.0 - (stt.pairs_fp64 + stt.pairs_fp32 - stt.pairs_redone) / nom), stt.ms_conv, ste.ms_conv, dY, dphi,
              float(np.abs(pe[fin]).max()) if fin.any() else float("nan"), int((~ft).sum()), int((~fe).sum()), "" if (fe == ft).all() else "  (sets differ)", len(ks)),
          flush=True)
print("worst max|dY| over the cases: %.2e   (budget of the tests: 1e-8)" % worst)

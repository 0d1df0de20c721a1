"""Where does the tiered fp64 Step 1 differ most from the all-fp64 kernel, and why?  For the nodes of largest |Y_tiered - Y_exact| on sampled z-planes the host
evaluates the node's sum term by term in fp64: |X| against the L1 norm of its terms (the amplification of any per-term error by Y = X / |X| where the sheets of
the source geometry cancel), the share of the terms below e^-8 of the node's largest term (what a per-pair rule would send to the packed-fp32 tier) and
lambda * r of the nearest source.     python tools/tier_worst_nodes.py [file hCoef]..."""
import os
import os as _os; _os.environ.setdefault("SHM_DEBUG_KNOBS", "1")   # this tool drives the library's experiment knobs (read only behind this gate)
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import shm_import

shm = shm_import.load()
from signed_heat_3d_amd.host_abi import HostSolver

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = sys.argv[1:]
CASES = [(args[i], float(args[i + 1])) for i in range(0, len(args), 2)] or [("knot.obj", 4.0), ("rocker.obj", 5.0), ("chair.obj", 5.0), ("bunny_small.obj", 5.0)]
for f, hc in CASES:
    pre = HostSolver(os.path.join(ROOT, "data", f)).preprocess(hCoef=hc)
    n, lam, cell, b0 = pre["n"], pre["lam"], pre["cell"], np.asarray(pre["bbox_min"])
    ks = sorted(set(list(range(0, n, max(1, n // 32))) + [n - 1]))
    Y = {}
    s = shm.GridSolver()
    s.set_problem(pre["pos"], pre["wnormal"], pre["area"], lam, n, pre["bbox_min"], cell)
    for arith in ("exact_f64", "auto"):
        s.run_conv(step1=arith)
        Y[arith] = np.stack([np.stack([s.get_field_planes(c, k, k + 1) for c in (0, 1, 2)], axis=1) for k in ks])
    s.close()
    d = np.abs(Y["auto"] - Y["exact_f64"]).max(axis=-1)
    d[~np.isfinite(d)] = 0.0
    order = np.argsort(d.ravel())[::-1]
    print("%s n=%d lambda*cell=%.3f: max|dY| %.2e; nodes above 1e-9: %d of %d sampled; above 3e-9: %d; above 1e-8: %d" % (
        f, n, lam * cell, d.max(), int((d > 1e-9).sum()), d.size, int((d > 3e-9).sum()), int((d > 1e-8).sum())))
    pos, wn = np.asarray(pre["pos"]), np.asarray(pre["wnormal"])
    wmag = np.linalg.norm(wn, axis=1)

    def stats(flat):
        pk, rem = divmod(int(flat), n * n)
        j, i = divmod(rem, n)
        x = b0 + cell * np.array([i, j, ks[pk]])
        r = np.linalg.norm(pos - x, axis=1)
        g = np.exp(-lam * (r - r.min())) / r
        terms = wmag * g
        X = (wn * g[:, None]).sum(axis=0)
        far = terms < np.exp(-8.0) * terms.max()
        return (i, j, ks[pk]), lam * r.min(), np.linalg.norm(X) / terms.sum(), terms[far].sum() / terms.sum(), terms[far].sum() / np.linalg.norm(X), int(far.sum())

    for flat in order[:6]:
        ijk, lr, xr, fs, fx, nf = stats(flat)
        print("   worst node %s  |dY| %.2e  lambda*r_near %.1f  |X|/L1 %.3e  L1_far/L1 %.3e  L1_far/|X| %.3e  (%d far sources by the per-pair rule)" % (ijk, d.ravel()[flat], lr, xr, fs, fx, nf))
    rng = np.random.default_rng(0)
    typ = rng.choice(d.size, 2000, replace=False)
    st = np.array([stats(t)[2:5] for t in typ])
    print("   2000 random sampled nodes: median |X|/L1 %.3f, 1%% quantile %.2e; L1_far/|X| median %.2e, 90%% %.2e, 99%% %.2e, max %.2e; their |dY| median %.1e max %.1e" % (
        np.median(st[:, 0]), np.quantile(st[:, 0], 0.01), np.median(st[:, 2]), np.quantile(st[:, 2], 0.9), np.quantile(st[:, 2], 0.99), st[:, 2].max(),
        np.median(d.ravel()[typ]), d.ravel()[typ].max()))

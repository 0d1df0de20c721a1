"""Where does the tiered Step 1 differ most from the all-fp64 kernel, and how far (in kernel widths) is that node from the sources?  For the planes given: the worst node's
indices, lambda * (distance to the nearest source), |X|-cancellation is not available here -- the point is to tell a tier error (near the surface / on the medial axis) from
the reference's own subnormal-|X|^2 zone (lambda r >~ 335).   python tools/r06_worst_node_diag.py file hCoef k [k ...]"""
import os
import os as _os; _os.environ.setdefault("SHM_DEBUG_KNOBS", "1")
import sys
import numpy as np
from scipy.spatial import cKDTree
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import shm_import
shm = shm_import.load()
from signed_heat_3d_amd.host_abi import HostSolver
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
f, hc = sys.argv[1], float(sys.argv[2])
ks = [int(a) for a in sys.argv[3:]]
pre = HostSolver(os.path.join(ROOT, "data", f)).preprocess(hCoef=hc)
n, lam, cell, b0 = pre["n"], pre["lam"], pre["cell"], np.asarray(pre["bbox_min"])
tree = cKDTree(np.asarray(pre["pos"]).reshape(-1, 3))
Y = {}
for arith in ("exact_f64", "auto"):
    s = shm.GridSolver()
    s.set_problem(pre["pos"], pre["wnormal"], pre["area"], pre["lam"], n, pre["bbox_min"], pre["cell"])
    s.run_conv(step1=arith)
    Y[arith] = {k: np.stack([s.get_field_planes(c, k, k + 1) for c in (0, 1, 2)], axis=1) for k in ks}
    s.close()
print("%s n=%d S=%d lambda*cell=%.4f" % (f, n, pre["S"], lam * cell))
jj, ii = np.meshgrid(np.arange(n), np.arange(n), indexing="ij")
for k in ks:
    d = np.abs(Y["auto"][k] - Y["exact_f64"][k]).max(axis=1)
    d[~np.isfinite(d)] = 0.0
    P = np.stack([ii.ravel() * cell + b0[0], jj.ravel() * cell + b0[1], np.full(n * n, k * cell + b0[2])], axis=1)
    r, _ = tree.query(P)
    lr = lam * r
    order = np.argsort(-d)[:5]
    print("plane %d: " % k + "; ".join("dY %.2e at (i %d, j %d) lambda r %.1f" % (d[o], o % n, o // n, lr[o]) for o in order))
    for lim in (335.0, 300.0, 200.0, 100.0):
        m = lr < lim
        print("    max|dY| over nodes with lambda r < %.0f: %.2e  (%d of %d nodes)" % (lim, d[m].max() if m.any() else float("nan"), int(m.sum()), n * n))

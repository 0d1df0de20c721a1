"""The 'shell' adversarial input of tests/test_gpu_parity.py (closed surface seen from inside) at other source counts and kernel widths: max|dY| of the tiered Step 1
against the all-fp64 arithmetic.   python tools/r06_shell_probe.py [n]"""
import os, sys
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import shm_import
shm = shm_import.load()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
cell = 2.0 / (n - 1)
CASES = ((40000, 0.46), (160000, 0.46), (160000, 0.25), (40000, 0.15), (400000, 0.46), (160000, 1.0))
if len(sys.argv) > 2:   # S lc S lc ...
    CASES = tuple((int(sys.argv[i]), float(sys.argv[i + 1])) for i in range(2, len(sys.argv) - 1, 2))
for S, lc in CASES:
    rng = np.random.default_rng(7)
    v = rng.normal(size=(S, 3)); v /= np.linalg.norm(v, axis=1, keepdims=True)
    ax3 = np.array([0.62, 0.5, 0.41])
    pos = v * ax3 * (1.0 + 0.01 * rng.normal(size=(S, 1))) + np.array([0.03, -0.02, 0.04])
    nrm = v / ax3; nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    area = np.full(S, 4 * np.pi * 0.25 / S) * (0.7 + 0.6 * rng.random(S))
    lam = lc / cell
    s = shm.GridSolver()
    s.set_problem(pos, nrm * area[:, None], area, lam, n, np.array([-1.0, -1.0, -1.0]), cell)
    s.run_conv(); Yt = np.stack([s.get_field(k) for k in (0, 1, 2)], axis=1)
    st = s.last_stats() if hasattr(s, "last_stats") else None
    s.run_conv(step1="exact_f64"); Ye = np.stack([s.get_field(k) for k in (0, 1, 2)], axis=1)
    s.close()
    ok = np.isfinite(Ye).all(1) & np.isfinite(Yt).all(1)
    d = np.abs(Yt - Ye).max(axis=1); d[~ok] = 0
    i = int(np.argmax(d))
    from scipy.spatial import cKDTree
    tree = cKDTree(pos)
    top = np.argsort(-d)[:2000]
    P = np.stack([top % n, (top // n) % n, top // (n * n)], axis=1) * cell - 1.0
    lr = lam * tree.query(P)[0]
    in_zone = lr < 335.0     # (beyond it the reference's own normalisation has lost its bits: DESIGN.md section 2a)
    print("shell S=%6d lambda*cell=%.2f n=%d: max|dY| %.2e at node (%d,%d,%d) lambda r %.1f; over the 2000 worst nodes with lambda r < 335: %.2e  finite %.3f" % (
        S, lc, n, d.max(), i % n, (i // n) % n, i // (n * n), lr[0], d[top][in_zone].max() if in_zone.any() else 0.0, ok.mean()), flush=True)

import os, sys
import numpy as np
sys.path.insert(0, '/root/repo')
import shm_import
shm = shm_import.load()
n = 256; cell = 2.0 / (n - 1)
for S, lc in ((40000, 0.46), (160000, 0.46), (400000, 0.46)):
    rng = np.random.default_rng(7)
    v = rng.normal(size=(S, 3)); v /= np.linalg.norm(v, axis=1, keepdims=True)
    ax3 = np.array([0.62, 0.5, 0.41])
    pos = v * ax3 * (1.0 + 0.01 * rng.normal(size=(S, 1))) + np.array([0.03, -0.02, 0.04])
    nrm = v / ax3; nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    area = np.full(S, 4 * np.pi * 0.25 / S) * (0.7 + 0.6 * rng.random(S))
    lam = lc / cell
    out = {}
    for prec in (64, 32):
        s = shm.GridSolver(precision=prec)
        s.set_problem(pos, nrm * area[:, None], area, lam, n, np.array([-1.0, -1.0, -1.0]), cell)
        st = s.solve(scrub=False, allow_noconv=True)
        phi, _ = s.get_phi()
        s.run_conv(); Y = np.stack([s.get_field(k) for k in (0, 1, 2)], axis=1).astype(np.float64)
        out[prec] = (Y, phi, st)
        s.close()
    dY = np.abs(out[32][0] - out[64][0]).max(axis=1); ok = np.isfinite(dY)
    print("shell S=%d: max|Y32 - Y64| %.2e (99.9th pct %.2e), L_inf(phi32 - phi64) %.2e of max|phi| %.2f; iters %d/%d" % (S, dY[ok].max(), np.percentile(dY[ok], 99.9), np.abs(out[32][1] - out[64][1]).max(), np.abs(out[64][1]).max(), out[64][2].iters, out[32][2].iters), flush=True)

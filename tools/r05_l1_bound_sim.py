"""Host model (no GPU): share of 8x8x4 blocks that fail the a-posteriori test of the packed-fp32 tier (eps_far L1_far <= budget |X| at every node) under the box rule
and under the two-source differential rule, and -- `ub` -- the share an A-PRIORI upper bound of L1_far (sum over the far sources of |w|_1 e^{-lambda d_box} / d_box)
would send to the second pass.   python tools/r05_l1_bound_sim.py <file> <hCoef> <blocks> [G]"""
import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import shm_import
shm = shm_import.load()
from signed_heat_3d_amd.host_abi import HostSolver
path = sys.argv[1]; hc = float(sys.argv[2]); nb = int(sys.argv[3])
G = float(sys.argv[4]) if len(sys.argv) > 4 else 8.0
pre = HostSolver(path).preprocess(hCoef=hc)
pos = np.asarray(pre["pos"]).reshape(-1, 3); wn = np.asarray(pre["wnormal"]).reshape(-1, 3)
n = pre["n"]; cell = pre["cell"]; lam = pre["lam"]; b0 = np.asarray(pre["bbox_min"]); S = len(pos)
w = np.linalg.norm(wn, axis=1); lw = np.log(w); w1 = np.abs(wn).sum(1)
skip = np.log(S / 2e-9)
rng = np.random.default_rng(1)
shape = (8, 8, 4); h = 0.5 * (np.array(shape) - 1) * cell; rt = np.linalg.norm(h)
ii, jj, kk = np.meshgrid(np.arange(8), np.arange(8), np.arange(4), indexing="ij")
off = np.stack([ii, jj, kk], -1).reshape(-1, 3)
res = {r: np.zeros(7) for r in ("box", "diff")}
for _ in range(nb):
    i0 = rng.integers(0, n // 8) * 8; j0 = rng.integers(0, n // 8) * 8; k0 = rng.integers(0, n // 4) * 4
    c = b0 + (np.array([i0, j0, k0]) + 0.5 * (np.array(shape) - 1)) * cell
    d = pos - c; dc = np.linalg.norm(d, axis=1); s_star = np.argmin(dc)
    r_hi = np.linalg.norm(np.abs(d[s_star]) + h)
    dist = np.linalg.norm(np.maximum(np.abs(d) - h, 0.0), axis=1)
    rel = lw - lw[s_star]; lhs = lam * (dist - r_hi)
    us = -d / np.maximum(dc, 1e-300)[:, None]
    du = np.linalg.norm(us - us[s_star], axis=1)
    lip = du + rt * (1.0 / np.maximum(dist, 1e-300) + 1.0 / max(dist[s_star], 1e-300))
    lhs2 = np.maximum(lhs, lam * (dc - dc[s_star] - rt * lip))
    x = b0 + (off + [i0, j0, k0]) * cell
    r = np.linalg.norm(x[:, None, :] - pos[None, :, :], axis=2)
    dmin = r.min()
    g = np.exp(-lam * (r - dmin)) / r
    for rule, L in (("box", lhs), ("diff", lhs2)):
        drop = lhs > skip + rel
        far = (L > G + rel) & ~drop
        X = g @ wn
        L1 = g[:, far] @ w1[far]
        nx = np.linalg.norm(X, axis=1)
        fail = (3e-6 * L1 > 1e-8 * nx).any()
        ub = (w1[far] * np.exp(-lam * (dist[far] - dmin)) / np.maximum(dist[far],1e-300)).sum()
        failub = 3e-6 * ub > 1e-8 * nx.min()
        nf = far.mean(); nn = (~far & ~drop).mean()
        res[rule] += [nn, nf, float(fail), float(fail) * nf, (L1 / nx).max(), float(failub), float(failub)*nf]
for rule in res:
    a = res[rule] / nb
    print("%s n=%d G=%.1f rule %-5s near %.3f far %.3f  blocks failing %.4f  redo share of far pairs %.4f  mean max(L1/|X|) %.2e   cost %.3f | ub: blocks failing %.4f redo share %.4f" % (path, n, G, rule, a[0], a[1], a[2], a[3] / a[1], a[4], a[0] + 0.43 * a[1] + a[3], a[5], a[6]/a[1]))

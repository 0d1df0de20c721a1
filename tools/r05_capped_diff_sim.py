"""Host model (no GPU) of a capped differential far rule for the Step-1 tiers: a source that the shipped box rule calls near but the two-source differential rule calls far
("marginal") goes to the packed-fp32 tier only while the running sum of the marginal sources' bounds e^{rel - lhs2} (their terms relative to the block's dominant term), taken
cluster by cluster (64 sources in input order), stays below CAP.   python tools/r05_capped_diff_sim.py <file> <hCoef> <blocks> <cap> [<cap> ...]"""
import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import shm_import
shm = shm_import.load()
from signed_heat_3d_amd.host_abi import HostSolver
path = sys.argv[1]; hc = float(sys.argv[2]); nb = int(sys.argv[3]); caps = [float(a) for a in sys.argv[4:]] or [1e-3]
G = 8.0
pre = HostSolver(path).preprocess(hCoef=hc)
pos = np.asarray(pre["pos"]).reshape(-1, 3); wn = np.asarray(pre["wnormal"]).reshape(-1, 3)
n = pre["n"]; cell = pre["cell"]; lam = pre["lam"]; b0 = np.asarray(pre["bbox_min"]); S = len(pos)
w = np.linalg.norm(wn, axis=1); lw = np.log(w); w1 = np.abs(wn).sum(1)
skip = np.log(S / 2e-9)
rng = np.random.default_rng(1)
shape = (8, 8, 4); h = 0.5 * (np.array(shape) - 1) * cell; rt = np.linalg.norm(h)
ii, jj, kk = np.meshgrid(np.arange(8), np.arange(8), np.arange(4), indexing="ij")
off = np.stack([ii, jj, kk], -1).reshape(-1, 3)
rules = ["box", "diff"] + ["cap %.0e" % c for c in caps]
res = {r: np.zeros(4) for r in rules}
cl = np.arange(S) // 64
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
    g = np.exp(-lam * (r - r.min())) / r
    X = g @ wn; nx = np.linalg.norm(X, axis=1)
    drop = lhs > skip + rel
    far_box = (lhs > G + rel) & ~drop
    far_diff = (lhs2 > G + rel) & ~drop
    marg = far_diff & ~far_box
    bound = np.where(marg, np.exp(np.minimum(rel - lhs2, 50.0)), 0.0)
    per_cluster = np.bincount(cl, weights=bound, minlength=cl.max() + 1)
    run = np.cumsum(per_cluster)
    for rule in rules:
        if rule == "box": far = far_box
        elif rule == "diff": far = far_diff
        else:
            cap = float(rule.split()[1])
            ok_cluster = run <= cap          # the cluster's marginal sources are admitted while the running sum (this cluster included) stays below the cap
            far = far_box | (marg & ok_cluster[cl])
        L1 = g[:, far] @ w1[far]
        fail = bool((3e-6 * L1 > 1e-8 * nx).any())
        nf = far.mean(); nn = (~far & ~drop).mean()
        res[rule] += [nn, nf, float(fail), float(fail) * nf]
print(path, "n", n, "S", S)
for rule in rules:
    a = res[rule] / nb
    print("  %-10s near %.3f far %.3f  blocks failing %.4f  cost %.3f (near + 0.43 far + redo)" % (rule, a[0], a[1], a[2], a[0] + 0.43 * a[1] + a[3]))

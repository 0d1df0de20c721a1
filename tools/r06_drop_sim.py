"""Host model of the Step-1 drop rule (csrc/shm_conv_tiered.hip.h): share of (node, source) pairs kept under
  old:   drop s  <=>  b_s <= eps / S                       (S times the worst case, rounds 3-5)
  acc:   drop s  <=>  b_s <= eps / K  and  running sum of dropped bounds + b_s <= eps     (round 6: accumulated bound, K a per-problem estimate)
  opt:   drop the largest set whose bounds sum to <= eps (sorted prefix): what any rigorous rule on these bounds can reach
with b_s = (|w_s| / |w_*|) e^{-lambda (d_box(s) - r_hi)} [r_hi / d_box(s)], s* the source nearest to the block's centre.
Random sample of 8 x 8 x 4 blocks; no GPU.   python tools/r06_drop_sim.py path hCoef eps [nblocks]"""
import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import shm_import
shm = shm_import.load()
from signed_heat_3d_amd.host_abi import HostSolver

path = sys.argv[1] if len(sys.argv) > 1 else "data/rocker.obj"
hc = float(sys.argv[2]) if len(sys.argv) > 2 else 5.0
eps = float(sys.argv[3]) if len(sys.argv) > 3 else 6e-8
nb = int(sys.argv[4]) if len(sys.argv) > 4 else 400
pre = HostSolver(path).preprocess(hCoef=hc)
pos = np.asarray(pre["pos"]).reshape(-1, 3); wn = np.asarray(pre["wnormal"]).reshape(-1, 3)
n = pre["n"]; cell = pre["cell"]; lam = pre["lam"]; b0 = np.asarray(pre["bbox_min"]); S = len(pos)
w = np.linalg.norm(wn, axis=1)
abar = w.mean()
Kest = 2.0 * 157.0 / (abar * lam * lam)
print(path, "n", n, "S", S, "lam*cell %.3f" % (lam * cell), "eps", eps, "K estimate %.0f" % Kest)
rng = np.random.default_rng(1)
shape = np.array([8, 8, 4]); h = 0.5 * (shape - 1) * cell
res = {}
for _ in range(nb):
    ijk = np.array([rng.integers(0, n // shape[a]) * shape[a] for a in range(3)])
    c = b0 + (ijk + 0.5 * (shape - 1)) * cell
    d = pos - c
    dc = np.linalg.norm(d, axis=1)
    s_star = np.argmin(dc)
    r_hi = np.linalg.norm(np.abs(d[s_star]) + h)
    dist = np.linalg.norm(np.maximum(np.abs(d) - h, 0.0), axis=1)
    rt = np.linalg.norm(h)
    us = -d / np.maximum(dc, 1e-300)[:, None]
    du = np.linalg.norm(us - us[s_star], axis=1)
    lip = du + rt * (1.0 / np.maximum(dist, 1e-300) + 1.0 / max(dist[s_star], 1e-300))
    gap2 = dc - dc[s_star] - rt * lip          # differential bound of r_s(x) - r_s*(x) over the block
    geo = np.minimum(1.0, r_hi / np.maximum(dist, 1e-300))
    for tag, gap in (("", dist - r_hi), ("+diff", np.maximum(dist - r_hi, gap2))):
        b = (w / w[s_star]) * np.exp(-lam * gap) * geo
        b = np.where(gap > 0, b, np.inf)
        res.setdefault("old" + tag, []).append((b <= eps / S).mean())
        order = np.argsort(b)
        cs = np.cumsum(b[order])
        res.setdefault("opt" + tag, []).append((cs <= eps).mean())
        for K in (256, 512, 1024, Kest):
            cand = b <= eps / K
            # greedy in storage order, cluster by cluster (64 sources): a cluster's candidates go together or not at all
            R = 0.0; dropped = 0
            for c0 in range(0, S, 64):
                bc = b[c0:c0 + 64][cand[c0:c0 + 64]]
                if len(bc) and R + bc.sum() <= eps:
                    R += bc.sum(); dropped += len(bc)
            res.setdefault("acc K=%d%s" % (K, tag), []).append(dropped / S)
for k, v in res.items():
    print("%-18s dropped %.3f  kept %.3f" % (k, np.mean(v), 1 - np.mean(v)))

"""Host model of the Step-1 tier classification (csrc/shm_conv_tiered.hip.h): the share of (node, source) pairs that lands in the fp64 / packed-fp32 / dropped
sets under different block shapes and bounds.  Random sample of blocks; no GPU.   python tools/r05_classify_sim.py [path hCoef [nblocks]]"""
import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import shm_import
shm = shm_import.load()
from signed_heat_3d_amd.host_abi import HostSolver

path = sys.argv[1] if len(sys.argv) > 1 else "data/bunny_small.obj"
hc = float(sys.argv[2]) if len(sys.argv) > 2 else 4.0
nb = int(sys.argv[3]) if len(sys.argv) > 3 else 1500
G = 8.0
pre = HostSolver(path).preprocess(hCoef=hc)
pos = np.asarray(pre["pos"]).reshape(-1, 3); wn = np.asarray(pre["wnormal"]).reshape(-1, 3)
n = pre["n"]; cell = pre["cell"]; lam = pre["lam"]; b0 = np.asarray(pre["bbox_min"]); S = len(pos)
w = np.linalg.norm(wn, axis=1); lw = np.log(w)
skip = np.log(S / 2e-9)
print(path, "n", n, "S", S, "lam*cell %.3f" % (lam * cell), "G", G, "skip %.1f" % skip)
rng = np.random.default_rng(1)


def run(shape, rule):
    bx, by, bz = shape
    h = 0.5 * (np.array(shape) - 1) * cell
    tot = np.zeros(3)
    for _ in range(nb):
        i0 = rng.integers(0, n // bx) * bx; j0 = rng.integers(0, n // by) * by; k0 = rng.integers(0, n // bz) * bz
        c = b0 + (np.array([i0, j0, k0]) + 0.5 * (np.array(shape) - 1)) * cell
        d = pos - c
        dc = np.linalg.norm(d, axis=1)
        s_star = np.argmin(dc)
        r_hi = np.linalg.norm(np.abs(d[s_star]) + h)
        box = np.maximum(np.abs(d) - h, 0.0)
        dist = np.linalg.norm(box, axis=1)
        rel = lw - lw[s_star]
        lhs = lam * (dist - r_hi)
        if rule == "pair":
            # every pair on its own: nodes of the block
            ii, jj, kk = np.meshgrid(np.arange(bx), np.arange(by), np.arange(bz), indexing="ij")
            x = b0 + (np.stack([ii + i0, jj + j0, kk + k0], -1).reshape(-1, 3)) * cell
            r = np.linalg.norm(x[:, None, :] - pos[None, :, :], axis=2)
            term = lw[None, :] - lam * r - np.log(r)
            dom = term.max(axis=1, keepdims=True)
            g = dom - term
            tot += [(g <= G).mean(), ((g > G) & (g <= skip)).mean(), (g > skip).mean()]
            continue
        if rule == "diff":
            # r_s(x) - r_near(x) >= r_s(x) - r_s*(x) =: f(x);  f(x) >= f(c) - rt (|u_s(c) - u_s*(c)| + rt (1/dist_s + 1/dist_s*))
            rt = np.linalg.norm(h)
            us = -d / np.maximum(dc, 1e-300)[:, None]
            du = np.linalg.norm(us - us[s_star], axis=1)
            lip = du + rt * (1.0 / np.maximum(dist, 1e-300) + 1.0 / max(dist[s_star], 1e-300))
            lhs2 = lam * (dc - dc[s_star] - rt * lip)
            lhs_far = np.maximum(lhs, lhs2)
        elif rule == "corners":
            # (not sound) f at the 8 corners + centre, minimum
            cs = np.array([[sx, sy, sz] for sx in (-1, 1) for sy in (-1, 1) for sz in (-1, 1)]) * h
            pts = np.vstack([cs, np.zeros((1, 3))])
            rr = np.linalg.norm((c + pts)[:, None, :] - pos[None, :, :], axis=2)
            lhs_far = np.maximum(lhs, lam * (rr - rr.min(axis=1, keepdims=True)).min(axis=0))
        else:
            lhs_far = lhs
        drop = lhs > skip + rel
        far = (lhs_far > G + rel) & ~drop
        near = ~far & ~drop
        tot += [near.mean(), far.mean(), drop.mean()]
    return tot / nb


for shape in ((8, 8, 4), (4, 4, 4), (4, 4, 8), (4, 8, 4), (8, 8, 2)):
    for rule in ("box", "diff", "corners"):
        f = run(shape, rule)
        print("block %s rule %-8s near %.3f far %.3f drop %.3f   cost %.3f" % (shape, rule, f[0], f[1], f[2], f[0] + 0.43 * f[1]))
nb = max(50, nb // 10)
f = run((8, 8, 4), "pair")
print("per pair: near %.3f far %.3f drop %.3f   cost %.3f" % (f[0], f[1], f[2], f[0] + 0.43 * f[1]))

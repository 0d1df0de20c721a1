"""Where the tiered Step 1 leaves its budget on an adversarial input (tests/test_gpu_parity.py::_adversarial_sources): tiered and all-fp64 (GPU) against each other and
against the C oracle on the plane of the worst node, with the node's own statistics.   python tools/r06_adv_diag.py kind n [seed]      (one process per library setting)"""
import os, sys, ctypes
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import shm_import
shm = shm_import.load()
import importlib.util
spec = importlib.util.spec_from_file_location("tgp", os.path.join(R, "tests", "test_gpu_parity.py")); tgp = importlib.util.module_from_spec(spec); spec.loader.exec_module(tgp)
kind, n = sys.argv[1], int(sys.argv[2]); seed = int(sys.argv[3]) if len(sys.argv) > 3 else 1
d = tgp._adversarial_sources(kind, n, seed)
s = tgp.make_solver(shm, d)
s.run_conv(); Yt = np.stack([s.get_field(k) for k in (0, 1, 2)], axis=1)
st = s.solve(scrub=True, allow_noconv=True, max_iters=1)
s.run_conv(step1="exact_f64"); Ye = np.stack([s.get_field(k) for k in (0, 1, 2)], axis=1)
s.close()
ok = np.isfinite(Ye).all(axis=1)
err = np.where(ok, np.abs(Yt - Ye).max(axis=1), 0.0)
nom = float(n) ** 3 * len(d["area"])
print("%s n=%d seed %d knobs %s: max|Yt - Ye| = %.3e; pairs fp64 %.3f fp32 %.3f redone %.4f of nominal" % (kind, n, seed, {k: v for k, v in os.environ.items() if k.startswith("SHM_") and k != "SHM_DEBUG_KNOBS"},
      err.max(), st.pairs_fp64 / nom, st.pairs_fp32 / nom, st.pairs_redone / nom))
worst = np.argsort(-err)[:5]
pos, wn, lam, cell, b0 = d["pos"], d["wnormal"], d["lam"], d["cell"], d["bbox_min"]
w = np.linalg.norm(wn, axis=1)
ctr = pos.mean(axis=0)
for idx in worst:
    k, j, i = idx // (n * n), (idx // n) % n, idx % n
    x = b0 + np.array([i, j, k]) * cell
    r = np.linalg.norm(pos - x, axis=1)
    terms = wn * (np.exp(-lam * (r - r.min())) / r)[:, None]
    X = terms.sum(axis=0); L1 = np.abs(terms).sum()
    tmag = np.linalg.norm(terms, axis=1)
    Yh = X / np.linalg.norm(X)
    print("  node (%d,%d,%d) |x - cloud centre| %.3f: err %.2e  lambda r_min %.1f  |X|/L1 %.2e  |X|/dominant %.2e   |Yt - Yhost| %.2e  |Ye - Yhost| %.2e" % (
        i, j, k, np.linalg.norm(x - ctr), err[idx], lam * r.min(), np.linalg.norm(X) / L1, np.linalg.norm(X) / tmag.max(), np.abs(Yt[idx] - Yh).max(), np.abs(Ye[idx] - Yh).max()))
if os.environ.get("R06_DIAG_SAVE"):
    kk = int(worst[0]) // (n * n)
    np.savez(os.environ["R06_DIAG_SAVE"], Yt=Yt.reshape(n, n, n, 3)[kk], Ye=Ye.reshape(n, n, n, 3)[kk], k=kk)

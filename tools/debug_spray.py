import os, sys, ctypes, numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import shm_import
shm = shm_import.load()
from signed_heat_3d_amd.host_abi import HostSolver
pre = HostSolver("data/SprayBottle.pc").preprocess(hCoef=1.0)
n, S = pre["n"], pre["S"]
print("n", n, "S", S, "lam", pre["lam"], "cell", pre["cell"], "lam*cell", pre["lam"] * pre["cell"], "area min/max", pre["area"].min(), pre["area"].max())
lib = ctypes.CDLL("oracle/_build/libshm_oracle.so")
f64 = np.ctypeslib.ndpointer(np.float64, flags="C")
ci, cd = ctypes.c_int, ctypes.c_double
lib.shmo_conv_normalize.argtypes = [ci, f64, cd, ci, f64, f64, cd, ci, ci, f64]
Y = np.zeros(3 * n ** 3)
lib.shmo_conv_normalize(n, np.ascontiguousarray(pre["bbox_min"]), pre["cell"], S, np.ascontiguousarray(pre["pos"]).reshape(-1), np.ascontiguousarray(pre["wnormal"]).reshape(-1), pre["lam"], 0, n, Y)
Y = Y.reshape(-1, 3)
print("oracle Y non-finite nodes:", (~np.isfinite(Y).all(1)).sum())
s = shm.GridSolver()
s.set_problem(pre["pos"], pre["wnormal"], pre["area"], pre["lam"], n, pre["bbox_min"], pre["cell"])
s.run_conv()
G = np.stack([s.get_field(k) for k in (0, 1, 2)], 1)
bad = ~np.isfinite(G).all(1)
print("gpu Y non-finite nodes:", bad.sum())
ok = np.isfinite(G).all(1) & np.isfinite(Y).all(1)
print("max |dY| on finite nodes", np.abs(G[ok] - Y[ok]).max())
if bad.sum():
    idx = np.nonzero(bad)[0][:5]
    for v in idx:
        k, j, i = v // (n * n), (v // n) % n, v % n
        x = np.array([i, j, k]) * pre["cell"] + pre["bbox_min"]
        d = np.sqrt(((pre["pos"] - x) ** 2).sum(1))
        print("node", (i, j, k), "dmin", d.min(), "lam*dmin", pre["lam"] * d.min(), "oracle", Y[v], "gpu", G[v])

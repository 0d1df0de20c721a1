"""Conditioning of the explicit Schur complement S = A K^+ A^T (is a direct dense inverse accurate enough?):  python tools/schur_cond_probe.py data/bunny_small.obj 4"""
import os, sys, numpy as np
sys.path.insert(0, os.getcwd())
import shm_import
shm = shm_import.load()
from signed_heat_3d_amd.host_abi import HostSolver
path, hc = sys.argv[1], float(sys.argv[2])
pre = HostSolver(path).preprocess(hCoef=hc)
s = shm.GridSolver()
s.set_problem(pre["pos"], pre["wnormal"], pre["area"], pre["lam"], pre["n"], pre["bbox_min"], pre["cell"])
S = s.get_schur()
m = len(S)
w = np.linalg.eigvalsh(S)
print(path, "n", pre["n"], "m", m, "eig min %.3e max %.3e cond %.3e" % (w[0], w[-1], w[-1] / w[0]))
L = np.linalg.cholesky(S)
rng = np.random.default_rng(0)
x = rng.standard_normal(m); b = S @ x
y = np.linalg.solve(L.T, np.linalg.solve(L, b))
print("Cholesky solve rel. error %.2e; after one refinement %.2e" % (np.abs(y - x).max() / np.abs(x).max(), np.abs(y + np.linalg.solve(L.T, np.linalg.solve(L, b - S @ y)) - x).max() / np.abs(x).max()))

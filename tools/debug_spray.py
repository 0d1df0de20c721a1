import os, sys, numpy as np
sys.path.insert(0, os.getcwd())
import shm_import
shm = shm_import.load()
from signed_heat_3d_amd.host_abi import HostSolver
pre = HostSolver("data/SprayBottle.pc").preprocess(hCoef=0.0)
print("n", pre["n"], "S", pre["S"], "lam", pre["lam"], "h", pre["h"], "cell", pre["cell"], "area min/max", pre["area"].min(), pre["area"].max(), "nan areas", np.isnan(pre["area"]).sum())
print("zero areas", (pre["area"] == 0).sum(), "wn nan", np.isnan(pre["wnormal"]).sum())
for prec in (64, 32):
    s = shm.GridSolver(precision=prec)
    s.set_problem(pre["pos"], pre["wnormal"], pre["area"], pre["lam"], pre["n"], pre["bbox_min"], pre["cell"])
    s.run_conv()
    Y = np.stack([s.get_field(k) for k in (0, 1, 2)], axis=1)
    print(prec, "Y nan nodes", np.isnan(Y).any(axis=1).sum(), "of", Y.shape[0])
    s.run_divergence(False)
    b = s.get_field(3)
    print(prec, "b nan", np.isnan(b).sum())

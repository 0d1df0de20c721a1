import os, sys, numpy as np
sys.path.insert(0, os.getcwd())
import shm_import
shm = shm_import.load()
from signed_heat_3d_amd.host_abi import HostSolver
pre = HostSolver("data/bunny_small.obj").preprocess(hCoef=4.0)
ref = None
print("lambda*cell", pre["lam"] * pre["cell"], "S", pre["S"])
for fl in ("25", "16", "12", "9", "7", "5", "3"):
    os.environ["SHM_CONV_FAR_LOG"] = fl
    s = shm.GridSolver()
    s.set_problem(pre["pos"], pre["wnormal"], pre["area"], pre["lam"], pre["n"], pre["bbox_min"], pre["cell"])
    st = s.solve(); st = s.solve()
    phi, _ = s.get_phi()
    s.run_conv()
    Y = np.stack([s.get_field(k) for k in (0, 1, 2)], axis=1)
    if ref is None: ref = (Y, phi)
    print(fl, "conv ms %.1f total %.1f" % (st.ms_conv, st.ms_total), "dY %.2e dphi %.2e" % (np.abs(Y - ref[0]).max(), np.abs(phi - ref[1]).max()))
    s.close()

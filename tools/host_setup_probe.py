import os, sys
sys.path.insert(0, os.getcwd())
import shm_import
shm = shm_import.load()
from signed_heat_3d_amd.host_abi import HostSolver
for path, hc in (("data/bunny_small.obj", 3.0), ("data/bunny_small.obj", 4.0), ("data/rocker.obj", 4.0)):
    pre = HostSolver(path).preprocess(hCoef=hc)
    s = shm.GridSolver(verbose=True)
    s.set_problem(pre["pos"], pre["wnormal"], pre["area"], pre["lam"], pre["n"], pre["bbox_min"], pre["cell"])
    s.solve(); s.solve()

import os,sys
sys.path.insert(0,'/root/repo')
import shm_import
shm=shm_import.load()
from signed_heat_3d_amd.host_abi import HostSolver
pre=HostSolver('/root/repo/data/bunny_small.obj').preprocess(hCoef=2.0)
s=shm.GridSolver(verbose=False)
s.set_problem(pre["pos"],pre["wnormal"],pre["area"],pre["lam"],pre["n"],pre["bbox_min"],pre["cell"])
for _ in range(5): st=s.solve()
s.close()
s=shm.GridSolver(verbose=True)
s.set_problem(pre["pos"],pre["wnormal"],pre["area"],pre["lam"],pre["n"],pre["bbox_min"],pre["cell"])
for _ in range(3): st=s.solve()
print({k:round(getattr(st,k),3) for k in ("ms_conv","ms_div","ms_setup","ms_wait_setup","ms_pcg","ms_shift","ms_total")})

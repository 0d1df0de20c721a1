import sys,os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import shm_import; shm=shm_import.load()
from signed_heat_3d_amd.host_abi import HostSolver
pre=HostSolver('data/bunny_small.obj').preprocess(hCoef=2.0)
s=shm.GridSolver(verbose=True) if 'verbose' in shm.GridSolver.__init__.__code__.co_varnames else shm.GridSolver()
s.set_problem(pre["pos"],pre["wnormal"],pre["area"],pre["lam"],pre["n"],pre["bbox_min"],pre["cell"])
for _ in range(3): st=s.solve()
print(st.ms_total, st.ms_conv, st.ms_setup, st.ms_wait_setup, st.ms_pcg)

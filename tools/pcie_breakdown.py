import os, sys, time, numpy as np
sys.path.insert(0, os.getcwd())
import shm_import
shm = shm_import.load()
from signed_heat_3d_amd.host_abi import HostSolver
h = HostSolver("data/bunny_small.obj")
t=time.perf_counter(); pre = h.preprocess(hCoef=4.0); t_pre=time.perf_counter()-t
s = shm.GridSolver()
for it in range(3):
    t0=time.perf_counter()
    s.set_problem(pre["pos"], pre["wnormal"], pre["area"], pre["lam"], pre["n"], pre["bbox_min"], pre["cell"])
    t1=time.perf_counter()
    st=s.solve()
    t2=time.perf_counter()
    phi,_=s.get_phi()
    t3=time.perf_counter()
    print("preprocess %.1f ms | set_problem %.1f | solve %.1f | get_phi (alloc+D2H) %.1f" % (t_pre*1e3,(t1-t0)*1e3,(t2-t1)*1e3,(t3-t2)*1e3))
buf=np.empty(pre["n"]**3)
buf[:]=0
import ctypes as C
for it in range(2):
    t=time.perf_counter(); s._lib.shm_grid_get_phi(s._h, buf.ctypes.data, None, None); print("D2H into touched buffer %.1f ms" % ((time.perf_counter()-t)*1e3))
t=time.perf_counter(); v=np.zeros(pre["n"]**3); v[0]=1; print("np.zeros alloc+touch? %.1f ms" % ((time.perf_counter()-t)*1e3))
t=time.perf_counter(); v=np.empty(pre["n"]**3); v[:]=1.0; print("first-touch fill 134MB %.1f ms" % ((time.perf_counter()-t)*1e3))

"""Multi-slab (packed all-to-all layout) DCT sweeps at 512^3 fp64 on one GPU: primal + DCT preconditioner with local_slabs = 2, a few iterations.
python tools/seg_dct_probe.py   (SHM_GRID_LIB selects the build)"""
import os, sys
sys.path.insert(0, os.getcwd())
import shm_import
shm = shm_import.load()
from signed_heat_3d_amd.host_abi import HostSolver
pre = HostSolver("data/bunny_small.obj").preprocess(hCoef=5.0)
s = shm.GridSolver(local_slabs=2)
s.set_problem(pre["pos"], pre["wnormal"], pre["area"], pre["lam"], pre["n"], pre["bbox_min"], pre["cell"])
for _ in range(2):
    st = s.solve(solver="primal", precond="dct", max_iters=12, allow_noconv=True)
print(os.environ.get("SHM_GRID_LIB", "default"), "iters", st.iters, "pcg %.2f ms  precond avg %.3f ms" % (st.ms_pcg, st.ms_precond_avg))

import os, sys, json, subprocess
ROOT=os.getcwd()
sys.path.insert(0, ROOT)
import shm_import
shm = shm_import.load()
from signed_heat_3d_amd.host_abi import HostSolver
hc, prec = float(sys.argv[1]), int(sys.argv[2])
pre = HostSolver("data/bunny_small.obj").preprocess(hCoef=hc)
s = shm.GridSolver(precision=prec)
s.set_problem(pre["pos"], pre["wnormal"], pre["area"], pre["lam"], pre["n"], pre["bbox_min"], pre["cell"])
for _ in range(2):
    st = s.solve(solver="primal", precond="none", max_iters=200, allow_noconv=True)
N=pre["n"]**3; T=prec//8
print("waves", os.environ.get("SHM_FUSED_WAVES","dflt"), "n", pre["n"], "fp%d"%prec, "ms/iter %.4f  DIR %.0f RES %.0f XU %.0f GB/s  loop %.3f of 8TB/s" % (st.ms_pcg/st.iters, 3*N*T/st.ms_stencil_avg/1e6, 3*N*T/st.ms_update_xr_avg/1e6, 4*N*T/st.ms_update_p_avg/1e6, 8*N*T/(st.ms_pcg/st.iters)/1e6/8000))

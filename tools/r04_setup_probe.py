import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import shm_import
shm = shm_import.load()
from signed_heat_3d_amd.host_abi import HostSolver
pre = HostSolver(os.path.join(R, "data/rocker.obj")).preprocess(hCoef=float(sys.argv[1]) if len(sys.argv) > 1 else 4.0)
s = shm.GridSolver(verbose=True)
s.set_problem(pre["pos"], pre["wnormal"], pre["area"], pre["lam"], pre["n"], pre["bbox_min"], pre["cell"])
s.solve()
print("---- second solve", file=sys.stderr, flush=True)
st = s.solve()
print("conv %.1f wait %.2f setup %.1f pcg %.1f" % (st.ms_conv, st.ms_wait_setup, st.ms_setup, st.ms_pcg))

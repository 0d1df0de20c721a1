"""Step 1+2 alone (shm_grid_run_conv: no constraint set-up beside it) -- the program the PMC passes for the Step-1 kernel's own HBM traffic run:
with the set-up co-resident, device-wide counters sampled around the Step-1 dispatch also see the set-up kernels' bytes.
    python tools/conv_only.py [file hCoef precision reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import shm_import
shm = shm_import.load()
from signed_heat_3d_amd.host_abi import HostSolver
path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "data/bunny_small.obj")
hc = float(sys.argv[2]) if len(sys.argv) > 2 else 4.0
prec = int(sys.argv[3]) if len(sys.argv) > 3 else 64
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 2
pre = HostSolver(path).preprocess(hCoef=hc)
s = shm.GridSolver(precision=prec)
s.set_problem(pre["pos"], pre["wnormal"], pre["area"], pre["lam"], pre["n"], pre["bbox_min"], pre["cell"])
for _ in range(reps):
    s.run_conv()
print("conv_only: %s n=%d S=%d fp%d, %d x run_conv" % (os.path.basename(path), pre["n"], pre["S"], prec, reps))

"""Probe: the stencil-PCG legs of bench.py in different orders (why did the rocker leg read 0.31 ms of projection inside the default bench run?)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import shm_import
shm = shm_import.load()
from signed_heat_3d_amd.host_abi import HostSolver
import bench
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
order = sys.argv[1:] or ["rocker32"]
pre = {"bunny": HostSolver(os.path.join(ROOT, "data/bunny_small.obj")).preprocess(hCoef=5.0), "rocker": HostSolver(os.path.join(ROOT, "data/rocker.obj")).preprocess(hCoef=5.0)}
for leg in order:
    name, prec = leg[:-2], int(leg[-2:])
    r = bench.stencil_pcg_leg(shm, pre[name], prec, 0, name)
    print(leg, "ms/iter %.4f loop %.3f project %.3f" % (r["ms_per_iter"], r["loop_frac_of_hbm_peak"], r["ms_project_avg"]), {k: round(v["frac_of_hbm_peak"], 3) for k, v in r["kernels"].items()}, flush=True)

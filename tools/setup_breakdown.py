#!/usr/bin/env python3
"""Verbose constraint set-up breakdown ([shm] setup ... lines on stderr) for one workload:  python tools/setup_breakdown.py data/SprayBottle.pc 6 32"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import shm_import
shm = shm_import.load()
from signed_heat_3d_amd.host_abi import HostSolver
path, hCoef, prec = sys.argv[1], float(sys.argv[2]), int(sys.argv[3])
pre = HostSolver(os.path.join(ROOT, path)).preprocess(hCoef=hCoef)
s = shm.GridSolver(precision=prec, verbose=True)
s.set_problem(pre["pos"], pre["wnormal"], pre["area"], pre["lam"], pre["n"], pre["bbox_min"], pre["cell"])
for _ in range(2):
    st = s.solve(scrub=not path.endswith(".pc"), allow_noconv=True)
print("iters", st.iters, "conv %.1f setup %.1f wait %.2f pcg %.1f" % (st.ms_conv, st.ms_setup, st.ms_wait_setup, st.ms_pcg))

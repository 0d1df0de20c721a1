#!/usr/bin/env python3
"""Constraint set-up per workload: its wall time on an otherwise idle GPU (SHM_SETUP_ALONE=1: the solve waits for Step 1 before it starts the set-up) and
beside Step 1 (the shipped schedule: co-resident with the tiered fp64 kernel, time-sliced against the fp32 one), with what the solve then still waits
for.  Feeds the constants of tools/scaling_model.py.      python tools/setup_alone.py [workload ...]"""
import os, subprocess, sys
import os as _os; _os.environ.setdefault("SHM_DEBUG_KNOBS", "1")   # this tool drives the library's experiment knobs (read only behind this gate)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import WORKLOADS
CHILD = r'''
import os, sys
sys.path.insert(0, %r)
import shm_import
shm = shm_import.load()
from signed_heat_3d_amd.host_abi import HostSolver
path, hc, prec = sys.argv[1], float(sys.argv[2]), int(sys.argv[3])
pre = HostSolver(path).preprocess(hCoef=hc)
s = shm.GridSolver(precision=prec)
s.set_problem(pre["pos"], pre["wnormal"], pre["area"], pre["lam"], pre["n"], pre["bbox_min"], pre["cell"])
for _ in range(3):
    st = s.solve(scrub=not path.endswith(".pc"), allow_noconv=True)
print("m %%d conv %%.2f setup %%.2f wait %%.2f pcg %%.2f total %%.2f" %% (st.m, st.ms_conv, st.ms_setup, st.ms_wait_setup, st.ms_pcg, st.ms_total))
''' % ROOT
for wl in (sys.argv[1:] or ["bunny_small_64_f64", "bunny_small_128_f64", "bunny_small_256_f64", "bunny_small_512_f64", "bunny_pc_512_f64", "rocker_512_f32"]):
    path, hc, prec = WORKLOADS[wl]
    for alone in (False, True):
        env = dict(os.environ)
        if alone:
            env["SHM_SETUP_ALONE"] = "1"
        p = subprocess.run([sys.executable, "-c", CHILD, os.path.join(ROOT, path), str(hc), str(prec)], capture_output=True, text=True, env=env)
        print("%-26s %s  %s" % (wl, "set-up ALONE    " if alone else "set-up beside S1", p.stdout.strip() or p.stderr[-300:]))

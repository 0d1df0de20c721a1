#!/usr/bin/env python3
"""Host-side laps of the constraint set-up (library log, verbose) of one workload, beside Step 1 and alone.   python tools/setup_laps.py rocker_512_f32"""
import os, subprocess, sys
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
s = shm.GridSolver(precision=prec, verbose=True)
s.set_problem(pre["pos"], pre["wnormal"], pre["area"], pre["lam"], pre["n"], pre["bbox_min"], pre["cell"])
for i in range(2):
    sys.stderr.write("=== solve %%d\n" %% i); sys.stderr.flush()
    st = s.solve(scrub=not path.endswith(".pc"), allow_noconv=True)
print("m %%d conv %%.2f setup %%.2f wait %%.2f pcg %%.2f total %%.2f" %% (st.m, st.ms_conv, st.ms_setup, st.ms_wait_setup, st.ms_pcg, st.ms_total))
''' % ROOT
for wl in (sys.argv[1:] or ["rocker_512_f32"]):
    path, hc, prec = WORKLOADS[wl]
    for alone in (False, True):
        env = dict(os.environ)
        if alone:
            env["SHM_SETUP_ALONE"] = "1"
        p = subprocess.run([sys.executable, "-c", CHILD, os.path.join(ROOT, path), str(hc), str(prec)], capture_output=True, text=True, env=env)
        print("==== %s %s: %s" % (wl, "set-up ALONE" if alone else "set-up beside Step 1", p.stdout.strip()))
        err = p.stderr.split("=== solve 1")[-1]
        print("\n".join(l for l in err.splitlines() if "[shm]" in l)[:6000])

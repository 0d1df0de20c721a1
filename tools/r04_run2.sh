#!/bin/bash
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -x -q -s -k "step1_full_size or far_tier_exponent or translation_invariant or bench_py_multi or tiered_conv_stays or preconditioner_is_the_dct or phi_matches_lu_golden or matches_c_oracle_128 or every_data_file" > gpurun_out/r04_tests_run2.log 2>&1
echo "pytest rc $?" >> gpurun_out/r04_tests_run2.log
python - > gpurun_out/r04_n362.txt 2>&1 <<'PY'
import os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
import shm_import
shm = shm_import.load()
from signed_heat_3d_amd.host_abi import HostSolver
for f, hc, prec in [("bunny_small.obj", 4.5, 64), ("bunny_small.obj", 3.5, 64), ("rocker.obj", 4.5, 64), ("rocker.obj", 4.5, 32), ("bunny_small.obj", 5.5, 64)]:
    pre = HostSolver(os.path.join("data", f)).preprocess(hCoef=hc)
    s = shm.GridSolver(precision=prec)
    s.set_problem(pre["pos"], pre["wnormal"], pre["area"], pre["lam"], pre["n"], pre["bbox_min"], pre["cell"])
    for rep in range(3):
        t0 = time.time()
        st = s.solve()
        dt = time.time() - t0
    print("%s hCoef %.1f n=%d fp%d: %.1f ms/solve (conv %.1f, wait %.2f, pcg %.2f, shift %.2f) solver %d cg_form %d iters %d rel %.2e m %d" % (
        f, hc, pre["n"], prec, 1e3 * dt, st.ms_conv, st.ms_wait_setup, st.ms_pcg, st.ms_shift, st.solver, st.cg_form, st.iters, st.rel_residual, st.m), flush=True)
    phi, _ = s.get_phi()
    if pre["n"] <= 400:
        st2 = s.solve(solver="primal", precond="dct", tol=1e-10)
        phi2, _ = s.get_phi()
        print("    primal+dct: %d iters, pcg %.1f ms, L_inf(dual - primal) %.2e" % (st2.iters, st2.ms_pcg, np.abs(phi - phi2).max()), flush=True)
    s.close()
PY
timeout 1500 python tools/tier_worst_nodes.py > gpurun_out/r04_tier_worst_nodes.txt 2>&1

"""Projection cost in the 512^3 stencil-PCG leg of BASELINE.json configs[2] (rocker.obj, fp32 and fp64; 200 iterations of the plain projected CG) and, at 128^3, the
converged plain projected CG (iterations, residual, L_inf against the fp64 default solve) -- what a change to the projector must leave alone.
    python tools/r05_proj_probe.py [label]"""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import shm_import
shm = shm_import.load()
from signed_heat_3d_amd.host_abi import HostSolver
label = sys.argv[1] if len(sys.argv) > 1 else ""
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pre = HostSolver(os.path.join(R, "data/rocker.obj")).preprocess(hCoef=5.0)
n = pre["n"]
for prec in (32, 64):
    s = shm.GridSolver(precision=prec)
    s.set_problem(pre["pos"], pre["wnormal"], pre["area"], pre["lam"], n, pre["bbox_min"], pre["cell"])
    s.solve(tol=1e-30, solver="primal", precond="none", max_iters=16, allow_noconv=True)
    best = None
    for _ in range(3):
        st = s.solve(tol=1e-30, solver="primal", precond="none", max_iters=200, allow_noconv=True).as_dict()
        per = st["ms_pcg"] / st["iters"]
        best = per if best is None else min(best, per)
    print("%-10s rocker 512^3 fp%d: %.4f ms/iter  loop frac of 8 TB/s %.4f  ms_project_avg %.4f" % (label, prec, best, st["bytes_per_iter"] / (best * 1e-3) / 8e12, st["ms_project_avg"]), flush=True)
    s.close()
if os.environ.get("SHM_PROBE_QUICK"): sys.exit(0)
pre = HostSolver(os.path.join(R, "data/rocker.obj")).preprocess(hCoef=3.0)
n = pre["n"]
s = shm.GridSolver(precision=64)
s.set_problem(pre["pos"], pre["wnormal"], pre["area"], pre["lam"], n, pre["bbox_min"], pre["cell"])
s.solve(); ref = s.get_phi()[0]; s.close()
for prec in (32, 64):
    s = shm.GridSolver(precision=prec)
    s.set_problem(pre["pos"], pre["wnormal"], pre["area"], pre["lam"], n, pre["bbox_min"], pre["cell"])
    st = s.solve(solver="primal", precond="none", allow_noconv=True)
    phi = s.get_phi()[0]
    print("%-10s rocker 128^3 fp%d plain projected CG: m %d iters %d rel %.2e  L_inf vs fp64 default %.2e" % (label, prec, st.m, st.iters, st.rel_residual, np.abs(phi - ref).max()), flush=True)
    s.close()

"""A/B per process (the knobs are read once): CG on the explicit Schur complement against S applied through the grid, mid-size constraint sets, fp64.
    python tools/r04_dense_s_256.py"""
import os, subprocess, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1:
    sys.path.insert(0, R)
    import shm_import
    shm = shm_import.load()
    from signed_heat_3d_amd.host_abi import HostSolver
    f, hc, prec = sys.argv[1], float(sys.argv[2]), int(sys.argv[4])
    pre = HostSolver(os.path.join(R, "data", f)).preprocess(hCoef=hc)
    s = shm.GridSolver(precision=prec)
    s.set_problem(pre["pos"], pre["wnormal"], pre["area"], pre["lam"], pre["n"], pre["bbox_min"], pre["cell"])
    for _ in range(3):
        st = s.solve(scrub=not f.endswith(".pc"), allow_noconv=True)
    print("%-15s fp%d n=%d m=%5d %-22s total %.1f conv %.1f wait %.2f pcg %.2f iters %d cg_form %d rel %.1e" % (f, prec, pre["n"], st.m, sys.argv[3], st.ms_total, st.ms_conv, st.ms_wait_setup, st.ms_pcg, st.iters, st.cg_form, st.rel_residual), flush=True)
    sys.exit(0)
CASES = [("rocker.obj", 4.0, 64), ("SprayBottle.pc", 4.0, 64), ("chair.obj", 4.0, 64), ("rocker.obj", 3.0, 64), ("rocker.obj", 5.0, 64), ("chair.obj", 5.0, 64),
         ("chair.obj", 4.0, 32), ("chair.obj", 5.0, 32), ("rocker.obj", 4.0, 32), ("rocker.obj", 5.0, 32)]
for f, hc, prec in CASES:
    for name, env in (("shipped", {}), ("S through the grid", {"SHM_DUAL_NO_DENSE_S": "1"}), ("explicit S forced", {"SHM_DUAL_DENSE_S_ALWAYS": "1"})):
        p = subprocess.run([sys.executable, os.path.abspath(__file__), f, str(hc), name, str(prec)], env=dict(os.environ, **env), capture_output=True, text=True)
        print(p.stdout.strip() or p.stderr[-300:], flush=True)

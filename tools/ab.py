"""Generic A/B of library knobs, one process per setting (most knobs are read once per process):
    python tools/ab.py "<file>:<hCoef>:<precision>,..." "name=ENV1=v;ENV2=v" "name2=" ...
prints phases of the third solve for every (case, setting)."""
import os, subprocess, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if sys.argv[1] == "--child":
    sys.path.insert(0, R)
    import shm_import
    shm = shm_import.load()
    from signed_heat_3d_amd.host_abi import HostSolver
    f, hc, prec, name = sys.argv[2], float(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
    pre = HostSolver(os.path.join(R, "data", f)).preprocess(hCoef=hc)
    s = shm.GridSolver(precision=prec)
    s.set_problem(pre["pos"], pre["wnormal"], pre["area"], pre["lam"], pre["n"], pre["bbox_min"], pre["cell"])
    for _ in range(3):
        st = s.solve(scrub=not f.endswith(".pc"), allow_noconv=True)
    import time
    alone = []
    for _ in range(4):   # Step 1 with nothing beside it (shm_grid_run_conv synchronises): wall clock, minimum of four
        t0 = time.perf_counter(); s.run_conv(); alone.append((time.perf_counter() - t0) * 1e3)
    nom = float(pre["n"]) ** 3 * pre["S"]
    print("%-15s fp%d n=%d m=%5d %-24s total %.2f conv %.2f wait %.2f pcg %.2f iters %d solver %d cg_form %d rel %.1e conv_alone %.2f pairs64 %.3f pairs32 %.3f" % (
        f, prec, pre["n"], st.m, name, st.ms_total, st.ms_conv, st.ms_wait_setup, st.ms_pcg, st.iters, st.solver, st.cg_form, st.rel_residual, min(alone),
        st.pairs_fp64 / nom, st.pairs_fp32 / nom), flush=True)
    sys.exit(0)
cases = [c.split(":") for c in sys.argv[1].split(",")]
settings = []
for a in sys.argv[2:]:
    name, _, envs = a.partition("=")
    settings.append((name, dict(e.split("=", 1) for e in envs.split(";") if e)))
for f, hc, prec in cases:
    for name, env in settings:
        p = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", f, hc, prec, name], env=dict(os.environ, SHM_DEBUG_KNOBS="1", **env), capture_output=True, text=True)
        print(p.stdout.strip() or p.stderr[-300:], flush=True)

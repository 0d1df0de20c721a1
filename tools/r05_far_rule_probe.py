"""What Solver::choose_far_rule decides for the data files (the [shm] far rule ... line of a verbose handle) and what set_problem costs with the pilot.
    python tools/r05_far_rule_probe.py "<file>:<hCoef>,..." """
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import shm_import
shm = shm_import.load()
from signed_heat_3d_amd.host_abi import HostSolver
for spec in sys.argv[1].split(","):
    f, hc = spec.split(":")
    pre = HostSolver(os.path.join(R, "data", f)).preprocess(hCoef=float(hc))
    s = shm.GridSolver(precision=64, verbose=True)
    t0 = time.perf_counter()
    s.set_problem(pre["pos"], pre["wnormal"], pre["area"], pre["lam"], pre["n"], pre["bbox_min"], pre["cell"])
    t1 = time.perf_counter()
    s.set_problem(pre["pos"], pre["wnormal"], pre["area"], pre["lam"], pre["n"], pre["bbox_min"], pre["cell"])
    t2 = time.perf_counter()
    print("%-16s n=%d S=%d set_problem %.1f ms (first call %.1f)" % (f, pre["n"], len(pre["area"]), (t2 - t1) * 1e3, (t1 - t0) * 1e3), flush=True)
    s.close()

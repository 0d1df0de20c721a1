"""Does the source-aware z-slab plan (shm_config.slab_plan = SHM_SLAB_PLAN_STEP1) equalise Step 1?  Runs Step 1 slab by slab on ONE GPU (local_slabs = P:
the same launches a P-rank run would make, one slab after the other) with the equal-plane plan and with the weighted plan, and prints each slab's own
Step-1 time and evaluated pairs (SHM_CONV_SLAB_LOG=1) beside the host estimate the plan was cut by.
    python tools/slab_plan_check.py [workload ...]      # default: rocker_512_f32 bunny_small_256_f64 ; P = 4, 8"""
import os, re, subprocess, sys
import os as _os; _os.environ.setdefault("SHM_DEBUG_KNOBS", "1")   # this tool drives the library's experiment knobs (read only behind this gate)
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import WORKLOADS

CHILD = r'''
import os, sys
sys.path.insert(0, %r)
import shm_import
shm = shm_import.load()
from signed_heat_3d_amd.host_abi import HostSolver
path, hc, prec, P, plan = sys.argv[1], float(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
pre = HostSolver(path).preprocess(hCoef=hc)
s = shm.GridSolver(precision=prec, local_slabs=P, slab_plan=plan)
s.set_problem(pre["pos"], pre["wnormal"], pre["area"], pre["lam"], pre["n"], pre["bbox_min"], pre["cell"])
s.run_conv()
os.environ["SHM_CONV_SLAB_LOG"] = "1"
s.run_conv()
''' % ROOT

def run(path, hc, prec, P, plan):
    p = subprocess.run([sys.executable, "-c", CHILD, os.path.join(ROOT, path), str(hc), str(prec), str(P), str(plan)], capture_output=True, text=True)
    rows = [(int(a), int(b), float(ms), float(p64), float(p32)) for a, b, ms, p64, p32 in
            re.findall(r"step1 slab planes \[(\d+),(\d+)\) ms ([\d.]+) pairs_fp64 ([\d.e+]+) pairs_fp32 ([\d.e+]+)", p.stderr)]
    if not rows:
        print(p.stdout, p.stderr)
    return rows

import shm_import
shm = shm_import.load()
from signed_heat_3d_amd.host_abi import HostSolver
for wl in (sys.argv[1:] or ["rocker_512_f32", "bunny_small_256_f64"]):
    path, hc, prec = WORKLOADS[wl]
    pre = HostSolver(os.path.join(ROOT, path)).preprocess(hCoef=hc)
    w = shm.step1_plane_weights(pre["pos"], pre["wnormal"], pre["lam"], pre["n"], pre["bbox_min"], pre["cell"], prec)
    for P in (4, 8):
        for plan, name in ((0, "equal planes"), (1, "weighted    ")):
            rows = run(path, hc, prec, P, plan)
            ms = np.array([r[2] for r in rows])
            est = np.array([w[r[0]:r[1]].sum() for r in rows])
            print("%s P=%d %s planes %s" % (wl, P, name, [r[1] - r[0] for r in rows]))
            print("    Step-1 ms per slab %s  max/mean %.3f   (host estimate of the same slabs: max/mean %.3f)" % (np.round(ms, 1).tolist(), ms.max() / ms.mean(), est.max() / est.mean()))

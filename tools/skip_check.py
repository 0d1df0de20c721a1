"""Step-1 cluster skipping is exact to rounding: same Y with and without it (SHM_CONV_NO_SKIP=1), fp32 and fp64; and the fp32 kernel with 8 nodes
per lane (two culled halves per tile) gives the bits of the one with 4 (SHM_CONV_NPT4=1)."""
import os, sys, numpy as np
import os as _os; _os.environ.setdefault("SHM_DEBUG_KNOBS", "1")   # this tool drives the library's experiment knobs (read only behind this gate)
sys.path.insert(0, os.getcwd())
import shm_import
shm = shm_import.load()
from signed_heat_3d_amd.host_abi import HostSolver
for path, hc in (("data/SprayBottle.pc", 3.0), ("data/SprayBottle.pc", 4.0), ("data/rocker.obj", 4.0), ("data/knot.obj", 4.0)):
    pre = HostSolver(path).preprocess(hCoef=hc)
    for prec in (32, 64):
        out = {}
        for skip in (True, False):
            if skip: os.environ.pop("SHM_CONV_NO_SKIP", None)
            else: os.environ["SHM_CONV_NO_SKIP"] = "1"
            s = shm.GridSolver(precision=prec)
            s.set_problem(pre["pos"], pre["wnormal"], pre["area"], pre["lam"], pre["n"], pre["bbox_min"], pre["cell"])
            import time
            s.run_conv(); t = time.time(); s.run_conv(); dt = time.time() - t
            out[skip] = (np.stack([s.get_field(k) for k in (0, 1, 2)], 1), dt)
            s.close()
        a, b = out[True][0], out[False][0]
        ok = np.isfinite(a).all(1) & np.isfinite(b).all(1)
        print("%-20s n=%d fp%d: conv %.1f ms (skip) vs %.1f ms; max |dY| %.2e; non-finite nodes %d / %d" % (path, pre["n"], prec, out[True][1] * 1e3, out[False][1] * 1e3, np.abs(a[ok] - b[ok]).max(), (~np.isfinite(a).all(1)).sum(), (~np.isfinite(b).all(1)).sum()))
        if prec == 32:
            os.environ["SHM_CONV_NPT4"] = "1"
            s = shm.GridSolver(precision=prec)
            s.set_problem(pre["pos"], pre["wnormal"], pre["area"], pre["lam"], pre["n"], pre["bbox_min"], pre["cell"])
            s.run_conv(); t = time.time(); s.run_conv(); dt = time.time() - t
            c = np.stack([s.get_field(k) for k in (0, 1, 2)], 1)
            s.close()
            os.environ.pop("SHM_CONV_NPT4")
            okc = ok & np.isfinite(c).all(1)
            print("%-20s n=%d fp32: 8 nodes per lane %.1f ms vs 4 nodes %.1f ms; max |dY| %.2e" % (path, pre["n"], out[True][1] * 1e3, dt * 1e3, np.abs(a[okc] - c[okc]).max()))

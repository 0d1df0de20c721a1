#!/usr/bin/env python3
"""Full-size parity run (GPU box): HIP path vs the C oracle (all host cores, projected CG converged to 1e-12) on the SAME
inputs at BASELINE.json's grid sizes, plus the tolerance sweep that justifies the library's default tolerance.
Writes a JSON report (committed under profiles/).   usage: tools/parity_fullsize.py [workload] [out.json]"""
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import shm_import  # noqa: E402
from bench import WORKLOADS  # noqa: E402


def main():
    wl = sys.argv[1] if len(sys.argv) > 1 else "bunny_small_256_f64"
    out_path = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "gpurun_out", "parity_%s.json" % wl)
    shm = shm_import.load()
    from signed_heat_3d_amd.host_abi import HostSolver
    path, hCoef, precision = WORKLOADS[wl]
    host = HostSolver(os.path.join(ROOT, path))
    pre = host.preprocess(hCoef=hCoef)
    n, S = pre["n"], pre["S"]
    N = n ** 3
    scrub = not path.endswith(".pc")
    rep = {"workload": wl, "n": n, "S": int(S), "precision": precision}

    s = shm.GridSolver(precision=precision)
    s.set_problem(pre["pos"], pre["wnormal"], pre["area"], pre["lam"], n, pre["bbox_min"], pre["cell"])
    st = s.solve(tol=1e-12, scrub=scrub, max_iters=20000)
    phi_ref_gpu, _ = s.get_phi()
    rep["gpu_tight"] = {"tol": 1e-12, "iters": st.iters, "rel_residual": st.rel_residual}
    sweep = []
    for precond in ("dct", "none"):
        for tol in (1e-5, 1e-6, 1e-7, 1e-8, 1e-9, 1e-10):
            if precond == "none" and tol < 1e-8:
                continue
            st = s.solve(tol=tol, scrub=scrub, precond=precond, max_iters=20000)
            phi, _ = s.get_phi()
            sweep.append({"precond": precond, "tol": tol, "iters": st.iters, "rel_residual": st.rel_residual, "ms_pcg": st.ms_pcg,
                          "linf_vs_gpu_tight": float(np.abs(phi - phi_ref_gpu).max())})
            print(sweep[-1], flush=True)
    rep["tolerance_sweep"] = sweep

    # ---- C oracle on all host cores
    so = os.path.join(ROOT, "oracle", "_build", "libshm_oracle.so")
    lib = ctypes.CDLL(so)
    f64 = np.ctypeslib.ndpointer(np.float64, flags="C")
    ci, cd = ctypes.c_int, ctypes.c_double
    lib.shmo_compute_distance.argtypes = [ci, f64, cd, ci, f64, f64, f64, cd, ci, ci, cd, ci, f64, f64]
    lib.shmo_set_threads.argtypes = [ci]
    cores = min(os.cpu_count() or 1, 128)
    lib.shmo_set_threads(cores)
    phi_cpu = np.zeros(N)
    stc = np.zeros(5)
    t = time.time()
    rc = lib.shmo_compute_distance(n, np.ascontiguousarray(pre["bbox_min"]), pre["cell"], S, np.ascontiguousarray(pre["pos"]).reshape(-1),
                                   np.ascontiguousarray(pre["wnormal"]).reshape(-1), np.ascontiguousarray(pre["area"]), pre["lam"], int(scrub), 0,
                                   1e-12, 100000, phi_cpu, stc)
    rep["cpu_oracle"] = {"rc": rc, "threads": cores, "seconds": time.time() - t, "m": int(stc[0]), "iters": int(stc[1]), "rel_residual": stc[2],
                         "max_abs_Ax": stc[3], "shift": stc[4]}
    rep["linf_gpu_tight_vs_cpu_oracle"] = float(np.abs(phi_ref_gpu - phi_cpu).max())
    st = s.solve(scrub=scrub)  # library defaults
    phi, _ = s.get_phi()
    rep["default"] = {"tol": "library default", "iters": st.iters, "preconditioner": st.preconditioner,
                      "linf_vs_cpu_oracle": float(np.abs(phi - phi_cpu).max()), "phi_min": float(phi.min()), "phi_max": float(phi.max())}
    print(json.dumps(rep, indent=1))
    os.makedirs(os.path.dirname(out_path), exist_ok=True)
    json.dump(rep, open(out_path, "w"), indent=1)


if __name__ == "__main__":
    main()

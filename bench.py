#!/usr/bin/env python3
"""bench.py -- end-to-end grid Signed Heat Method throughput on MI355X (BASELINE.json metric).

One "step" = one full device-resident pass of the hot path (Step 1+2 direct summation, divergence,
constraint set-up -- rows, the explicit Schur complement S = A K^+ A^T and its inverse (or (A A^T)^-1 for the
iterative solvers) -- the constrained Poisson solve to the stated tolerance, shift) over one synthetic-free
input (the reference's own data files) whose sources and grid are already resident in HBM.  Everything the solve
derives from the sources or the right-hand side is recomputed in every step, the Green's table of the dual solver
included; only the transforms' twiddle / eigenvalue tables (a few KB, functions of n and h alone) are kept.

    python bench.py                       # N=1, workload = BASELINE.json configs[1]: bunny_small.obj, 256^3, fp64
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W          # z-slab partition over N GPUs, RCCL halo + all-reduce

Prints ONE JSON line on rank 0.  `value` = grid nodes / s of the whole job (strong scaling: the grid is fixed,
its z-slabs are spread over the ranks).  `roofline` is for the kernel that dominates the step: Step 1+2 (vector-ALU bound,
fp64 vector peak = fp64 matrix peak = 78.6 TFLOP/s on MI355X) when it outlasts the CG loop, else the CG loop's dominant kernel
against HBM; `roofline_pcg` always carries the latter.  Durations are measured live by HIP events on the solver's stream.  `cpu_baseline` times the C oracle (a port of the
reference's serial loops, oracle/shm_oracle.c) on this box's host cores over a bounded sample.
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (file, hCoef, precision)    n = 2*2^(hCoef+3)
    "bunny_small_256_f64": ("data/bunny_small.obj", 4.0, 64),   # BASELINE.json configs[1]
    "bunny_small_64_f64": ("data/bunny_small.obj", 2.0, 64),    # configs[0] (the reference's CPU-runnable case)
    "bunny_small_128_f64": ("data/bunny_small.obj", 3.0, 64),
    "rocker_512_f32": ("data/rocker.obj", 5.0, 32),             # configs[2]
    "bunny_pc_512_f64": ("data/bunny.pc", 5.0, 64),             # configs[3]
    "spraybottle_pc_1024_f32": ("data/SprayBottle.pc", 6.0, 32),  # configs[4] (.obj missing upstream -> .pc)
    "bunny_small_512_f64": ("data/bunny_small.obj", 5.0, 64),
    "bunny_small_512_f32": ("data/bunny_small.obj", 5.0, 32),
    "rocker_512_f64": ("data/rocker.obj", 5.0, 64),             # configs[2] / [4] in the reference's own arithmetic (BASELINE names fp32 for them)
    "spraybottle_pc_1024_f64": ("data/SprayBottle.pc", 6.0, 64),
}
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is the measured streaming ceiling


def cpu_baseline(pre, iters_cpu, tol, seconds_budget=25.0, threads=1):
    """Time the C oracle (kind "port": the reference's serial loops restated in C, 1 thread like the reference)
    on a bounded sample: `planes` z-planes of the Step-1 summation (linear in planes) and `cg_its` iterations of
    the host projected CG incl. its set-up, both extrapolated to the full job.  `iters_cpu` is the iteration count
    THE PORT'S OWN ALGORITHM (plain projected CG, no fast Poisson solve) needs at this tolerance -- measured by running
    that same algorithm on the GPU once, untimed (the reference's sparse LU is infeasible at this size)."""
    import subprocess
    so = os.path.join(ROOT, "oracle", "_build", "libshm_oracle.so")
    if not os.path.exists(so):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")], stdout=subprocess.DEVNULL)
    lib = ctypes.CDLL(so)
    f64 = np.ctypeslib.ndpointer(np.float64, flags="C")
    i64 = np.ctypeslib.ndpointer(np.int64, flags="C")
    ci, cd = ctypes.c_int, ctypes.c_double
    lib.shmo_conv_normalize.argtypes = [ci, f64, cd, ci, f64, f64, cd, ci, ci, f64]
    lib.shmo_divergence.argtypes = [ci, cd, f64, ci, f64]
    lib.shmo_constraint_rows.argtypes = [ci, f64, cd, ci, f64, i64, f64]
    lib.shmo_constraint_rows.restype = ci
    lib.shmo_constrained_solve.argtypes = [ci, cd, f64, ci, i64, f64, cd, ci, f64, f64]
    lib.shmo_set_threads.argtypes = [ci]
    lib.shmo_set_threads(threads)
    n, S = pre["n"], pre["S"]
    N = n ** 3
    bbox = np.ascontiguousarray(pre["bbox_min"])
    pos = np.ascontiguousarray(pre["pos"]).reshape(-1)
    wn = np.ascontiguousarray(pre["wnormal"]).reshape(-1)
    # --- Step 1+2 on a few z-planes
    pairs_per_plane = n * n * S
    planes = int(max(1, min(n, round(0.4 * seconds_budget * threads / (pairs_per_plane * 22e-9)))))
    Y = np.zeros(3 * N)
    k0 = n // 2
    t = time.perf_counter()
    lib.shmo_conv_normalize(n, bbox, pre["cell"], S, pos, wn, pre["lam"], k0, k0 + planes, Y)
    t_conv_sample = time.perf_counter() - t
    t_conv = t_conv_sample * n / planes
    # --- divergence (full) on a synthetic unit field: cost does not depend on the values
    Y[:] = 1.0 / np.sqrt(3.0)
    b = np.zeros(N)
    t = time.perf_counter()
    lib.shmo_divergence(n, pre["cell"], Y, 1, b)
    t_div = time.perf_counter() - t
    del Y
    # --- constrained solve: set-up + a few CG iterations on a smooth synthetic right-hand side
    nodes = np.zeros(8 * S, dtype=np.int64)
    coeffs = np.zeros(8 * S)
    m = lib.shmo_constraint_rows(n, bbox, pre["cell"], S, pos, nodes, coeffs)
    rng = np.random.default_rng(0)
    b = rng.standard_normal(N)
    phi = np.zeros(N)
    st = np.zeros(3)
    t = time.perf_counter()
    lib.shmo_constrained_solve(n, pre["cell"], b, m, nodes, coeffs, 0.0, 0, phi, st)   # set-up only (0 iterations)
    t_setup = time.perf_counter() - t
    cg_its = int(max(2, min(40, round(0.4 * seconds_budget / (N * 9e-9)))))
    t = time.perf_counter()
    lib.shmo_constrained_solve(n, pre["cell"], b, m, nodes, coeffs, 0.0, cg_its, phi, st)
    t_iter = max(1e-9, (time.perf_counter() - t - t_setup)) / cg_its
    total = t_conv + t_div + t_setup + t_iter * iters_cpu
    return {
        "value": N / total, "unit": "grid-nodes/s", "cores": threads, "kind": "port",
        "sample": "C port of the reference's serial loops (oracle/shm_oracle.c, gcc -O3 -march=x86-64-v3, %d thread(s)): Step 1+2 on %d of %d "
                  "z-planes (%.2f s, extrapolated linearly to %.0f s), divergence in full (%.2f s), dense-Cholesky projector set-up "
                  "(%.2f s), %d plain projected-CG iterations (%.3f s each) extrapolated to the %d iterations that algorithm needs at "
                  "tolerance %.1e (count taken from an untimed run of the same plain CG on the GPU)"
                  % (threads, planes, n, t_conv_sample, t_conv, t_div, t_setup, cg_its, t_iter, iters_cpu, tol),
        "seconds_extrapolated": total,
    }


def live_step1_traffic(path, hCoef, precision, timeout_s=150.0):
    """HBM bytes of Step 1 per step, measured NOW on this device (the line used to quote a committed record): two rocprofv3 passes over
    tools/conv_only.py (2 x shm_grid_run_conv, nothing else on the device) as CHILD processes -- `--pmc FETCH_SIZE` and `--pmc WRITE_SIZE` each in its own
    run with `--kernel-trace` only, bytes = (2 FETCH_SIZE + WRITE_SIZE) * 1024 (MI355X_MICROARCH.md, HBM section: both counters in KB, FETCH_SIZE tallies
    128-byte requests at 64 on gfx950), divided by the Step-1 executions of the run.  Returns (bytes or None, how it was obtained / why not)."""
    import shutil
    import signal
    import subprocess
    import tempfile
    prof = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if prof is None:
        return None, "rocprofv3 not found"
    if any(k.startswith(("ROCP_", "ROCPROF")) for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", ""):
        return None, "this process is itself being profiled (no profiler inside a profiler)"
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import pmc_report
    reps = 2
    tmp = tempfile.mkdtemp(prefix="shm_pmc_")
    env = dict(os.environ, TMPDIR=tmp)
    sums = {}
    t0 = time.time()
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            cmd = [prof, "--kernel-trace", "--pmc", counter, "-d", tmp, "-o", counter.lower(), "--",
                   sys.executable, os.path.join(ROOT, "tools", "conv_only.py"), os.path.join(ROOT, path), str(hCoef), str(precision), str(reps)]
            left = timeout_s - (time.time() - t0)
            if left <= 5.0:
                return None, "time budget of %.0f s spent before the %s pass" % (timeout_s, counter)
            child = subprocess.Popen(cmd, cwd=tmp, env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, start_new_session=True)
            try:
                rc = child.wait(timeout=left)
            except subprocess.TimeoutExpired:
                try:
                    os.killpg(child.pid, signal.SIGKILL)   # the process group this call started, nothing else
                except OSError:
                    pass
                child.wait()
                return None, "%s pass timed out" % counter
            if rc != 0:
                return None, "%s pass exited with status %d" % (counter, rc)
            db = next((os.path.join(dp, f) for dp, _, fs in os.walk(tmp) for f in fs if f.startswith(counter.lower()) and f.endswith("_results.db")), None)
            if db is None:
                return None, "%s pass left no results database" % counter
            per = {k: v for k, v in pmc_report.per_kernel(db, counter).items() if "conv_" in k and "kernel" in k}
            if not per:
                return None, "no Step-1 dispatch in the %s pass" % counter
            sums[counter] = (sum(v[0] for v in per.values()), sum(v[1] for v in per.values()))
        nbytes = (2.0 * sums["FETCH_SIZE"][0] + sums["WRITE_SIZE"][0]) * 1024.0 / reps
        return nbytes, ("measured in this run: rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (one pass each) over tools/conv_only.py, %d Step-1 launches per step, "
                        "(2 FETCH_SIZE + WRITE_SIZE) * 1024; %.0f s" % (sums["FETCH_SIZE"][1] // reps, time.time() - t0))
    except Exception as e:   # informational: never lose the line over it
        return None, "failed: %r" % (e,)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def step1_roofline(avg, nominal_pairs, precision, tiered=None):
    """Step 1+2 against the vector-ALU roofline, from the pairs the kernel evaluated (per rank)."""
    p64, p32 = float(avg.get("pairs_fp64", 0.0)), float(avg.get("pairs_fp32", 0.0))
    t = avg["ms_conv"] * 1e-3
    ev = p64 + p32
    t_at_peak = 18.0 * p64 / 78.6e12 + 18.0 * p32 / 157.3e12
    if tiered is None:   # (SHM_CONV_EXACT / SHM_CONV32_CLASSIC are experiment knobs of the library: read only under SHM_DEBUG_KNOBS=1)
        knobs = os.environ.get("SHM_DEBUG_KNOBS", "0") not in ("", "0")
        tiered = not (knobs and os.environ.get("SHM_CONV_EXACT" if precision == 64 else "SHM_CONV32_CLASSIC"))
    return {"kernel": "conv_tiered_kernel" if tiered else "conv_normalize_kernel", "bound": "valu",
            "pairs_nominal": nominal_pairs, "pairs_fp64": p64, "pairs_fp32": p32,
            "pairs_evaluated_over_nominal": ev / nominal_pairs if nominal_pairs else None,
            "pairs_note": "pairs_fp64 / pairs_fp32 = (node, source) pairs the kernel evaluated in fp64 / packed fp32 (its own counters; shm_stats); nominal = N*S per rank. "
                          "fp64 solve: terms below e^-8 of a node block's dominant terms take the packed-fp32 tier (error budget on Y: 1e-8; shm_opts.step1_arith = EXACT_F64: all "
                          "fp64); fp32 solve: every kept pair in packed fp32 (the few sources outside the fp32 exponent range of a block in fp64); sources whose terms vanish "
                          "against the budget are dropped and count in neither",
            "nominal_pairs_per_s": nominal_pairs / t, "evaluated_pairs_per_s": ev / t,
            "achieved_TFLOPs_18_per_evaluated_pair": 18.0 * ev / t / 1e12,
            "peak_TFLOPs_weighted": (18.0 * ev / t_at_peak / 1e12) if t_at_peak > 0 else None,
            "frac": t_at_peak / t if t > 0 else None,
            "launches_per_step": int(round(avg.get("conv_launches", 0)))}


def kernel_table(avg, n_local, T, world, gathered):
    """Per-kernel algorithmic bytes of the decomposition actually launched against the HIP-event launch durations the
    library sampled.  name -> (algorithmic bytes per launch, avg ms per launch, launches per CG iteration)."""
    has_pre = int(avg["preconditioner"]) == 2
    is_dual = int(avg["solver"]) in (2, 3)
    TP = T  # the preconditioner sweeps run in the solve precision
    if int(avg.get("cg_form", 0)) in (1, 4):
        # fused sweeps (shm_cg_fused.hip.h): q = Kp is never stored, x is updated every other iteration: 3 + 3 + 4/2 = 8NT per iteration
        # (cg_form 4: x on half the grid in every iteration, beside the projection: 2NT per launch)
        half = int(avg.get("cg_form", 0)) == 4
        kernels = {
            "cg_fused_kernel<DIR>": (3 * n_local * T, avg["ms_stencil_avg"], 1),      # reads z, p; writes p'; partial p'.Kp'
            "cg_fused_kernel<RES>": (3 * n_local * T, avg["ms_update_xr_avg"], 1),    # reads r, p'; writes r; partial ||r||^2
            "cg_x_update2_kernel": ((2 if half else 4) * n_local * T, avg["ms_update_p_avg"], 1 if half else 0.5),    # reads x, p_a, p_b; writes x
        }
    else:
        kernels = {
            "stencil_dot_kernel": (2 * n_local * T, avg["ms_stencil_avg"], 1),
            "update_xr_kernel": (6 * n_local * T, avg["ms_update_xr_avg"], 1),
            "update_p_kernel": (3 * n_local * T, avg["ms_update_p_avg"], 1),
        }
    if has_pre:  # five DCT sweeps (x-fwd, y-fwd, z-fused, y-inv, x-inv[+dot]): 3T + 8TP bytes per node (2T + 8TP without the dot)
        kernels["dct_lines_kernel"] = (n_local * ((2 if is_dual else 3) * T + 8 * TP) / 5.0, avg["ms_precond_avg"] / 5.0, 5)
    if is_dual:
        # the dual solver has no N-sized CG sweeps (its vectors are m-dimensional); on one GPU its five sweeps per iteration are
        # sparse (active x tiles, active z-planes, masked z I/O) and the library reports the bytes they actually move
        kernels = {"dct_lines_kernel": (avg["bytes_per_iter"] / (1 if gathered else world) / 5.0, avg["ms_precond_avg"] / 5.0, 5)}
        if int(avg.get("cg_form", 0)) in (2, 3):
            # explicit Schur complement (csrc/shm_schur.hip.h): the iteration's grid part is one dense m x m mat-vec (cg_form 3: S p in the CG;
            # cg_form 2, direct solve: S^-1 r per pass, the residual check S mu is in ms_project_avg)
            m = float(avg["m"])
            kernels = {"ginv_matvec_kernel<double>": (m * m * 8.0, avg["ms_precond_avg"], 1)}
    kinfo = {k: {"algorithmic_bytes_per_launch": b, "avg_ms_per_launch": ms, "launches_per_iter": cnt,
                 "achieved_GBps": (b / (ms * 1e-3) / 1e9 if ms > 0 else None),
                 "frac_of_hbm_peak": (b / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS if ms > 0 else None)}
             for k, (b, ms, cnt) in kernels.items()}
    return kernels, kinfo


def stencil_pcg_leg(shm, pre, precision, device, label, iters=200):
    """The north star's stencil-PCG at a fixed iteration count: the primal projected stencil CG (fused sweeps, 8NT per iteration) with the constraint projector
    of THIS source set in the loop -- dense (A A^T)^-1 for m <= 6144 (bunny), two-level (boxes + separator) above (rocker at 512^3: m = 12 612)."""
    n = pre["n"]
    N = n ** 3
    T = precision // 8
    s = shm.GridSolver(device=device, precision=precision)
    s.set_problem(pre["pos"], pre["wnormal"], pre["area"], pre["lam"], n, pre["bbox_min"], pre["cell"])
    s.solve(tol=1e-30, solver="primal", precond="none", max_iters=16, allow_noconv=True)
    st = s.solve(tol=1e-30, solver="primal", precond="none", max_iters=iters, allow_noconv=True).as_dict()
    s.close()
    _, kinfo = kernel_table(st, N, T, 1, False)
    per_iter = st["ms_pcg"] / max(1.0, st["iters"])
    return {
        "workload": "%s 512^3, primal projected stencil CG, %d iterations (fixed count: kernel rates, not a converged solve)" % (label, iters),
        "dtype": "f%d" % precision, "iters": int(st["iters"]), "constraint_rows": int(st["m"]),
        "projector": "dense (A A^T)^-1" if int(st["m"]) <= 6144 else "two-level (A A^T)^-1 (boxes + separator)",
        "ms_per_iter": per_iter, "algorithmic_bytes_per_iter": st["bytes_per_iter"],
        "loop_achieved_GBps": st["bytes_per_iter"] / (per_iter * 1e-3) / 1e9, "loop_frac_of_hbm_peak": st["bytes_per_iter"] / (per_iter * 1e-3) / 1e9 / HBM_PEAK_GBS,
        "ms_project_avg": st["ms_project_avg"], "kernels": kinfo,
        "note": "N-sized launches per iteration: DIR (3NT) + RES (3NT) + x_update2 (4NT every other iteration) = 8NT; the loop figure also contains "
                "the m-sized projection (gather, (A A^T)^-1 mat-vec, scatter: ms_project_avg) and the scalar reductions"}


def also_legs(shm, HostSolver, device, tol, pre256, scrub256):
    """Extra N=1 legs so that the driver's line carries the whole BASELINE.json metric ("256^3 & 512^3") and the north star's "stencil-PCG at 512^3" as
    driver-observed numbers: (0) the headline workload with Step 1 entirely in fp64 (shm_opts.step1_arith = EXACT_F64: the reference's arithmetic), (1) the
    512^3 end-to-end solve (default dual solver), (2) the primal stencil-PCG at 512^3 in fp64 and fp32 for a fixed iteration count with per-kernel achieved
    GB/s -- on bunny_small.obj AND on rocker.obj, BASELINE.json configs[2]'s "HBM-roofline run", whose 12 612 constraint rows put the two-level
    (A A^T)^-1 into every projection -- (3) configs[2] (rocker 512^3 fp32) end to end with the L_inf of its fp32 phi against an fp64 run of the same
    configuration ("fp32 configs: report only")."""
    out = {}
    # (0) all-fp64 Step 1 on the headline workload
    s = shm.GridSolver(device=device, precision=64)
    s.set_problem(pre256["pos"], pre256["wnormal"], pre256["area"], pre256["lam"], pre256["n"], pre256["bbox_min"], pre256["cell"])
    s.solve(tol=tol, scrub=scrub256, step1="exact_f64")
    reps = 5
    t0 = time.perf_counter()
    sts = [s.solve(tol=tol, scrub=scrub256, step1="exact_f64").as_dict() for _ in range(reps)]
    dt = (time.perf_counter() - t0) / reps
    s.close()
    a = {k: float(np.mean([x[k] for x in sts])) for k in sts[0]}
    N256 = pre256["n"] ** 3
    out["exact_fp64"] = {
        "workload": "the headline workload (bunny_small.obj 256^3 fp64) with shm_opts.step1_arith = SHM_STEP1_EXACT_F64: every (node, source) pair of Step 1 in fp64 "
                    "like the reference (signed_heat_3d.cpp:45-49); %d timed solves" % reps,
        "value": N256 / dt, "unit": "grid-nodes/s", "ms_per_step": 1e3 * dt, "dtype": "f64",
        "phases_ms": {k: a[k] for k in ("ms_conv", "ms_div", "ms_wait_setup", "ms_pcg", "ms_shift", "ms_total")},
        # (round 5: EXACT_F64 runs the tiered kernel with nothing far and nothing dropped where the grid fits a block's exponent span -- it does at 256^3)
        "step1": step1_roofline(a, float(N256) * float(pre256["S"]), 64, tiered=True)}
    # (0b) round 6 -- what the budget knob of Step 1 spans (shm_opts.step1_budget; the default stays 1e-8): ms per solve and L_inf of phi against the solve whose Step 1
    # is all fp64 (itself held to the C oracle at this size by tests/test_gpu_parity.py::test_phi_of_the_default_solve_against_c_oracle_at_full_size: 3.6e-10, where
    # the same curve is asserted against the oracle directly -- 126 s of host time that do not belong in a bench run)
    s = shm.GridSolver(device=device, precision=64)
    s.set_problem(pre256["pos"], pre256["wnormal"], pre256["area"], pre256["lam"], pre256["n"], pre256["bbox_min"], pre256["cell"])
    s.solve(tol=tol, scrub=scrub256, step1="exact_f64")
    phi_exact = s.get_phi()[0]
    curve = []
    for budget in (1e-8, 1e-7, 1e-6):
        s.solve(tol=tol, scrub=scrub256, step1_budget=budget)
        t0 = time.perf_counter()
        stb = [s.solve(tol=tol, scrub=scrub256, step1_budget=budget).as_dict() for _ in range(reps)]
        dtb = (time.perf_counter() - t0) / reps
        ab = {k: float(np.mean([x[k] for x in stb])) for k in stb[0]}
        curve.append({"step1_budget": budget, "ms_per_step": 1e3 * dtb, "ms_conv": ab["ms_conv"], "value": N256 / dtb,
                      "packed_fp32_share_of_evaluated_pairs": ab["pairs_fp32"] / max(1.0, ab["pairs_fp32"] + ab["pairs_fp64"]),
                      "linf_phi_vs_exact_f64_step1": float(np.abs(s.get_phi()[0] - phi_exact).max())})
    s.close()
    out["step1_budget_curve"] = {"workload": "bunny_small.obj 256^3 fp64, %d timed solves per point" % reps, "unit": "grid-nodes/s", "points": curve,
                                 "note": "a reported curve, not a new default: the library default is 1e-8 (tests hold Y to it against the C oracle at full size)"}
    # (0c) configs[0] -- the only size the reference itself can run (64^3, CPU Eigen path) -- as a first-class number of this record
    pre64 = HostSolver(os.path.join(ROOT, "data/bunny_small.obj")).preprocess(hCoef=2.0)
    s = shm.GridSolver(device=device, precision=64)
    s.set_problem(pre64["pos"], pre64["wnormal"], pre64["area"], pre64["lam"], pre64["n"], pre64["bbox_min"], pre64["cell"])
    for _ in range(3):
        s.solve(tol=tol)
    reps64 = 50
    t0 = time.perf_counter()
    sts = [s.solve(tol=tol).as_dict() for _ in range(reps64)]
    dt = (time.perf_counter() - t0) / reps64
    s.close()
    a = {k: float(np.mean([x[k] for x in sts])) for k in sts[0]}
    out["bunny_small_64_f64"] = {"workload": "BASELINE.json configs[0]: bunny_small.obj at 64^3 (hCoef 2) fp64, library defaults; %d timed solves" % reps64,
                                 "value": pre64["n"] ** 3 / dt, "unit": "grid-nodes/s", "ms_per_step": 1e3 * dt, "constraint_rows": int(a["m"]), "cg_iters": int(a["iters"]),
                                 "phases_ms": {k: a[k] for k in ("ms_conv", "ms_div", "ms_setup", "ms_wait_setup", "ms_pcg", "ms_shift", "ms_total")}}
    pre = HostSolver(os.path.join(ROOT, "data/bunny_small.obj")).preprocess(hCoef=5.0)
    n = pre["n"]
    N = n ** 3
    s = shm.GridSolver(device=device, precision=64)
    s.set_problem(pre["pos"], pre["wnormal"], pre["area"], pre["lam"], n, pre["bbox_min"], pre["cell"])
    s.solve(tol=tol)
    t0 = time.perf_counter()
    reps = 3
    sts = [s.solve(tol=tol).as_dict() for _ in range(reps)]
    dt = (time.perf_counter() - t0) / reps
    s.close()
    a = {k: float(np.mean([x[k] for x in sts])) for k in sts[0]}
    _, kinfo = kernel_table(a, N, 8, 1, False)
    out["bunny_small_512_f64_end_to_end"] = {
        "value": N / dt, "unit": "grid-nodes/s", "ms_per_step": 1e3 * dt, "solver": "dual (library default)", "cg_iters": int(a["iters"]),
        "phases_ms": {k: a[k] for k in ("ms_conv", "ms_div", "ms_wait_setup", "ms_pcg", "ms_shift", "ms_total")},
        "ms_per_iter": a["ms_pcg"] / max(1.0, a["iters"]), "kernels": kinfo}
    pre_r = HostSolver(os.path.join(ROOT, "data/rocker.obj")).preprocess(hCoef=5.0)
    for precision, name in ((64, "f64"), (32, "f32")):
        out["stencil_pcg_512_" + name] = stencil_pcg_leg(shm, pre, precision, device, "bunny_small.obj")
        out["stencil_pcg_512_" + name + "_rocker"] = stencil_pcg_leg(shm, pre_r, precision, device, "rocker.obj (BASELINE.json configs[2])")
    # configs[2]: rocker.obj 512^3 fp32, and the same configuration in fp64 for the error of the fp32 path
    pre = pre_r
    n = pre["n"]
    phis = {}
    for precision in (32, 64):
        s = shm.GridSolver(device=device, precision=precision)
        s.set_problem(pre["pos"], pre["wnormal"], pre["area"], pre["lam"], n, pre["bbox_min"], pre["cell"])
        s.solve()
        reps = 3 if precision == 32 else 1
        t0 = time.perf_counter()
        for _ in range(reps):
            st = s.solve().as_dict()
        dt = (time.perf_counter() - t0) / reps
        phis[precision] = s.get_phi()[0]
        if precision == 32:
            out["rocker_512_f32"] = {"value": n ** 3 / dt, "unit": "grid-nodes/s", "ms_per_step": 1e3 * dt, "cg_iters": int(st["iters"]),
                                     "constraint_rows": int(st["m"]), "phases_ms": {k: st[k] for k in ("ms_conv", "ms_div", "ms_wait_setup", "ms_pcg", "ms_shift")}}
        s.close()
    out["rocker_512_f32"]["linf_fp32_vs_fp64"] = float(np.abs(phis[32] - phis[64]).max())
    out["rocker_512_f32"]["max_abs_phi"] = float(np.abs(phis[64]).max())
    return out


def multi_gpu_legs(shm, HostSolver, dist, torch, args, pre, precision, scrub, rank, world, local_rank, barrier, out=None):
    """N > 1 only.  The timed default above is the library's AUTO: since round 6 the slab-distributed explicit-S dual solve where it applies (256^3 ... 512^3, S <= 16384:
    S and S^-1 replicated beside every rank's Step 1, K^+ on the z-slabs), the gathered dual solve elsewhere (Steps 1-2 on z-slabs, D^T Y gathered, whole-grid solve
    replicated) -- which also runs here as the leg "gathered_dual" for the comparison.  BASELINE.json's own multi-GPU configurations come first.  The split the
    north star names -- z-slab stencil PCG with a one-plane halo exchange per sweep and an all-reduce per dot product -- is SHM_SOLVER_PRIMAL; it is
    run here as further legs on the SAME ranks so that whichever multi-GPU run the driver gets covers both: the DCT-preconditioned stencil PCG to the
    tolerance (z-slab transforms: two all-to-alls per application) and the plain stencil CG for a fixed 200 iterations (halo + two all-reduces per
    iteration, nothing else).  Collective: every rank calls this with the same arguments.  `out` is filled leg by leg: should a later leg hang, the watchdog of main()
    prints what the earlier ones measured."""
    out = {} if out is None else out
    # BASELINE.json's own multi-GPU configurations, end to end with the library defaults on the SAME ranks: configs[3] (bunny.pc 512^3 fp64) and configs[4]
    # (SprayBottle.pc 1024^3 fp32, z-slabs weighted by the Step-1 work the source culling leaves in them) -- so that whichever multi-GPU record the driver
    # gets carries them whatever --workload it timed.  SHM_BENCH_MULTI_HCOEF: stand-in grid size for the flow test on a one-GPU box.
    for wl in ("bunny_pc_512_f64", "spraybottle_pc_1024_f32"):
        if wl == args.workload:
            continue
        try:
            path2, hc2, prec2 = WORKLOADS[wl]
            hc2 = float(os.environ.get("SHM_BENCH_MULTI_HCOEF", hc2))
            try:
                pre2 = HostSolver(os.path.join(ROOT, path2)).preprocess(hCoef=hc2)
            except Exception as e:   #  a rank that fails BEFORE the collectives below must not leave the others waiting in them: everyone learns of it first
                pre2, pre_err = None, repr(e)
            oks = [None] * world
            dist.all_gather_object(oks, pre2 is not None)
            if not all(oks):
                raise RuntimeError("host pre-processing failed on rank(s) %s%s" % ([r for r, ok in enumerate(oks) if not ok], "" if pre2 is not None else ": " + pre_err))
            box = [shm.comm_unique_id() if rank == 0 else None]
            dist.broadcast_object_list(box, src=0)
            plan = 1 if prec2 == 32 else 0
            s2 = shm.GridSolver(device=local_rank, precision=prec2, rank=rank, world=world, rccl_unique_id=box[0], slab_plan=plan)
            n2 = pre2["n"]
            s2.set_problem(pre2["pos"], pre2["wnormal"], pre2["area"], pre2["lam"], n2, pre2["bbox_min"], pre2["cell"])
            scrub2 = not path2.endswith(".pc")
            s2.solve(scrub=scrub2)
            barrier()
            reps = 2 if n2 <= 512 else 1
            t0 = time.perf_counter()
            sts = [s2.solve(scrub=scrub2).as_dict() for _ in range(reps)]
            barrier()
            dt = time.perf_counter() - t0
            tt = torch.tensor([dt], dtype=torch.float64, device="cuda" if args.dist_backend == "nccl" else "cpu")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item()) / reps
            a = {k: float(np.mean([x[k] for x in sts])) for k in sts[0]}
            mine = {"rank": rank, "ms_conv": a["ms_conv"], "pairs_fp64": a.get("pairs_fp64", 0.0), "pairs_fp32": a.get("pairs_fp32", 0.0), "ms_total": a["ms_total"]}
            per_rank = [None] * world
            dist.all_gather_object(per_rank, mine)
            out[wl + "_end_to_end"] = {"value": n2 ** 3 / dt, "unit": "grid-nodes/s", "ms_per_step": 1e3 * dt, "steps": reps, "grid": "%d^3" % n2, "sources": int(pre2["S"]),
                                       "constraint_rows": int(a["m"]), "dtype": "f64 (Step 1: f64 / packed-f32 tiers)" if prec2 == 64 else "f32",
                                       "partition": "z-slabs x%d%s" % (world, " (planes weighted by Step-1 work)" if plan else ""),
                                       "solver": int(a["solver"]), "cg_form": int(a["cg_form"]), "cg_iters": int(a["iters"]), "rel_residual": a["rel_residual"],
                                       "phases_ms": {k: a[k] for k in ("ms_conv", "ms_div", "ms_wait_setup", "ms_pcg", "ms_shift", "ms_total")},
                                       "per_rank": per_rank}
            s2.close()
        except Exception as e:
            out[wl + "_end_to_end"] = {"failed": repr(e)}
    box = [shm.comm_unique_id() if rank == 0 else None]
    dist.broadcast_object_list(box, src=0)
    s = shm.GridSolver(device=local_rank, precision=precision, rank=rank, world=world, rccl_unique_id=box[0], slab_plan=0)   # equal planes: what the z-slab transforms need
    n = pre["n"]
    N = n ** 3
    s.set_problem(pre["pos"], pre["wnormal"], pre["area"], pre["lam"], n, pre["bbox_min"], pre["cell"])
    # (round 6: the timed default is the slab-distributed explicit-S dual solve where it applies; "gathered_dual" is the default of rounds 1-5 -- D^T Y gathered, the whole-grid
    # solve replicated on every rank -- on the same ranks, for the comparison)
    legs = [("gathered_dual", dict(solver="dual", tol=args.tol), max(1, min(3, args.steps))),
            ("primal_pcg", dict(solver="primal", precond="auto", tol=args.tol), max(1, min(3, args.steps))),
            ("primal_plain_cg_200", dict(solver="primal", precond="none", tol=1e-30, max_iters=200, allow_noconv=True), 1)]
    for name, kw, reps in legs:
        try:
            s.solve(scrub=scrub, **kw)
            barrier()
            t0 = time.perf_counter()
            sts = [s.solve(scrub=scrub, **kw).as_dict() for _ in range(reps)]
            barrier()
            dt = time.perf_counter() - t0
            tt = torch.tensor([dt], dtype=torch.float64, device="cuda" if args.dist_backend == "nccl" else "cpu")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item()) / reps
            a = {k: float(np.mean([x[k] for x in sts])) for k in sts[0]}
            _, kinfo = kernel_table(a, N / world, precision // 8, world, False)
            per_iter = a["ms_pcg"] / max(1.0, a["iters"])
            out[name] = {"value": N / dt, "unit": "grid-nodes/s", "ms_per_step": 1e3 * dt, "steps": reps, "cg_iters": int(a["iters"]), "rel_residual": a["rel_residual"],
                         "solver": int(a["solver"]), "cg_form": int(a["cg_form"]),
                         "preconditioner": "dct (z-slab transforms, two all-to-alls per application)" if int(a["preconditioner"]) == 2 else "none",
                         "per_iteration": "gather of D^T Y, whole-grid dual solve on every rank" if name == "gathered_dual" else
                                          "one-plane halo of the direction to each slab neighbour before the DIR sweep, all-reduce of p.Kp, all-reduce of [||r||^2, A r]",
                         "phases_ms": {k: a[k] for k in ("ms_conv", "ms_div", "ms_wait_setup", "ms_pcg", "ms_shift", "ms_total")},
                         "ms_per_iter": per_iter, "algorithmic_bytes_per_iter_per_rank": a["bytes_per_iter"] / world,
                         "loop_frac_of_hbm_peak_per_rank": a["bytes_per_iter"] / world / (per_iter * 1e-3) / 1e9 / HBM_PEAK_GBS if per_iter > 0 else None,
                         "kernels": kinfo}
        except Exception as e:   # never lose the headline over an extra leg (every rank fails alike: the library's checks are rank-independent)
            out[name] = {"failed": repr(e)}
    s.close()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="bunny_small_256_f64", choices=sorted(WORKLOADS))
    ap.add_argument("--tol", type=float, default=0.0, help="projected-CG relative residual tolerance (0 = library default)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-live-traffic", action="store_true", help="take roofline.traffic from the committed PMC record instead of measuring it (two rocprofv3 child runs, ~30 s)")
    ap.add_argument("--no-also", action="store_true", help="skip the extra legs (N=1: all-fp64 Step 1, 512^3 end-to-end, 512^3 stencil-PCG fp64/fp32 on bunny and rocker, "
                                                            "rocker 512^3 fp32 vs fp64; N>1: the z-slab stencil-PCG legs)")
    ap.add_argument("--max-iters", type=int, default=0, help="cap the CG iterations (0 = library default 20 n); a capped run reports kernel rates, not a converged solve")
    ap.add_argument("--precond", default="auto", choices=["auto", "none", "dct"])
    ap.add_argument("--solver", default="auto", choices=["auto", "primal", "dual", "dual_slabs"])
    ap.add_argument("--slab-plan", default="auto", choices=["auto", "equal", "step1"],
                    help="z-slab plan of a multi-GPU run: equal planes, or planes weighted by the Step-1 work that source culling leaves in them "
                         "(shm_config.slab_plan; auto = step1 for the culled fp32 workloads, equal otherwise -- DESIGN.md section 5)")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend of the bootstrap/timing collectives (gloo + SHM_RCCL_LIB + SHM_BENCH_ONE_DEVICE=1 lets "
                         "several ranks share one GPU in tests)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d" % (args.gpus, args.gpus))
        args.gpus = world

    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a HIP device (there is no CPU fallback for the product path)")
    if os.environ.get("SHM_BENCH_ONE_DEVICE"):
        local_rank = 0
    torch.cuda.set_device(local_rank)

    import shm_import
    shm = shm_import.load()
    from signed_heat_3d_amd.host_abi import HostSolver

    uid = None
    if world > 1 or os.environ.get("SHM_BENCH_FORCE_DIST"):  # the env knob exercises the RCCL bootstrap with one rank
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if args.dist_backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend="gloo")
        box = [shm.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        uid = box[0]

    path, hCoef, precision = WORKLOADS[args.workload]
    host = HostSolver(os.path.join(ROOT, path))          # C++ host mirror: loaders + signed_heat_3d.cpp helpers
    pre = host.preprocess(hCoef=hCoef)                    # centroid/radius/h/areas/barycenters on the host (cheap)
    n, N = pre["n"], pre["n"] ** 3

    # auto: the weighted plan serves the default (gathered dual) solve of the culled fp32 workloads; the slab-distributed transforms (dual_slabs, primal + DCT)
    # need equal slabs, and the plain stencil CG is an explicit request for the north star's split -- those keep equal planes
    auto_weighted = precision == 32 and world > 1 and args.solver in ("auto", "dual") and args.precond == "auto"
    slab_plan = {"equal": 0, "step1": 1, "auto": 1 if auto_weighted else 0}[args.slab_plan]
    solver = shm.GridSolver(device=local_rank, precision=precision, rank=rank, world=world, rccl_unique_id=uid, slab_plan=slab_plan)
    solver.set_problem(pre["pos"], pre["wnormal"], pre["area"], pre["lam"], n, pre["bbox_min"], pre["cell"])  # inputs resident in HBM
    scrub = not path.endswith(".pc")

    def barrier():
        torch.cuda.synchronize()
        if dist.is_initialized():
            dist.barrier()
        torch.cuda.synchronize()

    st = None
    for _ in range(args.warmup):
        st = solver.solve(tol=args.tol, scrub=scrub, precond=args.precond, solver=args.solver, max_iters=args.max_iters, allow_noconv=args.max_iters > 0)
    barrier()
    t0 = time.perf_counter()
    stats = []
    for _ in range(args.steps):
        st = solver.solve(tol=args.tol, scrub=scrub, precond=args.precond, solver=args.solver, max_iters=args.max_iters, allow_noconv=args.max_iters > 0)
        stats.append(st.as_dict())
    barrier()
    elapsed = time.perf_counter() - t0
    if dist.is_initialized():
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if args.dist_backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    out = None
    if rank == 0:
        T = precision // 8
        avg = {k: float(np.mean([s[k] for s in stats])) for k in stats[0]}
        # per-kernel algorithmic bytes of the decomposition launched; per rank = per launch
        n_local = N / world
        has_pre = int(avg["preconditioner"]) == 2
        is_dual = int(avg["solver"]) in (2, 3)
        gathered = world > 1 and int(avg["solver"]) == 2   # every rank solved the whole grid after gathering D^T Y
        kernels, kinfo = kernel_table(avg, n_local, T, world, gathered)
        dominant = max(kernels, key=lambda k: kernels[k][1] * kernels[k][2])
        ach = kinfo[dominant]["achieved_GBps"] or 0.0
        traffic = None
        # HBM bytes per launch from the rocprofv3 PMC passes of THIS command (tools/pmc_traffic.py; FETCH_SIZE doubled per the gfx950
        # correction), committed per round; PMC counters cannot be collected from inside the run
        tfile = next((f for f in (os.path.join(ROOT, "profiles", "r06_pmc_traffic.json"), os.path.join(ROOT, "profiles", "r05_pmc_traffic.json"))
                      if os.path.exists(f)), "")
        if os.path.exists(tfile):
            try:
                traffic = json.load(open(tfile)).get(args.workload, {}).get(dominant)
            except Exception:
                traffic = None
        out = {
            "metric": "grid-nodes/sec end-to-end SHM (conv+PCG)", "value": N * args.steps / elapsed, "unit": "grid-nodes/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            # the arithmetic the path computes in.  fp64 handles: everything in fp64 except Step 1's far tier -- (node, source) pairs whose terms are below e^-8 of
            # their node block's dominant terms are summed in packed fp32 (step1.pairs_fp32; error budget on Y 1e-8, asserted against the C oracle at full
            # size in tests/); `also.exact_fp64` is the same workload with every pair in fp64
            "dtype": "f64 (Step 1: f64 / packed-f32 tiers)" if precision == 64 else "f32",
            "data": "reference data file %s (no RNG; sources + grid resident in HBM before the timed region)" % path,
            "config": {"workload": args.workload, "grid": "%d^3" % n, "sources": int(pre["S"]), "constraint_rows": int(avg["m"]),
                       "tol": args.tol if args.tol > 0 else (1e-8 if precision == 64 else 1e-5), "cg_iters": int(avg["iters"]),
                       "rel_residual": avg["rel_residual"], "partition": "z-slabs x%d%s" % (world, " (planes weighted by Step-1 work)" if slab_plan and world > 1 else ""),
                       "solver": (("dual, direct: the Schur complement S = A K^+ A^T is assembled explicitly (image-sum Green's table) and inverted beside Step 1; "
                                   "after Step 1: g = A K^+ b, mu = S^-1-solve of the bordered system, x = K^+ (A^T mu - b); cg_iters = passes (the first is "
                                   "the solve, further ones iterative refinement); K^+ = DCT fast Poisson solve"
                                   if int(avg.get("cg_form", 0)) == 2 else
                                   "dual: CG on the explicit Schur complement A K^+ A^T (dense mat-vec), K^+ = DCT fast Poisson solve"
                                   if int(avg.get("cg_form", 0)) == 3 else
                                   "dual: CG on the Schur complement A K^+ A^T, K^+ = DCT fast Poisson solve")
                                  + ("; Steps 1-2 on z-slabs, D^T Y gathered over RCCL, whole-grid solve on every rank" if gathered else
                                     "; z-slab DCT with two all-to-alls per application" if int(avg["solver"]) == 3 else "")) if is_dual
                       else "primal: projected stencil CG",
                       "preconditioner": (("none (direct)" if int(avg.get("cg_form", 0)) == 2 else "G^-1 (A K A^T) G^-1") if is_dual else "dct (exact fast Poisson, sandwiched P M^-1 P)") if has_pre else "none"},
            "phases_ms": {k: avg[k] for k in ("ms_conv", "ms_div", "ms_setup", "ms_wait_setup", "ms_pcg", "ms_shift", "ms_total")},
            "pcg": {"ms_per_iter": avg["ms_pcg"] / max(1.0, avg["iters"]), "algorithmic_bytes_per_iter": avg["bytes_per_iter"] / (1 if gathered else world),
                    "achieved_GBps": avg["bytes_per_iter"] / (1 if gathered else world) / (avg["ms_pcg"] / max(1.0, avg["iters"]) * 1e-3) / 1e9,
                    "frac_of_hbm_peak": avg["bytes_per_iter"] / (1 if gathered else world) / (avg["ms_pcg"] / max(1.0, avg["iters"]) * 1e-3) / 1e9 / HBM_PEAK_GBS,
                    "ms_project_avg": avg["ms_project_avg"]},
            "kernels": kinfo,
            # Step 1+2 is compute-bound on the vector ALU: 18 nominal flop per (node, source) pair against the fp64 / fp32 vector
            # peak.  Computed from the pairs the kernel actually EVALUATED (shm_stats.pairs_fp64 / pairs_fp32, counted by the kernel itself):
            # culled / dropped pairs earn nothing, and a pair evaluated in packed fp32 is priced against the fp32 peak.
            "step1": step1_roofline(avg, float(N) * float(pre["S"]) / world, precision),
            # the CG loop's dominant kernel against the HBM roofline (achieved = algorithmic bytes per launch / avg launch duration,
            # HIP events on the solver's stream)
            "roofline_pcg": {"kernel": dominant, "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": ach / HBM_PEAK_GBS, "traffic": traffic,
                             "note": "achieved = algorithmic bytes per launch / avg launch duration (HIP events on the solver stream, "
                                     "%d sampled launches per solve)" % int(avg["kernel_samples"])},
        }
        # `roofline` is for the kernel that dominates the step.  Step 1+2 takes > 90% of it at 256^3 and is bound by vector-ALU
        # issue (no GEMM shape, HBM traffic 3 words per node), so its roofline is the fp64/fp32 vector peak at the
        # nominal 18 flop per EVALUATED pair; when the CG loop takes longer than Step 1 the HBM roofline of its dominant kernel is reported
        s1 = out["step1"]
        conv_traffic = None   # PMC-measured HBM bytes of Step 1 per step (all its launches), profiles/r0N_pmc_traffic.json (latest round)
        tfile3 = tfile
        if os.path.exists(tfile3):
            try:
                conv_traffic = json.load(open(tfile3)).get(args.workload, {}).get("step1_bytes_per_step")
            except Exception:
                conv_traffic = None
        if avg["ms_conv"] >= avg["ms_pcg"]:
            # the kernel is bound by vector-ALU instruction issue -- neither of the contract's two labels ("hbm", "mfma") describes it, so it
            # is called what it is; its peak is the VECTOR peak (78.6 fp64 / 157.3 fp32 TFLOP/s on MI355X), weighted by the tiers' shares
            out["roofline"] = {"kernel": s1["kernel"], "bound": "valu",
                               "bound_detail": "compute: vector-ALU issue (no matrix-core shape in the kernel, no MFMA instruction); peak = the vector peaks of the two "
                                               "arithmetic tiers (78.6 TFLOP/s fp64, 157.3 TFLOP/s fp32) weighted by the pairs each tier evaluated; SQ counters of this "
                                               "kernel: profiles/r06_sq_counters_conv.txt",
                               "achieved": s1["achieved_TFLOPs_18_per_evaluated_pair"], "peak": s1["peak_TFLOPs_weighted"], "unit": "TFLOP/s", "frac": s1["frac"],
                               "traffic": conv_traffic,
                               "note": "%d launch(es) per step, duration = phases_ms.ms_conv (HIP events on the solver's stream around all of them); achieved = 18 nominal "
                                       "flop (SURVEY 8(d)) x pairs EVALUATED / duration, pairs counted by the kernel per arithmetic tier (step1.pairs_*); "
                                       "frac = (18 pairs_fp64 / 78.6e12 + 18 pairs_fp32 / 157.3e12) / duration; traffic = PMC-measured HBM bytes of Step 1 per step, "
                                       "summed over its launches (3 words per node written; irrelevant to the bound)" % s1["launches_per_step"]}
        else:
            out["roofline"] = dict(out["roofline_pcg"])
        if not args.no_cpu_baseline and world == 1:  # rank 0 at N=1 only (the extra plain-CG solve below is a collective at N>1)
            try:
                st_plain = solver.solve(tol=args.tol, scrub=scrub, solver="primal", precond="none")  # untimed: iteration count of the port's algorithm
                out["cpu_baseline"] = cpu_baseline(pre, int(st_plain.iters), out["config"]["tol"])
                out["cpu_baseline"]["host_cores_available"] = os.cpu_count()
                # the reference is single-threaded; an OpenMP build of the same port on many cores is reported alongside
                nthr = min(os.cpu_count() or 1, 64)
                if nthr > 1:
                    allc = cpu_baseline(pre, int(st_plain.iters), out["config"]["tol"], seconds_budget=8.0, threads=nthr)
                    out["cpu_baseline"]["openmp"] = {"value": allc["value"], "cores": nthr, "seconds_extrapolated": allc["seconds_extrapolated"]}
            except Exception as e:  # the baseline is informational; never lose the GPU line over it
                out["cpu_baseline"] = {"value": None, "unit": "grid-nodes/s", "cores": 1, "kind": "port", "sample": "failed: %r" % (e,)}
        out["roofline"]["traffic_source"] = "committed record %s (rocprofv3 PMC passes of tools/profile_r06.sh)" % os.path.relpath(tfile, ROOT) if out["roofline"].get("traffic") else None
        if world == 1 and not args.no_also and not args.no_live_traffic and out["roofline"].get("bound") == "valu":
            live, how = live_step1_traffic(path, hCoef, precision)
            if live is not None:
                out["roofline"]["traffic_committed_record"] = out["roofline"].get("traffic")
                out["roofline"]["traffic"] = live
                out["roofline"]["traffic_source"] = how
            else:
                out["roofline"]["traffic_live_attempt"] = how
        if world == 1 and not args.no_also and args.workload == "bunny_small_256_f64":
            try:
                solver.close()
                out["also"] = also_legs(shm, HostSolver, local_rank, args.tol, pre, scrub)
            except Exception as e:  # never lose the headline over an extra leg
                out["also"] = {"failed": repr(e)}
    if world > 1 and not args.no_also and args.solver == "auto" and args.precond == "auto" and args.max_iters == 0:
        # The extra multi-rank legs run AFTER the headline record is complete, under a watchdog: should one of them hang on a transport this build has never
        # run on (real multi-rank RCCL: the development pool has one-GPU boxes), every rank gives up after SHM_BENCH_LEGS_TIMEOUT seconds, rank 0 prints the
        # line without them, and the processes exit -- the timed result above is never lost to an extra.
        import threading
        legs_done = threading.Event()
        line_lock = threading.Lock()   # the record is printed exactly once: by the watchdog (legs hung) or by the main thread (legs done), whoever takes the lock first
        line_printed = [False]

        def watchdog():
            #  budgeted per leg: 60 s for configs[3], 180 s for configs[4] (host pre-processing of 52 290 points, a 1024^3 set-up and two solves per rank), 40 s for
            # each of the three legs on the timed workload's grid (scaled with the grid beyond 256^3); SHM_BENCH_LEGS_TIMEOUT overrides the sum.  The legs that finished
            # before a hang are printed with the headline (multi_gpu_legs fills `multi_partial` leg by leg)
            budget = 60.0 + 180.0 + 3 * 40.0 * max(1.0, (pre["n"] / 256.0) ** 3)
            if not legs_done.wait(float(os.environ.get("SHM_BENCH_LEGS_TIMEOUT", budget))):
                with line_lock:
                    if line_printed[0]:
                        return
                    line_printed[0] = True
                    if rank == 0:
                        headline = dict(out)   # (the main thread only ever ADDS "also_multi" to `out`, after the legs: this copy cannot see a half-written record)
                        headline["also_multi"] = dict(multi_partial)
                        headline["also_multi"]["failed"] = "timed out after SHM_BENCH_LEGS_TIMEOUT; the headline record is complete, the legs listed here finished before the hang; exit status 3"
                        print(json.dumps(headline), flush=True)
                    os._exit(3)        # a hang in an extra leg is not a green run: the headline line is on stdout, the launcher sees a failing status

        multi_partial = {}
        threading.Thread(target=watchdog, daemon=True).start()
        solver.close()   # (its communicator and whole-grid arrays go first: the legs build their own solver on a fresh communicator)
        multi = multi_gpu_legs(shm, HostSolver, dist, torch, args, pre, precision, scrub, rank, world, local_rank, barrier, multi_partial)
        with line_lock:
            if line_printed[0]:       # the watchdog got there first and is taking the process down
                return
            line_printed[0] = True
            legs_done.set()
            if rank == 0:
                out["also_multi"] = multi
                print(json.dumps(out), flush=True)
    elif rank == 0:
        print(json.dumps(out), flush=True)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

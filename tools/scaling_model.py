#!/usr/bin/env python3
"""Predicted strong-scaling curve (1, 2, 4, 8 GPUs of one node) of a bench.py workload from its measured single-GPU phases.
NO HARDWARE CURVE EXISTS YET (the development pool has one-GPU boxes; the driver's 8-GPU leg has not run): this is the cost model
DESIGN.md section 5 quotes, so that the solver choice per (n, m, P) is an argument and not a guess.

    python tools/scaling_model.py profiles/r03_bench_default.json [profiles/r03_bench_256_primal_plain.json]

Model, per solve on P ranks (z-slabs; constants at the top of the file, each marked MEASURED (with its profiles/ file) or ASSUMED):
  gathered dual (AUTO):  t = imb(P) (conv + div)/P + exposed_setup + gather + pcg + shift
        Steps 1-2 and the divergence shard up to the measured slab imbalance imb(P) (tools/slab_plan_check.py); the constraint set-up (replicated) runs
        beside Step 1 -- co-resident with the tiered fp64 kernel, where it takes `beside` ms, time-sliced to Step 1's end against the fp32 one -- and what is
        left of it when Step 1 ends runs at its idle-GPU rate: exposed = max(0, 1 - (conv/P)/beside) * alone; the right-hand side is all-gathered (every rank
        receives (P-1)/P of one N-vector over its xGMI links); the m-dimensional dual solve is replicated (latency-bound: slicing it buys nothing).
  slab-distributed explicit-S dual (round 6; AUTO for 256^3 ... 512^3 with S <= 16384):
                         t = imb(P) conv/P + div/P + exposed_setup + [replicated m-sized part + n_sized/P + 4 all-to-alls + 3 small all-reduces] + shift
        S and S^-1 are replicated beside every rank's Step 1 (the same exposed_setup as above); the solve applies K^+ twice on the slabs -- the transform sweeps shard, every
        application pays two all-to-alls in which a rank sends (P-1)/P of its slab -- and solves the dual system on m-vectors that every rank holds (direct: two dense
        mat-vecs; CG on the explicit S: the iterations, replicated).  No gather of D^T Y, no whole-grid K^+ per rank.
  primal stencil CG:     t = (conv + div)/P + max(0, setup - conv/P) + iters * (sweeps/P + halo + 2 allreduce + project) + shift
        the N-sized sweeps shard, the m-sized projection is replicated, one ghost plane of z per neighbour and two all-reduces per iteration.
"""
import json
import sys

XGMI_LINK_GBS = 50.0     # ASSUMED: sustained per link and direction for large send/recv (153 GB/s bidirectional peak per link, 7 links per GPU)
LINKS = 7                # hardware: xGMI links per GPU
P2P_LAT_US = 15.0        # ASSUMED: one grouped ncclSend/ncclRecv pair on an idle stream
ALLREDUCE_LAT_US = 20.0  # ASSUMED: small (<= 400 KB) all-reduce over 8 ranks
SETUP_GJ_US_PER_STEP = 40.0  # MEASURED (profiles/r04_gj_step.txt, idle-GPU launches of gj_step_kernel: one launch per 64-row pivot block; rounds 1-3: 80 us in three launches)
# MEASURED (profiles/r06_setup_alone.txt, tools/setup_alone.py): constraint set-up wall time in ms, (alone on an idle GPU, beside Step 1), by constraint rows m.
# The two 512^3 "beside" figures are round 3's (raised wave priority, which is what a rank with an eighth of Step 1 runs with: Solver::setup_prio) scaled by the
# ratio of the "alone" times; the round-4 file has them at the low priority a 200 ms Step 1 selects (58.7 / 51.0 ms, hidden all the same).
SETUP_MS = {1129: (1.17, 1.51), 2496: (2.73, 3.47), 2842: (3.95, 6.48), 2856: (9.32, 33.30), 1430: (5.45, 25.07), 12612: (36.27, 51.56), 48893: (145.0, None)}   # round 6: profiles/r06_setup_alone.txt (SprayBottle: round 4)
# MEASURED (profiles/r03_slab_plan_check*.txt, tools/slab_plan_check.py): max / mean of the slabs' own Step-1 times, by workload and slab count;
# fp32 culled workloads with the weighted plan (shm_config.slab_plan = SHM_SLAB_PLAN_STEP1), the others with equal planes
IMBALANCE = {"bunny_small_256_f64": {4: 1.04, 8: 1.09}, "bunny_small_512_f64": {4: 1.03, 8: 1.10}, "bunny_pc_512_f64": {4: 1.05, 8: 1.12},
             "rocker_512_f32": {4: 1.027, 8: 1.069}, "spraybottle_pc_1024_f32": {4: 1.016, 8: 1.014}}


def setup_alone_ms(m, host_ms):
    nb = (m + 63) // 64
    return host_ms + nb * SETUP_GJ_US_PER_STEP * 1e-3


def main():
    text = open(sys.argv[1]).read().strip()
    try:
        d = json.loads(text)                      # profiles/*.json (pretty-printed)
    except ValueError:
        d = json.loads(text.splitlines()[-1])     # raw bench.py output: the JSON line is the last one
    ph, cfg = d["phases_ms"], d["config"]
    n = int(cfg["grid"].split("^")[0])
    T = 8 if str(d["dtype"]).startswith("f64") else 4
    N = n ** 3
    m = cfg["constraint_rows"]
    host_ms = 0.6e-3 * m + 0.3
    alone, beside = SETUP_MS.get(m, (setup_alone_ms(m, host_ms) if m <= 6144 else 3.0e-3 * m, None))
    setup = alone
    imb = IMBALANCE.get(cfg["workload"], {})
    print("workload %s  n=%d  m=%d  single-GPU: conv %.1f  pcg %.2f ms (%d iterations)  set-up alone %.1f ms, beside Step 1 %s" % (
        cfg["workload"], n, m, ph["ms_conv"], ph["ms_pcg"], cfg["cg_iters"], alone, ("%.1f ms" % beside) if beside else "time-sliced to Step 1's end"))
    print("%-28s %10s %10s %10s %10s" % ("gathered dual (AUTO)", "P=1", "P=2", "P=4", "P=8"))
    rows = {"ms_per_solve": [], "speedup": [], "nodes_per_s": []}
    t1 = None
    for P in (1, 2, 4, 8):
        gather = 0.0 if P == 1 else (N * T * (P - 1) / P) / (min(P - 1, LINKS) * XGMI_LINK_GBS * 1e9) * 1e3 + P2P_LAT_US * 1e-3
        conv_p = ph["ms_conv"] / P * (imb.get(P, 1.0) if P > 1 else 1.0)
        if beside:   # co-resident: progresses at alone/beside of its idle rate while Step 1 runs, at full rate afterwards
            exposed = max(0.0, 1.0 - conv_p / beside) * alone
        else:        # time-sliced: the host part of the set-up runs under Step 1 (staged uploads, round 3), its kernels queue up behind Step 1 and run when it
            # ends -- what one GPU shows as ms_wait_setup; a slab's Step 1 shorter than the whole set-up leaves the rest exposed as well
            exposed = max(ph["ms_wait_setup"], alone - conv_p)
        t = conv_p + ph["ms_div"] / P + exposed + gather + ph["ms_pcg"] + ph["ms_shift"]
        t1 = t1 or t
        rows["ms_per_solve"].append(t)
        rows["speedup"].append(t1 / t)
        rows["nodes_per_s"].append(N / (t * 1e-3))
    for k, v in rows.items():
        print("%-28s " % k + " ".join("%10.3g" % x for x in v))
    # ---- slab-distributed explicit-S dual (round 6)
    if m <= 16384 and n <= 512:
        # N-sized part of the one-GPU solve phase: two dense K^+ applications (10 N T bytes each at the measured transform rate: 3.7 TB/s at 512^3, 5.1 TB/s at 256^3,
        # profiles/r03_bench_{512,256}_primal_dct.json); what is left of the measured phase is m-sized (mat-vecs, CG iterations on S) and stays replicated
        app_ms = 10.0 * N * T / ((3.7e12 if n >= 512 else 5.1e12)) * 1e3
        n_sized = min(ph["ms_pcg"], 2.0 * app_ms)
        replicated = ph["ms_pcg"] - n_sized
        print("%-28s %10s %10s %10s %10s   (solve phase on one GPU %.2f ms = %.2f N-sized + %.2f m-sized)" % ("slab-distributed explicit S", "P=1", "P=2", "P=4", "P=8", ph["ms_pcg"], n_sized, replicated))
        rows = {"ms_per_solve": [], "speedup": []}
        t1 = None
        for P in (1, 2, 4, 8):
            conv_p = ph["ms_conv"] / P * (imb.get(P, 1.0) if P > 1 else 1.0)
            exposed = (max(0.0, 1.0 - conv_p / beside) * alone) if beside else max(ph["ms_wait_setup"], alone - conv_p)
            a2a = 0.0 if P == 1 else (N * T * (P - 1) / (P * P)) / (min(P - 1, LINKS) * XGMI_LINK_GBS * 1e9) * 1e3 + P2P_LAT_US * 1e-3
            small = 0.0 if P == 1 else 3 * ALLREDUCE_LAT_US * 1e-3
            t = conv_p + ph["ms_div"] / P + exposed + replicated + n_sized / P + 4 * a2a + small + ph["ms_shift"]
            t1 = t1 or t
            rows["ms_per_solve"].append(t)
            rows["speedup"].append(t1 / t)
        for k, v in rows.items():
            print("%-28s " % k + " ".join("%10.3g" % x for x in v))
    if len(sys.argv) > 2:
        p = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
        pp, it = p["phases_ms"], p["config"]["cg_iters"]
        sweeps = sum(v["avg_ms_per_launch"] * v["launches_per_iter"] for v in p["kernels"].values())
        proj = p["pcg"]["ms_project_avg"]
        other = max(0.0, pp["ms_pcg"] / it - sweeps - proj)
        print("%-28s %10s %10s %10s %10s   (%d iterations; sweeps %.3f + projection %.3f + rest %.3f ms per iteration on one GPU)" % (
            "primal stencil CG (slabs)", "P=1", "P=2", "P=4", "P=8", it, sweeps, proj, other))
        vals = []
        for P in (1, 2, 4, 8):
            comm = 0.0 if P == 1 else (n * n * T / (XGMI_LINK_GBS * 1e9) * 1e3 + P2P_LAT_US * 1e-3 + 2 * ALLREDUCE_LAT_US * 1e-3)
            t = (pp["ms_conv"] + pp["ms_div"]) / P + max(0.0, setup - pp["ms_conv"] / P) + it * (sweeps / P + proj + other + comm) + pp["ms_shift"]
            vals.append(t)
        print("%-28s " % "ms_per_solve" + " ".join("%10.3g" % x for x in vals))


if __name__ == "__main__":
    main()

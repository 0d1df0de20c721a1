#!/usr/bin/env python3
"""Predicted strong-scaling curve (1, 2, 4, 8 GPUs of one node) of a bench.py workload from its measured single-GPU phases.
NO HARDWARE CURVE EXISTS YET (the development pool has one-GPU boxes; the driver's 8-GPU leg has not run): this is the cost model
DESIGN.md section 5 quotes, so that the solver choice per (n, m, P) is an argument and not a guess.

    python tools/scaling_model.py profiles/r02_bench_default.json [profiles/r02_bench_256_primal_plain.json]

Model, per solve on P ranks (z-slabs; constants at the top of the file):
  gathered dual (AUTO):  t = (conv + div)/P + max(0, setup - conv/P) + gather + pcg + shift
        Steps 1-2 and the divergence shard perfectly; the constraint set-up (host + Gauss-Jordan, replicated) hides behind Step 1 until Step 1
        gets shorter than it; the right-hand side is all-gathered (every rank receives (P-1)/P of one N-vector over its xGMI links); the
        m-dimensional dual iteration is replicated (latency-bound: slicing it buys nothing).
  primal stencil CG:     t = (conv + div)/P + max(0, setup - conv/P) + iters * (sweeps/P + halo + 2 allreduce + project) + shift
        the N-sized sweeps shard, the m-sized projection is replicated, one ghost plane of z per neighbour and two all-reduces per iteration.
"""
import json
import sys

XGMI_LINK_GBS = 50.0     # sustained per link and direction for large send/recv (153 GB/s bidirectional peak per link, 7 links per GPU)
LINKS = 7
P2P_LAT_US = 15.0        # one grouped ncclSend/ncclRecv pair on an idle stream
ALLREDUCE_LAT_US = 20.0  # small (<= 400 KB) all-reduce over 8 ranks
SETUP_GJ_US_PER_STEP = 45.0  # three launch-bound kernels per 64-row pivot block (dense inverse, m <= 6144)


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
    T = 8 if d["dtype"] == "f64" else 4
    N = n ** 3
    m = cfg["constraint_rows"]
    host_ms = 0.6e-3 * m + 0.3
    # set-up on an otherwise idle GPU, measured with SHM_SETUP_ALONE=1 (MI355X).  Direct dual solve (m <= 4096: S assembled and inverted): m = 1129: 2.3 ms,
    # 1430: 4.4, 2496: 5.6, 2842: 8.4;  through the grid + two-level G^-1: 26 ms at m = 12 612,
    # 145 ms at m = 48 893; in between by the launch-bound Gauss-Jordan step count
    measured = {1129: 2.3, 1430: 4.4, 2496: 5.6, 2842: 8.4, 2856: 8.4, 12612: 26.0, 48893: 145.0}
    setup = measured.get(m, setup_alone_ms(m, host_ms) if m <= 6144 else 3.0e-3 * m)
    print("workload %s  n=%d  m=%d  single-GPU: conv %.1f  pcg %.2f ms (%d iterations)  set-up alone ~%.1f ms" % (cfg["workload"], n, m, ph["ms_conv"], ph["ms_pcg"], cfg["cg_iters"], setup))
    print("%-28s %10s %10s %10s %10s" % ("gathered dual (AUTO)", "P=1", "P=2", "P=4", "P=8"))
    rows = {"ms_per_solve": [], "speedup": [], "nodes_per_s": []}
    t1 = None
    for P in (1, 2, 4, 8):
        gather = 0.0 if P == 1 else (N * T * (P - 1) / P) / (min(P - 1, LINKS) * XGMI_LINK_GBS * 1e9) * 1e3 + P2P_LAT_US * 1e-3
        t = (ph["ms_conv"] + ph["ms_div"]) / P + max(0.0, setup - ph["ms_conv"] / P) + gather + ph["ms_pcg"] + ph["ms_shift"]
        t1 = t1 or t
        rows["ms_per_solve"].append(t)
        rows["speedup"].append(t1 / t)
        rows["nodes_per_s"].append(N / (t * 1e-3))
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

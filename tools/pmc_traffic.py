"""Build profiles/rNN_pmc_traffic.json from the separate FETCH_SIZE / WRITE_SIZE passes of tools/profile_r01.sh.
HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: on gfx950 FETCH_SIZE tallies 128-byte requests at 64 bytes
(MI355X_MICROARCH.md, HBM / rocprofv3 section), both counters are in KB."""
import json
import re
import sqlite3
import sys
from collections import defaultdict


def per_kernel(db, counter):
    c = sqlite3.connect(db)
    cols = [r[1] for r in c.execute("pragma table_info(counters_collection)").fetchall()]
    kn = "kernel_name" if "kernel_name" in cols else "name"
    # one row per (dispatch, counter instance): sum the instances (XCDs / channels) of a dispatch, then average over dispatches
    did = "dispatch_id" if "dispatch_id" in cols else "id"
    rows = c.execute("select %s, %s, sum(value) from counters_collection where counter_name = ? group by %s, %s" % (kn, did, kn, did), (counter,)).fetchall()
    acc = defaultdict(list)
    for k, _, v in rows:
        acc[k].append(v)
    return {k: (sum(v) / len(v), len(v)) for k, v in acc.items()}


def short(name):
    name = re.sub(r"^void ", "", name)
    name = name.split("(")[0]
    return name.replace("shm::", "")


def build(fetch_db, write_db, n_nodes, tbytes):
    f = per_kernel(fetch_db, "FETCH_SIZE")
    w = per_kernel(write_db, "WRITE_SIZE")
    out = {}
    for k in sorted(set(f) | set(w)):
        fk, nf = f.get(k, (0.0, 0))
        wk, nw = w.get(k, (0.0, 0))
        b = (2.0 * fk + wk) * 1024.0
        out[short(k)] = {"launches": max(nf, nw), "FETCH_SIZE_KB_raw": fk, "WRITE_SIZE_KB_raw": wk, "hbm_bytes_per_launch_corrected": b,
                         "in_units_of_N_T": b / (n_nodes * tbytes)}
    return out


def family(per, prefix):
    tot = sum(v["hbm_bytes_per_launch_corrected"] * v["launches"] for k, v in per.items() if k.startswith(prefix))
    cnt = sum(v["launches"] for k, v in per.items() if k.startswith(prefix))
    return tot / cnt if cnt else None


if __name__ == "__main__":
    d, out = sys.argv[1], sys.argv[2]
    N, T = 256 ** 3, 8
    dual = build(d + "/pmc_fetch_results.db", d + "/pmc_write_results.db", N, T)
    primal = build(d + "/pmc_fetch_primal_results.db", d + "/pmc_write_primal_results.db", N, T)
    res = {
        "_note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes over `bench.py --steps 1 --warmup 0 [--solver primal]` "
                 "(bunny_small 256^3 fp64); hbm bytes per launch = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950 correction of MI355X_MICROARCH.md: "
                 "FETCH_SIZE tallies 128-B requests at 64 B).  dct_lines_kernel = average over the five sweeps of an iteration "
                 "(dual solver: sparse sweeps; the two dense solves at start and end are in the average too).",
        "per_kernel_dual": dual,
        "per_kernel_primal": primal,
        "bunny_small_256_f64": {
            "dct_lines_kernel": family(dual, "dct_lines_kernel"),
            "conv_normalize_kernel": family(dual, "conv_normalize_kernel"),
            "stencil_dot_kernel": family(primal, "stencil_dot_kernel"),
            "update_xr_kernel": family(primal, "update_xr_kernel"),
            "update_p_kernel": family(primal, "update_p_kernel"),
        },
    }
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res["bunny_small_256_f64"], indent=1))

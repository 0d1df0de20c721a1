"""Post-process the PMC passes of tools/profile_r02.sh into profiles-ready files:
  r02_pmc_traffic.json   HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 per kernel (gfx950: FETCH_SIZE tallies 128-byte requests at 64
                         bytes -- MI355X_MICROARCH.md, HBM section; both counters in KB), for the 256^3 default bench and the 512^3 stencil-PCG leg
  r02_sq_counters_conv.txt   SQ / GRBM counters of conv_normalize_kernel (VALU issue, LDS bank conflicts, effective clock)"""
import json
import re
import sqlite3
import sys
from collections import defaultdict


def per_kernel(db, counter):
    c = sqlite3.connect(db)
    cols = [r[1] for r in c.execute("pragma table_info(counters_collection)").fetchall()]
    kn = "kernel_name" if "kernel_name" in cols else "name"
    did = "dispatch_id" if "dispatch_id" in cols else "id"
    rows = c.execute("select %s, %s, sum(value) from counters_collection where counter_name = ? group by %s, %s" % (kn, did, kn, did), (counter,)).fetchall()
    acc = defaultdict(list)
    for k, _, v in rows:
        acc[k].append(v)
    return {k: (sum(v) / len(v), len(v)) for k, v in acc.items()}


def durations(db):
    c = sqlite3.connect(db)
    return {r[0]: (r[1], r[2]) for r in c.execute("select name, avg(duration), count(*) from kernels group by name").fetchall()}


def short(name):
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*$", "", name)
    return name.replace("shm::", "")


def traffic(fetch_db, write_db, n_nodes, tbytes):
    f, w = per_kernel(fetch_db, "FETCH_SIZE"), per_kernel(write_db, "WRITE_SIZE")
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
    t256 = traffic(d + "/pmc_fetch_results.db", d + "/pmc_write_results.db", 256 ** 3, 8)
    t512 = traffic(d + "/pmc_fetch_pcg512_results.db", d + "/pmc_write_pcg512_results.db", 512 ** 3, 8)
    res = {
        "_note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes (tools/profile_r02.sh); hbm bytes per launch = (2*FETCH_SIZE + WRITE_SIZE)*1024 "
                 "(gfx950 correction of MI355X_MICROARCH.md: FETCH_SIZE tallies 128-B requests at 64 B; Infinity-Cache hits are counted as fetches).  "
                 "per_kernel_256 = `bench.py --steps 1 --warmup 0` (bunny_small 256^3 fp64, dual solver); per_kernel_pcg512 = `bench.py --workload "
                 "bunny_small_512_f64 --solver primal --precond none --max-iters 12` (the fused stencil-PCG sweeps; N T = 512^3 * 8 bytes).",
        "per_kernel_256": t256, "per_kernel_pcg512": t512,
        "bunny_small_256_f64": {"conv_normalize_kernel": family(t256, "conv_normalize_kernel"), "dct_lines_kernel": family(t256, "dct_lines_kernel"),
                                "zsolve_sparse_kernel": family(t256, "zsolve_sparse_kernel")},
        "bunny_small_512_f64": {k: family(t512, k) for k in ("cg_fused_kernel", "cg_x_update2_kernel", "conv_normalize_kernel")},
    }
    for k, v in t512.items():
        if k.startswith("cg_fused_kernel"):
            res["bunny_small_512_f64"][k] = v["hbm_bytes_per_launch_corrected"]
    json.dump(res, open(out + "/r02_pmc_traffic.json", "w"), indent=1)
    # ---- SQ counters of the Step-1 kernel
    lines = ["SQ / GRBM counters of conv_normalize_kernel<double, 4> in `bench.py --steps 1 --warmup 0` (bunny_small 256^3 fp64; 3 launches: solve + 2 of the set-up of",
             "the untimed legs are not in this run).  Two passes (SQ has 8 counter slots).  Sums over all SEs / XCDs of a dispatch, averaged over dispatches.", ""]
    vals = {}
    for db, names in ((d + "/pmc_sq1_results.db", ["SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAVES", "GRBM_GUI_ACTIVE"]),
                      (d + "/pmc_sq2_results.db", ["SQ_INSTS_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_ACTIVE_INST_LDS", "SQ_INSTS_SALU", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY"])):
        try:
            dur = durations(db)
            for nm in names:
                pk = per_kernel(db, nm)
                for k, (v, cnt) in pk.items():
                    if "conv_normalize_kernel" in k:
                        vals[nm] = v
                        vals["_launches"] = cnt
                        vals["_kernel"] = short(k)
                        vals.setdefault("_dur_ns_" + db.split("/")[-1], dur.get(k, (0, 0))[0])
        except Exception as e:
            lines.append("(%s: %r)" % (db, e))
    for k, v in vals.items():
        lines.append("%-28s %s" % (k, v))
    pairs = 256.0 ** 3 * 2856
    if "SQ_INSTS_VALU" in vals:
        lines += ["", "derived:",
                  "  VALU wave-instructions per (node, source) pair = SQ_INSTS_VALU * 64 / pairs = %.2f  (pairs = 256^3 * 2856; includes the per-tile set-up and the normalisation)" % (vals["SQ_INSTS_VALU"] * 64 / pairs)]
    if "GRBM_GUI_ACTIVE" in vals:
        for k in vals:
            if k.startswith("_dur_ns_pmc_sq1"):
                lines.append("  effective shader clock = GRBM_GUI_ACTIVE / 8 XCD instances / duration = %.3f GHz (profiled pass)" % (vals["GRBM_GUI_ACTIVE"] / 8.0 / vals[k]))
    if "SQ_LDS_BANK_CONFLICT" in vals and "SQ_LDS_IDX_ACTIVE" in vals:
        lines.append("  LDS bank-conflict cycles / LDS active cycles = %.3f ; per LDS instruction = %.3f" % (vals["SQ_LDS_BANK_CONFLICT"] / max(vals["SQ_LDS_IDX_ACTIVE"], 1), vals["SQ_LDS_BANK_CONFLICT"] / max(vals.get("SQ_INSTS_LDS", 1), 1)))
    if "SQ_ACTIVE_INST_VALU" in vals and "SQ_WAVE_CYCLES" in vals:
        lines.append("  SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES = %.3f ; SQ_ACTIVE_INST_VALU*4 / SQ_BUSY_CYCLES = %.3f" % (vals["SQ_ACTIVE_INST_VALU"] / vals["SQ_WAVE_CYCLES"], 4 * vals["SQ_ACTIVE_INST_VALU"] / max(vals.get("SQ_BUSY_CYCLES", 1), 1)))
    open(out + "/r02_sq_counters_conv.txt", "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))
    print(json.dumps(res["bunny_small_512_f64"], indent=1))

"""Post-process the PMC passes of tools/profile_r05.sh into profiles-ready files.  Everything is accounted PER STEP (= per solve): a counter is summed over
all launches of a kernel inside one solve, and the launch count is printed beside it.
  r05_pmc_traffic.json      HBM bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950: FETCH_SIZE tallies 128-byte requests at 64 bytes -- MI355X_MICROARCH.md,
                            HBM section; both counters in KB) per kernel: per launch, launches per step, per step
  r05_sq_counters_conv.txt  SQ / GRBM counters of the Step-1 kernel of one solve (VALU issue, LDS, effective clock), with the executed pairs of that solve
    python tools/pmc_report.py <dir with the *_results.db> <out dir> [executed_pairs_fp64 executed_pairs_fp32]"""
import json
import os
import re
import sqlite3
import sys
from collections import defaultdict


TAG = os.environ.get("SHM_PROFILE_TAG", "r06")   # round tag of the files written (profiles/<tag>_pmc_traffic.json, <tag>_sq_counters_conv.txt)


def per_kernel(db, counter):
    """kernel -> (sum over the dispatches of the run, number of dispatches).  The runs are `--steps 1 --warmup 0`: one solve."""
    c = sqlite3.connect(db)
    cols = [r[1] for r in c.execute("pragma table_info(counters_collection)").fetchall()]
    kn = "kernel_name" if "kernel_name" in cols else "name"
    did = "dispatch_id" if "dispatch_id" in cols else "id"
    rows = c.execute("select %s, %s, sum(value) from counters_collection where counter_name = ? group by %s, %s" % (kn, did, kn, did), (counter,)).fetchall()
    acc = defaultdict(list)
    for k, _, v in rows:
        acc[k].append(v)
    return {k: (sum(v), len(v)) for k, v in acc.items()}


def durations(db):
    c = sqlite3.connect(db)
    return {r[0]: (r[1], r[2]) for r in c.execute("select name, sum(duration), count(*) from kernels group by name").fetchall()}


def short(name):
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*$", "", name)
    return name.replace("shm::", "")


def traffic(fetch_db, write_db, n_nodes, tbytes, iters=1):
    f, w = per_kernel(fetch_db, "FETCH_SIZE"), per_kernel(write_db, "WRITE_SIZE")
    out = {}
    for k in sorted(set(f) | set(w)):
        fk, nf = f.get(k, (0.0, 0))
        wk, nw = w.get(k, (0.0, 0))
        n = max(nf, nw, 1)
        b = (2.0 * fk + wk) * 1024.0
        out[short(k)] = {"launches_per_step": n, "FETCH_SIZE_KB_raw_per_step": fk, "WRITE_SIZE_KB_raw_per_step": wk, "hbm_bytes_per_step_corrected": b,
                         "hbm_bytes_per_launch_corrected": b / n, "per_launch_in_units_of_N_T": b / n / (n_nodes * tbytes),
                         "per_step_in_units_of_N_T": b / (n_nodes * tbytes)}
    return out


def fam(per, prefix, key):
    return sum(v[key] for k, v in per.items() if k.startswith(prefix)) or None


if __name__ == "__main__":
    d, out = sys.argv[1], sys.argv[2]
    t256 = traffic(d + "/pmc_fetch_results.db", d + "/pmc_write_results.db", 256 ** 3, 8)
    res = {
        "_note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes (tools/profile_r05.sh), each over ONE solve (`bench.py --steps 1 --warmup 0`); "
                 "hbm bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950 correction of MI355X_MICROARCH.md: FETCH_SIZE tallies 128-B requests at 64 B; Infinity-Cache hits "
                 "are counted as fetches).  Every figure is per STEP (sum over the kernel's launches in the solve), with the launch count beside it.  "
                 "per_kernel_256 = bunny_small 256^3 fp64, dual solver; per_kernel_pcg512 / _f32 = `--workload bunny_small_512_f64|f32 --solver primal --precond none "
                 "--max-iters 12` (the fused stencil-PCG sweeps; N T = 512^3 * 8 or 4 bytes; per launch = per step / launches).",
        "per_kernel_256": t256,
    }
    conv = {k: v for k, v in t256.items() if k.startswith("conv_")}
    # Step 1's own traffic comes from the passes over tools/conv_only.py (2 x shm_grid_run_conv, nothing else on the device): inside a solve the set-up kernels
    # run on the same SIMDs at the same time, and the device-wide counters sampled around the Step-1 dispatch contain their bytes too
    alone = {}
    try:
        ta = traffic(d + "/pmc_fetch_conv_results.db", d + "/pmc_write_conv_results.db", 256 ** 3, 8)
        alone = {k: v for k, v in ta.items() if k.startswith("conv_")}
        res["per_kernel_256_step1_alone"] = alone
    except Exception as e:
        res["per_kernel_256_step1_alone"] = {"failed": repr(e)}
    runs = 2.0   # conv_only.py launches Step 1 twice
    res["bunny_small_256_f64"] = {
        "step1_kernel": ", ".join(conv) or None,
        "step1_launches_per_step": sum(v["launches_per_step"] for v in conv.values()) or None,
        "step1_bytes_per_step": (sum(v["hbm_bytes_per_step_corrected"] for v in alone.values()) / runs) if alone else None,
        "step1_bytes_per_step_in_units_of_N_T": (sum(v["per_step_in_units_of_N_T"] for v in alone.values()) / runs) if alone else None,
        "step1_algorithmic_bytes_per_step": 3.0 * 256 ** 3 * 8,
        "step1_bytes_per_step_inside_a_solve_with_the_setup_coresident": fam(t256, "conv_", "hbm_bytes_per_step_corrected"),
        "_step1_note": "step1_bytes_per_step: kernel alone (tools/conv_only.py); the in-solve figure includes the bytes of the set-up kernels that run meanwhile",
    }
    try:   # the 512^3 default solve: divergence kernel, dense transform sweeps (N T = 512^3 * 8)
        t512 = traffic(d + "/pmc_fetch_512_results.db", d + "/pmc_write_512_results.db", 512 ** 3, 8)
        res["per_kernel_512"] = {k: v for k, v in t512.items() if not k.startswith("conv_") and v["hbm_bytes_per_step_corrected"] > 1e8}
    except Exception as e:
        res["per_kernel_512"] = {"failed": repr(e)}
    for tag, T in (("pcg512", 8), ("pcg512f32", 4), ("pcg512rockerf32", 4)):
        try:
            t = traffic(d + "/pmc_fetch_%s_results.db" % tag, d + "/pmc_write_%s_results.db" % tag, 512 ** 3, T)
            res["per_kernel_" + tag] = t
            res[("rocker_512_" if "rocker" in tag else "bunny_small_512_") + ("f64" if T == 8 else "f32") + ("_primal" if "rocker" in tag else "")] = {k: {"hbm_bytes_per_launch": v["hbm_bytes_per_launch_corrected"], "launches": v["launches_per_step"],
                                                                         "per_launch_in_units_of_N_T": v["per_launch_in_units_of_N_T"]}
                                                                     for k, v in t.items() if k.startswith("cg_")}
        except Exception as e:
            res["per_kernel_" + tag] = {"failed": repr(e)}
    json.dump(res, open(out + "/" + TAG + "_pmc_traffic.json", "w"), indent=1)
    # ---- SQ counters of the Step-1 kernel, per step
    lines = ["SQ / GRBM counters of the Step-1 kernel ALONE (tools/conv_only.py: 2 x shm_grid_run_conv on bunny_small 256^3 fp64, no set-up kernels on the device).",
             "Two passes (SQ has 8 counter slots).  Every counter is the sum over all SEs / XCDs, PER STEP (= per Step 1: the sum over the run's launches divided by the",
             "2 Step-1 executions of the run; launches per step printed).", ""]
    vals = {}
    for db, names in ((d + "/pmc_sq1_results.db", ["SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAVES", "GRBM_GUI_ACTIVE"]),
                      (d + "/pmc_sq2_results.db", ["SQ_INSTS_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_ACTIVE_INST_LDS", "SQ_INSTS_SALU", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY"])):
        try:
            dur = durations(db)
            for nm in names:
                for k, (v, cnt) in per_kernel(db, nm).items():
                    if "conv_" in k and "kernel" in k:
                        vals[nm] = vals.get(nm, 0.0) + v / 2.0
                        vals["_launches_per_step"] = cnt / 2.0
                        vals["_kernel"] = short(k)
                        vals["_dur_ns_per_step_" + db.split("/")[-1]] = dur.get(k, (0, 0))[0] / 2.0
        except Exception as e:
            lines.append("(%s: %r)" % (db, e))
    for k, v in vals.items():
        lines.append("%-40s %s" % (k, v))
    nominal = 256.0 ** 3 * 2856
    p64 = float(sys.argv[3]) if len(sys.argv) > 3 else None
    p32 = float(sys.argv[4]) if len(sys.argv) > 4 else None
    if p64 is None:   # the bench line of the traced run of the same command
        try:
            for ln in open(d + "/bench256.log"):
                if ln.startswith("{") and '"step1"' in ln:
                    s1 = json.loads(ln)["step1"]
                    p64, p32 = float(s1["pairs_fp64"]), float(s1["pairs_fp32"])
        except Exception:
            pass
    lines += ["", "derived (per step):", "  nominal pairs N*S = %.4e" % nominal]
    if p64 is not None:
        lines.append("  evaluated pairs (shm_stats of the same configuration): fp64 %.4e (%.3f of nominal), packed fp32 %.4e (%.3f)" % (p64, p64 / nominal, p32, p32 / nominal))
    if "SQ_INSTS_VALU" in vals:
        lines.append("  VALU wave-instructions per nominal pair = SQ_INSTS_VALU * 64 / (N*S) = %.2f  (all tiers, the per-block source scan, classification and normalisation included)" % (vals["SQ_INSTS_VALU"] * 64 / nominal))
    d1 = next((vals[k] for k in vals if k.startswith("_dur_ns_per_step_pmc_sq1")), None)
    if "GRBM_GUI_ACTIVE" in vals and d1:
        clk = vals["GRBM_GUI_ACTIVE"] / 8.0 / d1
        lines.append("  effective shader clock = GRBM_GUI_ACTIVE / 8 XCD instances / duration = %.3f GHz (profiled pass, duration %.3f ms)" % (clk, d1 * 1e-6))
        if "SQ_ACTIVE_INST_VALU" in vals:
            lines.append("  VALU busy = SQ_ACTIVE_INST_VALU * 4 cycles / (1024 SIMDs * duration * clock) = %.3f" % (vals["SQ_ACTIVE_INST_VALU"] * 4.0 / (1024.0 * d1 * clk)))
    if "SQ_LDS_BANK_CONFLICT" in vals and "SQ_INSTS_LDS" in vals:
        lines.append("  LDS bank-conflict cycles per LDS instruction = %.3f" % (vals["SQ_LDS_BANK_CONFLICT"] / max(vals["SQ_INSTS_LDS"], 1)))
    if "SQ_WAIT_INST_ANY" in vals and "SQ_WAVE_CYCLES" in vals:
        lines.append("  SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES = %.3f" % (vals["SQ_WAIT_INST_ANY"] / max(vals["SQ_WAVE_CYCLES"], 1)))
    open(out + "/" + TAG + "_sq_counters_conv.txt", "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))
    print(json.dumps(res["bunny_small_256_f64"], indent=1))

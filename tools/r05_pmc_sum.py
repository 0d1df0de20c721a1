"""Sum the SQ / GRBM counters of the Step-1 kernel over the rocprofv3 --pmc passes of tools/r05_step1_ab.sh (kernel alone, tools/conv_only.py: 2 x shm_grid_run_conv),
per Step 1.   python tools/r05_pmc_sum.py <pass dir> [<pass dir> ...]"""
import glob, sqlite3, sys
tot = {}; dur = None; name = None; n_launch = 0
for d in sys.argv[1:]:
    for db in glob.glob(d + "/**/*_results.db", recursive=True):
        c = sqlite3.connect(db)
        cols = [r[1] for r in c.execute("pragma table_info(counters_collection)").fetchall()]
        kn = "kernel_name" if "kernel_name" in cols else "name"
        for k, cn, v, nd in c.execute("select %s, counter_name, sum(value), count(distinct dispatch_id) from counters_collection where %s like '%%conv_%%' group by %s, counter_name" % (kn, kn, kn)).fetchall():
            tot[cn] = v; name = k; n_launch = nd
        r = c.execute("select sum(duration), count(*) from kernels where name like '%conv_%'").fetchone()
        if r and r[0]: dur = (r[0], r[1])
steps = 2.0
print("kernel", (name or "?")[:60], "dispatches", n_launch, "duration per step %.3f ms" % (dur[0] / steps * 1e-6) if dur else "")
for k in sorted(tot): print("%-28s %.4e" % (k, tot[k] / steps))
N_S = 256.0 ** 3 * 2856
g = lambda k: tot.get(k, float("nan")) / steps
if dur:
    t = dur[0] / steps * 1e-9
    clk = g("GRBM_GUI_ACTIVE") / 8 / t
    print("VALU instr per nominal pair (bunny 256^3) %.2f   clock %.3f GHz   VALU busy %.3f   trans share %.3f   LDS instr per pair %.3f   conflict cycles / LDS instr %.2f   WAIT_INST_ANY / WAVE_CYCLES %.3f" % (
        g("SQ_INSTS_VALU") * 64 / N_S, clk * 1e-9, g("SQ_ACTIVE_INST_VALU") * 4 / (1024 * t * clk), g("SQ_INSTS_VALU_TRANS") / g("SQ_INSTS_VALU"), g("SQ_INSTS_LDS") * 64 / N_S,
        g("SQ_LDS_BANK_CONFLICT") / g("SQ_INSTS_LDS"), g("SQ_WAIT_INST_ANY") / g("SQ_WAVE_CYCLES")))

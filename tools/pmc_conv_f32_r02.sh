set -x
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_f32
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --no-cpu-baseline --no-also --workload rocker_512_f32"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES GRBM_GUI_ACTIVE -d $OUT -o sq -- $B --steps 1 --warmup 0 > $OUT/sq.log 2>&1
rocprofv3 -L 2>/dev/null | grep -i "trans\|SQ_INSTS_VALU_\|VALU_" | head -20 > $OUT/counters.txt
python3 - <<'P'
import sqlite3,os,glob
from collections import defaultdict
R=os.environ["GRAFT_REPO_ROOT"]
for db in sorted(glob.glob(R+"/gpurun_out/prof_f32/*_results.db")):
    c=sqlite3.connect(db)
    cols=[r[1] for r in c.execute("pragma table_info(counters_collection)").fetchall()]
    kn="kernel_name" if "kernel_name" in cols else "name"; did="dispatch_id" if "dispatch_id" in cols else "id"
    rows=c.execute("select %s, counter_name, %s, sum(value) from counters_collection group by %s, counter_name, %s"%(kn,did,kn,did)).fetchall()
    acc=defaultdict(list)
    for k,cn,_,v in rows:
        if "conv_normalize" in k: acc[(k.split("(")[0][-40:],cn)].append(v)
    dur={r[0].split("(")[0][-40:]:r[1] for r in c.execute("select name, avg(duration) from kernels group by name").fetchall()}
    for (k,cn),v in sorted(acc.items()): print(k, cn, "%.5g"%(sum(v)/len(v)), "dur_ms %.2f"%(dur.get(k,0)/1e6))
P
cat $OUT/counters.txt

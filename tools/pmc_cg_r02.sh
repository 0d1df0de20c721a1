set -x
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_cg
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B512="python3 $R/bench.py --no-cpu-baseline --no-also --workload bunny_small_512_f64 --solver primal --precond none"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS -d $OUT -o sq -- $B512 --steps 1 --warmup 0 --max-iters 12 > $OUT/sq.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_WAVES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $OUT -o sq2 -- $B512 --steps 1 --warmup 0 --max-iters 12 > $OUT/sq2.log 2>&1
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum -d $OUT -o tcc -- $B512 --steps 1 --warmup 0 --max-iters 12 > $OUT/tcc.log 2>&1
python3 - <<'P'
import sqlite3,os,glob
from collections import defaultdict
R=os.environ["GRAFT_REPO_ROOT"]
for db in sorted(glob.glob(R+"/gpurun_out/prof_cg/*_results.db")):
    c=sqlite3.connect(db)
    cols=[r[1] for r in c.execute("pragma table_info(counters_collection)").fetchall()]
    kn="kernel_name" if "kernel_name" in cols else "name"; did="dispatch_id" if "dispatch_id" in cols else "id"
    rows=c.execute("select %s, counter_name, %s, sum(value) from counters_collection group by %s, counter_name, %s"%(kn,did,kn,did)).fetchall()
    acc=defaultdict(list)
    for k,cn,_,v in rows:
        if "cg_fused" in k or "x_update2" in k: acc[(k.split("(")[0][-40:],cn)].append(v)
    dur={r[0].split("(")[0][-40:]:r[1] for r in c.execute("select name, avg(duration) from kernels group by name").fetchall()}
    for (k,cn),v in sorted(acc.items()): print(os.path.basename(db), k, cn, "%.4g"%(sum(v)/len(v)), "dur_us %.1f"%(dur.get(k,0)/1e3))
P

R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r03/convpmc; rm -rf $OUT; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
export SHM_DEBUG_KNOBS=1   # experiment knobs of the library are read only behind this gate
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT -o f -- python3 $R/tools/conv_only.py > $OUT/f.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT -o w -- python3 $R/tools/conv_only.py > $OUT/w.log 2>&1
cd $R; python3 - <<'P'
import sqlite3,os
for tag,ctr in (("f","FETCH_SIZE"),("w","WRITE_SIZE")):
    c=sqlite3.connect(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r03/convpmc/%s_results.db"%tag)
    cols=[r[1] for r in c.execute("pragma table_info(counters_collection)").fetchall()]
    kn="kernel_name" if "kernel_name" in cols else "name"; did="dispatch_id" if "dispatch_id" in cols else "id"
    for r in c.execute("select %s,%s,sum(value) from counters_collection where counter_name=? group by %s,%s"%(kn,did,kn,did),(ctr,)).fetchall():
        if "conv" in r[0]: print(ctr, r[0][:40], r[1], "%.1f MB raw"%(r[2]/1024))
P

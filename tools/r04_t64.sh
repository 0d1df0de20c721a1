#!/bin/bash
# timeline of a 64^3 solve (the reference's own CPU-runnable case): which kernels make up its ~1.8 ms
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_t64
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT -o b64 -- python3 $R/bench.py --no-cpu-baseline --no-also --workload bunny_small_64_f64 --steps 3 --warmup 1 > $OUT/b64.log 2>&1
python3 $R/tools/timeline.py $OUT/b64_results.db 140 > $R/gpurun_out/r04_timeline_64.txt 2>&1
python3 $R/profiles/rocpd_summary.py $OUT/b64_results.db $R/gpurun_out/r04_bench64_kernel_stats.txt > /dev/null
tail -2 $OUT/b64.log | cut -c1-600

#!/bin/bash
export SHM_DEBUG_KNOBS=1   # experiment knobs of the library are read only behind this gate
# rocprofv3 kernel trace of one bench.py workload (2 timed solves): per-kernel totals.   bash tools/trace_workload.sh rocker_512_f32 [extra bench.py flags]
R="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
WL=${1:-rocker_512_f32}; shift
OUT="$R/gpurun_out/trace_$WL"
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT -o t -- python3 $R/bench.py --no-cpu-baseline --no-also --workload $WL --steps 2 --warmup 1 "$@" > $OUT/bench.log 2>&1
python3 $R/profiles/rocpd_summary.py $OUT/t_results.db $OUT/kernel_stats.txt > /dev/null
python3 $R/tools/timeline.py $OUT/t_results.db 70 > $OUT/timeline.txt 2>&1
rm -f $OUT/*.db
head -40 $OUT/kernel_stats.txt | cut -c1-200

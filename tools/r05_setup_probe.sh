#!/bin/bash
export SHM_DEBUG_KNOBS=1   # experiment knobs of the library are read only behind this gate
# Set-up beside Step 1 at 256^3: the A/B of the Green's-table GEMM shape, then a kernel trace of one solve (timeline of the set-up kernels inside Step 1's span).
R="$(cd "$(dirname "$0")/.." && pwd)"
OUT="$(mkdir -p "$1" && cd "$1" && pwd)"
cd "$R"
for rep in 1 2 3; do python3 tools/ab.py bunny_small.obj:4:64 "default=" "wn2=SHM_GREEN_WN=2" "wide=SHM_GREEN_WIDE=1"; done > "$OUT/ab.txt" 2>&1
cd /tmp && export TMPDIR=/tmp
for v in default wn2; do
  if [ $v = wn2 ]; then export SHM_GREEN_WN=2; fi
  rocprofv3 --kernel-trace -d "$OUT/trace_$v" -o t -- python3 "$R/bench.py" --no-cpu-baseline --no-also --steps 2 --warmup 1 > "$OUT/trace_$v.log" 2>&1
  python3 "$R/tools/timeline.py" "$OUT/trace_$v/"*/t_results.db 120 > "$OUT/timeline_$v.txt" 2>&1 || python3 "$R/tools/timeline.py" "$OUT/trace_$v/t_results.db" 120 > "$OUT/timeline_$v.txt" 2>&1
  rm -rf "$OUT/trace_$v"
done

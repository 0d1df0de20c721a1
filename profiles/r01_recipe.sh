#!/bin/bash
# Round-1 profiling recipe (run on the GPU box through gpurun):  tools/profile_r01.sh
# 1) kernel trace + stats of the default bench (rocpd database -> profiles/rocpd_summary.py), default (dual) and primal solver
# 2) separate PMC passes for FETCH_SIZE and WRITE_SIZE (never combined with a trace domain other than --kernel-trace)
set -x
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_r01
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT -o bench256 -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench256.log 2>&1
rocprofv3 --kernel-trace --stats -d $OUT -o bench256_primal -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --solver primal > $OUT/bench256_primal.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT -o pmc_fetch -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $OUT/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT -o pmc_write -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $OUT/pmc_write.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT -o pmc_fetch_primal -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --solver primal > $OUT/pmc_fetch_primal.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT -o pmc_write_primal -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --solver primal > $OUT/pmc_write_primal.log 2>&1
ls -la $OUT

#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out/r04
python -m pytest tests -m gpu -q -x > gpurun_out/r04_gpu_tests_full.log 2>&1
echo "pytest rc $?" >> gpurun_out/r04_gpu_tests_full.log
grep -E "passed|failed|pytest rc|^FAILED" gpurun_out/r04_gpu_tests_full.log | tail -4 > gpurun_out/r04_gpu_tests.txt
cat gpurun_out/r04_gpu_tests.txt
python bench.py --workload rocker_512_f64 --no-cpu-baseline --no-also --steps 3 --warmup 1 > gpurun_out/r04/bench_rocker_512_f64.json 2>/dev/null
tail -c 600 gpurun_out/r04/bench_rocker_512_f64.json
python tools/r04_ab.py "rocker.obj:3:64,SprayBottle.pc:3:64,chair.obj:4:64,rocker.obj:4:64,SprayBottle.pc:4:64,chair.obj:5:64,knot.obj:3:64,rocker.pc:4:64,chair.pc:4:64,knot.pc:3:64,rocker.obj:5:64" "shipped=" "limit4096=SHM_DUAL_DIRECT_EST_OFF=1" "direct16k=SHM_DUAL_DIRECT_MAX_M=16384;SHM_DENSE_S_MAX_M=16384" > gpurun_out/r04_direct_rule.txt 2>&1

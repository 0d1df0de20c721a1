#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out/r04
python -m pytest tests -m gpu -q -x > gpurun_out/r04_gpu_tests_full.log 2>&1
echo "pytest rc $?" >> gpurun_out/r04_gpu_tests_full.log
grep -E "passed|failed|pytest rc|^FAILED" gpurun_out/r04_gpu_tests_full.log | tail -4 > gpurun_out/r04_gpu_tests.txt
cat gpurun_out/r04_gpu_tests.txt
python bench.py --workload rocker_512_f64 --no-cpu-baseline --no-also --steps 3 --warmup 1 > gpurun_out/r04/bench_rocker_512_f64.json 2>/dev/null
tail -c 600 gpurun_out/r04/bench_rocker_512_f64.json

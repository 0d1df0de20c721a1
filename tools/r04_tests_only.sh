#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
python -m pytest tests -m gpu -q -x > gpurun_out/r04_gpu_tests_full.log 2>&1
echo "pytest rc $?" >> gpurun_out/r04_gpu_tests_full.log
grep -E "passed|failed|pytest rc|^FAILED" gpurun_out/r04_gpu_tests_full.log | tail -4 > gpurun_out/r04_gpu_tests.txt
cat gpurun_out/r04_gpu_tests.txt
python tools/setup_alone.py bunny_small_64_f64 bunny_small_128_f64 bunny_small_256_f64 > gpurun_out/r04_setup_alone_pairs.txt 2>&1
cat gpurun_out/r04_setup_alone_pairs.txt

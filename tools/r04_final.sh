#!/bin/bash
# final verification of the round: the whole GPU suite (summary kept), then the measurement sweep
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
python -m pytest tests -m gpu -q -x > gpurun_out/r04_gpu_tests_full.log 2>&1
echo "pytest rc $?" >> gpurun_out/r04_gpu_tests_full.log
grep -E "passed|failed|pytest rc" gpurun_out/r04_gpu_tests_full.log | tail -3 > gpurun_out/r04_gpu_tests.txt
cat gpurun_out/r04_gpu_tests.txt
bash tools/collect_r04.sh > gpurun_out/collect_r04.log 2>&1
tail -30 gpurun_out/collect_r04.log | cut -c1-420

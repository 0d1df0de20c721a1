#!/bin/bash
# round 4, first GPU call: the new Step-1 tests, the bench line without the long legs, the tier table at full sizes
mkdir -p gpurun_out
nproc > gpurun_out/r04_nproc.txt; free -g >> gpurun_out/r04_nproc.txt
python bench.py --no-also --no-cpu-baseline > gpurun_out/r04_bench_quick.json 2> gpurun_out/r04_bench_quick.err
timeout 1500 python -m pytest tests -m gpu -x -q -s -k "step1_full_size or far_tier_exponent or translation_invariant or bench_py_multi or conv_normalize_matches or tiered_conv_stays or far_clusters" > gpurun_out/r04_tests_step1.log 2>&1
echo "pytest rc $?" >> gpurun_out/r04_tests_step1.log
timeout 1500 python tools/tier_robustness_big.py > gpurun_out/r04_tier_robustness_big.txt 2>&1
echo "rc $?" >> gpurun_out/r04_tier_robustness_big.txt

#!/bin/bash
mkdir -p gpurun_out
python bench.py --no-also --no-cpu-baseline > gpurun_out/r04_bench_quick3.json 2> gpurun_out/r04_bench_quick3.err
SHM_CONV_REDO_RATIO=0 python bench.py --no-also --no-cpu-baseline > gpurun_out/r04_bench_quick3_noredo.json 2>> gpurun_out/r04_bench_quick3.err
SHM_DIV_CLASSIC=1 python bench.py --no-also --no-cpu-baseline --workload bunny_small_512_f64 --steps 5 > gpurun_out/r04_bench_512_divclassic.json 2>> gpurun_out/r04_bench_quick3.err
python bench.py --no-also --no-cpu-baseline --workload bunny_small_512_f64 --steps 5 > gpurun_out/r04_bench_512.json 2>> gpurun_out/r04_bench_quick3.err
python bench.py --no-also --no-cpu-baseline --workload rocker_512_f32 --steps 3 > gpurun_out/r04_bench_rocker_f32.json 2>> gpurun_out/r04_bench_quick3.err
timeout 2400 python -m pytest tests -m gpu -q -s -k "step1_full_size or far_tier_exponent or translation_invariant or bench_py_multi or tiered_conv_stays or preconditioner_is_the_dct or phi_matches_lu_golden or matches_c_oracle_128 or every_data_file or divergence or conv_normalize or far_clusters or fp32_conv_exponent or full_size" > gpurun_out/r04_tests_run3.log 2>&1
echo "pytest rc $?" >> gpurun_out/r04_tests_run3.log
timeout 1500 python tools/tier_robustness_big.py > gpurun_out/r04_tier_robustness_big2.txt 2>&1

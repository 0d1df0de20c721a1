#!/bin/bash
mkdir -p gpurun_out
bash tools/collect_r04.sh > gpurun_out/r04_collect.log 2>&1
tail -25 gpurun_out/r04_collect.log | cut -c1-400
bash tools/profile_r04.sh > gpurun_out/r04_profile.log 2>&1
tail -30 gpurun_out/r04_profile.log | cut -c1-300
timeout 1500 python tools/tier_robustness_big.py > gpurun_out/r04_tier_robustness_big4.txt 2>&1
tail -3 gpurun_out/r04_tier_robustness_big4.txt

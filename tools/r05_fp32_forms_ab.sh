#!/bin/bash
# fp32 solves now run their set-up BESIDE Step 1 (tiered kernel, two waves per SIMD): are the dual solver's form rules, drawn in round 4 for an fp32 Step 1 that left the
# set-up no room, still right?   bash tools/r05_fp32_forms_ab.sh <out-file>
R="$(cd "$(dirname "$0")/.." && pwd)"
cd "$R"
for rep in 1 2; do
python3 tools/ab.py "rocker.obj:5:32,chair.obj:5:32,bunny_small.obj:5:32,bunny.pc:5:32,SprayBottle.pc:4:32,rocker.obj:4:32" "default=" "denseS=SHM_DUAL_DENSE_S_ALWAYS=1;SHM_DENSE_S_MAX_M=16384" "direct=SHM_DUAL_DIRECT_ALWAYS=1;SHM_DUAL_DIRECT_MAX_M=16384"
done > "$1" 2>&1

#!/bin/bash
# gj_step_kernel at three waves per SIMD (144 registers, no spill) against four (128 registers, 21 spilled), beside the round-5 Step 1 (171 registers): variant built by
#   hipcc ... -DSHM_GJ_STEP_WAVES=3 into signed-heat-3d_amd/lib/variants/libshm_grid_gj3.so
R="$(cd "$(dirname "$0")/.." && pwd)"; cd "$R"
V=signed-heat-3d_amd/lib/variants
for rep in 1 2 3; do python3 tools/ab.py "bunny_small.obj:2:64,bunny_small.obj:3:64,bunny_small.obj:4:64" "w4=" "w3=SHM_GRID_LIB=$V/libshm_grid_gj3.so" "w4alone=SHM_SETUP_ALONE=1" "w3alone=SHM_GRID_LIB=$V/libshm_grid_gj3.so;SHM_SETUP_ALONE=1"; done
python3 tools/r05_proj_probe.py samples12

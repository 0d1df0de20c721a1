#!/bin/bash
# Round 4: the one-launch-per-pivot-block Gauss-Jordan (gj_step_kernel) against the three-launch chain (SHM_GJ_CLASSIC=1): parity tests, then the set-up alone / beside Step 1
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
python -m pytest tests -m gpu -q -x -k "projector or lu_golden or c_oracle_128 or refine or direct or schur or two_level or n24" 2>&1 | tail -5 > gpurun_out/r04_gj_tests.txt
WL="bunny_small_64_f64 bunny_small_128_f64 bunny_small_256_f64"
{
echo "== three launches per pivot block (SHM_GJ_CLASSIC=1)"
SHM_GJ_CLASSIC=1 python tools/setup_alone.py $WL
echo "== one launch per pivot block"
python tools/setup_alone.py $WL
echo "== one launch per pivot block, scalar pivot inversion (SHM_GJ_PIVOT_E=4)"
SHM_GJ_PIVOT_E=4 python tools/setup_alone.py $WL
if [ -f signed-heat-3d_amd/lib/variants/libshm_grid_nocap.so ]; then
echo "== one launch per pivot block, no register cap (144 / 148 registers)"
SHM_GRID_LIB=signed-heat-3d_amd/lib/variants/libshm_grid_nocap.so python tools/setup_alone.py $WL
fi
} > gpurun_out/r04_gj_probe.txt 2>&1

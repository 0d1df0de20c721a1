#!/bin/bash
mkdir -p gpurun_out
O=gpurun_out/r04_ab4.txt; : > $O
for rep in 1 2 3; do
for v in "" CA R3; do
  if [ -n "$v" ]; then export SHM_GRID_LIB=$PWD/signed-heat-3d_amd/lib/variants/libshm_grid_$v.so; else unset SHM_GRID_LIB; fi
  python bench.py --no-also --no-cpu-baseline --steps 10 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('rep $rep variant [$v] 256: ms/step %.2f conv %.2f wait %.2f div %.3f' % (d['ms_per_step'], d['phases_ms']['ms_conv'], d['phases_ms']['ms_wait_setup'], d['phases_ms']['ms_div']))" >> $O
done
done
for rep in 1 2; do
for v in "" NT; do
  if [ -n "$v" ]; then export SHM_GRID_LIB=$PWD/signed-heat-3d_amd/lib/variants/libshm_grid_$v.so; else unset SHM_GRID_LIB; fi
  for w in bunny_small_512_f64 bunny_small_512_f32; do
  python bench.py --no-also --no-cpu-baseline --workload $w --steps 3 --warmup 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('rep $rep variant [$v] $w: ms/step %.2f conv %.2f div %.3f' % (d['ms_per_step'], d['phases_ms']['ms_conv'], d['phases_ms']['ms_div']))" >> $O
  done
done
done
unset SHM_GRID_LIB
sort -k3,3 -s $O
timeout 3000 python -m pytest tests -m gpu -q -x > gpurun_out/r04_tests_full.log 2>&1
echo "pytest rc $?" >> gpurun_out/r04_tests_full.log
tail -15 gpurun_out/r04_tests_full.log
timeout 1500 python tools/tier_robustness_big.py > gpurun_out/r04_tier_robustness_big3.txt 2>&1
cat gpurun_out/r04_tier_robustness_big3.txt

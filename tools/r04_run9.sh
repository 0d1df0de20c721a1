#!/bin/bash
mkdir -p gpurun_out
: > gpurun_out/r04_g_sweep.txt
for g in 8 7.5 7 6.5 6; do
  echo "=== SHM_CONV_TIER_LOG=$g (a-posteriori threshold 3.3e-3)" >> gpurun_out/r04_g_sweep.txt
  SHM_CONV_TIER_LOG=$g timeout 1500 python tools/tier_robustness_big.py --cases bunny_small.obj 4.0 bunny_small.obj 5.0 rocker.obj 5.0 knot.obj 5.0 chair.obj 5.0 SprayBottle.pc 6.0 >> gpurun_out/r04_g_sweep.txt 2>&1
  for i in 1 2; do
  SHM_CONV_TIER_LOG=$g python bench.py --no-also --no-cpu-baseline --steps 10 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('G=$g bench 256: ms/step %.2f conv %.2f' % (d['ms_per_step'], d['phases_ms']['ms_conv']))" >> gpurun_out/r04_g_sweep.txt
  done
done
cat gpurun_out/r04_g_sweep.txt

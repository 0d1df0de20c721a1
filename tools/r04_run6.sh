#!/bin/bash
mkdir -p gpurun_out
O=gpurun_out/r04_ab3.txt; : > $O
for rep in 1 2 3; do
for v in "" V2 V3 V4 V5 V6; do
  if [ -n "$v" ]; then export SHM_GRID_LIB=$PWD/signed-heat-3d_amd/lib/variants/libshm_grid_$v.so; else unset SHM_GRID_LIB; fi
  python bench.py --no-also --no-cpu-baseline --steps 10 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('rep $rep variant [$v] 256: ms/step %.2f conv %.2f wait %.2f' % (d['ms_per_step'], d['phases_ms']['ms_conv'], d['phases_ms']['ms_wait_setup']))" >> $O
done
done
for v in "" V2 V3 V4 V5; do
  if [ -n "$v" ]; then export SHM_GRID_LIB=$PWD/signed-heat-3d_amd/lib/variants/libshm_grid_$v.so; else unset SHM_GRID_LIB; fi
  python bench.py --no-also --no-cpu-baseline --workload bunny_small_512_f64 --steps 3 --warmup 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('variant [$v] 512: ms/step %.2f conv %.2f' % (d['ms_per_step'], d['phases_ms']['ms_conv']))" >> $O
done
unset SHM_GRID_LIB
sort -k3,3 -s $O
: > gpurun_out/r04_drop_sweep.txt
for db in 2e-9 2e-10 2e-11; do
  echo "=== SHM_CONV_REDO_RATIO=3.3e-3 SHM_CONV_DROP_BUDGET=$db" >> gpurun_out/r04_drop_sweep.txt
  SHM_CONV_REDO_RATIO=3.3e-3 SHM_CONV_DROP_BUDGET=$db timeout 1500 python tools/tier_robustness_big.py --cases SprayBottle.pc 6.0 knot.obj 6.0 >> gpurun_out/r04_drop_sweep.txt 2>&1
done
cat gpurun_out/r04_drop_sweep.txt

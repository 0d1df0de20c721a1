#!/bin/bash
mkdir -p gpurun_out
O=gpurun_out/r04_ab2.txt; : > $O
for v in "" U4 U4R U4N; do
  if [ -n "$v" ]; then export SHM_GRID_LIB=$PWD/signed-heat-3d_amd/lib/variants/libshm_grid_$v.so; else unset SHM_GRID_LIB; fi
  for i in 1 2; do
    python bench.py --no-also --no-cpu-baseline --steps 10 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('variant [$v] 256: ms/step %.2f conv %.2f wait %.2f' % (d['ms_per_step'], d['phases_ms']['ms_conv'], d['phases_ms']['ms_wait_setup']))" >> $O
  done
  python bench.py --no-also --no-cpu-baseline --workload bunny_small_512_f64 --steps 3 --warmup 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('variant [$v] 512: ms/step %.2f conv %.2f div %.3f pcg %.2f' % (d['ms_per_step'], d['phases_ms']['ms_conv'], d['phases_ms']['ms_div'], d['phases_ms']['ms_pcg']))" >> $O
done
unset SHM_GRID_LIB
for zc in 2 4 8; do
  SHM_DIV_ZC=$zc python bench.py --no-also --no-cpu-baseline --workload bunny_small_512_f64 --steps 3 --warmup 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('div ZC=$zc 512 f64: div %.3f ms' % (d['phases_ms']['ms_div']))" >> $O
  SHM_DIV_ZC=$zc python bench.py --no-also --no-cpu-baseline --workload bunny_small_512_f32 --steps 3 --warmup 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('div ZC=$zc 512 f32: div %.3f ms' % (d['phases_ms']['ms_div']))" >> $O
  SHM_DIV_ZC=$zc python bench.py --no-also --no-cpu-baseline --steps 3 --warmup 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('div ZC=$zc 256 f64: div %.3f ms' % (d['phases_ms']['ms_div']))" >> $O
done
cat $O
: > gpurun_out/r04_redo_sweep.txt
for r in 1e-2 5e-3 3.3e-3 2e-3; do
  echo "=== SHM_CONV_REDO_RATIO=$r" >> gpurun_out/r04_redo_sweep.txt
  SHM_CONV_REDO_RATIO=$r timeout 1500 python tools/tier_robustness_big.py --cases bunny_small.obj 4.0 bunny_small.obj 5.0 rocker.obj 5.0 knot.obj 5.0 SprayBottle.pc 6.0 >> gpurun_out/r04_redo_sweep.txt 2>&1
done
cat gpurun_out/r04_redo_sweep.txt

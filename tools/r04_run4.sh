#!/bin/bash
mkdir -p gpurun_out
O=gpurun_out/r04_ab.txt; : > $O
for v in "" B C D E; do
  if [ -n "$v" ]; then export SHM_GRID_LIB=$PWD/signed-heat-3d_amd/lib/variants/libshm_grid_$v.so; else unset SHM_GRID_LIB; fi
  for i in 1 2; do
    python bench.py --no-also --no-cpu-baseline --steps 10 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('variant [$v] 256: ms/step %.2f conv %.2f' % (d['ms_per_step'], d['phases_ms']['ms_conv']))" >> $O
  done
  python bench.py --no-also --no-cpu-baseline --workload bunny_small_512_f64 --steps 3 --warmup 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('variant [$v] 512: ms/step %.2f conv %.2f div %.3f pcg %.2f' % (d['ms_per_step'], d['phases_ms']['ms_conv'], d['phases_ms']['ms_div'], d['phases_ms']['ms_pcg']))" >> $O
done
unset SHM_GRID_LIB
SHM_DCT_NO_PF=1 python bench.py --no-also --no-cpu-baseline --workload bunny_small_512_f64 --steps 3 --warmup 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('no-PF 512: ms/step %.2f conv %.2f div %.3f pcg %.2f' % (d['ms_per_step'], d['phases_ms']['ms_conv'], d['phases_ms']['ms_div'], d['phases_ms']['ms_pcg']))" >> $O
for pf in 0 1; do
  if [ $pf = 0 ]; then export SHM_DCT_NO_PF=1; else unset SHM_DCT_NO_PF; fi
  for w in bunny_small_512_f64 bunny_small_512_f32; do
  python bench.py --no-also --no-cpu-baseline --workload $w --solver primal --steps 1 --warmup 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('PF=$pf $w primal+dct: iters %d ms/iter %.3f' % (d['config']['cg_iters'], d['pcg']['ms_per_iter']), {k:(round(v['avg_ms_per_launch'],4), round(v['frac_of_hbm_peak'] or 0,3)) for k,v in d['kernels'].items()})" >> $O
  done
done
unset SHM_DCT_NO_PF
for zc in 8 16 32 64; do
  SHM_DIV_ZC=$zc python bench.py --no-also --no-cpu-baseline --workload bunny_small_512_f64 --steps 3 --warmup 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('div ZC=$zc 512 f64: div %.3f ms' % (d['phases_ms']['ms_div']))" >> $O
done
cat $O
timeout 1200 python -m pytest tests -m gpu -q -s -k "far_tier_exponent or translation_invariant or tiered_conv_stays or matches_c_oracle_128 or far_clusters or conv_normalize or fp32_conv_exponent or divergence or every_data_file or fuzz" > gpurun_out/r04_tests_run4.log 2>&1
echo "pytest rc $?" >> gpurun_out/r04_tests_run4.log
tail -12 gpurun_out/r04_tests_run4.log
timeout 1500 python tools/tier_worst_nodes.py SprayBottle.pc 6.0 SprayBottle.pc 5.0 > gpurun_out/r04_tier_worst_nodes_spray.txt 2>&1
cat gpurun_out/r04_tier_worst_nodes_spray.txt

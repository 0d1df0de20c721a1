#!/bin/bash
mkdir -p gpurun_out
: > gpurun_out/r04_eps_sweep.txt
for r in 0.0833 0.05 0.03; do
  echo "=== SHM_CONV_REDO_RATIO=$r (per unit of lambda' r: threshold on L1_far (coff + 16) / |X|)" >> gpurun_out/r04_eps_sweep.txt
  SHM_CONV_REDO_RATIO=$r timeout 1500 python tools/tier_robustness_big.py --cases bunny_small.obj 4.0 bunny_small.obj 5.0 rocker.obj 5.0 knot.obj 5.0 SprayBottle.pc 5.0 SprayBottle.pc 6.0 knot.obj 6.0 >> gpurun_out/r04_eps_sweep.txt 2>&1
  for i in 1 2; do
  SHM_CONV_REDO_RATIO=$r python bench.py --no-also --no-cpu-baseline --steps 10 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ratio=$r bench 256: ms/step %.2f conv %.2f redone %.3e' % (d['ms_per_step'], d['phases_ms']['ms_conv'], 0))" >> gpurun_out/r04_eps_sweep.txt
  done
done
cat gpurun_out/r04_eps_sweep.txt
# the projection changes: rocker stencil-PCG legs and the two-level tests
for w in rocker_512_f32 rocker_512_f64 bunny_small_512_f32 bunny_small_512_f64; do
 for x in 0 1; do
  if [ $x = 0 ]; then export SHM_CG_NO_XOVERLAP=1; else unset SHM_CG_NO_XOVERLAP; fi
  python bench.py --no-cpu-baseline --no-also --workload $w --solver primal --precond none --max-iters 200 --steps 1 --warmup 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w xoverlap=$x: ms/iter %.4f loop frac %.3f project %.3f' % (d['pcg']['ms_per_iter'], d['pcg']['frac_of_hbm_peak'], d['pcg']['ms_project_avg']), {k:(round(v['avg_ms_per_launch'],4), round(v['frac_of_hbm_peak'] or 0,3)) for k,v in d['kernels'].items()})"
 done
done
unset SHM_CG_NO_XOVERLAP
python bench.py --no-cpu-baseline --no-also --workload rocker_512_f32 --steps 3 --warmup 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('rocker_512_f32 default: ms/step %.2f' % d['ms_per_step'], d['phases_ms'], 'ms/iter %.3f' % d['pcg']['ms_per_iter'])"
timeout 1500 python -m pytest tests -m gpu -q -x -k "two_level or phi_matches_lu or fp32_configs or config2 or full_size or fused_sweeps or phi_64 or degenerate" 2>&1 | tail -5

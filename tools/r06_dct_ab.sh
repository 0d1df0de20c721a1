#!/bin/bash
# Round 6: the LDS-DMA double buffer of the dense transform sweeps (shm_dct.hip.h, PF) against the plain kernel (SHM_DCT_NO_PF=1), same box, same build.
R="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"; cd "$R" || exit 1
export SHM_DEBUG_KNOBS=1
run() {   # label, env assignment, bench args...
  local label="$1" envs="$2"; shift 2
  env $envs python3 bench.py --no-cpu-baseline --no-also "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d['phases_ms']; k=d.get('kernels',{})
print('%-44s ms/step %8.2f  pcg %8.3f  iters %3d  ms/iter %.4f  %s' % ('$label', d['ms_per_step'], p['ms_pcg'], d['config']['cg_iters'], d['pcg']['ms_per_iter'], {a:(round(b['avg_ms_per_launch'],4), round(b['frac_of_hbm_peak'] or 0,3)) for a,b in k.items() if 'dct' in a}))"
}
for rep in 1 2; do
  for v in "dma SHM_X=0" "plain SHM_DCT_NO_PF=1"; do
    set -- $v
    run "512 fp64 primal+dct [$1]" "$2" --workload bunny_small_512_f64 --solver primal --steps 2 --warmup 1
    run "512 fp64 default (direct dual) [$1]" "$2" --workload bunny_small_512_f64 --steps 3 --warmup 1
    run "512 fp32 primal+dct [$1]" "$2" --workload bunny_small_512_f32 --solver primal --steps 2 --warmup 1
    run "256 fp64 primal+dct [$1]" "$2" --solver primal --steps 3 --warmup 1
    run "256 fp64 default [$1]" "$2" --steps 5 --warmup 1
  done
done

#!/bin/bash
# Round-3 A/B of the fused stencil-CG sweeps at 512^3 fp32 (build-time shapes selected by env knobs of Solver::fused_cfg): rows per lane, waves per workgroup,
# planes per z chunk.  Prints per-kernel fractions of the 8 TB/s HBM peak.     bash tools/r03_cg_probe.sh [f32|f64]
P=${1:-f32}
R=${GRAFT_REPO_ROOT:-.}
run() { # label, env...
  label=$1; shift
  env "$@" python3 $R/bench.py --no-cpu-baseline --no-also --workload bunny_small_512_$P --solver primal --precond none --max-iters 200 --steps 1 --warmup 1 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); k=d['kernels']
print('%-28s ms/iter %.4f loop %.4f | '%('$label', d['pcg']['ms_per_iter'], d['pcg']['frac_of_hbm_peak']) + ' '.join('%s %.4f'%(n.split('<')[-1].rstrip('>')[:8], v['frac_of_hbm_peak']) for n,v in k.items()))"
}
run default SHM_X=0
run ry4 SHM_FUSED_RY=4
run waves16 SHM_FUSED_WAVES=16
run waves4 SHM_FUSED_WAVES=4
run zc8 SHM_FUSED_ZC=8
run zc16 SHM_FUSED_ZC=16
run zc32 SHM_FUSED_ZC=32
run zc64 SHM_FUSED_ZC=64
run ry4_zc32 SHM_FUSED_RY=4 SHM_FUSED_ZC=32
run waves16_zc64 SHM_FUSED_WAVES=16 SHM_FUSED_ZC=64

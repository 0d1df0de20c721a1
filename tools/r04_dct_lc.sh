#!/bin/bash
# A/B: complex lines per transform tile at n = 512 fp64 (4 shipped, 8 = full 128-byte rows) on the DENSE sweeps (direct dual solve phase, primal + DCT)
cd "$(dirname "$0")/.." || exit 1
run() { python bench.py --no-cpu-baseline --no-also --steps 3 --warmup 1 "$@" 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d['phases_ms']; print('   ms/step %.2f iters %d ms/iter %.3f'%(d['ms_per_step'], d['config']['cg_iters'], d['pcg']['ms_per_iter']), {k:round(v,2) for k,v in p.items()})"; }
for lib in "" signed-heat-3d_amd/lib/variants/libshm_grid_lc9_8.so; do
echo "lib=${lib:-shipped}"
SHM_GRID_LIB=$lib run --workload bunny_small_512_f64
SHM_GRID_LIB=$lib run --workload bunny_small_512_f64 --solver primal
SHM_GRID_LIB=$lib run --workload rocker_512_f64
done

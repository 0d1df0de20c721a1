#!/bin/bash
# experiment: CG on the explicit S beyond m = 8192 now that its assembly is a third shorter (rocker 512^3: m = 12 612)
cd "$(dirname "$0")/.." || exit 1
run() { python bench.py --no-cpu-baseline --no-also --steps 3 --warmup 1 "$@" 2>&1 | tail -1 | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d['phases_ms']; print('   ms/step %.2f iters %d rel %.2e'%(d['ms_per_step'], d['config']['cg_iters'], d['config']['rel_residual']), {k:round(v,2) for k,v in p.items()})
except Exception as e: print('   FAILED', e)"; }
for w in rocker_512_f32 rocker_512_f64; do
echo "$w default"; run --workload $w
echo "$w explicit S up to m = 16384"; SHM_DENSE_S_MAX_M=16384 run --workload $w
done

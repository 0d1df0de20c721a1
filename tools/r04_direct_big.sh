#!/bin/bash
# experiment: the direct dual solve (explicit S inverted beside Step 1) beyond m = 4096, where Step 1 is long enough to hide a bigger inversion
cd "$(dirname "$0")/.." || exit 1
run() { python bench.py --no-cpu-baseline --no-also --steps 2 --warmup 1 "$@" 2>&1 | tail -1 | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d['phases_ms']; print('   ms/step %.2f iters %d rel %.2e'%(d['ms_per_step'], d['config']['cg_iters'], d['config']['rel_residual']), {k:round(v,2) for k,v in p.items()})
except Exception as e: print('   FAILED', e)"; }
for w in rocker_512_f64; do
echo "$w default"; run --workload $w
echo "$w direct up to m = 16384"; SHM_DUAL_DIRECT_MAX_M=16384 SHM_DENSE_S_MAX_M=16384 run --workload $w
done

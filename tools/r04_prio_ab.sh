#!/bin/bash
# A/B: wave priority of the set-up kernels beside Step 1 (SHM_SETUP_PRIO) now that the set-up is shorter
cd "$(dirname "$0")/.." || exit 1
run() { python bench.py --no-cpu-baseline --no-also --steps 10 --warmup 2 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d['phases_ms']; print('   ms/step %.2f'%d['ms_per_step'], {k:round(v,2) for k,v in p.items()})"; }
for w in bunny_small_256_f64 bunny_small_128_f64 bunny_small_64_f64; do
for prio in 1 0 1 0; do echo "$w SHM_SETUP_PRIO=$prio"; SHM_SETUP_PRIO=$prio run --workload $w; done; done

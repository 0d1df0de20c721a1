for w in rocker_512_f32 spraybottle_pc_1024_f32; do python bench.py --workload $w --no-also --no-cpu-baseline --steps 3 --warmup 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d['phases_ms']; print(d['config']['workload'], 'ms/step %.1f'%d['ms_per_step'], {k:round(v,2) for k,v in p.items()})"; done

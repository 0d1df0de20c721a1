#!/bin/bash
mkdir -p gpurun_out
for w in rocker_512_f32 rocker_512_f64; do
  python bench.py --no-cpu-baseline --no-also --workload $w --solver primal --precond none --max-iters 200 --steps 1 --warmup 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w: ms/iter %.4f loop frac %.3f project %.3f' % (d['pcg']['ms_per_iter'], d['pcg']['frac_of_hbm_peak'], d['pcg']['ms_project_avg']), {k:(round(v['avg_ms_per_launch'],4), round(v['frac_of_hbm_peak'] or 0,3)) for k,v in d['kernels'].items()})"
done
timeout 3000 python -m pytest tests -m gpu -q -x > gpurun_out/r04_tests_full2.log 2>&1
echo "pytest rc $?" >> gpurun_out/r04_tests_full2.log
tail -6 gpurun_out/r04_tests_full2.log
bash tools/collect_r04.sh > gpurun_out/r04_collect.log 2>&1
tail -22 gpurun_out/r04_collect.log | cut -c1-330
bash tools/profile_r04.sh > gpurun_out/r04_profile.log 2>&1
tail -4 gpurun_out/r04_profile.log | cut -c1-300
timeout 1500 python tools/tier_robustness_big.py > gpurun_out/r04_tier_robustness_big5.txt 2>&1
tail -14 gpurun_out/r04_tier_robustness_big5.txt
python tools/pcie_inclusive.py > gpurun_out/r04_pcie_inclusive.json 2>/dev/null; cat gpurun_out/r04_pcie_inclusive.json | head -c 600

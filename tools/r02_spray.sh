#!/bin/bash
# SprayBottle.pc 1024^3 fp32 (configs[4] on one GPU) and rocker 512^3 fp32: two-level inverse of A A^T vs the dense one
cd "$(dirname "$0")/.."
O=gpurun_out/spray; mkdir -p $O
for v in twolevel dense; do
  for WL in ${WLS:-spraybottle_pc_1024_f32 rocker_512_f32}; do
    if [ $v = dense ]; then export SHM_TL_MIN_M=1000000; else unset SHM_TL_MIN_M; fi
    python bench.py --no-cpu-baseline --no-also --steps 2 --warmup 1 --workload $WL > $O/${WL}_$v.json 2> $O/${WL}_$v.err
    python - $O/${WL}_$v.json $v <<'P'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], d["config"]["workload"], "ms/step %.1f"%d["ms_per_step"], "iters", d["config"]["cg_iters"], {k:round(v,2) for k,v in d["phases_ms"].items()}, "per-iter %.3f project %.3f dct %.3f"%(d["pcg"]["ms_per_iter"], d["pcg"]["ms_project_avg"], 5*d["kernels"]["dct_lines_kernel"]["avg_ms_per_launch"]))
except Exception as e: print(sys.argv[2],"FAILED",e, open(sys.argv[1].replace(".json",".err")).read()[-500:])
P
  done
done 2>&1 | tee $O/summary.txt

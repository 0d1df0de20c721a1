#!/bin/bash
# Round-2 probe: fused stencil-CG sweeps (512^3, capped iterations): A/B of builds / knobs.
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/cg_probe; mkdir -p $O
run() { # name, env...
  local name=$1; shift
  env "$@" python bench.py --no-cpu-baseline --no-also --steps 1 --warmup 1 --solver primal --precond none --max-iters 200 --workload ${WL} > $O/${WL}_$name.json 2> $O/${WL}_$name.err
  python - $O/${WL}_$name.json $name <<'P'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    k=d["kernels"]; print(sys.argv[2], d["config"]["workload"], "ms/iter %.3f"%d["pcg"]["ms_per_iter"], {a:(round(b["avg_ms_per_launch"],4), round(b["achieved_GBps"] or 0)) for a,b in k.items()}, "proj %.3f"%d["pcg"]["ms_project_avg"])
except Exception as e: print(sys.argv[2],"FAILED",e)
P
}
for WL in bunny_small_512_f64 rocker_512_f32; do
  run skew0 X=1
  run skew256 SHM_ARRAY_SKEW_BYTES=256
  run skew4352 SHM_ARRAY_SKEW_BYTES=4352
  run skew69888 SHM_ARRAY_SKEW_BYTES=69888
  run skew1M SHM_ARRAY_SKEW_BYTES=1052928
  run skew0_again X=1
done 2>&1 | tee $O/summary.txt

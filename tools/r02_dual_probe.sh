#!/bin/bash
# Round-2 probe: dual solver per-iteration time, A/B knobs.
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/dual_probe; mkdir -p $O
if [ "${1:-}" = "test" ]; then
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "lu_golden or phi_64 or every_data_file or fuzz or full_size or fp32_path or local_slabs_with" 2>&1 | tail -8 > $O/pytest.txt
fi
run() { # name, env...
  local name=$1; shift
  env "$@" python bench.py --no-cpu-baseline --no-also --steps 3 --warmup 1 --workload ${WL} > $O/${WL}_$name.json 2> $O/${WL}_$name.err
  python - $O/${WL}_$name.json $name <<'P'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], d["config"]["workload"], "ms/step %.2f"%d["ms_per_step"], "iters", d["config"]["cg_iters"], "pcg %.3f ms  per-iter %.4f  dct/iter %.4f  rest/iter %.4f"%(d["phases_ms"]["ms_pcg"], d["pcg"]["ms_per_iter"], 5*d["kernels"]["dct_lines_kernel"]["avg_ms_per_launch"], d["pcg"]["ms_project_avg"]), "rel_res %.2e"%d["config"]["rel_residual"])
except Exception as e: print(sys.argv[2],"FAILED",e)
P
}
for WL in ${WLS:-bunny_small_256_f64 bunny_small_512_f64 rocker_512_f32}; do
  run new X=1
  run zfft SHM_DUAL_Z_FFT=1
done 2>&1 | tee $O/summary.txt

#!/bin/bash
# Round-6 check on the GPU box (through gpurun): the whole -m gpu suite with its -s output kept, then the bench legs the drop rule moves.
#   bash tools/r06_check.sh [tag]      -> gpurun_out/r06<tag>/
R="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
O="$R/gpurun_out/r06$1"
mkdir -p "$O"
cd "$R" || exit 1
[ -f bench.py ] || { echo "bench.py not found under $R" >&2; exit 1; }
if [ -z "$R06_NO_TESTS" ]; then
  python -m pytest tests -m gpu -q -s ${R06_PYTEST_ARGS} > "$O/gpu_tests.txt" 2>&1
  tail -15 "$O/gpu_tests.txt"
fi
for w in ${R06_WORKLOADS:-bunny_small_256_f64 rocker_512_f32 spraybottle_pc_1024_f32 rocker_512_f64 bunny_small_512_f64}; do
  python bench.py --workload $w --no-cpu-baseline --no-also --steps 3 --warmup 1 > "$O/bench_$w.json" 2> "$O/bench_$w.err"
done
python - "$O" <<'P'
import json,glob,os,sys
for f in sorted(glob.glob(os.path.join(sys.argv[1],"bench_*.json"))):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        s1=d.get("step1",{})
        print(os.path.basename(f), "ms/step %.2f nodes/s %.3e"%(d["ms_per_step"], d["value"]), {k:round(v,2) for k,v in d["phases_ms"].items()},
              "pairs fp64 %.3f fp32 %.3f of nominal"%(s1.get("pairs_fp64",0)/max(1,s1.get("pairs_nominal",1)), s1.get("pairs_fp32",0)/max(1,s1.get("pairs_nominal",1))))
    except Exception as e: print(f,"FAILED",e)
P

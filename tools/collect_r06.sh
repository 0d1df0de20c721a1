#!/bin/bash
# Round-6 measurement sweep (run on the GPU box through gpurun): every workload of bench.py, the solver variants at 256^3 / 512^3 and the PCIe-inclusive one-shot rate.
# Outputs land in gpurun_out/r06/ (copied to profiles/r06_*).   bash tools/collect_r06.sh
R="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
O="$R/gpurun_out/r06"
mkdir -p "$O"
cd "$R" || exit 1
python bench.py > "$O/bench_default.json" 2> "$O/bench_default.err"
for w in bunny_small_64_f64 bunny_small_128_f64 bunny_small_512_f64 bunny_pc_512_f64 rocker_512_f32 bunny_small_512_f32; do
  python bench.py --workload $w --no-cpu-baseline --no-also --steps 3 --warmup 1 > "$O/bench_$w.json" 2> "$O/bench_$w.err"
done
python bench.py --no-cpu-baseline --no-also --solver primal --steps 3 --warmup 1 > "$O/bench_256_primal_dct.json" 2> "$O/bench_256_primal_dct.err"
python bench.py --no-cpu-baseline --no-also --solver primal --precond none --steps 2 --warmup 1 > "$O/bench_256_primal_plain.json" 2> "$O/bench_256_primal_plain.err"
python bench.py --no-cpu-baseline --no-also --workload bunny_small_512_f64 --solver primal --steps 2 --warmup 1 > "$O/bench_512_primal_dct.json" 2> "$O/bench_512_primal_dct.err"
python bench.py --no-cpu-baseline --no-also --workload bunny_small_512_f64 --solver primal --precond none --max-iters 200 --steps 1 --warmup 1 > "$O/bench_512_primal_plain_200.json" 2> "$O/bench_512_primal_plain_200.err"
python bench.py --no-cpu-baseline --no-also --workload spraybottle_pc_1024_f32 --steps 1 --warmup 1 > "$O/bench_spraybottle_pc_1024_f32.json" 2> "$O/bench_spraybottle.err"
for w in rocker_512_f64 spraybottle_pc_1024_f64; do
  python bench.py --workload $w --no-cpu-baseline --no-also --steps 1 --warmup 1 > "$O/bench_$w.json" 2> "$O/bench_$w.err"
done
python bench.py --no-cpu-baseline --no-also --workload rocker_512_f32 --solver primal --precond none --max-iters 200 --steps 1 --warmup 1 > "$O/bench_rocker_512_f32_primal_plain_200.json" 2> "$O/bench_rocker_primal.err"
python bench.py --no-cpu-baseline --no-also --workload rocker_512_f64 --solver primal --precond none --max-iters 200 --steps 1 --warmup 1 > "$O/bench_rocker_512_f64_primal_plain_200.json" 2>> "$O/bench_rocker_primal.err"
python tools/pcie_inclusive.py > "$O/pcie_inclusive.json" 2> "$O/pcie.err"
SHM_DEBUG_KNOBS=1 python tools/setup_alone.py > "$O/setup_alone.txt" 2>&1
python - "$O" <<'P'
import json,glob,os,sys
for f in sorted(glob.glob(os.path.join(sys.argv[1],"bench_*.json"))):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(os.path.basename(f), "ms/step %.2f nodes/s %.3e iters %d"%(d["ms_per_step"], d["value"], d["config"]["cg_iters"]), {k:round(v,2) for k,v in d["phases_ms"].items()}, "ms/iter %.4f"%d["pcg"]["ms_per_iter"], {k:(round(v["avg_ms_per_launch"],4), round(v["achieved_GBps"] or 0)) for k,v in d["kernels"].items()})
    except Exception as e: print(f,"FAILED",e)
P

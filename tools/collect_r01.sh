#!/bin/bash
# Round-1 measurement sweep (run on the GPU box through gpurun): every workload of bench.py, the solver variants at 256^3 / 512^3,
# the full-size parity report and the PCIe-inclusive one-shot rate.  Outputs land in gpurun_out/r01/ (copied to profiles/ by hand).
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r01
mkdir -p $O
cd $R
python bench.py > $O/bench_default.json 2> $O/bench_default.err
for w in bunny_small_64_f64 bunny_small_128_f64 bunny_small_512_f64 bunny_pc_512_f64 rocker_512_f32; do
  python bench.py --workload $w --no-cpu-baseline --steps 3 --warmup 1 > $O/bench_$w.json 2> $O/bench_$w.err
done
python bench.py --no-cpu-baseline --solver primal --steps 3 --warmup 1 > $O/bench_primal.json 2> $O/bench_primal.err
python bench.py --no-cpu-baseline --solver primal --precond none --steps 2 --warmup 1 > $O/bench_primal_plain.json 2> $O/bench_primal_plain.err
python bench.py --no-cpu-baseline --workload bunny_small_512_f64 --solver primal --steps 2 --warmup 1 > $O/bench_512_primal.json 2> $O/bench_512_primal.err
python bench.py --no-cpu-baseline --workload spraybottle_pc_1024_f32 --steps 1 --warmup 0 > $O/bench_spraybottle.json 2> $O/bench_spraybottle.err
python tools/parity_fullsize.py bunny_small_256_f64 $O/parity_bunny_small_256_f64.json > $O/parity.log 2>&1
python tools/pcie_inclusive.py > $O/pcie_inclusive.json 2> $O/pcie.err
ls -la $O

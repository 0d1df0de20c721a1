#!/bin/bash
# Round-6 profiling recipe (run on the GPU box through gpurun):  bash tools/profile_r06.sh
# 1) rocprofv3 --kernel-trace --stats of the default bench (256^3 fp64, dual) and of the 512^3 stencil-PCG legs (primal, capped iterations; fp64 and fp32)
# 2) separate PMC passes (never combined with a trace domain other than --kernel-trace): FETCH_SIZE, WRITE_SIZE (HBM traffic), SQ / GRBM counters of Step 1
# The program itself follows `--` (python3 bench.py ...): no env / bash -c hop under the profiler.
set -x
# (the repository is found from this file, not from the environment; every path is quoted -- ADVICE r4)
R="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
[ -f "$R/bench.py" ] || { echo "bench.py not found under $R" >&2; exit 1; }
OUT="$R/gpurun_out/prof_r06"
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
B256=(python3 "$R/bench.py" --no-cpu-baseline --no-also)
B512=(python3 "$R/bench.py" --no-cpu-baseline --no-also --workload bunny_small_512_f64 --solver primal --precond none)
B512F=(python3 "$R/bench.py" --no-cpu-baseline --no-also --workload bunny_small_512_f32 --solver primal --precond none)
# BASELINE.json configs[2]: rocker.obj 512^3 -- the "HBM-roofline run": 12 612 constraint rows, two-level (A A^T)^-1 in every projection
R512=(python3 "$R/bench.py" --no-cpu-baseline --no-also --workload rocker_512_f64 --solver primal --precond none)
R512F=(python3 "$R/bench.py" --no-cpu-baseline --no-also --workload rocker_512_f32 --solver primal --precond none)
rocprofv3 --kernel-trace --stats -d "$OUT" -o bench256 -- "${B256[@]}" --steps 3 --warmup 1 > "$OUT/bench256.log" 2>&1
rocprofv3 --kernel-trace --stats -d "$OUT" -o pcg512 -- "${B512[@]}" --steps 1 --warmup 1 --max-iters 200 > "$OUT/pcg512.log" 2>&1
rocprofv3 --kernel-trace --stats -d "$OUT" -o pcg512f32 -- "${B512F[@]}" --steps 1 --warmup 1 --max-iters 200 > "$OUT/pcg512f32.log" 2>&1
rocprofv3 --kernel-trace --stats -d "$OUT" -o pcg512rocker -- "${R512[@]}" --steps 1 --warmup 1 --max-iters 200 > "$OUT/pcg512rocker.log" 2>&1
rocprofv3 --kernel-trace --stats -d "$OUT" -o pcg512rockerf32 -- "${R512F[@]}" --steps 1 --warmup 1 --max-iters 200 > "$OUT/pcg512rockerf32.log" 2>&1
rocprofv3 --kernel-trace --stats -d "$OUT" -o bench512 -- python3 "$R"/bench.py --no-cpu-baseline --no-also --workload bunny_small_512_f64 --steps 2 --warmup 1 > "$OUT/bench512.log" 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$OUT" -o pmc_fetch_pcg512rockerf32 -- "${R512F[@]}" --steps 1 --warmup 0 --max-iters 12 > "$OUT/pmc_fetch_pcg512rockerf32.log" 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$OUT" -o pmc_write_pcg512rockerf32 -- "${R512F[@]}" --steps 1 --warmup 0 --max-iters 12 > "$OUT/pmc_write_pcg512rockerf32.log" 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$OUT" -o pmc_fetch_512 -- python3 "$R"/bench.py --no-cpu-baseline --no-also --workload bunny_small_512_f64 --steps 1 --warmup 0 > "$OUT/pmc_fetch_512.log" 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$OUT" -o pmc_write_512 -- python3 "$R"/bench.py --no-cpu-baseline --no-also --workload bunny_small_512_f64 --steps 1 --warmup 0 > "$OUT/pmc_write_512.log" 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$OUT" -o pmc_fetch -- "${B256[@]}" --steps 1 --warmup 0 > "$OUT/pmc_fetch.log" 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$OUT" -o pmc_write -- "${B256[@]}" --steps 1 --warmup 0 > "$OUT/pmc_write.log" 2>&1
# Step 1 alone (no set-up kernels co-resident: device-wide counters sampled around a dispatch see every kernel running meanwhile)
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$OUT" -o pmc_fetch_conv -- python3 "$R"/tools/conv_only.py > "$OUT/pmc_fetch_conv.log" 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$OUT" -o pmc_write_conv -- python3 "$R"/tools/conv_only.py > "$OUT/pmc_write_conv.log" 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$OUT" -o pmc_fetch_pcg512 -- "${B512[@]}" --steps 1 --warmup 0 --max-iters 12 > "$OUT/pmc_fetch_pcg512.log" 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$OUT" -o pmc_write_pcg512 -- "${B512[@]}" --steps 1 --warmup 0 --max-iters 12 > "$OUT/pmc_write_pcg512.log" 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$OUT" -o pmc_fetch_pcg512f32 -- "${B512F[@]}" --steps 1 --warmup 0 --max-iters 12 > "$OUT/pmc_fetch_pcg512f32.log" 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$OUT" -o pmc_write_pcg512f32 -- "${B512F[@]}" --steps 1 --warmup 0 --max-iters 12 > "$OUT/pmc_write_pcg512f32.log" 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE -d "$OUT" -o pmc_sq1 -- python3 "$R"/tools/conv_only.py > "$OUT/pmc_sq1.log" 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_WAIT_ANY -d "$OUT" -o pmc_sq2 -- python3 "$R"/tools/conv_only.py > "$OUT/pmc_sq2.log" 2>&1
# round 5: the fp32 solve's Step 1 goes through the tiered kernel too -- its trace on configs[2]
rocprofv3 --kernel-trace --stats -d "$OUT" -o rocker512f32 -- python3 "$R"/bench.py --no-cpu-baseline --no-also --workload rocker_512_f32 --steps 2 --warmup 1 > "$OUT/rocker512f32.log" 2>&1
ls -la "$OUT"
for f in bench256 bench512 pcg512 pcg512f32 pcg512rocker pcg512rockerf32 rocker512f32; do python3 "$R"/profiles/rocpd_summary.py "$OUT/${f}_results.db" "$OUT/${f}_kernel_stats.txt" > /dev/null; done
python3 "$R"/tools/pmc_report.py "$OUT" "$OUT" > "$OUT/pmc_report.log" 2>&1
python3 "$R"/tools/timeline.py "$OUT/bench256_results.db" 60 > "$OUT/timeline_256.txt" 2>&1
tail -3 "$OUT/bench256.log" "$OUT/pcg512.log" "$OUT/pcg512f32.log" "$OUT/pcg512rocker.log" "$OUT/pcg512rockerf32.log" | cut -c1-400

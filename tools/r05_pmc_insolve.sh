cd /tmp && export TMPDIR=/tmp
R="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
[ -f "$R/bench.py" ] || { echo "bench.py not found under $R" >&2; exit 1; }
O="$R/gpurun_out/pmc_insolve"; rm -rf "$O"; mkdir -p "$O"
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_LDS -d "$O/solve" -o p -- python3 "$R/bench.py" --no-cpu-baseline --no-also --steps 2 --warmup 0 > "$O/solve.log" 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_LDS -d "$O/alone" -o p -- python3 "$R/tools/conv_only.py" > "$O/alone.log" 2>&1
python3 "$R/tools/r05_pmc_sum.py" "$O/solve"; python3 "$R/tools/r05_pmc_sum.py" "$O/alone"

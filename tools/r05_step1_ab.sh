#!/bin/bash
# Same-box A/B of Step-1 kernel builds (signed-heat-3d_amd/lib/variants/libshm_grid_<name>.so, built by tools/r05_build_variant.sh), three interleaved rounds, then the SQ
# counters of the kernel alone for the first two names.   bash tools/r05_step1_ab.sh <out-dir> <name> [<name> ...]    ("default" = the shipped library)
R="$(cd "$(dirname "$0")/.." && pwd)"
OUT="$(mkdir -p "$1" && cd "$1" && pwd)"; shift
: "${OUT:?usage: r05_step1_ab.sh out-dir name...}"
mkdir -p "$OUT"
V="$R/signed-heat-3d_amd/lib/variants"
args=()
for n in "$@"; do
  if [ "$n" = default ]; then args+=("default="); else args+=("$n=SHM_GRID_LIB=$V/libshm_grid_$n.so"); fi
done
for rep in 1 2 3; do
  python3 "$R/tools/ab.py" "${CASES:-bunny_small.obj:4:64}" "${args[@]}" >> "$OUT/ab.txt" 2>&1
done
cat "$OUT/ab.txt"
cd /tmp && export TMPDIR=/tmp
i=0
for n in "$@"; do
  i=$((i+1)); [ $i -gt "${PMC_N:-2}" ] && break
  if [ "$n" = default ]; then unset SHM_GRID_LIB; else export SHM_GRID_LIB="$V/libshm_grid_$n.so"; fi
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE SQ_INSTS_LDS -d "$OUT/pmc1_$n" -o p -- python3 "$R/tools/conv_only.py" > "$OUT/pmc1_$n.log" 2>&1
  rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU_TRANS SQ_WAIT_INST_LDS -d "$OUT/pmc2_$n" -o p -- python3 "$R/tools/conv_only.py" > "$OUT/pmc2_$n.log" 2>&1
  python3 "$R/tools/r05_pmc_sum.py" "$OUT/pmc1_$n" "$OUT/pmc2_$n" > "$OUT/counters_$n.txt" 2>&1
  cat "$OUT/counters_$n.txt"
done

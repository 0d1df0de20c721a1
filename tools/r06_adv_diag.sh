#!/bin/bash
R="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"; cd "$R" || exit 1
V="$R/signed-heat-3d_amd/lib/variants"
export SHM_DEBUG_KNOBS=1
for c in "cloud 128" "cloud 256"; do
  python3 tools/r06_adv_diag.py $c
  SHM_CONV_NO_SKIP=1 python3 tools/r06_adv_diag.py $c
  SHM_CONV_REDO_RATIO=1e-5 python3 tools/r06_adv_diag.py $c
  SHM_CONV_TIER_LOG=30 python3 tools/r06_adv_diag.py $c
  SHM_GRID_LIB="$V/libshm_grid_r05.so" python3 tools/r06_adv_diag.py $c
  SHM_GRID_LIB="$V/libshm_grid_r05.so" SHM_CONV_NO_SKIP=1 python3 tools/r06_adv_diag.py $c
done

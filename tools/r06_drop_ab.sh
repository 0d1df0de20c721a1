#!/bin/bash
# Round 6: the accumulated drop rule against the round-5 build (libshm_grid_r05.so, built from the round-5 sources)
#   bash tools/r06_drop_ab.sh [cases]
R="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
V="$R/signed-heat-3d_amd/lib/variants"
cd "$R" || exit 1
CASES="${1:-bunny_small.obj:4:64,rocker.obj:4:64,rocker.obj:5:32,SprayBottle.pc:4:32,SprayBottle.pc:5:32,rocker.obj:4:32}"
ARGS=("r06=")
[ -f "$V/libshm_grid_r05.so" ] && ARGS+=("r05=SHM_GRID_LIB=$V/libshm_grid_r05.so")

for extra in $R06_AB_VARIANTS; do ARGS+=("$extra=SHM_GRID_LIB=$V/libshm_grid_$extra.so"); done
for rep in 1 2; do python3 tools/ab.py "$CASES" "${ARGS[@]}"; done

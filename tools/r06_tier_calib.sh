#!/bin/bash
# Round 6 (late): the a-posteriori test's price of a packed-fp32 term re-examined with the sums flushed into fp64 (SHM_TIER_FLUSH), a price that grows with the exponent
# (SHM_TIER_U0), the price itself (SHM_CONV_REDO_RATIO = budget / eps_far) and the far threshold G (SHM_CONV_TIER_LOG): max|dY| against the all-fp64 arithmetic and the time
# of Step 1 for each setting.   bash tools/r06_tier_calib.sh  -> gpurun_out/tier_calib.txt
# (The sweep of profiles/r06_tier_calib.txt was run on the build BEFORE its outcome became the default -- there "base" is the flat 3e-6 without flush.  On the current build
#  the old behaviour is SHM_TIER_FLUSH=0;SHM_TIER_U0=0;SHM_CONV_REDO_RATIO=3.333e-3 and "base" is the adopted setting.)
R="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
cd "$R" || exit 1
O="$R/gpurun_out/tier_calib.txt"; : > "$O"
SETTINGS=(
 "base="
 "r05_flat3e-6_noflush=SHM_TIER_FLUSH=0;SHM_TIER_U0=0;SHM_CONV_REDO_RATIO=3.333e-3"
 "F256=SHM_TIER_FLUSH=256;SHM_TIER_U0=0;SHM_CONV_REDO_RATIO=3.333e-3"
 "F256_e1.5_u36=SHM_TIER_FLUSH=256;SHM_TIER_U0=36;SHM_CONV_REDO_RATIO=6.667e-3"
 "F256_e1.5_flat=SHM_TIER_FLUSH=256;SHM_CONV_REDO_RATIO=6.667e-3"
 "F256_e1_u24=SHM_TIER_FLUSH=256;SHM_TIER_U0=24;SHM_CONV_REDO_RATIO=1e-2"
 "F512_e1.5_u36=SHM_TIER_FLUSH=512;SHM_TIER_U0=36;SHM_CONV_REDO_RATIO=6.667e-3"
)
[ -n "$CALIB_SETTINGS" ] && IFS=' ' read -r -a SETTINGS <<< "$CALIB_SETTINGS"
for s in "${SETTINGS[@]}"; do
  name="${s%%=*}"; envs="${s#*=}"
  echo "== $name  [$envs]" >> "$O"
  ( IFS=';'; for e in $envs; do export "$e"; done; unset IFS; export SHM_DEBUG_KNOBS=1
    python3 tools/tier_robustness_big.py --cases ${CALIB_CASES:-SprayBottle.pc 6.0 knot.obj 6.0 SprayBottle.pc 5.0 knot.obj 5.0 rocker.obj 5.0 chair.obj 5.0 bunny_small.obj 5.0 bunny_small.obj 4.0 rocker.obj 4.0 knot.obj 4.0} 2>&1 | grep 'max|dY|' | cut -c1-175 ) >> "$O"
done
[ -z "$CALIB_NO_AB" ] && python3 tools/ab.py 'bunny_small.obj:4:64,bunny_small.obj:5:64,rocker.obj:4:64,rocker.obj:5:64,knot.obj:5:64,bunny.pc:5:64,SprayBottle.pc:5:64,chair.obj:5:64' "${SETTINGS[@]}" >> "$O" 2>&1

#!/bin/bash
# Every shipped data file at several grid sizes / precisions through the headless CLI: finite min/max, no error exit.
cd $(dirname $0)/..
CLI=signed-heat-3d_amd/bin/shm_grid_cli
for f in data/bunny_small.obj data/polygon-bear.obj data/rocker.obj data/chair.obj data/knot.obj data/bunny.pc data/rocker.pc data/chair.pc data/knot.pc data/SprayBottle.pc; do
  for h in 0 1 2 3 4; do
    for p in "" "--fp32"; do
      if [ "$h" = "4" ] && [ "$p" = "" ] && [ "$f" = "data/SprayBottle.pc" ]; then continue; fi
      out=$( $CLI $f --g --V --h $h $p 2>&1 | grep -E "min:|error|gfx950|Solve time" | tr '\n' ' ')
      echo "$f h=$h $p :: $out"
    done
  done
done

#!/bin/bash
# One box, every combination of the round-5 schedule changes of the plain stencil CG (rocker.obj 512^3): two-level projector in 3 launches vs 5 (SHM_TL_CLASSIC),
# p.Kp summed by the RES sweep vs finalize_sum_kernel (SHM_CG_NO_PQFOLD), x on half the grid every iteration vs all of it every other (SHM_CG_NO_XSPLIT).
R="$(cd "$(dirname "$0")/.." && pwd)"; cd "$R"
export SHM_DEBUG_KNOBS=1 SHM_PROBE_QUICK=1
for rep in 1 2; do
for tl in 0 1; do for pq in 0 1; do for xs in 0 1; do
  env $( [ $tl = 1 ] && echo SHM_TL_CLASSIC=1 ) $( [ $pq = 1 ] && echo SHM_CG_NO_PQFOLD=1 ) $( [ $xs = 1 ] && echo SHM_CG_NO_XSPLIT=1 ) python3 tools/r05_proj_probe.py "tl5=$tl nofold=$pq nosplit=$xs"
done; done; done; done

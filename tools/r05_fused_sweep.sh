#!/bin/bash
# Tile-shape sweep of the fused stencil-CG sweeps on the 512^3 legs (rocker.obj, fp32 and fp64): rows per lane, waves per workgroup, planes per chunk.
R="$(cd "$(dirname "$0")/.." && pwd)"; cd "$R"
export SHM_DEBUG_KNOBS=1 SHM_PROBE_QUICK=1
python3 tools/r05_proj_probe.py "default"
for ry in 2 4; do for nw in 4 8 16; do for zc in 32 64 128; do
  SHM_FUSED_RY=$ry SHM_FUSED_WAVES=$nw SHM_FUSED_ZC=$zc python3 tools/r05_proj_probe.py "ry=$ry nw=$nw zc=$zc"
done; done; done
python3 tools/r05_proj_probe.py "default"

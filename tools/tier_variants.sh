#!/bin/bash
# Build A/B variants of libshm_grid.so for the tiered fp64 Step 1 (csrc/shm_conv_tiered.hip.h build-time knobs) into signed-heat-3d_amd/lib/variants/ ; run with
#   SHM_GRID_LIB=signed-heat-3d_amd/lib/variants/libshm_grid_<name>.so python tools/tier_sweep.py data/bunny_small.obj 4 exact 7
cd $(dirname $0)/../signed-heat-3d_amd/csrc
mkdir -p ../lib/variants
build() { name=$1; shift; /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-parameter "$@" -shared shm_grid.hip -o ../lib/variants/libshm_grid_$name.so -ldl 2>&1 | grep -E "error" ; echo built $name; }
build u4 &
build u3 -DSHM_CONV32_UNROLL=3 &
build u2 -DSHM_CONV32_UNROLL=2 &
wait
ls -la ../lib/variants

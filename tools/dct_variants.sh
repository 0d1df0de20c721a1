#!/bin/bash
# Build A/B variants of libshm_grid.so (transform-kernel build-time knobs) into signed-heat-3d_amd/lib/variants/ ; run with
#   SHM_GRID_LIB=signed-heat-3d_amd/lib/variants/libshm_grid_<name>.so python bench.py ...
cd $(dirname $0)/../signed-heat-3d_amd/csrc
mkdir -p ../lib/variants
build() { name=$1; shift; /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-parameter "$@" -shared shm_grid.hip -o ../lib/variants/libshm_grid_$name.so -ldl 2>&1 | grep -E "error" ; echo built $name; }
build v0 &
build hint -DSHM_DCT_WAVES_HINT=1 &
build lc4 -DSHM_DCT_LC8=4 &
build lc4_hint -DSHM_DCT_LC8=4 -DSHM_DCT_WAVES_HINT=1 &
wait
build hoist_hint -DSHM_DCT_HOIST=1 -DSHM_DCT_WAVES_HINT=1 &
build twg_hint -DSHM_DCT_TW_LDS_MAX=7 -DSHM_DCT_WAVES_HINT=1 &
build lc4_twg_hint -DSHM_DCT_LC8=4 -DSHM_DCT_TW_LDS_MAX=7 -DSHM_DCT_WAVES_HINT=1 &
wait
ls -la ../lib/variants

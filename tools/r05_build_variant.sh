#!/bin/bash
# Build an A/B variant of libshm_grid.so with extra -D flags into signed-heat-3d_amd/lib/variants/libshm_grid_<name>.so (what SHM_GRID_LIB points tools/ab.py at)
# and print the compiler's resource report of the Step-1 kernel.   bash tools/r05_build_variant.sh <name> [-DSHM_TIER_NEAR_BATCH=2 ...]
# (round 6: through the library's own Makefile -- three translation units in parallel -- with its output directory redirected)
R="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
name="${1:?usage: r05_build_variant.sh name [-D...]}"; shift
T="$(mktemp -d)"
mkdir -p "$R/signed-heat-3d_amd/lib/variants"
make -s -C "$R/signed-heat-3d_amd/csrc" OUT="$T" HIPFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-parameter -Wno-unused-function $*" || exit 1
cp "$T/libshm_grid.so" "$R/signed-heat-3d_amd/lib/variants/libshm_grid_$name.so"
grep -A12 "Function Name: _ZN3shm18conv_tiered_kernelILi4EdLb1" "$T/kernel_resources.txt" | grep -E " VGPRs:|SGPRs Spill|VGPRs Spill|LDS Size" | sed 's/.*remark: *//' | tr '\n' ' '; echo " [$name]"
rm -rf "$T"

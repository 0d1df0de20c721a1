#!/bin/bash
# Build an A/B variant of libshm_grid.so with extra -D flags into signed-heat-3d_amd/lib/variants/libshm_grid_<name>.so (what SHM_GRID_LIB points tools/ab.py at)
# and print the compiler's resource report of the Step-1 kernel.   bash tools/r05_build_variant.sh <name> [-DSHM_TIER_NEAR_BATCH=2 ...]
R="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
name="${1:?usage: r05_build_variant.sh name [-D...]}"; shift
T="$(mktemp -d)"
mkdir -p "$R/signed-heat-3d_amd/lib/variants"
cd "$R/signed-heat-3d_amd/csrc" || exit 1
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-parameter "$@" -Rpass-analysis=kernel-resource-usage -shared shm_grid.hip -o "$T/lib.so" -ldl 2> "$T/res.txt" || { grep -E "error" -A3 "$T/res.txt"; exit 1; }
cp "$T/lib.so" "$R/signed-heat-3d_amd/lib/variants/libshm_grid_$name.so"
grep -A12 "Function Name: _ZN3shm18conv_tiered_kernelILi4EdLb1" "$T/res.txt" | grep -E " VGPRs:|SGPRs Spill|VGPRs Spill|LDS Size" | sed 's/.*remark: *//' | tr '\n' ' '; echo " [$name]"
rm -rf "$T"

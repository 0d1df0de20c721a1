#!/bin/bash
# CPU sanitizer run (AddressSanitizer + UBSan; the reference builds Debug with -fsanitize=address, CMakeLists.txt:32): builds the C++ host mirror, the headless
# CLI and the C oracle with `make SAN=1` and runs the CPU test suite on those builds (python with libasan preloaded; no GPU involved -- GPU ASan is not
# available on this pool).  Leak checking is off: the interpreter itself never frees most of what it allocates.
#     bash tools/san_check.sh            # exit code of pytest; sanitizer reports abort the run (halt_on_error)
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
make -C $R/oracle SAN=1 > /dev/null
make -C $R/signed-heat-3d_amd/host SAN=1 > /dev/null
ASAN=$(gcc -print-file-name=libasan.so)
UBSAN=$(gcc -print-file-name=libubsan.so)
export ASAN_OPTIONS=detect_leaks=0:halt_on_error=1:abort_on_error=1:verify_asan_link_order=0
export UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1
export SHM_HOST_LIB=$R/signed-heat-3d_amd/lib/san/libshm_host.so
export SHM_ORACLE_LIB=$R/oracle/_build/san/libshm_oracle.so
export SHM_CLI=$R/signed-heat-3d_amd/bin/san/shm_grid_cli
cd $R
# the CLI on its own first (no interpreter in the way): argument parsing, the loaders, the pre-processing; it stops at shm_grid_create without a GPU
LD_PRELOAD="$ASAN $UBSAN" $SHM_CLI data/bunny_small.obj --g --h 0 > /tmp/san_cli.log 2>&1 || true
if grep -E "AddressSanitizer|runtime error" /tmp/san_cli.log; then echo "sanitizer report in the CLI run"; exit 1; fi
LD_PRELOAD="$ASAN $UBSAN" python -m pytest tests -q -x -m "not gpu" -p no:cacheprovider "$@"

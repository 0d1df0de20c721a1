"""signed-heat-3d_amd -- MI355X (gfx950) implementation of the regular-grid Signed Heat Method solver.

The directory name contains '-' (it mirrors the reference's repository name), so import it with
`shm_import.load()` (repo root) or by file path; inside, the modules are plain relative imports.

Layout:
  csrc/   hand-written HIP kernels + the C ABI of include/shm_grid.h      -> lib/libshm_grid.so
  host/   C++ mirror of the reference's SignedHeatGridSolver / SignedHeat3DOptions surface,
          OBJ/.pc loaders, headless CLI                                   -> lib/libshm_host.so, bin/shm_grid_cli
  grid_abi.py   ctypes binding of the C ABI (tests, bench.py)
  host_abi.py   ctypes binding of the C++ host layer's flat wrapper
"""
from .grid_abi import (GridSolver, ShmError, ShmStats, lib_path, load_library, SHM_F32, SHM_F64,  # noqa: F401
                       plan_slab, comm_unique_id, step1_plane_weights, plan_slab_weighted)

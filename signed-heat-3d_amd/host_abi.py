"""ctypes binding of the C++ host layer's flat wrapper (host/host_c_api.cpp -> lib/libshm_host.so).

The host layer is the C++ mirror of the reference's SignedHeatGridSolver / SignedHeat3DOptions surface
(signed-heat-3d_amd/host/).  This module only marshals arguments.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from .grid_abi import ShmStats, load_library

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def load_host_library():
    global _LIB
    if _LIB is not None:
        return _LIB
    load_library()  # libshm_grid.so first (libshm_host.so links it through $ORIGIN)
    path = os.environ.get("SHM_HOST_LIB") or os.path.join(_HERE, "lib", "libshm_host.so")   # SHM_HOST_LIB: another build (tools/san_check.sh: the sanitizer build)
    if not os.path.exists(path):
        raise OSError("libshm_host.so not built (%s): run __graft_entry__.build()" % path)
    lib = C.CDLL(path)
    lib.shmh_last_error.restype = C.c_char_p
    lib.shmh_new.restype = C.c_void_p
    lib.shmh_new.argtypes = [C.c_int, C.c_int, C.c_double, C.c_int, C.c_int, C.c_int]
    lib.shmh_delete.argtypes = [C.c_void_p]
    lib.shmh_delete.restype = None
    lib.shmh_load.argtypes = [C.c_void_p, C.c_char_p]
    lib.shmh_counts.argtypes = [C.c_void_p, C.c_void_p]
    lib.shmh_counts.restype = None
    lib.shmh_set_point_areas.argtypes = [C.c_void_p, C.c_void_p, C.c_double]
    lib.shmh_preprocess.argtypes = [C.c_void_p, C.c_double, C.c_double, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                    C.POINTER(C.c_int64)]
    lib.shmh_compute_distance.argtypes = [C.c_void_p, C.c_double, C.c_double, C.c_double, C.c_int, C.c_int, C.c_void_p, C.POINTER(ShmStats)]
    lib.shmh_grid_info.argtypes = [C.c_void_p, C.c_void_p]
    lib.shmh_grid_info.restype = None
    _LIB = lib
    return lib


class HostSolver:
    """SignedHeatGridSolver (C++ mirror) + a loaded mesh / point cloud."""

    def __init__(self, path=None, device=0, precision=64, tol=0.0, max_iters=0, local_slabs=1, verbose=False):
        self._lib = load_host_library()
        self._h = C.c_void_p(self._lib.shmh_new(device, precision, tol, max_iters, local_slabs, int(verbose)))
        if path is not None:
            self.load(path)

    def close(self):
        if getattr(self, "_h", None):
            self._lib.shmh_delete(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc != 0:
            raise RuntimeError(self._lib.shmh_last_error().decode())

    def load(self, path):
        self._chk(self._lib.shmh_load(self._h, os.fsencode(path)))

    def counts(self):
        c = np.zeros(2, dtype=np.int64)
        self._lib.shmh_counts(self._h, c.ctypes.data)
        return int(c[0]), int(c[1])

    def set_point_areas(self, areas, h):
        a = np.ascontiguousarray(areas, dtype=np.float64)
        self._chk(self._lib.shmh_set_point_areas(self._h, a.ctypes.data, float(h)))

    def preprocess(self, tCoef=1.0, hCoef=0.0, scale=2.0, arrays=True):
        """Host pre-processing only (no GPU).  Returns a dict with centroid, radius, h, lam, n, bbox_min, cell
        and (arrays=True) pos [S,3], wnormal [S,3], area [S]."""
        out = np.zeros(11)
        S = C.c_int64()
        self._chk(self._lib.shmh_preprocess(self._h, tCoef, hCoef, scale, out.ctypes.data, None, None, None, C.byref(S)))
        res = dict(centroid=out[0:3].copy(), radius=out[3], h=out[4], lam=out[5], n=int(out[6]), bbox_min=out[7:10].copy(), cell=out[10],
                   S=S.value)
        if arrays:
            pos = np.zeros((S.value, 3))
            wn = np.zeros((S.value, 3))
            area = np.zeros(S.value)
            self._chk(self._lib.shmh_preprocess(self._h, tCoef, hCoef, scale, out.ctypes.data, pos.ctypes.data, wn.ctypes.data,
                                                area.ctypes.data, C.byref(S)))
            res.update(pos=pos, wnormal=wn, area=area)
        return res

    def grid_info(self):
        """Grid block the solver object holds right now: dict(n, bbox_min, bbox_max, cell); n == 0 before the first build."""
        out = np.zeros(8)
        self._lib.shmh_grid_info(self._h, out.ctypes.data)
        return dict(n=int(out[0]), bbox_min=out[1:4].copy(), bbox_max=out[4:7].copy(), cell=float(out[7]))

    def compute_distance(self, tCoef=1.0, hCoef=0.0, scale=2.0, rebuild=True, fast=False):
        # with rebuild=False a mesh solve keeps the previous call's grid (signed_heat_grid_solver.cpp:8): size the buffer for either
        n = max(int(2 * 2.0 ** (hCoef + 3)), self.grid_info()["n"])
        phi = np.empty(n ** 3, dtype=np.float64)
        st = ShmStats()
        self._chk(self._lib.shmh_compute_distance(self._h, tCoef, hCoef, scale, int(rebuild), int(fast), phi.ctypes.data, C.byref(st)))
        return phi[:self.grid_info()["n"] ** 3], st

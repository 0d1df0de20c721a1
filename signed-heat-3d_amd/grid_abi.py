"""ctypes binding of include/shm_grid.h (libshm_grid.so).  Plumbing only: no arithmetic happens here.

There is deliberately no CPU fallback: if the shared library is missing, or no HIP device is visible,
construction raises (ShmError / OSError).
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

SHM_F64, SHM_F32 = 64, 32
_HERE = os.path.dirname(os.path.abspath(__file__))

STATUS_NAMES = {0: "SHM_OK", 1: "SHM_ERR_INVALID", 2: "SHM_ERR_HIP", 3: "SHM_ERR_NOMEM", 4: "SHM_ERR_BREAKDOWN",
                5: "SHM_ERR_NOCONV", 6: "SHM_ERR_RCCL", 7: "SHM_ERR_STATE", 8: "SHM_ERR_SINGULAR"}

# every symbol include/shm_grid.h declares (tests check the library exports all of them)
ABI_SYMBOLS = ["shm_grid_owned_planes", "shm_grid_create", "shm_grid_destroy", "shm_grid_last_error", "shm_grid_abi_version", "shm_grid_set_problem",
               "shm_grid_solve", "shm_grid_get_phi", "shm_grid_compute_distance", "shm_grid_run_conv", "shm_grid_run_conv_arith", "shm_grid_run_divergence",
               "shm_grid_get_field", "shm_grid_get_field_planes", "shm_grid_apply_laplacian", "shm_grid_get_constraints", "shm_grid_get_schur", "shm_grid_apply_projector", "shm_grid_apply_preconditioner", "shm_grid_isosurface", "shm_grid_isosurface_ex", "shm_grid_get_isosurface",
               "shm_comm_unique_id", "shm_plan_slab", "shm_step1_plane_weights", "shm_plan_slab_weighted"]


class ShmError(RuntimeError):
    def __init__(self, status, message):
        super().__init__("%s: %s" % (STATUS_NAMES.get(status, status), message))
        self.status = status


class _Config(C.Structure):
    _fields_ = [("device", C.c_int32), ("precision", C.c_int32), ("local_slabs", C.c_int32), ("rank", C.c_int32),
                ("world", C.c_int32), ("verbose", C.c_int32), ("rccl_unique_id", C.c_void_p), ("slab_plan", C.c_int32)]


class _Sources(C.Structure):
    _fields_ = [("S", C.c_int64), ("pos", C.c_void_p), ("wnormal", C.c_void_p), ("area", C.c_void_p), ("lam", C.c_double)]


class _Grid(C.Structure):
    _fields_ = [("n", C.c_int32), ("bbox_min", C.c_double * 3), ("cell", C.c_double)]


class _Opts(C.Structure):
    _fields_ = [("fast_integration", C.c_int32), ("scrub_nonfinite", C.c_int32), ("tol", C.c_double), ("max_iters", C.c_int32),
                ("check_every", C.c_int32), ("preconditioner", C.c_int32), ("solver", C.c_int32), ("step1_arith", C.c_int32),
                ("dual_form", C.c_int32), ("step1_budget", C.c_double)]


class ShmStats(C.Structure):
    _fields_ = [("n", C.c_int32), ("m", C.c_int32), ("S", C.c_int64), ("iters", C.c_int32), ("rel_residual", C.c_double),
                ("shift", C.c_double), ("ms_conv", C.c_double), ("ms_div", C.c_double), ("ms_setup", C.c_double), ("ms_wait_setup", C.c_double),
                ("ms_pcg", C.c_double), ("ms_shift", C.c_double), ("ms_total", C.c_double), ("ms_stencil_avg", C.c_double),
                ("ms_update_xr_avg", C.c_double), ("ms_project_avg", C.c_double), ("ms_update_p_avg", C.c_double),
                ("ms_precond_avg", C.c_double), ("kernel_samples", C.c_int32), ("preconditioner", C.c_int32),
                ("solver", C.c_int32), ("bytes_per_iter", C.c_double), ("cg_form", C.c_int32), ("pairs_fp64", C.c_double), ("pairs_fp32", C.c_double), ("conv_launches", C.c_int32), ("pairs_redone", C.c_double)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


def lib_path():
    """In-tree library; SHM_GRID_LIB selects another build of it (tools/dct_variants.sh A/B runs)."""
    return os.environ.get("SHM_GRID_LIB") or os.path.join(_HERE, "lib", "libshm_grid.so")


_LIB = None


def load_library():
    """Load libshm_grid.so (built in-tree by __graft_entry__.build() / make -C signed-heat-3d_amd/csrc)."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = lib_path()
    if not os.path.exists(path):
        raise OSError("libshm_grid.so not built: run `python -c 'import __graft_entry__ as g; g.build()'` (%s)" % path)
    lib = C.CDLL(path)
    lib.shm_grid_last_error.restype = C.c_char_p
    lib.shm_grid_last_error.argtypes = [C.c_void_p]
    lib.shm_grid_create.argtypes = [C.POINTER(_Config), C.POINTER(C.c_void_p)]
    lib.shm_grid_destroy.argtypes = [C.c_void_p]
    lib.shm_grid_destroy.restype = None
    lib.shm_grid_set_problem.argtypes = [C.c_void_p, C.POINTER(_Sources), C.POINTER(_Grid)]
    lib.shm_grid_solve.argtypes = [C.c_void_p, C.POINTER(_Opts), C.POINTER(ShmStats)]
    lib.shm_grid_get_phi.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    lib.shm_grid_compute_distance.argtypes = [C.c_void_p, C.POINTER(_Sources), C.POINTER(_Grid), C.POINTER(_Opts), C.c_void_p,
                                              C.POINTER(ShmStats)]
    lib.shm_grid_run_conv.argtypes = [C.c_void_p]
    lib.shm_grid_run_conv_arith.argtypes = [C.c_void_p, C.c_int32]
    lib.shm_grid_run_divergence.argtypes = [C.c_void_p, C.c_int32]
    lib.shm_grid_get_field.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    lib.shm_grid_get_field_planes.argtypes = [C.c_void_p, C.c_int, C.c_int32, C.c_int32, C.c_void_p]
    lib.shm_grid_apply_laplacian.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    lib.shm_grid_get_constraints.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int32)]
    lib.shm_grid_get_schur.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_int32)]
    lib.shm_grid_apply_projector.argtypes = [C.c_void_p, C.c_void_p]
    lib.shm_grid_apply_preconditioner.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    lib.shm_grid_isosurface.argtypes = [C.c_void_p, C.c_double, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    lib.shm_grid_isosurface_ex.argtypes = [C.c_void_p, C.c_double, C.c_int32, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    lib.shm_grid_get_isosurface.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    lib.shm_grid_owned_planes.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    lib.shm_comm_unique_id.argtypes = [C.c_void_p]
    lib.shm_plan_slab.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    lib.shm_plan_slab.restype = None
    lib.shm_step1_plane_weights.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]
    lib.shm_plan_slab_weighted.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    lib.shm_plan_slab_weighted.restype = None
    _LIB = lib
    return lib


def plan_slab(n, nslabs, slab):
    lib = load_library()
    k0, k1 = C.c_int32(), C.c_int32()
    lib.shm_plan_slab(n, nslabs, slab, C.byref(k0), C.byref(k1))
    return k0.value, k1.value


def step1_plane_weights(pos, wnormal, lam, n, bbox_min, cell, precision=SHM_F64):
    """Relative Step-1 cost of every z-plane (shm_step1_plane_weights: pure host logic, no GPU)."""
    lib = load_library()
    pos, wn = _f64(pos).reshape(-1), _f64(wnormal).reshape(-1)
    area = np.zeros(pos.size // 3)
    src = _Sources(pos.size // 3, pos.ctypes.data, wn.ctypes.data, area.ctypes.data, float(lam))
    g = _Grid(int(n), (C.c_double * 3)(*[float(x) for x in bbox_min]), float(cell))
    w = np.zeros(int(n))
    rc = lib.shm_step1_plane_weights(C.byref(src), C.byref(g), int(precision), w.ctypes.data)
    if rc != 0:
        raise ShmError(rc, "shm_step1_plane_weights: invalid argument")
    return w


def plan_slab_weighted(n, nslabs, slab, weights, granule):
    lib = load_library()
    w = _f64(weights)
    k0, k1 = C.c_int32(), C.c_int32()
    lib.shm_plan_slab_weighted(int(n), int(nslabs), int(slab), w.ctypes.data, int(granule), C.byref(k0), C.byref(k1))
    return k0.value, k1.value


def comm_unique_id():
    lib = load_library()
    buf = C.create_string_buffer(128)
    rc = lib.shm_comm_unique_id(buf)
    if rc != 0:
        raise ShmError(rc, lib.shm_grid_last_error(None).decode())
    return buf.raw


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


class GridSolver:
    """Thin handle around shm_solver*.  Mirrors the C ABI one to one."""

    FIELD_Y0, FIELD_Y1, FIELD_Y2, FIELD_DIV, FIELD_PHI = 0, 1, 2, 3, 4

    def __init__(self, device=0, precision=SHM_F64, local_slabs=1, rank=0, world=1, verbose=False, rccl_unique_id=None, slab_plan=0):
        self._lib = load_library()
        self._h = C.c_void_p()
        self._uid = C.create_string_buffer(rccl_unique_id, 128) if rccl_unique_id is not None else None
        cfg = _Config(device, precision, local_slabs, rank, world, int(verbose),
                      C.cast(self._uid, C.c_void_p) if self._uid is not None else None, int(slab_plan))
        rc = self._lib.shm_grid_create(C.byref(cfg), C.byref(self._h))
        if rc != 0:
            raise ShmError(rc, self._lib.shm_grid_last_error(None).decode())
        self.n = 0
        self.world, self.rank, self.local_slabs = world, rank, local_slabs
        self._keep = None

    def close(self):
        if getattr(self, "_h", None):
            self._lib.shm_grid_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc, allow=()):
        if rc != 0 and rc not in allow:
            raise ShmError(rc, self._lib.shm_grid_last_error(self._h).decode())
        return rc

    def set_problem(self, pos, wnormal, area, lam, n, bbox_min, cell):
        pos, wnormal, area = _f64(pos).reshape(-1), _f64(wnormal).reshape(-1), _f64(area).reshape(-1)
        S = area.shape[0]
        assert pos.shape[0] == 3 * S and wnormal.shape[0] == 3 * S
        src = _Sources(S, pos.ctypes.data, wnormal.ctypes.data, area.ctypes.data, float(lam))
        g = _Grid(int(n), (C.c_double * 3)(*[float(v) for v in bbox_min]), float(cell))
        self._keep = (pos, wnormal, area)
        self._chk(self._lib.shm_grid_set_problem(self._h, C.byref(src), C.byref(g)))
        self.n, self.S = int(n), S

    def owned_planes(self):
        """z-planes [k0, k1) this handle owns under the slab plan in force (shm_grid_owned_planes: the weighted plan is not the equal-plane one)."""
        k0, k1 = C.c_int32(), C.c_int32()
        self._chk(self._lib.shm_grid_owned_planes(self._h, C.byref(k0), C.byref(k1)))
        return k0.value, k1.value

    PRECOND = {"auto": 0, "none": 1, "dct": 2}

    SOLVER = {"auto": 0, "primal": 1, "dual": 2, "dual_slabs": 3}

    STEP1 = {"auto": 0, "exact_f64": 1}

    DUAL_FORM = {"auto": 0, "direct": 1, "explicit_s_cg": 2, "through_grid": 3}

    def solve(self, tol=0.0, max_iters=0, check_every=0, scrub=True, fast=False, allow_noconv=False, precond="auto", solver="auto", step1="auto",
              dual_form="auto", step1_budget=0.0):
        o = _Opts(int(fast), int(scrub), float(tol), int(max_iters), int(check_every), self.PRECOND[precond], self.SOLVER[solver], self.STEP1[step1],
                  self.DUAL_FORM[dual_form], float(step1_budget))
        st = ShmStats()
        self._chk(self._lib.shm_grid_solve(self._h, C.byref(o), C.byref(st)), allow=(5,) if allow_noconv else ())
        return st

    def _owned_count(self):
        k0, k1 = self.owned_planes()
        return (k1 - k0) * self.n * self.n

    def get_phi(self):
        out = np.empty(self._owned_count(), dtype=np.float64)
        k0, k1 = C.c_int32(), C.c_int32()
        self._chk(self._lib.shm_grid_get_phi(self._h, out.ctypes.data, C.byref(k0), C.byref(k1)))
        return out, (k0.value, k1.value)

    def run_conv(self, step1="auto"):
        if step1 == "auto":
            self._chk(self._lib.shm_grid_run_conv(self._h))
        else:
            self._chk(self._lib.shm_grid_run_conv_arith(self._h, self.STEP1[step1]))

    def run_divergence(self, scrub=True):
        self._chk(self._lib.shm_grid_run_divergence(self._h, int(scrub)))

    def get_field(self, which):
        out = np.empty(self._owned_count(), dtype=np.float64)
        self._chk(self._lib.shm_grid_get_field(self._h, int(which), out.ctypes.data))
        return out

    def get_field_planes(self, which, k_begin, k_end):
        """z-planes [k_begin, k_end) of a field (shm_grid_get_field_planes): (k_end - k_begin) * n * n values."""
        out = np.empty((int(k_end) - int(k_begin)) * self.n * self.n, dtype=np.float64)
        self._chk(self._lib.shm_grid_get_field_planes(self._h, int(which), int(k_begin), int(k_end), out.ctypes.data))
        return out

    def apply_laplacian(self, u):
        u = _f64(u).reshape(-1)
        out = np.empty_like(u)
        self._chk(self._lib.shm_grid_apply_laplacian(self._h, u.ctypes.data, out.ctypes.data))
        return out

    def get_constraints(self):
        nodes = np.empty(8 * self.S, dtype=np.int64)
        coeffs = np.empty(8 * self.S, dtype=np.float64)
        m = C.c_int32()
        self._chk(self._lib.shm_grid_get_constraints(self._h, nodes.ctypes.data, coeffs.ctypes.data, C.byref(m)))
        return nodes[:8 * m.value].reshape(-1, 8).copy(), coeffs[:8 * m.value].reshape(-1, 8).copy()

    def get_schur(self):
        """The explicit m x m Schur complement A K^+ A^T of the dual solver (ShmError SHM_ERR_STATE when the problem does not qualify)."""
        m = len(self.get_constraints()[0])
        out = np.empty((m, m), dtype=np.float64)
        mm = C.c_int32(0)
        self._chk(self._lib.shm_grid_get_schur(self._h, out.ctypes.data, C.byref(mm)))
        assert mm.value == m
        return out

    def apply_projector(self, v):
        v = _f64(v).reshape(-1).copy()
        self._chk(self._lib.shm_grid_apply_projector(self._h, v.ctypes.data))
        return v

    def apply_preconditioner(self, v):
        v = _f64(v).reshape(-1)
        out = np.empty_like(v)
        self._chk(self._lib.shm_grid_apply_preconditioner(self._h, v.ctypes.data, out.ctypes.data))
        return out

    ISO_METHOD = {"marching_cubes": 0, "marching_tets": 1}

    def isosurface(self, isovalue=0.0, method="marching_cubes"):
        """Isosurface of the resident phi: (vertices [nv,3] float64, triangles [nt,3] int64).  method: "marching_cubes" (the demo's contour,
        src/main.cpp:121-124) or "marching_tets" (Kuhn split)."""
        nv, nt = C.c_int64(), C.c_int64()
        if method == "marching_cubes":
            self._chk(self._lib.shm_grid_isosurface(self._h, float(isovalue), C.byref(nv), C.byref(nt)))
        else:
            self._chk(self._lib.shm_grid_isosurface_ex(self._h, float(isovalue), self.ISO_METHOD[method], C.byref(nv), C.byref(nt)))
        V = np.empty((nv.value, 3), dtype=np.float64)
        F = np.empty((nt.value, 3), dtype=np.int64)
        self._chk(self._lib.shm_grid_get_isosurface(self._h, V.ctypes.data, F.ctypes.data))
        return V, F

    def compute_distance(self, pos, wnormal, area, lam, n, bbox_min, cell, **kw):
        """One-shot convenience mirroring shm_grid_compute_distance (set_problem + solve + get_phi)."""
        self.set_problem(pos, wnormal, area, lam, n, bbox_min, cell)
        st = self.solve(**kw)
        phi, _ = self.get_phi()
        return phi, st

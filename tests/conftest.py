import ctypes
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
# The library reads its experiment knobs (SHM_CONV_EXACT, SHM_DUAL_NO_DIRECT, SHM_GJ_CLASSIC, ... : alternative forms the tests hold against each other) only when
# this is set; a product run never sees them (csrc/shm_host.hip.h `knob`).  Test processes and the children they spawn inherit it.
os.environ.setdefault("SHM_DEBUG_KNOBS", "1")
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def shm():
    import shm_import
    return shm_import.load()


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


@pytest.fixture(scope="session")
def oracle_c():
    """ctypes handle of the C oracle (test infrastructure); built on demand with gcc."""
    so = os.environ.get("SHM_ORACLE_LIB") or os.path.join(ROOT, "oracle", "_build", "libshm_oracle.so")   # SHM_ORACLE_LIB: the sanitizer build (tools/san_check.sh)
    if not os.path.exists(so):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])
    lib = ctypes.CDLL(so)
    f64 = np.ctypeslib.ndpointer(np.float64, flags="C")
    i64 = np.ctypeslib.ndpointer(np.int64, flags="C")
    ci, cd = ctypes.c_int, ctypes.c_double
    lib.shmo_conv_normalize.argtypes = [ci, f64, cd, ci, f64, f64, cd, ci, ci, f64]
    lib.shmo_conv_normalize.restype = None
    lib.shmo_conv_normalize_planes.argtypes = [ci, f64, cd, ci, f64, f64, cd, ci, ci, f64]
    lib.shmo_conv_normalize_planes.restype = None
    lib.shmo_divergence.argtypes = [ci, cd, f64, ci, f64]
    lib.shmo_divergence.restype = None
    lib.shmo_laplacian_apply.argtypes = [ci, cd, f64, f64]
    lib.shmo_laplacian_apply.restype = None
    lib.shmo_constraint_rows.argtypes = [ci, f64, cd, ci, f64, i64, f64]
    lib.shmo_constraint_rows.restype = ci
    lib.shmo_source_average.argtypes = [ci, f64, cd, f64, ci, f64, f64]
    lib.shmo_source_average.restype = cd
    lib.shmo_constrained_solve.argtypes = [ci, cd, f64, ci, i64, f64, cd, ci, f64, f64]
    lib.shmo_constrained_solve.restype = ci
    lib.shmo_integrate_greedily.argtypes = [ci, f64, cd, f64, f64]
    lib.shmo_integrate_greedily.restype = None
    lib.shmo_compute_distance.argtypes = [ci, f64, cd, ci, f64, f64, f64, cd, ci, ci, cd, ci, f64, f64]
    lib.shmo_compute_distance.restype = ci
    lib.shmo_set_threads.argtypes = [ci]
    lib.shmo_set_threads.restype = None
    lib.shmo_max_threads.restype = ci
    lib.shmo_set_threads(min(8, os.cpu_count() or 1))  # tiny problems: hundreds of OpenMP threads would only spin
    return lib


def c_(a, dt=np.float64):
    return np.ascontiguousarray(a, dtype=dt)

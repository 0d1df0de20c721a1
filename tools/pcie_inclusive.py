#!/usr/bin/env python3
"""PCIe-inclusive rate (never bench.py's `value`): the one-shot drop-in call, i.e. host pre-processing + H2D of the sources +
device allocation on first use + solve + D2H of phi, through the C++ host mirror's computeDistance()."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import shm_import  # noqa: E402

shm = shm_import.load()
from signed_heat_3d_amd.host_abi import HostSolver  # noqa: E402

out = {}
for name, path, hc in (("bunny_small_256_f64", "data/bunny_small.obj", 4.0), ("bunny_small_512_f64", "data/bunny_small.obj", 5.0)):
    h = HostSolver(os.path.join(ROOT, path))
    n = int(2 * 2 ** (hc + 3))
    import ctypes as C
    from signed_heat_3d_amd.grid_abi import ShmStats
    ts = []
    st = ShmStats()
    sec = C.c_double()
    h._lib.shmh_time_compute_distance.argtypes = [C.c_void_p, C.c_double, C.c_double, C.c_double, C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(ShmStats)]
    for rep in range(4):
        # the C++ call itself (computeDistance() returning its vector, as main.cpp:90-91 consumes it), timed inside the library
        rc = h._lib.shmh_time_compute_distance(h._h, 1.0, hc, 2.0, int(rep == 0), 0, C.byref(sec), C.byref(st))
        assert rc == 0
        ts.append(sec.value)
    out[name] = {"first_call_s": ts[0], "steady_call_s": min(ts[1:]), "nodes_per_s_steady": n ** 3 / min(ts[1:]), "device_ms_total": st.ms_total,
                 "phi_bytes": n ** 3 * 8}
print(json.dumps(out, indent=1))

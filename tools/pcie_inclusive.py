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
    ts = []
    for rep in range(4):
        t = time.perf_counter()
        phi, st = h.compute_distance(hCoef=hc, rebuild=(rep == 0))
        ts.append(time.perf_counter() - t)
    out[name] = {"first_call_s": ts[0], "steady_call_s": min(ts[1:]), "nodes_per_s_steady": n ** 3 / min(ts[1:]), "device_ms_total": st.ms_total,
                 "phi_bytes": n ** 3 * 8}
print(json.dumps(out, indent=1))

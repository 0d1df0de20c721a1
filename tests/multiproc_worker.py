"""Worker of tests/test_gpu_parity.py::test_multiprocess_ranks_on_one_gpu: one rank of a `world`-process solve whose RCCL calls go
through the shared-memory test double (tests/native/rccl_mock.c)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import shm_import  # noqa: E402


def main():
    rank, world, uid_hex, case, mode, out_dir = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4], sys.argv[5], sys.argv[6]
    shm = shm_import.load()
    d = np.load(os.path.join(ROOT, "tests", "golden", case + ".npz"))
    s = shm.GridSolver(device=0, rank=rank, world=world, rccl_unique_id=bytes.fromhex(uid_hex))
    s.set_problem(d["pos"], d["wnormal"], d["area"], float(d["lam"]), int(d["n"]), d["bbox_min"], float(d["cell"]))
    kw = {"primal-plain": dict(solver="primal", precond="none"), "primal-dct": dict(solver="primal", precond="dct"),
          "dual": dict(solver="dual"), "dual-slabs": dict(solver="dual_slabs"), "fast": dict(fast=True)}[mode]
    st = s.solve(tol=1e-10, **kw)
    assert st.solver == {"dual": 2, "dual-slabs": 3}.get(mode, st.solver)
    phi, (k0, k1) = s.get_phi()
    np.save(os.path.join(out_dir, "phi_%d.npy" % rank), phi)
    np.save(os.path.join(out_dir, "meta_%d.npy" % rank), np.array([k0, k1, st.iters, st.shift]))
    s.close()


if __name__ == "__main__":
    main()

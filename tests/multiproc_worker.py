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
    precision = int(os.environ.get("SHM_WORKER_PRECISION", "64"))
    if case.startswith("file:"):   # file:<data file>:<hCoef> -- pre-processed by the C++ host mirror instead of a golden fixture (no oracle phi in it)
        from signed_heat_3d_amd.host_abi import HostSolver
        _, fname, hcoef = case.split(":")
        pre = HostSolver(os.path.join(ROOT, "data", fname)).preprocess(hCoef=float(hcoef))
        d = dict(pos=pre["pos"], wnormal=pre["wnormal"], area=pre["area"], lam=pre["lam"], n=pre["n"], bbox_min=pre["bbox_min"], cell=pre["cell"], m=-1)
        scrub = not fname.endswith(".pc")
    else:
        d = np.load(os.path.join(ROOT, "tests", "golden", case + ".npz"))
        scrub = "_pc_" not in case   # point overload: no divYt scrub (signed_heat_grid_solver.cpp:179-180)
    # SHM_WORKER_DEVICE_PER_RANK=1: one GPU per rank and the real librccl (test_multiprocess_ranks_real_rccl); default: all ranks share
    # device 0 through the shared-memory double of librccl
    device = rank if os.environ.get("SHM_WORKER_DEVICE_PER_RANK") else 0
    if uid_hex == "file":   # real RCCL: rank 0 creates the ncclUniqueId and hands it over through the scratch directory
        import time
        path = os.path.join(out_dir, "uid.bin")
        if rank == 0:
            uid = shm.comm_unique_id()
            with open(path + ".tmp", "wb") as fh:
                fh.write(uid)
            os.replace(path + ".tmp", path)
        else:
            t0 = time.time()
            while not os.path.exists(path):
                if time.time() - t0 > 120:
                    raise SystemExit("rank %d: no unique id from rank 0" % rank)
                time.sleep(0.05)
            uid = open(path, "rb").read()
        uid_hex = uid.hex()
    s = shm.GridSolver(device=device, precision=precision, rank=rank, world=world, rccl_unique_id=bytes.fromhex(uid_hex),
                       slab_plan=int(os.environ.get("SHM_WORKER_SLAB_PLAN", "0")))
    s.set_problem(d["pos"], d["wnormal"], d["area"], float(d["lam"]), int(d["n"]), d["bbox_min"], float(d["cell"]))
    kw = {"primal-plain": dict(solver="primal", precond="none"), "primal-dct": dict(solver="primal", precond="dct"),
          "dual": dict(solver="dual"), "dual-slabs": dict(solver="dual_slabs", dual_form="through_grid"), "fast": dict(fast=True),
          # round 6: the slab-distributed explicit-S forms (S and its inverse replicated, K^+ on the slabs): the direct solve and CG on the explicit S
          "dual-direct-slabs": dict(solver="dual_slabs", dual_form="direct"), "dual-scg-slabs": dict(solver="dual_slabs", dual_form="explicit_s_cg"),
          "auto": dict()}[mode]
    if os.environ.get("SHM_WORKER_MAX_ITERS"):
        # did-not-converge hand-over (include/shm_grid.h: SHM_ERR_NOCONV still leaves phi): the gathered multi-rank solve must copy phi to the
        # rank's slabs and fill the statistics before it reports the status
        st = s.solve(tol=1e-10, max_iters=int(os.environ["SHM_WORKER_MAX_ITERS"]), allow_noconv=True, scrub=scrub, **kw)
        assert st.iters == int(os.environ["SHM_WORKER_MAX_ITERS"]) and (int(d["m"]) < 0 or st.m == int(d["m"])) and st.ms_pcg > 0
    else:
        st = s.solve(tol=1e-10 if precision == 64 else 0.0, scrub=scrub, **kw)
    assert st.solver == {"dual": 2, "dual-slabs": 3, "dual-direct-slabs": 3, "dual-scg-slabs": 3}.get(mode, st.solver)
    assert st.cg_form == {"dual-direct-slabs": 2, "dual-scg-slabs": 3, "dual-slabs": 0}.get(mode, st.cg_form), st.cg_form
    if os.environ.get("SHM_WORKER_EXPECT_SOLVER"):   # what AUTO picked on this rank
        want_solver, want_form = (int(v) for v in os.environ["SHM_WORKER_EXPECT_SOLVER"].split(":"))
        assert st.solver == want_solver and (want_form < 0 and st.cg_form in (2, 3) or st.cg_form == want_form), (st.solver, st.cg_form)
    phi, (k0, k1) = s.get_phi()
    np.save(os.path.join(out_dir, "phi_%d.npy" % rank), phi)
    np.save(os.path.join(out_dir, "meta_%d.npy" % rank), np.array([k0, k1, st.iters, st.shift]))
    s.close()


if __name__ == "__main__":
    main()

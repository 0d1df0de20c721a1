"""world_size>1 on CPU (gloo): a numpy model of the z-slab protocol that libshm_grid.so implements with HIP kernels +
RCCL -- same slab plan (shm_plan_slab from the product's C ABI), same ownership rule for constraint-row entries and shift
items, same collectives (one-plane halo of p, all-reduce of p.q, all-reduce of [||r'||^2, A r'], scalar all-reduce for the
shift) and the same identity rho' = ||r'||^2 - u.w.  It must reproduce the single-process LU oracle.  The per-slab
arithmetic comes from the oracle (test infrastructure); the GPU implementation of the same protocol is covered by
tests/test_gpu_parity.py::test_local_slabs_match_single_slab on one GPU and by the driver's multi-GPU bench."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _worker(rank, world, port, case, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import shm_import
    shm = shm_import.load()
    d = np.load(os.path.join(ROOT, "tests", "golden", case + ".npz"))
    n = int(d["n"])
    plane = n * n
    h = float(d["cell"])
    k0, k1 = shm.plan_slab(n, world, rank)            # product host logic (C ABI, no GPU involved)
    nzl = k1 - k0
    lo, hi = k0 * plane, k1 * plane
    nodes, coeffs = d["c_nodes"], d["c_coeffs"]
    m = nodes.shape[0]
    own = (nodes >= lo) & (nodes < hi)                # entries of every row that THIS slab owns

    def allreduce(v):
        t = torch.from_numpy(np.ascontiguousarray(v, dtype=np.float64))
        dist.all_reduce(t)
        return t.numpy()

    # replicated (A A^T)^-1
    import scipy.sparse as sp
    A = sp.coo_matrix((coeffs.ravel(), (np.repeat(np.arange(m), 8), nodes.ravel())), shape=(m, n ** 3)).tocsr()
    Ginv = np.linalg.inv((A @ A.T).toarray())

    def gather_rows(v_owned):                          # partial A v over owned nodes
        loc = np.where(own, nodes - lo, 0)
        return (np.where(own, coeffs, 0.0) * v_owned[loc]).sum(axis=1)

    def scatter_rows(v_owned, u):                      # v -= A^T u on owned nodes
        np.subtract.at(v_owned, (nodes - lo)[own], (coeffs * u[:, None])[own])

    def stencil(p_owned):                              # q = -L p on owned planes, one-plane halo exchange of p
        P = p_owned.reshape(nzl, n, n)
        below = np.zeros((n, n))
        above = np.zeros((n, n))
        reqs = []
        if rank > 0:
            reqs.append(dist.isend(torch.from_numpy(P[0].copy()), rank - 1))
            tb = torch.zeros((n, n), dtype=torch.float64)
            reqs.append(dist.irecv(tb, rank - 1))
        if rank < world - 1:
            reqs.append(dist.isend(torch.from_numpy(P[-1].copy()), rank + 1))
            ta = torch.zeros((n, n), dtype=torch.float64)
            reqs.append(dist.irecv(ta, rank + 1))
        for r in reqs:
            r.wait()
        if rank > 0:
            below = tb.numpy()
        if rank < world - 1:
            above = ta.numpy()
        ext = np.concatenate([below[None], P, above[None]], axis=0)
        if k0 == 0:
            ext[0] = P[0]                              # out-of-grid neighbour -> the node itself
        if k1 == n:
            ext[-1] = P[-1]
        c = ext[1:-1]
        s = ext[2:] + ext[:-2] - 2 * c
        for ax in (1, 2):
            up = np.roll(c, -1, axis=ax)
            dn = np.roll(c, 1, axis=ax)
            idx_last = [slice(None)] * 3
            idx_last[ax] = -1
            idx_first = [slice(None)] * 3
            idx_first[ax] = 0
            up[tuple(idx_last)] = c[tuple(idx_last)]
            dn[tuple(idx_first)] = c[tuple(idx_first)]
            s += up + dn - 2 * c
        return (-s / (h * h)).reshape(-1)

    b = d["b"][lo:hi].copy()
    x = np.zeros(nzl * plane)
    r = b.copy()
    red = allreduce(np.concatenate([[r @ r], gather_rows(r)]))
    u = Ginv @ red[1:]
    scatter_rows(r, u)
    rho = red[0] - u @ red[1:]
    rho0 = rho
    p = -r
    it = 0
    while rho > 1e-24 * rho0 and it < 5000:
        q = stencil(p)
        pq = allreduce(np.array([p @ q]))[0]
        alpha = rho / pq
        x += alpha * p
        r += alpha * q
        red = allreduce(np.concatenate([[r @ r], gather_rows(r)]))
        u = Ginv @ red[1:]
        scatter_rows(r, u)
        rho_new = red[0] - u @ red[1:]
        p = -r + (rho_new / rho) * p
        rho = rho_new
        it += 1
    phi = -x
    # shift: every (source, z-plane) bilinear piece is evaluated by the owner of that plane
    pos, area = d["pos"], d["area"]
    part = 0.0
    for s in range(len(area)):
        t = (pos[s] - d["bbox_min"]) / h
        i, j, k = (int(np.floor(v)) for v in t)
        tx, ty, tz = (pos[s][0] - (i * h + d["bbox_min"][0])) / h, (pos[s][1] - (j * h + d["bbox_min"][1])) / h, \
                     (pos[s][2] - (k * h + d["bbox_min"][2])) / h
        for dz, w in ((0, 1 - tz), (1, tz)):
            kz = k + dz
            if not (k0 <= kz < k1):
                continue
            base = i + j * n + (kz - k0) * plane
            a0 = phi[base] * (1 - tx) + phi[base + 1] * tx
            a1 = phi[base + n] * (1 - tx) + phi[base + n + 1] * tx
            part += area[s] * w * (a0 * (1 - ty) + a1 * ty)
    shift = allreduce(np.array([part]))[0] / area.sum()
    phi -= shift
    np.save(os.path.join(out_dir, "phi_%d.npy" % rank), phi)
    np.save(os.path.join(out_dir, "meta_%d.npy" % rank), np.array([k0, k1, it, shift]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,case", [(2, "bunny_small_n16"), (3, "bunny_pc_n16")])
def test_zslab_protocol_matches_lu_oracle(tmp_path, world, case):
    port = 29500 + (os.getpid() % 2000) + world
    mp.spawn(_worker, args=(world, port, case, str(tmp_path)), nprocs=world, join=True)
    d = np.load(os.path.join(ROOT, "tests", "golden", case + ".npz"))
    n = int(d["n"])
    parts, covered = [], 0
    for r in range(world):
        k0, k1, it, shift = np.load(tmp_path / ("meta_%d.npy" % r))
        assert int(k0) == covered
        covered = int(k1)
        parts.append(np.load(tmp_path / ("phi_%d.npy" % r)))
        assert abs(shift - float(d["shift"])) < 1e-9
    assert covered == n
    phi = np.concatenate(parts)
    assert np.abs(phi - d["phi"]).max() < 1e-8


def _worker_direct_slabs(rank, world, port, case, out_dir):
    """Round 6: the slab-distributed DIRECT dual solve (libshm_grid.so: Solver::solve_dual with the explicit S on several z-slabs; SHM_SOLVER_DUAL_SLABS, and AUTO at
    256^3 ... 512^3) as a numpy model over gloo.  S = A K^+ A^T and its bordered inverse are REPLICATED (every rank forms them from the whole grid's operator, as every
    rank assembles S from the Green's table); K^+ runs on the z-slabs -- x and y transforms on the rank's planes, an all-to-all that turns z-slabs into y-pencils, the
    z transform with the spectral division, the all-to-all back, the inverse y and x transforms --; A K^+ b is summed over the slabs' owned row entries by one
    all-reduce; the multipliers are solved for on every rank alike; x = K^+ (A^T mu - b) on the slabs again; the shift by one scalar all-reduce."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import scipy.fft as sfft
    import scipy.sparse as sp
    import shm_import
    shm = shm_import.load()
    d = np.load(os.path.join(ROOT, "tests", "golden", case + ".npz"))
    n = int(d["n"])
    plane = n * n
    h = float(d["cell"])
    k0, k1 = shm.plan_slab(n, world, rank)
    nzl, nyl = k1 - k0, n // world
    assert n % world == 0 and nzl == n // world          # equal slabs: what the distributed transforms need (the library refuses anything else)
    lo, hi = k0 * plane, k1 * plane
    nodes, coeffs = d["c_nodes"], d["c_coeffs"]
    m = nodes.shape[0]
    own = (nodes >= lo) & (nodes < hi)
    lam = (2.0 - 2.0 * np.cos(np.pi * np.arange(n) / n)) / (h * h)

    def allreduce(v):
        t = torch.from_numpy(np.ascontiguousarray(v, dtype=np.float64))
        dist.all_reduce(t)
        return t.numpy()

    def alltoall(blocks):                                 # blocks[q] goes to rank q; returns what every rank sent here, by source rank (grouped send / recv pairs)
        got = [None] * world
        got[rank] = blocks[rank].copy()
        reqs, bufs = [], {}
        for q in range(world):
            if q == rank:
                continue
            reqs.append(dist.isend(torch.from_numpy(np.ascontiguousarray(blocks[q])), q))
            bufs[q] = torch.zeros(blocks[q].shape, dtype=torch.float64)
            reqs.append(dist.irecv(bufs[q], q))
        for r in reqs:
            r.wait()
        for q, t in bufs.items():
            got[q] = t.numpy()
        return got

    def kplus_slabs(v_owned):                             # K^+ on the rank's planes [nzl][n (y)][n (x)]
        V = sfft.dct(sfft.dct(v_owned.reshape(nzl, n, n), type=2, norm="ortho", axis=2), type=2, norm="ortho", axis=1)
        pencils = np.concatenate(alltoall([V[:, q * nyl:(q + 1) * nyl, :] for q in range(world)]), axis=0)      # [n (z)][nyl][n]: this rank's y rows of every plane
        W = sfft.dct(pencils, type=2, norm="ortho", axis=0)
        den = lam[:, None, None] + lam[None, rank * nyl:(rank + 1) * nyl, None] + lam[None, None, :]
        W = np.where(den > 0, W / np.where(den > 0, den, 1.0), 0.0)                                                # the constant mode -> 0: the pseudo-inverse
        pencils = sfft.idct(W, type=2, norm="ortho", axis=0)
        back = alltoall([pencils[q * nzl:(q + 1) * nzl] for q in range(world)])                                   # this rank's planes, the y rows by owner
        V = np.concatenate(back, axis=1)
        return sfft.idct(sfft.idct(V, type=2, norm="ortho", axis=1), type=2, norm="ortho", axis=2).reshape(-1)

    def gather_rows(v_owned):
        loc = np.where(own, nodes - lo, 0)
        return (np.where(own, coeffs, 0.0) * v_owned[loc]).sum(axis=1)

    # replicated set-up: S = A K^+ A^T on the whole grid (the library assembles it from the Green's table of the grid), bordered with the constant
    A = sp.coo_matrix((coeffs.ravel(), (np.repeat(np.arange(m), 8), nodes.ravel())), shape=(m, n ** 3)).tocsr()
    den = lam[:, None, None] + lam[None, :, None] + lam[None, None, :]
    AT = A.T.toarray().reshape(n, n, n, m)
    KAT = sfft.idctn(np.where(den[..., None] > 0, sfft.dctn(AT, type=2, norm="ortho", axes=(0, 1, 2)) / np.where(den > 0, den, 1.0)[..., None], 0.0), type=2, norm="ortho", axes=(0, 1, 2))
    S = A @ KAT.reshape(n ** 3, m)
    B = np.block([[S, np.ones((m, 1))], [np.ones((1, m)), np.zeros((1, 1))]])

    b = d["b"][lo:hi].copy()
    red = allreduce(np.concatenate([[b.sum()], gather_rows(kplus_slabs(b))]))      # [sum b, A K^+ b]: one all-reduce
    mu = np.linalg.solve(B, np.concatenate([red[1:], [red[0]]]))[:m]               # the same numbers on every rank
    rhs = -b
    np.add.at(rhs, (nodes - lo)[own], (coeffs * mu[:, None])[own])                 # A^T mu - b on the owned nodes
    x = kplus_slabs(rhs)
    ax = allreduce(gather_rows(x))
    assert np.abs(ax - ax.mean()).max() < 1e-9 * max(1.0, np.abs(x).max())         # A x is constant over the rows: the additive constant of the KKT solution
    phi = -x
    pos, area = d["pos"], d["area"]
    part = 0.0
    for s in range(len(area)):
        t = (pos[s] - d["bbox_min"]) / h
        i, j, k = (int(np.floor(v)) for v in t)
        tx, ty, tz = (pos[s][0] - (i * h + d["bbox_min"][0])) / h, (pos[s][1] - (j * h + d["bbox_min"][1])) / h, (pos[s][2] - (k * h + d["bbox_min"][2])) / h
        for dz, w in ((0, 1 - tz), (1, tz)):
            kz = k + dz
            if not (k0 <= kz < k1):
                continue
            base = i + j * n + (kz - k0) * plane
            a0 = phi[base] * (1 - tx) + phi[base + 1] * tx
            a1 = phi[base + n] * (1 - tx) + phi[base + n + 1] * tx
            part += area[s] * w * (a0 * (1 - ty) + a1 * ty)
    shift = allreduce(np.array([part]))[0] / area.sum()
    phi -= shift
    np.save(os.path.join(out_dir, "phi_%d.npy" % rank), phi)
    np.save(os.path.join(out_dir, "meta_%d.npy" % rank), np.array([k0, k1]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,case", [(2, "bunny_small_n16"), (4, "bunny_pc_n16")])
def test_slab_distributed_direct_dual_protocol_matches_lu_oracle(tmp_path, world, case):
    port = 31500 + (os.getpid() % 2000) + world
    mp.spawn(_worker_direct_slabs, args=(world, port, case, str(tmp_path)), nprocs=world, join=True)
    d = np.load(os.path.join(ROOT, "tests", "golden", case + ".npz"))
    parts, covered = [], 0
    for r in range(world):
        k0, k1 = np.load(tmp_path / ("meta_%d.npy" % r))
        assert int(k0) == covered
        covered = int(k1)
        parts.append(np.load(tmp_path / ("phi_%d.npy" % r)))
    assert covered == int(d["n"])
    assert np.abs(np.concatenate(parts) - d["phi"]).max() < 1e-8

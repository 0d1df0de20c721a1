"""CPU oracle (numpy/scipy) for the regular-grid Signed Heat Method solver.

TEST INFRASTRUCTURE ONLY.  Nothing in the product path (signed-heat-3d_amd/, bench.py's GPU leg)
may import this file; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may.

PARITY UNPINNED: the reference (nzfeng/signed-heat-3d) cannot be compiled in this image -- its
geometry-central / Eigen / Polyscope / TetGen / libigl submodules are empty and there is no network --
and the reference has no tests or golden vectors of its own.  This file therefore restates the
reference's arithmetic line by line (citations below are into /root/reference) and *literally*
assembles the same sparse matrices (L, D, A, KKT) the reference assembles, then LU-solves the KKT
system with SuperLU (scipy.sparse.linalg.splu), which stands in for geometry-central's
`solveSquare` -> Eigen::SparseLU (third-party, un-vendored, version unpinned).  It is pinned against
(i) the spot values of BASELINE.md section 2 (an independent restatement made by the surveyor),
(ii) operator identities, (iii) analytic known-answer cases (tests/test_oracle.py).

Node flattening everywhere: idx = i + j*n + k*n*n (x fastest), signed_heat_grid_solver.cpp:505-508.
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla


# --------------------------------------------------------------------------------------------------
# Loaders (geometry-central's readSurfaceMesh is absent; behaviour restated from SURVEY 8(c):
# polygons kept, unreferenced vertices stripped preserving file order).  main.cpp:196-225 for .pc.
# --------------------------------------------------------------------------------------------------
def read_obj(path):
    """Return (V [nv,3] float64, faces list[list[int]]) with unreferenced vertices dropped."""
    verts, faces = [], []
    with open(path, "r") as fh:
        for line in fh:
            t = line.split()
            if not t:
                continue
            if t[0] == "v":
                verts.append((float(t[1]), float(t[2]), float(t[3])))
            elif t[0] == "f":
                f = []
                for tok in t[1:]:
                    vi = int(tok.split("/")[0])
                    f.append(vi - 1 if vi > 0 else len(verts) + vi)
                faces.append(f)
    V = np.asarray(verts, dtype=np.float64)
    used = np.zeros(len(V), dtype=bool)
    for f in faces:
        used[f] = True
    remap = np.cumsum(used) - 1
    faces = [[int(remap[v]) for v in f] for f in faces]
    return V[used], faces


def read_pc(path):
    """main.cpp:196-225: 'v x y z' -> positions, 'vn x y z' -> normals, everything else ignored."""
    P, Nn = [], []
    with open(path, "r") as fh:
        for line in fh:
            t = line.split()
            if not t:
                continue
            if t[0] == "v":
                P.append((float(t[1]), float(t[2]), float(t[3])))
            elif t[0] == "vn":
                Nn.append((float(t[1]), float(t[2]), float(t[3])))
    return np.asarray(P, dtype=np.float64), np.asarray(Nn, dtype=np.float64)


# --------------------------------------------------------------------------------------------------
# signed_heat_3d.cpp helpers
# --------------------------------------------------------------------------------------------------
def centroid(V):
    """signed_heat_3d.cpp:3-12 / :24-33 -- sequential sum then divide."""
    c = np.zeros(3)
    for p in V:
        c = c + p
    return c / len(V)


def radius(V, c):
    """signed_heat_3d.cpp:14-22 / :35-43."""
    return float(np.max(np.sqrt(((c[None, :] - V) ** 2).sum(axis=1))))


def mean_edge_length(V, faces):
    """signed_heat_3d.cpp:51-60: mean length over UNIQUE edges (mesh.edges())."""
    seen = {}
    for f in faces:
        d = len(f)
        for a in range(d):
            u, v = f[a], f[(a + 1) % d]
            key = (u, v) if u < v else (v, u)
            if key not in seen:
                seen[key] = float(np.linalg.norm(V[u] - V[v]))
    h = 0.0
    for L in seen.values():
        h += L
    return h / len(seen)


def face_vector_areas(V, faces):
    """signed_heat_3d.cpp:74-88 (the shoelace branch always wins, :65-72 is overwritten)."""
    areas = np.zeros(len(faces))
    normals = np.zeros((len(faces), 3))
    for fi, f in enumerate(faces):
        N = np.zeros(3)
        d = len(f)
        for a in range(d):
            N = N + np.cross(V[f[a]], V[f[(a + 1) % d]])
        N = N * 0.5
        A = math.sqrt(N[0] * N[0] + N[1] * N[1] + N[2] * N[2])
        areas[fi] = A
        normals[fi] = N / A
    return areas, normals


def barycenters(V, faces):
    """signed_heat_grid_solver.cpp:498-503."""
    B = np.zeros((len(faces), 3))
    for fi, f in enumerate(faces):
        c = np.zeros(3)
        for v in f:
            c = c + V[v]
        B[fi] = c / len(f)
    return B


@dataclass
class Grid:
    n: int
    bbox_min: np.ndarray
    cell: float

    @property
    def N(self):
        return self.n ** 3


def grid_setup(V, scale=2.0, hCoef=0.0):
    """signed_heat_grid_solver.cpp:13-26."""
    c = centroid(V)
    r = radius(V, c)
    s = r * scale
    n = int(2 * math.pow(2.0, hCoef + 3))
    cell = 2.0 * s / (n - 1)
    return Grid(n=n, bbox_min=c - s, cell=cell)


@dataclass
class Sources:
    pos: np.ndarray      # [S,3] barycenters / point positions
    wnormal: np.ndarray  # [S,3] N*A  (the reference multiplies N*A first, :57)
    area: np.ndarray     # [S]
    lam: float


def mesh_sources(V, faces, tCoef=1.0):
    """signed_heat_grid_solver.cpp:42-47,54-57."""
    h = mean_edge_length(V, faces)
    short_time = tCoef * h * h
    lam = math.sqrt(1.0 / short_time)
    areas, normals = face_vector_areas(V, faces)
    return Sources(pos=barycenters(V, faces), wnormal=normals * areas[:, None], area=areas, lam=lam)


def point_sources(P, Nn, areas, h, tCoef=1.0):
    """signed_heat_grid_solver.cpp:151-153,163-166.  areas/h come from geometry-central's tufted
    triangulation in the reference (third-party, absent) -> they are INPUTS here."""
    short_time = tCoef * h * h
    lam = math.sqrt(1.0 / short_time)
    return Sources(pos=P.copy(), wnormal=Nn * areas[:, None], area=np.asarray(areas, float), lam=lam)


# --------------------------------------------------------------------------------------------------
# Steps 1+2: direct summation + normalisation, signed_heat_grid_solver.cpp:48-65 / :157-174
# --------------------------------------------------------------------------------------------------
def node_positions(g: Grid):
    """signed_heat_grid_solver.cpp:510-514: (i,j,k)*cellSize + bboxMin."""
    ax = np.arange(g.n, dtype=np.float64) * g.cell
    X = np.empty((g.n, g.n, g.n, 3))  # [k,j,i,:]
    X[..., 0] = ax[None, None, :] + g.bbox_min[0]
    X[..., 1] = ax[None, :, None] + g.bbox_min[1]
    X[..., 2] = ax[:, None, None] + g.bbox_min[2]
    return X.reshape(-1, 3)


def conv_raw(g: Grid, src: Sources, chunk=2048):
    """Unnormalised X(x) = sum_s wn_s * exp(-lam r)/r   (signed_heat_3d.cpp:45-49)."""
    X = node_positions(g)
    out = np.zeros((g.N, 3))
    for a in range(0, g.N, chunk):
        d = X[a:a + chunk, None, :] - src.pos[None, :, :]
        r = np.sqrt((d * d).sum(axis=2))
        with np.errstate(divide="ignore", invalid="ignore"):
            G = np.exp(-src.lam * r) / r
        out[a:a + chunk] = G @ src.wnormal
    return out


def conv_normalize(g: Grid, src: Sources, chunk=2048):
    """Returns Y as [N,3] (reference stores AoS Y[3*idx+p], :58-62)."""
    Xs = conv_raw(g, src, chunk)
    with np.errstate(divide="ignore", invalid="ignore"):
        nrm = np.sqrt((Xs * Xs).sum(axis=1))
        return Xs / nrm[:, None]


# --------------------------------------------------------------------------------------------------
# Literal sparse operators
# --------------------------------------------------------------------------------------------------
def _ijk(n):
    k, j, i = np.meshgrid(np.arange(n), np.arange(n), np.arange(n), indexing="ij")
    return i.ravel(), j.ravel(), k.ravel()


def laplacian_matrix(g: Grid):
    """signed_heat_grid_solver.cpp:278-334: 7 triplets per row; an out-of-grid neighbour is
    redirected to the node itself (:299-319); duplicates sum (setFromTriplets)."""
    n = g.n
    i, j, k = _ijk(n)
    cur = i + j * n + k * n * n
    rows, cols, vals = [], [], []

    def add(c, v):
        rows.append(cur)
        cols.append(c)
        vals.append(np.full(cur.shape, v, dtype=np.float64))

    add(np.where(i == n - 1, cur, cur + 1), 1.0)          # nextX
    add(np.where(j == n - 1, cur, cur + n), 1.0)          # nextY
    add(np.where(k == n - 1, cur, cur + n * n), 1.0)      # nextZ
    add(np.where(i == 0, cur, cur - 1), 1.0)              # prevX
    add(np.where(j == 0, cur, cur - n), 1.0)              # prevY
    add(np.where(k == 0, cur, cur - n * n), 1.0)          # prevZ
    add(cur, -6.0)
    L = sp.coo_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))),
                      shape=(g.N, g.N)).tocsc()
    return L / (g.cell * g.cell)


def gradient_matrix(g: Grid):
    """signed_heat_grid_solver.cpp:336-402: forward differences, last node mirrored backwards."""
    n = g.n
    i, j, k = _ijk(n)
    cur = i + j * n + k * n * n
    rows, cols, vals = [], [], []
    for p, (idx, stride) in enumerate(((i, 1), (j, n), (k, n * n))):
        last = idx == n - 1
        nxt = np.where(last, cur, cur + stride)
        c0 = np.where(last, cur - stride, cur)
        rows += [3 * cur + p, 3 * cur + p]
        cols += [nxt, c0]
        vals += [np.ones(cur.shape), -np.ones(cur.shape)]
    D = sp.coo_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))),
                      shape=(3 * g.N, g.N)).tocsc()
    return D / g.cell


def trilinear_row(g: Grid, q):
    """signed_heat_grid_solver.cpp:433-464: node order 000,100,010,001,110,101,011,111."""
    n, h = g.n, g.cell
    d = q - g.bbox_min
    i = int(math.floor(d[0] / h))
    j = int(math.floor(d[1] / h))
    k = int(math.floor(d[2] / h))
    p000 = np.array([i * h, j * h, k * h]) + g.bbox_min
    tx = (q[0] - p000[0]) / h
    ty = (q[1] - p000[1]) / h
    tz = (q[2] - p000[2]) / h

    def ix(a, b, c):
        return a + b * n + c * n * n

    nodes = [ix(i, j, k), ix(i + 1, j, k), ix(i, j + 1, k), ix(i, j, k + 1),
             ix(i + 1, j + 1, k), ix(i + 1, j, k + 1), ix(i, j + 1, k + 1), ix(i + 1, j + 1, k + 1)]
    coeffs = [(1. - tx) * (1. - ty) * (1. - tz), tx * (1. - ty) * (1. - tz), (1. - tx) * ty * (1. - tz),
              (1. - tx) * (1. - ty) * tz, tx * ty * (1. - tz), tx * (1. - ty) * tz,
              (1. - tx) * ty * tz, tx * ty * tz]
    return nodes, coeffs


def constraint_rows(g: Grid, pts):
    """signed_heat_grid_solver.cpp:80-98 / :186-204: one row per DISTINCT cell, in source order.
    Returns (nodes [m,8] int64, coeffs [m,8] float64)."""
    n, h = g.n, g.cell
    used = set()
    nodes, coeffs = [], []
    for b in pts:
        d = b - g.bbox_min
        i = int(math.floor(d[0] / h))
        j = int(math.floor(d[1] / h))
        k = int(math.floor(d[2] / h))
        cell_idx = i + j * n + k * n * n
        if cell_idx in used:
            continue
        nd, cf = trilinear_row(g, b)
        nodes.append(nd)
        coeffs.append(cf)
        used.add(cell_idx)
    return np.asarray(nodes, dtype=np.int64).reshape(-1, 8), np.asarray(coeffs, dtype=np.float64).reshape(-1, 8)


def constraint_matrix(g: Grid, pts):
    nodes, coeffs = constraint_rows(g, pts)
    m = nodes.shape[0]
    rows = np.repeat(np.arange(m), 8)
    return sp.coo_matrix((coeffs.ravel(), (rows, nodes.ravel())), shape=(m, g.N)).tocsr()


def divergence_rhs(g: Grid, Y, scrub=True):
    """divYt = D^T * Y (:70-71) and the non-finite scrub of the mesh overload (:72-74)."""
    D = gradient_matrix(g)
    Yflat = np.asarray(Y, dtype=np.float64).reshape(-1)  # AoS 3*idx+p
    with np.errstate(invalid="ignore"):
        b = D.T @ Yflat
    if scrub:
        b = np.where(np.isfinite(b), b, 0.0)
    return b


def evaluate_function(g: Grid, u, q):
    """signed_heat_grid_solver.cpp:405-431 (lerp order x, then y, then z)."""
    n, h = g.n, g.cell
    d = q - g.bbox_min
    i = int(math.floor(d[0] / h))
    j = int(math.floor(d[1] / h))
    k = int(math.floor(d[2] / h))

    def at(a, b, c):
        return u[a + b * n + c * n * n]

    tx = (q[0] - (i * h + g.bbox_min[0])) / h
    ty = (q[1] - (j * h + g.bbox_min[1])) / h
    tz = (q[2] - (k * h + g.bbox_min[2])) / h
    v00 = at(i, j, k) * (1. - tx) + at(i + 1, j, k) * tx
    v01 = at(i, j, k + 1) * (1. - tx) + at(i + 1, j, k + 1) * tx
    v10 = at(i, j + 1, k) * (1. - tx) + at(i + 1, j + 1, k) * tx
    v11 = at(i, j + 1, k + 1) * (1. - tx) + at(i + 1, j + 1, k + 1) * tx
    v0 = v00 * (1. - ty) + v10 * ty
    v1 = v01 * (1. - ty) + v11 * ty
    return v0 * (1. - tz) + v1 * tz


def source_average(g: Grid, u, src: Sources):
    """signed_heat_grid_solver.cpp:466-496: area-weighted mean over ALL sources."""
    shift = 0.0
    norm = 0.0
    for s in range(len(src.area)):
        shift += src.area[s] * evaluate_function(g, u, src.pos[s])
        norm += src.area[s]
    return shift / norm


def solve_kkt_lu(g: Grid, b, A):
    """signed_heat_grid_solver.cpp:101-108: [[L,A^T],[A,0]] [x;mu] = [b;0], phi = -x (sparse LU)."""
    L = laplacian_matrix(g)
    m = A.shape[0]
    K = sp.bmat([[L, A.T], [A, None]], format="csc")
    rhs = np.concatenate([b, np.zeros(m)])
    lu = spla.splu(K)
    sol = lu.solve(rhs)
    res = float(np.max(np.abs(K @ sol - rhs)))
    return -sol[:g.N], res


def projected_cg(g: Grid, b, A, tol=1e-13, maxit=20000):
    """Matrix-free equivalent of solve_kkt_lu: CG on K=-L restricted to null(A).
    Used as the oracle above 64^3 where the LU is infeasible; validated against the LU below that."""
    L = laplacian_matrix(g)
    K = -L
    AAt = (A @ A.T).tocsc()
    lu = spla.splu(AAt)

    def P(v):
        return v - A.T @ lu.solve(A @ v)

    x = np.zeros(g.N)
    r = P(b)
    p = -r
    rho = r @ r
    rho0 = rho
    it = 0
    while it < maxit and rho > tol * tol * rho0:
        q = K @ p
        alpha = rho / (p @ q)
        x += alpha * p
        r = P(r + alpha * q)
        rho_new = r @ r
        p = -r + (rho_new / rho) * p
        rho = rho_new
        it += 1
    return -x, it, math.sqrt(rho / rho0)


def integrate_greedily(g: Grid, Y):
    """signed_heat_grid_solver.cpp:224-275: FIFO BFS from node (0,0,0), first visitor wins."""
    from collections import deque
    n, h = g.n, g.cell
    Y = np.asarray(Y).reshape(-1, 3)
    phi = np.zeros(g.N)
    visited = np.zeros(g.N, dtype=bool)
    qd = deque([(0, 0, 0)])
    visited[0] = True
    while qd:
        cur = qd.popleft()
        ci = cur[0] + cur[1] * n + cur[2] * n * n
        Yp = Y[ci]
        for a in range(3):
            for step in (-1, +1):
                if step == -1 and cur[a] == 0:
                    continue
                if step == +1 and cur[a] >= n - 1:
                    continue
                nxt = list(cur)
                nxt[a] += step
                ni = nxt[0] + nxt[1] * n + nxt[2] * n * n
                if visited[ni]:
                    continue
                # edge = q - p computed from node positions (:245-246)
                pq = np.zeros(3)
                pq[a] = (nxt[a] * h + g.bbox_min[a]) - (cur[a] * h + g.bbox_min[a])
                Yavg = Y[ni] + Yp
                Yavg = Yavg / math.sqrt(Yavg[0] ** 2 + Yavg[1] ** 2 + Yavg[2] ** 2)
                phi[ni] = phi[ci] + (Yavg[0] * pq[0] + Yavg[1] * pq[1] + Yavg[2] * pq[2])
                visited[ni] = True
                qd.append(tuple(nxt))
    return phi


# --------------------------------------------------------------------------------------------------
# End to end, signed_heat_grid_solver.cpp:5-114 (mesh) and :116-222 (points)
# --------------------------------------------------------------------------------------------------
def compute_distance(g: Grid, src: Sources, *, scrub, fast=False, solver="lu", tol=1e-13, detail=False):
    Y = conv_normalize(g, src)
    b = divergence_rhs(g, Y, scrub=scrub)
    info = {}
    if fast:
        phi = integrate_greedily(g, Y)
    else:
        A = constraint_matrix(g, src.pos)
        info["m"] = A.shape[0]
        if solver == "lu":
            phi, res = solve_kkt_lu(g, b, A)
            info["kkt_residual"] = res
        else:
            phi, it, rel = projected_cg(g, b, A, tol=tol)
            info["iters"], info["rel_res"] = it, rel
    shift = source_average(g, phi, src)
    phi = phi - shift
    info["shift"] = shift
    if detail:
        info["Y"] = Y
        info["b"] = b
    return phi, info


def compute_distance_mesh(V, faces, *, tCoef=1.0, hCoef=0.0, scale=2.0, fast=False, **kw):
    g = grid_setup(V, scale, hCoef)
    src = mesh_sources(V, faces, tCoef)
    phi, info = compute_distance(g, src, scrub=True, fast=fast, **kw)
    return g, src, phi, info


def compute_distance_points(P, Nn, areas, h, *, tCoef=1.0, hCoef=0.0, scale=2.0, fast=False, **kw):
    g = grid_setup(P, scale, hCoef)
    src = point_sources(P, Nn, areas, h, tCoef)
    phi, info = compute_distance(g, src, scrub=False, fast=fast, **kw)
    return g, src, phi, info

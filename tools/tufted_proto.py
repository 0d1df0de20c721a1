"""Prototype (numpy / scipy, CPU): what would a faithful reading of geometry-central's point-cloud pipeline report for h = meanEdgeLength(tuftedGeom)
(signed_heat_grid_solver.cpp:149-151)?  k = 30 nearest neighbours, tangent-plane Delaunay triangulation per point, the triangles incident on the point, the UNION OF ALL
local triangles with multiplicity (a triangle all three of whose corners agree on appears three times), tufted double cover (front and back copy of every triangle, glued
around every edge in list order: Sharp & Crane 2020), mollification, intrinsic Delaunay flips (tools/delaunay_anchor.py), mean length over the cover's edges.
Result (round 4): bunny.pc 0.094596, rocker.pc 0.108667 -- 3.9 % / 2.6 % ABOVE the flipped-mesh anchors (0.091045 / 0.105880), where the shipped estimator
(agreed triangles once, flips on the manifold part) gives 0.091222 / 0.107997 (+0.2 % / +2.0 %): with the duplicates kept, most cover edges run between coincident copies of
one triangle and cannot be flipped.  Which of the two geometry-central reports cannot be decided without it; areas and h stay replaceable inputs of the ABI.   python tools/tufted_proto.py"""
import os, sys, time
import numpy as np
from collections import deque
from scipy.spatial import cKDTree, Delaunay
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
from oracle import shm_oracle as O
import delaunay_anchor as DA

def local_triangles(P, N, k=30):
    tree = cKDTree(P)
    _, nbr = tree.query(P, k=k + 1)
    tris = []
    for a in range(len(P)):
        n = N[a] / np.linalg.norm(N[a])
        t = np.array([1.0, 0, 0]) if abs(n[0]) < 0.9 else np.array([0, 1.0, 0])
        e1 = np.cross(n, t); e1 /= np.linalg.norm(e1); e2 = np.cross(n, e1)
        idx = nbr[a]
        d = P[idx] - P[a]
        uv = np.stack([d @ e1, d @ e2], axis=1)
        try:
            dl = Delaunay(uv)
        except Exception:
            continue
        for simp in dl.simplices:
            if 0 in simp:   # incident on the centre point (index 0 in the local list)
                g = [int(idx[s]) for s in simp]
                tris.append(tuple(g))
    return tris

def tufted(P, tris):
    """all local triangles with multiplicity -> front/back copies glued around every edge in list order (Sharp & Crane 2020)"""
    F = []
    for (a, b, c) in tris:
        F.append((a, b, c)); F.append((a, c, b))       # front = 2t, back = 2t+1
    F = np.array(F, dtype=np.int64)
    nF = len(F)
    L = np.zeros((nF, 3))
    for s in range(3):
        L[:, s] = np.linalg.norm(P[F[:, (s + 1) % 3]] - P[F[:, s]], axis=1)
    G = -np.ones((nF, 3, 2), dtype=np.int64)
    # sides by undirected edge, per original triangle t: front side s goes a->b; the back copy has the reversed side
    edges = {}
    for t in range(len(tris)):
        f = 2 * t
        for s in range(3):
            a, b = int(F[f, s]), int(F[f, (s + 1) % 3])
            edges.setdefault((min(a, b), max(a, b)), []).append((t, a, b))
    def side_of(f, a, b):   # side of face f running a -> b
        for s in range(3):
            if F[f, s] == a and F[f, (s + 1) % 3] == b: return s
        raise KeyError
    for (lo, hi), lst in edges.items():
        k = len(lst)
        ups, downs = [], []
        for (t, a, b) in lst:
            # copy of t whose side runs lo -> hi ("up") and the one running hi -> lo ("down")
            if (a, b) == (lo, hi): up, down = 2 * t, 2 * t + 1
            else: up, down = 2 * t + 1, 2 * t
            ups.append((up, side_of(up, lo, hi))); downs.append((down, side_of(down, hi, lo)))
        for i in range(k):
            (f0, s0), (f1, s1) = ups[i], downs[(i + 1) % k]
            G[f0, s0] = (f1, s1); G[f1, s1] = (f0, s0)
    assert (G[:, :, 0] >= 0).all()
    return F, L, G

def mollify(L, G, rel=1e-5):
    """geometry-central mollifyIntrinsic: add a constant to all lengths so that every triangle inequality holds with margin delta"""
    delta = rel * L.mean()
    eps = 0.0
    for s in range(3):
        a, b, c = L[:, s], L[:, (s + 1) % 3], L[:, (s + 2) % 3]
        eps = max(eps, float(np.max(delta + a - b - c)))
    eps = max(eps, 0.0)
    return L + eps

for name, mesh in (("bunny.pc", "bunny_small.obj"), ("rocker.pc", "rocker.obj")):
    P, N = O.read_pc(os.path.join(ROOT, "data", name))
    P = np.asarray(P); N = np.asarray(N)
    t0 = time.time()
    tris = local_triangles(P, N)
    F, L, G = tufted(P, tris)
    L = mollify(L, G)
    h0, ne = DA.mean_edge(F, L, G)
    h1 = DA.flip_to_delaunay(F, L, G)
    area = np.zeros(len(P))
    for f in range(len(F)):
        a, b, c = L[f]; sp = 0.5 * (a + b + c)
        ar = np.sqrt(max(0.0, sp * (sp - a) * (sp - b) * (sp - c)))
        for v in F[f]: area[v] += ar / 3
    print("%s: %d points, %d local triangles, cover faces %d, edges %d; mean edge before flips %.6f, tufted intrinsic Delaunay %.6f; area sum %.4f (x1/6 = %.4f)  [%.0f s]" % (
        name, len(P), len(tris), len(F), ne, h0, h1, area.sum(), area.sum() / 6, time.time() - t0), flush=True)

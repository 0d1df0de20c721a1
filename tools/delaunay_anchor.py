#!/usr/bin/env python3
"""What mean edge length would geometry-central's tufted intrinsic DELAUNAY triangulation report for bunny.pc / rocker.pc?

The point overload takes h = meanEdgeLength(tuftedGeom) (signed_heat_grid_solver.cpp:151) where the tufted triangulation has been flipped to
intrinsic Delaunay.  bunny.pc / rocker.pc are exactly the vertices of bunny_small.obj / rocker.obj, so a plausibility anchor for a
headless estimator is NOT the mean edge length of the mesh as modelled (0.0950 / 0.1081) but that of the same surface after intrinsic Delaunay
flips.  This script flips the given mesh (intrinsic edge lengths, gluing-map representation, flip while cot(a) + cot(b) < 0) and prints both.
Development tool (uses oracle/ loaders); not part of the product.
"""
import os
import sys
from collections import deque

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import shm_oracle as O  # noqa: E402


def build(V, F):
    F = np.asarray(F, dtype=np.int64)
    nF = len(F)
    L = np.zeros((nF, 3))
    for s in range(3):
        L[:, s] = np.linalg.norm(V[F[:, (s + 1) % 3]] - V[F[:, s]], axis=1)   # side s: from corner s to corner s+1
    G = -np.ones((nF, 3, 2), dtype=np.int64)
    edges = {}
    for f in range(nF):
        for s in range(3):
            a, b = int(F[f, s]), int(F[f, (s + 1) % 3])
            key = (min(a, b), max(a, b))
            edges.setdefault(key, []).append((f, s))
    nonman = 0
    for key, lst in edges.items():
        if len(lst) == 2:
            (f0, s0), (f1, s1) = lst
            G[f0, s0] = (f1, s1)
            G[f1, s1] = (f0, s0)
        elif len(lst) > 2:
            nonman += 1
    return F.copy(), L, G, nonman


def cot_opposite(L, f, s):
    a, b, c = L[f, s], L[f, (s + 1) % 3], L[f, (s + 2) % 3]   # angle opposite side s
    cosv = (b * b + c * c - a * a) / (2 * b * c)
    cosv = min(1.0, max(-1.0, cosv))
    sinv = max(1e-300, np.sqrt(1 - cosv * cosv))
    return cosv / sinv


def flip(F, L, G, f0, s0):
    f1, s1 = G[f0, s0]
    if f1 < 0 or f1 == f0:
        return False
    # triangle f0: corners (i, j, k) with side s0 = (i, j); f1: corners (j, i, m) with side s1 = (j, i)
    i, j, k = F[f0, s0], F[f0, (s0 + 1) % 3], F[f0, (s0 + 2) % 3]
    m = F[f1, (s1 + 2) % 3]
    l_ij = L[f0, s0]
    l_jk, l_ki = L[f0, (s0 + 1) % 3], L[f0, (s0 + 2) % 3]
    l_im, l_mj = L[f1, (s1 + 1) % 3], L[f1, (s1 + 2) % 3]
    # new diagonal k-m by laying the two triangles out in the plane
    def ang(a, b, c):  # angle between sides a, b opposite c
        return np.arccos(min(1.0, max(-1.0, (a * a + b * b - c * c) / (2 * a * b))))
    th = ang(l_ij, l_ki, l_jk) + ang(l_ij, l_im, l_mj)      # angle at i between ik and im
    if th >= np.pi - 1e-12:
        return False                                        # not flippable (non-convex quad)
    l_km = np.sqrt(max(0.0, l_ki * l_ki + l_im * l_im - 2 * l_ki * l_im * np.cos(th)))
    g_jk, g_ki = G[f0, (s0 + 1) % 3].copy(), G[f0, (s0 + 2) % 3].copy()
    g_im, g_mj = G[f1, (s1 + 1) % 3].copy(), G[f1, (s1 + 2) % 3].copy()
    # new faces: f0 = (k, i, m), f1 = (m, j, k)
    F[f0] = (k, i, m); L[f0] = (l_ki, l_im, l_km)
    F[f1] = (m, j, k); L[f1] = (l_mj, l_jk, l_km)
    def glue(f, s, g):
        G[f, s] = g
        if g[0] >= 0:
            G[g[0], g[1]] = (f, s)
    glue(f0, 0, g_ki); glue(f0, 1, g_im); glue(f1, 0, g_mj); glue(f1, 1, g_jk)
    G[f0, 2] = (f1, 2); G[f1, 2] = (f0, 2)
    return True


def mean_edge(F, L, G):
    tot, cnt = 0.0, 0
    for f in range(len(F)):
        for s in range(3):
            g = G[f, s]
            if g[0] < 0 or (f, s) < (int(g[0]), int(g[1])):
                tot += L[f, s]; cnt += 1
    return tot / cnt, cnt


def flip_to_delaunay(F, L, G):
    """Flip in place until every interior edge is locally (intrinsically) Delaunay; returns the mean edge length."""
    q = deque((f, s) for f in range(len(F)) for s in range(3))
    while q:
        f, s = q.popleft()
        g = G[f, s]
        if g[0] < 0:
            continue
        # Delaunay: the two angles opposite the edge sum to <= pi  <=>  cot(a) + cot(b) >= 0
        if cot_opposite(L, f, s) + cot_opposite(L, int(g[0]), int(g[1])) < -1e-12 and flip(F, L, G, f, s):
            for ff in (f, int(g[0])):
                for ss in range(3):
                    q.append((ff, ss))
    return mean_edge(F, L, G)[0]


def main():
    for name in ("bunny_small.obj", "rocker.obj"):
        V, faces = O.read_obj(os.path.join(ROOT, "data", name))
        F, L, G, nonman = build(V, [f for f in faces if len(f) == 3])
        h0, ne = mean_edge(F, L, G)
        h1 = flip_to_delaunay(F, L, G)
        print("%-16s faces %d edges %d non-manifold edges %d: mean edge length as modelled %.6f, intrinsic Delaunay %.6f (%.2f %%)" % (
            name, len(F), ne, nonman, h0, h1, 100 * (h1 / h0 - 1)))


if __name__ == "__main__":
    main()

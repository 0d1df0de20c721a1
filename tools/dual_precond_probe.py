#!/usr/bin/env python3
"""Experiment (CPU, numpy/scipy): iteration counts of the dual solver (CG on S = A K^+ A^T) under different
approximations of G^-1 (G = A A^T) inside its preconditioner  G^-1 (A K A^T) G^-1.

    python tools/dual_precond_probe.py data/bunny_small.obj 3      # file, hCoef

Development tool: imports oracle/ (never part of the product path).
"""
import os
import sys
import time

import numpy as np
import scipy.fft as sfft
import scipy.sparse as sp
import scipy.sparse.linalg as spla

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import shm_oracle as O  # noqa: E402


def setup(path, hCoef):
    if path.endswith(".pc"):
        P, Nn = O.read_pc(path)
        # crude areas / h: uniform (inputs of the ABI; irrelevant for the conditioning of G)
        c = O.centroid(P)
        r = O.radius(P, c)
        h = 2.0 * r / np.sqrt(len(P))
        areas = np.full(len(P), 4 * np.pi * r * r / len(P))
        g = O.grid_setup(P, hCoef=hCoef)
        src = O.point_sources(P, Nn, areas, h)
    else:
        V, faces = O.read_obj(path)
        g = O.grid_setup(V, hCoef=hCoef)
        src = O.mesh_sources(V, faces)
    return g, src


def main():
    path, hCoef = sys.argv[1], float(sys.argv[2])
    g, src = setup(os.path.join(ROOT, path), hCoef)
    n = g.n
    N = n ** 3
    h = g.cell
    A = O.constraint_matrix(g, src.pos).tocsr()
    m = A.shape[0]
    print("n=%d N=%d S=%d m=%d" % (n, N, len(src.pos), m))
    rng = np.random.default_rng(0)
    # right-hand side: a smooth-ish field with the character of D^T Y (use random; iteration counts are what matter)
    b = rng.standard_normal(N)
    lam1 = (2.0 - 2.0 * np.cos(np.pi * np.arange(n) / n)) / (h * h)
    lam = lam1[:, None, None] + lam1[None, :, None] + lam1[None, None, :]
    lam[0, 0, 0] = 1.0

    def Kplus(v):
        w = sfft.dctn(v.reshape(n, n, n), type=2, norm="ortho")
        w /= lam
        w[0, 0, 0] = 0.0
        return sfft.idctn(w, type=2, norm="ortho").reshape(-1)

    L = O.laplacian_matrix(g).tocsr()
    K = -L
    G = (A @ A.T).tocsc()
    B = (A @ K @ A.T).tocsr()
    d = G.diagonal()
    print("G: nnz/row %.1f  diag [%.3g, %.3g]" % (G.nnz / m, d.min(), d.max()))
    Dm12 = sp.diags(1.0 / np.sqrt(d))
    Gs = (Dm12 @ G @ Dm12).tocsr()
    t = time.time()
    try:
        emax = spla.eigsh(Gs, k=1, which="LA", return_eigenvectors=False, tol=1e-4)[0]
        lu = spla.splu(G)
        emin = 1.0 / spla.eigsh(spla.LinearOperator((m, m), matvec=lambda v: Dm12 @ lu.solve(Dm12 @ v) * 1.0, dtype=float), k=1, which="LA",
                                return_eigenvectors=False, tol=1e-4)[0]
        # (D^-1/2 G D^-1/2)^-1 = D^1/2 G^-1 D^1/2
        Dp12 = sp.diags(np.sqrt(d))
        emin = 1.0 / spla.eigsh(spla.LinearOperator((m, m), matvec=lambda v: Dp12 @ lu.solve(Dp12 @ v), dtype=float), k=1, which="LA",
                                return_eigenvectors=False, tol=1e-4)[0]
        print("Jacobi-scaled G: eig in [%.3e, %.3e]  cond %.3e  (%.1fs)" % (emin, emax, emax / emin, time.time() - t))
    except Exception as e:  # noqa
        print("eig failed", e)
        lu = spla.splu(G)
        emin, emax = 1e-4, 2.0

    def Pm(v):
        return v - v.mean()

    def S(v):
        return A @ Kplus(A.T @ v)

    gvec = A @ Kplus(b)

    def run(name, Ginv_apply, tol=1e-8, maxit=400):
        mu = np.full(m, b.sum() / m)
        r = Pm(gvec - S(mu))
        rr0 = r @ r

        def prec(r):
            return Pm(Ginv_apply(B @ Ginv_apply(r)))

        z = prec(r)
        p = z.copy()
        rz = r @ z
        it = 0
        hist = []
        while it < maxit:
            Sp = Pm(S(p))
            alpha = rz / (p @ Sp)
            mu += alpha * p
            r -= alpha * Sp
            it += 1
            rr = r @ r
            hist.append(np.sqrt(rr / rr0))
            if rr <= tol * tol * rr0:
                break
            z = prec(r)
            rz_new = r @ z
            p = z + (rz_new / rz) * p
            rz = rz_new
        print("%-34s iters %4d  final rel %.2e" % (name, it, hist[-1]))
        return it

    run("exact G^-1 (LU)", lambda v: lu.solve(v))
    run("identity", lambda v: v)
    run("diag(G)^-1", lambda v: v / d)

    def cheb(k, lo, hi):
        # Chebyshev semi-iteration for Gs y = v on [lo, hi] (Jacobi-scaled), zero start: a fixed polynomial in Gs
        theta, delta = 0.5 * (hi + lo), 0.5 * (hi - lo)
        sigma = theta / delta

        def apply(v):
            vs = v / np.sqrt(d)
            rho = 1.0 / sigma
            y = vs / theta
            dvec = y.copy()
            for _ in range(k - 1):
                rho_new = 1.0 / (2.0 * sigma - rho)
                res = vs - Gs @ y
                dvec = rho_new * rho * dvec + (2.0 * rho_new / delta) * res
                y = y + dvec
                rho = rho_new
            return y / np.sqrt(d)
        return apply

    for k in (2, 4, 8, 16, 32):
        for lo_frac in (None, 0.1, 0.03):
            lo = emin if lo_frac is None else lo_frac * emax
            run("cheb k=%d lo=%.3g hi=%.3g" % (k, lo, emax * 1.02), cheb(k, lo, emax * 1.02))

    # fixed number of Jacobi-PCG steps on G (non-linear in v, but often fine)
    def pcg_fixed(k):
        def apply(v):
            y = np.zeros(m)
            r = v.copy()
            z = r / d
            p = z.copy()
            rz = r @ z
            for _ in range(k):
                Gp = G @ p
                a = rz / (p @ Gp)
                y += a * p
                r -= a * Gp
                z = r / d
                rzn = r @ z
                p = z + (rzn / rz) * p
                rz = rzn
            return y
        return apply

    for k in (4, 8, 16, 32):
        run("inner Jacobi-PCG k=%d" % k, pcg_fixed(k))


if __name__ == "__main__":
    main()

"""The identities behind csrc/shm_schur.hip.h, checked in numpy at n = 8 (no GPU, no product code): the Neumann Green's function of the 7-point Laplacian
is the sum of eight images of one cosine table on the integer lattice, and the Schur complement entry of two trilinear rows is a 6 x 6 x 6 weighted sum of
table entries.  (The device kernels themselves are held to the operator they replace in tests/test_gpu_parity.py::test_explicit_schur_complement_is_A_Kplus_AT.)"""
import numpy as np

N_SIDE, H = 8, 0.37


def _setup():
    n, h = N_SIDE, H
    k = np.arange(n)
    lam1 = (2 - 2 * np.cos(np.pi * k / n)) / h ** 2
    V = np.cos(np.pi * np.outer(np.arange(n) + 0.5, k) / n)          # DCT-II basis V[x, k]
    nrm = np.where(k == 0, n, n / 2.0)
    lam = lam1[:, None, None] + lam1[None, :, None] + lam1[None, None, :]
    inv = np.zeros_like(lam)
    inv[lam > 0] = 1 / lam[lam > 0]
    gam = np.where(k == 0, 1 / (2 * n), 1.0 / n)
    C = np.cos(np.pi * np.outer(np.arange(n + 1), k) / n)            # C[d, k]
    W0 = gam[:, None, None] * gam[None, :, None] * gam[None, None, :] * inv
    T = np.einsum("ai,bj,ck,ijk->abc", C, C, C, W0)
    return n, h, V, nrm, inv, T


def _kplus(V, nrm, inv, x, y):
    f = [V[x[a], :] * V[y[a], :] / nrm for a in range(3)]
    return np.einsum("i,j,k,ijk->", f[0], f[1], f[2], inv)


def _fold(e, n):
    return e if e <= n else 2 * n - e


def test_green_function_is_eight_images_of_one_table():
    n, h, V, nrm, inv, T = _setup()
    rng = np.random.default_rng(1)
    for _ in range(100):
        x, y = rng.integers(0, n, 3), rng.integers(0, n, 3)
        s = 0.0
        for sg in np.ndindex(2, 2, 2):
            idx = [abs(x[a] - y[a]) if sg[a] == 0 else _fold(x[a] + y[a] + 1, n) for a in range(3)]
            s += T[idx[0], idx[1], idx[2]]
        assert abs(s - _kplus(V, nrm, inv, x, y)) < 1e-15


def test_kplus_is_the_pseudo_inverse_of_the_mirror_laplacian():
    n, h, V, nrm, inv, T = _setup()

    def K(u):   # -L with the reference's Neumann convention: an out-of-grid neighbour is the node itself
        u = u.reshape(n, n, n)
        out = np.zeros_like(u)
        for a in range(3):
            up, um = np.roll(u, -1, a), np.roll(u, 1, a)
            sl = [slice(None)] * 3
            sl[a] = n - 1
            up[tuple(sl)] = u[tuple(sl)]
            sl[a] = 0
            um[tuple(sl)] = u[tuple(sl)]
            out += (2 * u - up - um) / h ** 2
        return out.ravel()

    y = np.array([2, 5, 1])
    col = np.array([_kplus(V, nrm, inv, [i, j, l], y) for i in range(n) for j in range(n) for l in range(n)])
    e = np.zeros(n ** 3)
    e[(y[0] * n + y[1]) * n + y[2]] = 1
    assert np.abs(K(col) - (e - 1.0 / n ** 3)).max() < 1e-13


def test_schur_entry_of_two_trilinear_rows():
    n, h, V, nrm, inv, T = _setup()
    rng = np.random.default_rng(2)
    for _ in range(20):
        Xi, Xj = rng.integers(0, n - 1, 3), rng.integers(0, n - 1, 3)
        ti, tj = rng.random(3), rng.random(3)
        per = []
        for a in range(3):
            wi, wj = (1 - ti[a], ti[a]), (1 - tj[a], tj[a])
            D, E = Xi[a] - Xj[a], Xi[a] + Xj[a] + 1
            idx = [abs(D - 1), abs(D), abs(D + 1), _fold(E, n), _fold(E + 1, n), _fold(E + 2, n)]
            w = [wi[0] * wj[1], wi[0] * wj[0] + wi[1] * wj[1], wi[1] * wj[0], wi[0] * wj[0], wi[0] * wj[1] + wi[1] * wj[0], wi[1] * wj[1]]
            per.append((idx, w))
        assembled = sum(per[0][1][p] * per[1][1][q] * per[2][1][r] * T[per[0][0][p], per[1][0][q], per[2][0][r]]
                        for p in range(6) for q in range(6) for r in range(6))
        direct = 0.0
        for ci in np.ndindex(2, 2, 2):
            wi = np.prod([ti[a] if ci[a] else 1 - ti[a] for a in range(3)])
            for cj in np.ndindex(2, 2, 2):
                wj = np.prod([tj[a] if cj[a] else 1 - tj[a] for a in range(3)])
                direct += wi * wj * _kplus(V, nrm, inv, np.array(Xi) + ci, np.array(Xj) + cj)
        assert abs(assembled - direct) < 1e-15

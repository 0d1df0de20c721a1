"""CPU tests (no GPU): the oracle against its golden vectors, BASELINE.md's spot values, operator identities
and analytic known-answer cases.  The oracle is test infrastructure; these tests pin it."""
import math
import os
import sys

import numpy as np
import pytest

from conftest import ROOT, c_, load_golden

sys.path.insert(0, os.path.join(ROOT, "oracle"))
import shm_oracle as o  # noqa: E402


@pytest.fixture(scope="module")
def bunny():
    return o.read_obj(os.path.join(ROOT, "data", "bunny_small.obj"))


def test_preprocessing_matches_survey_values(bunny):
    """c=(-0.1645,-0.2038,0.1243), r=1.48649, h=0.095001, lambda=10.5262, cell(64)=0.094380."""
    V, F = bunny
    assert V.shape == (1430, 3) and len(F) == 2856
    c = o.centroid(V)
    assert np.allclose(c, [-0.1645, -0.2038, 0.1243], atol=5e-5)
    assert abs(o.radius(V, c) - 1.48649) < 1e-5
    assert abs(o.mean_edge_length(V, F) - 0.095001) < 1e-6
    src = o.mesh_sources(V, F)
    assert abs(src.lam - 10.5262) < 1e-4
    g = o.grid_setup(V, 2.0, 2.0)
    assert g.n == 64 and abs(g.cell - 0.094380) < 1e-6


def test_obj_loader_strips_unreferenced_vertices():
    """rocker.obj: 8884 v lines, 142 unreferenced -> 8742; centroid/radius use the stripped set."""
    V, F = o.read_obj(os.path.join(ROOT, "data", "rocker.obj"))
    assert V.shape[0] == 8742 and len(F) == 13819
    c = o.centroid(V)
    assert np.allclose(c, [-0.05023, 0.24140, 0.02235], atol=1e-5)
    assert abs(o.radius(V, c) - 4.17290) < 1e-5


# BASELINE.md section 2 spot values (independent restatement by the surveyor)
SPOT = {
    16: dict(p000=4.607456591, pend=4.570540191, mid=-0.053935477, q=1.259426334, e=3.129727516, mn=-0.187913473, amin=1913,
             mean=2.028955677, shift=-1.418e-3, m=79),
    32: dict(p000=4.537946781, pend=4.477278558, mid=-0.285922354, q=1.293558628, e=3.103254843, mn=-0.455887167, amin=16881,
             mean=2.009514617, shift=-2.039e-4, m=316),
    64: dict(p000=4.516073089, pend=4.464383064, mid=-0.349381879, q=1.335604869, e=3.108209551, mn=-0.552552469, amin=133027,
             mean=2.027988741, shift=-1.217e-4, m=1129),
}


@pytest.mark.parametrize("n", [16, 32, 64])
def test_golden_phi_matches_baseline_spot_values(n):
    name = "bunny_small_n%d" % n
    if not os.path.exists(os.path.join(ROOT, "tests", "golden", name + ".npz")):
        pytest.skip("fixture not generated")
    d = load_golden(name)
    phi, s = d["phi"], SPOT[n]
    at = lambda i, j, k: phi[i + j * n + k * n * n]  # noqa: E731
    tol = 5e-9
    assert abs(at(0, 0, 0) - s["p000"]) < tol and abs(phi.max() - s["p000"]) < tol
    assert abs(at(n - 1, n - 1, n - 1) - s["pend"]) < tol
    assert abs(at(n // 2, n // 2, n // 2) - s["mid"]) < tol
    assert abs(at(n // 4, n // 2, 3 * n // 4) - s["q"]) < tol
    assert abs(at(n - 1, 0, n // 2) - s["e"]) < tol
    assert abs(phi.min() - s["mn"]) < tol and int(phi.argmin()) == s["amin"]
    assert abs(phi.mean() - s["mean"]) < tol
    assert abs(float(d["shift"]) - s["shift"]) < 2e-6
    assert int(d["m"]) == s["m"]


def test_operator_identities():
    """L = L^T, L 1 = 0, D 1 = 0, negative semi-definite; closed forms used by the HIP kernels equal the assembled
    matrices."""
    g = o.Grid(n=9, bbox_min=np.array([-1.0, -2.0, 0.5]), cell=0.37)
    L = o.laplacian_matrix(g)
    D = o.gradient_matrix(g)
    assert abs(L - L.T).max() < 1e-12
    assert np.abs(L @ np.ones(g.N)).max() < 1e-12
    assert np.abs(D @ np.ones(g.N)).max() < 1e-12
    rng = np.random.default_rng(0)
    u = rng.standard_normal(g.N)
    assert u @ (L @ u) < 0
    # matrix-free Neumann graph Laplacian
    U = u.reshape(g.n, g.n, g.n)  # [k,j,i]
    acc = np.zeros_like(U)
    for ax in range(3):
        up = np.roll(U, -1, axis=ax)
        dn = np.roll(U, 1, axis=ax)
        sl_last = [slice(None)] * 3
        sl_last[ax] = -1
        sl_first = [slice(None)] * 3
        sl_first[ax] = 0
        up[tuple(sl_last)] = U[tuple(sl_last)]
        dn[tuple(sl_first)] = U[tuple(sl_first)]
        acc += up + dn - 2 * U
    assert np.abs(acc.ravel() / g.cell ** 2 - L @ u).max() < 1e-10
    # closed-form D^T
    Y = rng.standard_normal((g.N, 3))
    ref = D.T @ Y.reshape(-1)
    n = g.n
    b = np.zeros((n, n, n))
    Yg = Y.reshape(n, n, n, 3)
    for p, ax in ((0, 2), (1, 1), (2, 0)):
        Ya = np.moveaxis(Yg[..., p], ax, 0)
        ba = np.zeros_like(Ya)
        ba[1:] += Ya[:-1]
        ba[n - 1] += Ya[n - 1]
        ba[:n - 1] -= Ya[:n - 1]
        ba[n - 2] -= Ya[n - 1]
        b += np.moveaxis(ba, 0, ax)
    assert np.abs(b.ravel() / g.cell - ref).max() < 1e-10


@pytest.mark.parametrize("case", ["bunny_small_n16", "bunny_pc_n16", "polygon_bear_n16"])
def test_c_oracle_matches_lu_golden(oracle_c, case):
    d = load_golden(case)
    n = int(d["n"])
    phi = np.zeros(n ** 3)
    st = np.zeros(5)
    scrub = 0 if "pc" in case else 1
    rc = oracle_c.shmo_compute_distance(n, c_(d["bbox_min"]), float(d["cell"]), len(d["area"]), c_(d["pos"]).reshape(-1),
                                        c_(d["wnormal"]).reshape(-1), c_(d["area"]), float(d["lam"]), scrub, 0, 1e-13, 100000, phi, st)
    assert rc == 0 and int(st[0]) == int(d["m"])
    assert np.abs(phi - d["phi"]).max() < 1e-9
    assert st[3] < 1e-11  # max |A x|
    assert abs(st[4] - float(d["shift"])) < 1e-10


def test_c_oracle_stages_match_golden(oracle_c):
    d = load_golden("bunny_small_n16")
    n = int(d["n"])
    S = len(d["area"])
    Y = np.zeros(3 * n ** 3)
    oracle_c.shmo_conv_normalize(n, c_(d["bbox_min"]), float(d["cell"]), S, c_(d["pos"]).reshape(-1), c_(d["wnormal"]).reshape(-1),
                                 float(d["lam"]), 0, n, Y)
    assert np.abs(Y.reshape(-1, 3) - d["Y"]).max() < 1e-12
    b = np.zeros(n ** 3)
    oracle_c.shmo_divergence(n, float(d["cell"]), Y, 1, b)
    assert np.abs(b - d["b"]).max() < 1e-10 * np.abs(d["b"]).max()
    nodes = np.zeros(8 * S, dtype=np.int64)
    coeffs = np.zeros(8 * S)
    m = oracle_c.shmo_constraint_rows(n, c_(d["bbox_min"]), float(d["cell"]), S, c_(d["pos"]).reshape(-1), nodes, coeffs)
    assert m == int(d["m"])
    assert np.array_equal(nodes[:8 * m].reshape(-1, 8), d["c_nodes"])
    assert np.array_equal(coeffs[:8 * m].reshape(-1, 8), d["c_coeffs"])


@pytest.mark.parametrize("n", [16, 32])
def test_fast_integration_bfs_matches_golden(oracle_c, n):
    """integrateGreedily (:224-275): C restatement vs the python restatement's fixture (order-dependent BFS)."""
    d = load_golden("bunny_small_fast_n%d" % n)
    phi = np.zeros(n ** 3)
    st = np.zeros(5)
    rc = oracle_c.shmo_compute_distance(n, c_(d["bbox_min"]), float(d["cell"]), len(d["area"]), c_(d["pos"]).reshape(-1),
                                        c_(d["wnormal"]).reshape(-1), c_(d["area"]), float(d["lam"]), 1, 1, 0.0, 0, phi, st)
    assert rc == 0
    assert np.abs(phi - d["phi"]).max() < 1e-10


def test_projected_cg_equals_lu(bunny):
    """The matrix-free formulation the GPU uses equals the reference's KKT/LU solve."""
    V, F = bunny
    g = o.grid_setup(V, 2.0, 0.0)
    src = o.mesh_sources(V, F)
    d = load_golden("bunny_small_n16")
    A = o.constraint_matrix(g, src.pos)
    phi, it, rel = o.projected_cg(g, d["b"], A, tol=1e-13)
    phi = phi - o.source_average(g, phi, src)
    assert np.abs(phi - d["phi"]).max() < 1e-10
    # dropping the constraints is NOT the same problem (SURVEY trap #1)
    import scipy.sparse.linalg as spla
    L = o.laplacian_matrix(g)
    x = spla.lsqr(L, d["b"], atol=1e-12, btol=1e-12, iter_lim=5000)[0]
    phi_plain = -x
    phi_plain -= o.source_average(g, phi_plain, src)
    assert np.abs(phi_plain - d["phi"]).max() > 0.1


def test_non_finite_rhs_poisons_the_solution_like_the_lu(bunny, oracle_c):
    """A non-finite entry of D^T Y (point overload: no scrub, :180) makes the reference's LU solve return NaN; the C oracle reports
    that (rc 2, NaN phi) instead of leaving its CG loop through a false NaN comparison, and the HIP path raises SHM_ERR_BREAKDOWN
    (tests/test_gpu_parity.py::test_every_data_file_matches_c_oracle_32[SprayBottle.pc])."""
    V, F = bunny
    g = o.grid_setup(V, 2.0, 0.0)
    src = o.mesh_sources(V, F)
    d = load_golden("bunny_small_n16")
    b = d["b"].copy()
    b[1234] = np.inf
    A = o.constraint_matrix(g, src.pos)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        phi_lu, _ = o.solve_kkt_lu(g, b, A)
    assert not np.isfinite(phi_lu).any()     # SuperLU, like Eigen::SparseLU, has no NaN guard: the whole vector is lost
    nodes = np.ascontiguousarray(d["c_nodes"], dtype=np.int64).reshape(-1)
    coeffs = np.ascontiguousarray(d["c_coeffs"], dtype=np.float64).reshape(-1)
    phi = np.zeros(g.N)
    st = np.zeros(3)
    rc = oracle_c.shmo_constrained_solve(g.n, g.cell, b, len(nodes) // 8, nodes, coeffs, 1e-12, 1000, phi, st)
    assert rc == 2 and np.isnan(phi).all()


def test_sphere_known_answer(oracle_c):
    """Point samples on a unit sphere with outward normals: SHM -> phi ~ |x|-1 (sign, zero set, monotone in radius)."""
    # Fibonacci sphere
    P = 2000
    i = np.arange(P) + 0.5
    z = 1 - 2 * i / P
    th = math.pi * (1 + 5 ** 0.5) * i
    pts = np.stack([np.sqrt(1 - z * z) * np.cos(th), np.sqrt(1 - z * z) * np.sin(th), z], axis=1)
    areas = np.full(P, 4 * math.pi / P)
    h = math.sqrt(4 * math.pi / P)
    g, src, phi, info = o.compute_distance_points(pts, pts.copy(), areas, h, hCoef=1.0, solver="cg", tol=1e-10)
    n = g.n
    X = o.node_positions(g)
    r = np.linalg.norm(X, axis=1)
    exact = r - 1.0
    inner = r < 2.5
    assert np.all(phi[r < 0.7] < 0) and np.all(phi[(r > 1.3) & inner] > 0)
    assert np.abs(phi[inner] - exact[inner]).max() < 0.15   # O(h) discretisation, cell = 4/31
    # C oracle agrees with the numpy oracle on the same inputs
    out = np.zeros(n ** 3)
    st = np.zeros(5)
    oracle_c.shmo_compute_distance(n, c_(g.bbox_min), g.cell, P, c_(src.pos).reshape(-1), c_(src.wnormal).reshape(-1), c_(src.area),
                                   src.lam, 0, 0, 1e-12, 100000, out, st)
    assert np.abs(out - phi).max() < 1e-7

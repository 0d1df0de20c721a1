"""GPU parity tests: the HIP path, called through the C ABI (include/shm_grid.h), against
(i) the committed golden fixtures (numpy/scipy oracle: literal KKT + SuperLU) and (ii) the C oracle on
the same inputs.  Tolerances: fp64 L-infinity on phi < 1e-5 is the north-star gate; the intermediate
stages are held to much tighter bounds because they are the same arithmetic in a different order."""
import numpy as np
import pytest

from conftest import c_, load_golden

pytestmark = pytest.mark.gpu

PHI_GATE = 1e-5  # BASELINE.json north_star: L-inf vs CPU reference < 1e-5 (fp64)


def make_solver(shm, d, **kw):
    s = shm.GridSolver(**kw)
    s.set_problem(d["pos"], d["wnormal"], d["area"], float(d["lam"]), int(d["n"]), d["bbox_min"], float(d["cell"]))
    return s


Y_BUDGET = 1e-8   # error budget of the tiered fp64 Step 1 on the normalised field: terms below e^-8 of a node's dominant terms
                  # are evaluated in packed fp32.  phi inherits ~0.05 of it (measured), two decades inside the 1e-7 the phi tests hold.


@pytest.mark.parametrize("exact", [False, True])
@pytest.mark.parametrize("case", ["bunny_small_n16", "bunny_small_n24", "bunny_small_n32", "polygon_bear_n16", "bunny_pc_n16"])
def test_conv_normalize_matches_golden(shm, case, exact, monkeypatch):
    """Y against the oracle's fixtures: the shipped tiered kernel within its budget, the all-fp64 kernel (SHM_CONV_EXACT=1: every pair in the
    reference's arithmetic) to rounding."""
    d = load_golden(case)
    if exact:
        monkeypatch.setenv("SHM_CONV_EXACT", "1")
    s = make_solver(shm, d)
    s.run_conv()
    Y = np.stack([s.get_field(k) for k in (0, 1, 2)], axis=1)
    err = np.abs(Y - d["Y"]).max()
    assert err < (1e-11 if exact else Y_BUDGET), err
    assert np.abs(np.linalg.norm(Y, axis=1) - 1).max() < 1e-14


@pytest.mark.parametrize("path,hcoef", [("bunny_small.obj", 3.0), ("rocker.obj", 3.0), ("knot.obj", 3.0), ("chair.obj", 2.0), ("SprayBottle.pc", 3.0), ("bunny.pc", 3.0),
                                        # 256^3: the sizes from which the far rule is decided by a sample (round 5) -- SprayBottle.pc is the input that read 1.8e-8 before the rule
                                        # was confined to blocks near the sources, rocker.pc the worst of the shipped form (4.1e-9)
                                        ("bunny_small.obj", 4.0), ("SprayBottle.pc", 4.0), ("rocker.pc", 4.0)])
def test_tiered_conv_stays_within_its_budget(shm, path, hcoef, monkeypatch):
    """The precision tiers of the fp64 Step 1 (csrc/shm_conv_tiered.hip.h) at sizes where more than half of the (node, source) pairs take the packed-fp32
    tier: Y within the budget of the all-fp64 kernel, phi within a tenth of the 1e-7 the parity tests hold, and the executed-pair counters consistent."""
    import os
    from conftest import ROOT
    from signed_heat_3d_amd.host_abi import HostSolver
    pre = HostSolver(os.path.join(ROOT, "data", path)).preprocess(hCoef=hcoef)
    scrub = not path.endswith(".pc")
    out = {}
    for exact in (True, False):
        if exact:
            monkeypatch.setenv("SHM_CONV_EXACT", "1")
        else:
            monkeypatch.delenv("SHM_CONV_EXACT")
        s = shm.GridSolver()
        s.set_problem(pre["pos"], pre["wnormal"], pre["area"], pre["lam"], pre["n"], pre["bbox_min"], pre["cell"])
        st = s.solve(tol=1e-10, scrub=scrub)
        phi, _ = s.get_phi()
        s.run_conv()
        out[exact] = (np.stack([s.get_field(k) for k in (0, 1, 2)], axis=1), phi, st)
        s.close()
    (Ye, pe, ste), (Yt, pt, stt) = out[True], out[False]
    ok = np.isfinite(Ye).all(axis=1)
    assert (np.isfinite(Yt).all(axis=1) == ok).all()
    assert np.abs(Yt[ok] - Ye[ok]).max() < Y_BUDGET, np.abs(Yt[ok] - Ye[ok]).max()
    assert np.abs(pt - pe).max() < 1e-8 * max(1.0, np.abs(pe).max()), np.abs(pt - pe).max()
    nominal = float(pre["n"]) ** 3 * pre["S"]
    assert ste.pairs_fp32 == 0 or ste.pairs_fp64 > 0          # all-fp64 kernel: (almost) everything in fp64
    assert stt.pairs_fp64 + stt.pairs_fp32 <= 1.03 * nominal   # never more than the nominal N S (cluster padding aside)
    # the tiers are really in use at this size (SprayBottle.pc at 128^3, lambda * cell = 3.7: most far sources fail the exponent-range test of the packed-fp32
    # tier -- one offset per block -- and stay in fp64; 0.59 of its pairs are dropped)
    assert stt.pairs_fp32 > (0.1 if path != "SprayBottle.pc" else 0.0) * nominal and stt.pairs_fp64 < 0.7 * nominal
    assert stt.pairs_fp64 > 0


@pytest.mark.parametrize("exact", [False, True])
@pytest.mark.parametrize("lam_scale,n", [(4.0, 32), (8.0, 24)])
def test_conv_far_clusters_in_fp32_keep_fp64_accuracy(shm, oracle_c, lam_scale, n, exact, monkeypatch):
    """With a short diffusion length most sources are "far" for most node blocks.  The all-fp64 kernel (SHM_CONV_EXACT=1) sends a cluster to fp32 only when
    its terms are below e^-25 of the tile's dominant term: Y agrees with the all-fp64 C oracle to rounding.  The shipped tiered kernel (terms below e^-8
    of a block's dominant terms in packed fp32, vanishing ones dropped) stays within its budget."""
    if exact:
        monkeypatch.setenv("SHM_CONV_EXACT", "1")
    d = load_golden("bunny_small_n16")
    lam = float(d["lam"]) * lam_scale
    cell = float(d["cell"]) * 15 / (n - 1)
    s = shm.GridSolver()
    s.set_problem(d["pos"], d["wnormal"], d["area"], lam, n, d["bbox_min"], cell)
    s.run_conv()
    Y = np.stack([s.get_field(k) for k in (0, 1, 2)], axis=1)
    ref = np.zeros(3 * n ** 3)
    oracle_c.shmo_conv_normalize(n, c_(d["bbox_min"]), cell, len(d["area"]), c_(d["pos"]).reshape(-1), c_(d["wnormal"]).reshape(-1), lam, 0, n, ref)
    ref = ref.reshape(-1, 3)
    ok = np.isfinite(ref).all(axis=1)
    assert ok.mean() > 0.5            # beyond lambda*r ~ 745 the fp64 reference itself underflows to 0/0
    assert np.abs(Y[ok] - ref[ok]).max() < (1e-10 if exact else Y_BUDGET)


@pytest.mark.parametrize("lam_scale", [2.0, 4.0, 6.0, 10.0])
def test_far_tier_exponent_range_guard(shm, oracle_c, lam_scale):
    """A short diffusion length on a coarse grid (tCoef < 1 via --t: lambda = 1 / (h sqrt(tCoef)) does not depend on the cell): the packed-fp32 tier carries one
    exponent offset per 8 x 8 x 4 block, and beyond lambda * cell ~ 3 the spread of the nodes' own dominant terms over a block plus the drop threshold leaves
    the fp32 exponent range: such a source stays in fp64 (`in_range`, shm_conv_tiered.hip.h).  lam_scale 2 keeps the tier for 40 % of the pairs, 4 for a few,
    6 and 10 (tCoef = 0.01) for none; Y stays within the budget of the all-fp64 C oracle in every case."""
    d = load_golden("bunny_small_n64")
    n, lam, cell = int(d["n"]), float(d["lam"]) * lam_scale, float(d["cell"])
    s = shm.GridSolver()
    s.set_problem(d["pos"], d["wnormal"], d["area"], lam, n, d["bbox_min"], cell)
    st = s.solve(scrub=True, allow_noconv=True, max_iters=2)
    s.run_conv()
    Y = np.stack([s.get_field(k) for k in (0, 1, 2)], axis=1)
    ref = np.zeros(3 * n ** 3)
    oracle_c.shmo_conv_normalize(n, c_(d["bbox_min"]), cell, len(d["area"]), c_(d["pos"]).reshape(-1), c_(d["wnormal"]).reshape(-1), lam, 0, n, ref)
    ref = ref.reshape(-1, 3)
    ok = np.isfinite(ref).all(axis=1)
    assert ok.mean() > 0.5
    assert np.isfinite(Y[ok]).all()
    err = np.abs(Y[ok] - ref[ok]).max()
    print("\nlambda x %g (lambda * cell = %.2f): pairs fp64 %.3e packed fp32 %.3e, max|dY| = %.2e" % (lam_scale, lam * cell, st.pairs_fp64, st.pairs_fp32, err))
    assert err < Y_BUDGET, err
    if lam_scale >= 6.0:
        assert st.pairs_fp32 < 1e-3 * st.pairs_fp64   # outside the fp32 exponent range: (almost) everything that is not dropped is evaluated in fp64
    elif lam_scale <= 2.0:
        assert st.pairs_fp32 > 0.1 * st.pairs_fp64


@pytest.mark.parametrize("precision", [64, 32])
def test_tiered_kernel_hands_over_beyond_its_exponent_span(shm, oracle_c, precision):
    """The tiered kernel puts a term's power of two into the exponent field by an integer add, each 8 x 8 x 4 block relative to its own exponent: valid while a block's
    evaluated exponents span less than 2^-990 (Solver::tier_exponent_span_ok).  lambda x 20 on the 16^3 fixture (lambda * cell = 83: a block spans 4 x 5.2 x 83 nats)
    is beyond that: Step 1 must run in the all-fp64 kernel (fp64 handle: no packed-fp32 pairs at all, Y to 1e-10 of the C oracle, the reference's 0/0 nodes far from every
    source reproduced) or the classic fp32 kernel (fp32 handle: finite everywhere, directions of the fp64 oracle)."""
    d = load_golden("bunny_small_n16")
    n, lam, cell = int(d["n"]), float(d["lam"]) * 20.0, float(d["cell"])
    s = shm.GridSolver(precision=precision)
    s.set_problem(d["pos"], d["wnormal"], d["area"], lam, n, d["bbox_min"], cell)
    s.run_conv()
    Y = np.stack([s.get_field(k) for k in (0, 1, 2)], axis=1)
    st = s.solve(scrub=True, allow_noconv=True, max_iters=2)
    s.close()
    ref = np.zeros(3 * n ** 3)
    oracle_c.shmo_conv_normalize(n, c_(d["bbox_min"]), cell, len(d["area"]), c_(d["pos"]).reshape(-1), c_(d["wnormal"]).reshape(-1), lam, 0, n, ref)
    ref = ref.reshape(-1, 3)
    ok = np.isfinite(ref).all(axis=1)
    assert ok.mean() > 0.05      # (beyond lambda r ~ 745 the reference's own sum underflows: 0/0 -- the far corners of this grid)
    if precision == 64:
        assert st.pairs_fp32 == 0 and st.pairs_fp64 > 0
        assert (np.isfinite(Y).all(axis=1) == ok).all()
        assert np.abs(Y[ok] - ref[ok]).max() < 1e-10
    else:
        assert np.isfinite(Y).all()
        dots = (Y[ok] * ref[ok]).sum(axis=1)
        assert np.median(1 - dots) < 1e-6 and np.quantile(1 - dots, 0.99) < 1e-3


def test_opts_dual_form_and_step1_budget(shm):
    """ABI 5: shm_opts.dual_form selects the dual solver's form (stats.cg_form says what ran; the same phi in every form), shm_opts.step1_budget moves the
    tiers' thresholds together (a larger budget sends more pairs to the packed-fp32 tier and stays inside itself); out-of-range values are refused."""
    d = load_golden("bunny_small_n32")
    s = make_solver(shm, d)
    want = {"direct": 2, "explicit_s_cg": 3, "through_grid": 0}
    phis = {}
    for form, cg in want.items():
        st = s.solve(tol=1e-10, dual_form=form)
        assert st.solver == 2 and st.cg_form == cg, (form, st.solver, st.cg_form)
        phis[form] = s.get_phi()[0]
        assert np.abs(phis[form] - d["phi"]).max() < 1e-7
    assert np.abs(phis["direct"] - phis["through_grid"]).max() < 1e-8
    st_auto = s.solve(tol=1e-10)
    assert st_auto.cg_form == 2      # (what AUTO picks for 316 rows)
    # the Step-1 budget
    far = {}
    for budget in (0.0, 1e-6, 1e-10):
        st = s.solve(tol=1e-10, step1_budget=budget)
        far[budget] = st.pairs_fp32 / (st.pairs_fp32 + st.pairs_fp64)
        assert np.abs(s.get_phi()[0] - d["phi"]).max() < (1e-7 if budget < 1e-7 else 1e-5)
    assert far[1e-6] > far[0.0] > far[1e-10]
    for bad in (1e-2, 1e-14, -1.0):
        with pytest.raises(shm.ShmError):
            s.solve(step1_budget=bad)
    with pytest.raises(KeyError):
        s.solve(dual_form="no_such_form")
    s.close()


@pytest.mark.parametrize("precision", [64, 32])
def test_step1_is_translation_invariant(shm, precision):
    """Step 1 computes in grid-centred coordinates (Solver::set_problem): a mesh far from the origin -- here the 32^3 fixture moved by (1000, -2000, 500), where
    fp32 resolves coordinates to 1e-4 only -- gives the Y of the mesh at the origin (fp64: to rounding of the fp64 differences; fp32: to fp32 rounding)."""
    d = load_golden("bunny_small_n32")
    shift = np.array([1000.0, -2000.0, 500.0])
    out = []
    for t in (np.zeros(3), shift):
        s = shm.GridSolver(precision=precision)
        s.set_problem(d["pos"] + t, d["wnormal"], d["area"], float(d["lam"]), int(d["n"]), d["bbox_min"] + t, float(d["cell"]))
        s.run_conv()
        out.append(np.stack([s.get_field(k) for k in (0, 1, 2)], axis=1))
        s.close()
    err = np.abs(out[0] - out[1]).max()
    assert err < (1e-9 if precision == 64 else 2e-4), err
    assert np.abs(out[0] - d["Y"]).max() < (Y_BUDGET if precision == 64 else 1e-3)


# ---- Step 1 at BASELINE.json's full sizes against the C oracle (the reference's serial loops) on sampled z-planes ------------------------------------------
# The classification geometry of the tiered kernel changes with n (a block spans 2.5 e-folds of the kernel at 256^3, 1.3 at 512^3), so the budget is
# checked where it is used: configs[1] (256^3), configs[3] (bunny.pc 512^3), configs[2] / [4] in the reference's fp64 arithmetic (rocker 512^3, SprayBottle.pc
# 1024^3) and the file with the largest measured error (knot.obj), tiered and all-fp64; the fp32 kernel of configs[2] / [4] with its own bound.
STEP1_FULL = [("bunny_small.obj", 4.0, 64), ("bunny.pc", 5.0, 64), ("knot.obj", 4.0, 64), ("rocker.obj", 5.0, 64), ("knot.obj", 5.0, 64), ("chair.obj", 5.0, 64),
              ("SprayBottle.pc", 6.0, 64), ("rocker.obj", 5.0, 32), ("SprayBottle.pc", 6.0, 32)]
Y_BUDGET_F32 = 2e-3   # fp32 kernel: every pair in fp32 (relative error ~1e-5 per term incl. the exponent); measured worst 3e-4 where sheets cancel
# Where the tiers' error is largest: nodes on the medial axis of the source geometry (the sheets' contributions cancel: |X| / sum|terms| = 2e-3 ... 8e-3), located by
# tools/tier_worst_nodes.py on the round-3 kernel (profiles/r04_tier_worst_nodes.txt).  These planes come first; the centre / quarter / bbox planes follow while the
# oracle's time budget lasts.  (knot 512^3: the 256^3 planes doubled; SprayBottle 1024^3: its worst nodes sit on the centre plane, a second plane 32 below it.)
STEP1_WORST_PLANES = {("rocker.obj", 5.0): [272, 224, 304], ("chair.obj", 5.0): [240, 256], ("knot.obj", 4.0): [120, 128, 104], ("knot.obj", 5.0): [240, 256, 208],
                      ("bunny_small.obj", 4.0): [136, 56],
                      # (round 6, on the final kernel -- profiles/r06_tier_robustness_big.txt: k = 512 8.1e-9, 480 7.9e-9, then 1023 5.1e-9 -- the bbox plane, whose corners lie beyond
                      # lambda r = 350 where the reference's own normalisation is noise, DESIGN.md section 2a -- and 448 5.0e-9: the third oracle plane)
                      ("SprayBottle.pc", 6.0): [512, 480, 448]}


_ORACLE_PLANES = {}   # (file, n, plane) -> the oracle's Y on that plane: the fp32 case of a file reuses what its fp64 case computed (SprayBottle.pc at 1024^3: 110 s of host time)


def _oracle_planes(oracle_c, pre, ks, key=None):
    n = pre["n"]
    import os
    out = {}
    oracle_c.shmo_set_threads(os.cpu_count() or 8)    # (the session fixture keeps the oracle at 8 threads for the tiny cases)
    for k in ks:
        if key is not None and (key, n, k) in _ORACLE_PLANES:
            out[k] = _ORACLE_PLANES[(key, n, k)]
            continue
        Yp = np.zeros(3 * n * n)
        oracle_c.shmo_conv_normalize_planes(n, c_(pre["bbox_min"]), pre["cell"], pre["S"], c_(pre["pos"]).reshape(-1), c_(pre["wnormal"]).reshape(-1), pre["lam"], k, k + 1, Yp)
        out[k] = Yp.reshape(-1, 3)
        if key is not None:
            _ORACLE_PLANES[(key, n, k)] = out[k]
    oracle_c.shmo_set_threads(min(8, os.cpu_count() or 1))
    return out


@pytest.mark.parametrize("fname,hCoef,precision", STEP1_FULL)
def test_step1_full_size_against_c_oracle(shm, oracle_c, fname, hCoef, precision):
    import os
    import time
    import psutil
    pre = _preprocess(fname, hCoef)
    n, S = pre["n"], pre["S"]
    cores = os.cpu_count() or 1
    # one plane of the serial loops costs n^2 S pair evaluations at ~170 ns per pair and hardware thread (measured on the 256-thread host of the GPU boxes: exp, sqrt
    # and a division per pair): keep the oracle's share of the test under two minutes on this box's cores
    per_plane_s = n * n * S * 170e-9 / max(1, cores)
    if per_plane_s > 120.0:
        pytest.skip("one oracle plane would take %.0f s on %d cores" % (per_plane_s, cores))
    if n >= 1024 and (os.environ.get("SHM_SKIP_1024") or psutil.virtual_memory().available < 24 * 2 ** 30):
        pytest.skip("1024^3 skipped (SHM_SKIP_1024 / host memory)")
    want = []
    for k in STEP1_WORST_PLANES.get((fname, hCoef), []) + [n // 2, n // 4, 0, n - 1]:   # medial-axis planes, then centre, quarter, the planes through the bbox corners
        if k not in want:
            want.append(k)
    ks = want[:max(1, min(len(want), int(120.0 / max(per_plane_s, 1e-3))))]   # (120 s: three planes of SprayBottle.pc at 1024^3 on the 256-thread host)
    t0 = time.time()
    ref = _oracle_planes(oracle_c, pre, ks, key=fname)
    t_or = time.time() - t0
    s = shm.GridSolver(precision=precision)
    s.set_problem(pre["pos"], pre["wnormal"], pre["area"], pre["lam"], n, pre["bbox_min"], pre["cell"])
    worst = {}
    for arith in (("auto", "exact_f64") if precision == 64 else ("auto",)):
        s.run_conv(step1=arith)
        e = 0.0
        for k in ks:
            Y = np.stack([s.get_field_planes(f, k, k + 1) for f in (0, 1, 2)], axis=1)
            ok = np.isfinite(ref[k]).all(axis=1)
            assert ok.any()
            assert np.isfinite(Y[ok]).all()
            if precision == 64:
                assert (np.isfinite(Y).all(axis=1) == ok).all()      # same 0/0 nodes as the reference's arithmetic (beyond lambda r ~ 745)
            e = max(e, float(np.abs(Y[ok] - ref[k][ok]).max()))
        worst[arith] = e
    s.close()
    print("\nStep 1 %s n=%d S=%d fp%d: planes %s, oracle %.1f s on %d cores; max|Y_gpu - Y_oracle| %s" % (
        fname, n, S, precision, ks, t_or, cores, ", ".join("%s %.2e" % kv for kv in worst.items())))
    if precision == 64:
        assert worst["auto"] < Y_BUDGET, worst
        assert worst["exact_f64"] < 1e-10, worst
    else:
        assert worst["auto"] < Y_BUDGET_F32, worst


def test_far_rule_is_decided_by_the_sample_per_problem(shm, monkeypatch):
    """The tiered fp64 Step 1 classifies with the differential far rule where a sample of its own blocks says it pays (round 5; Solver::far_rule_plan, conv_tiered_kernel): on the
    bunny at 256^3 the default run sends clearly more pairs to the packed-fp32 tier than the box rule alone (SHM_TIER_FAR_RULE=0) and nearly as many as the rule forced on; on
    rocker at 256^3 the verdict is whatever its sample earns (between the two forced runs).  Either way Y keeps the budget against the all-fp64 kernel, and two runs agree
    bit for bit."""
    import os
    from conftest import ROOT
    from signed_heat_3d_amd.host_abi import HostSolver
    for path, expect_rule in (("bunny_small.obj", True), ("rocker.obj", None)):
        pre = HostSolver(os.path.join(ROOT, "data", path)).preprocess(hCoef=4.0)
        assert pre["n"] == 256
        share, Y = {}, {}
        for mode in ("default", "box", "rule", "default2", "exact"):
            monkeypatch.delenv("SHM_TIER_FAR_RULE", raising=False)
            monkeypatch.delenv("SHM_CONV_EXACT", raising=False)
            if mode == "box":
                monkeypatch.setenv("SHM_TIER_FAR_RULE", "0")
            if mode == "rule":
                monkeypatch.setenv("SHM_TIER_FAR_RULE", "1")
            if mode == "exact":
                monkeypatch.setenv("SHM_CONV_EXACT", "1")
            s = shm.GridSolver()
            s.set_problem(pre["pos"], pre["wnormal"], pre["area"], pre["lam"], pre["n"], pre["bbox_min"], pre["cell"])
            st = s.solve(scrub=True, allow_noconv=True)
            share[mode] = st.pairs_fp32 / max(1.0, st.pairs_fp32 + st.pairs_fp64)
            if mode in ("default", "default2", "exact"):
                s.run_conv()
                Y[mode] = np.stack([s.get_field_planes(f, 96, 160) for f in (0, 1, 2)], axis=1)
            s.close()
        print("\nfar rule %s 256^3: packed-fp32 share of the evaluated pairs: default %.3f, box rule %.3f, rule forced %.3f" % (path, share["default"], share["box"], share["rule"]))
        assert share["rule"] > share["box"] + (0.02 if expect_rule else 0.005)   # (rocker: the rule's far pairs that fail the a-posteriori test come back as fp64 pairs)
        assert share["box"] - 1e-3 <= share["default"] <= share["rule"] + 1e-3       # the sample's verdict lies between the two forced runs
        if expect_rule:
            assert share["default"] > share["box"] + 0.6 * (share["rule"] - share["box"])
        assert np.array_equal(Y["default"], Y["default2"], equal_nan=True)
        ok = np.isfinite(Y["exact"]).all(axis=1)
        assert np.abs(Y["default"][ok] - Y["exact"][ok]).max() < Y_BUDGET


def _adversarial_sources(kind, n, seed):
    """Seeded inputs built to break the precision tiers of Step 1 (round 6) -- not meshes anyone would draw, but valid inputs of the C ABI:
      sheets:  two parallel sheets of sources `sep` cells apart with opposite normals -- between and around them the two sheets' terms cancel in X, and
               Y = X / |X| amplifies every error of the packed-fp32 tier and of the drop rule by (sum of |terms|) / |X|;
      cloud:   a small dense point cloud in a large grid, lambda r up to ~1e3 at the far corners -- exponents in the hundreds, where a packed-fp32 term's relative
               error (~1e-7 per unit of exponent) is at its largest and most sources are dropped;
      areas:   a closed surface whose source weights span 1e4 : 1 -- the weight ratios enter every threshold (far, drop, exponent range of the fp32 tier).
    Returns the arguments of set_problem."""
    rng = np.random.default_rng(seed)
    cell = 2.0 / (n - 1)
    bbox_min = np.array([-1.0, -1.0, -1.0])
    if kind.startswith("sheets"):
        sep = float(kind.split(":")[1])             # separation in cells
        hs = 2.0 * cell                            # source spacing: two cells
        g = np.arange(-0.5, 0.5, hs) + 0.37 * cell
        X, Yg = np.meshgrid(g, g, indexing="ij")
        jit = lambda: (rng.random(X.shape) - 0.5) * 0.3 * hs   # noqa: E731
        z0 = 0.123 * cell
        # (gently undulating sheets: with flat ones every normal is +-z, X is parallel to z everywhere and Y = (0, 0, +-1) whatever the arithmetic)
        hgt = lambda x, y: 0.03 * np.sin(5.0 * x) * np.cos(4.0 * y)   # noqa: E731
        def sheet(dz, sign):
            x, y = X + jit(), Yg + jit()
            nv = np.stack([-0.15 * np.cos(5.0 * x) * np.cos(4.0 * y), 0.12 * np.sin(5.0 * x) * np.sin(4.0 * y), np.ones(x.shape)], -1)
            nv /= np.linalg.norm(nv, axis=-1, keepdims=True)
            return np.stack([x, y, hgt(x, y) + z0 + dz], -1).reshape(-1, 3), (sign * nv).reshape(-1, 3)
        lo, nlo = sheet(0.0, -1.0)
        hi, nhi = sheet(sep * cell, 1.0)
        pos = np.vstack([lo, hi])
        nrm = np.vstack([nlo, nhi])
        area = np.full(len(pos), hs * hs) * (0.8 + 0.4 * rng.random(len(pos)))
        lam = 1.0 / hs
    elif kind == "cloud":
        S = 4000
        v = rng.normal(size=(S, 3))
        v /= np.linalg.norm(v, axis=1, keepdims=True)
        R = 0.12
        pos = R * v * (1.0 + 0.02 * rng.normal(size=(S, 1))) + np.array([0.05, -0.03, 0.02])
        nrm = v
        area = np.full(S, 4 * np.pi * R * R / S)
        lam = 1.0e3 / (np.sqrt(3.0) * 1.0)          # lambda r ~ 1e3 at the corners of the grid
        lam = min(lam, 4.0 / cell)                 # (keep lambda * cell <= 4: beyond it the tiered kernel hands over to the all-fp64 one by design)
    elif kind == "areas":
        S = 6000
        v = rng.normal(size=(S, 3))
        v /= np.linalg.norm(v, axis=1, keepdims=True)
        R = 0.45
        pos = R * v * np.array([1.0, 0.7, 0.5])
        nrm = v / np.array([1.0, 0.7, 0.5])
        nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
        base = 4 * np.pi * R * R * 0.7 / S
        area = base * 10.0 ** (-4.0 * rng.random(S))     # 1e4 : 1
        lam = 1.0 / np.sqrt(base)
    elif kind == "shell":
        # a closed, finely sampled surface seen from INSIDE (round 6, late): the interior's nodes are tens of kernel widths from the nearest source, thousands of sources
        # lie within e^-8 ... e^-30 of it, and their terms cancel (outward normals: X would vanish for a uniform kernel) -- the place where the accumulation of the
        # packed-fp32 sums in fp32 showed in SprayBottle.pc at 1024^3 (8.1e-9 of the 1e-8 budget until the sums were flushed into fp64 every 256 sources)
        S = 40000
        v = rng.normal(size=(S, 3))
        v /= np.linalg.norm(v, axis=1, keepdims=True)
        ax3 = np.array([0.62, 0.5, 0.41])
        pos = v * ax3 * (1.0 + 0.01 * rng.normal(size=(S, 1))) + np.array([0.03, -0.02, 0.04])
        nrm = v / ax3
        nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
        area = np.full(S, 4 * np.pi * 0.5 * 0.5 / S) * (0.7 + 0.6 * rng.random(S))
        lam = 0.46 / cell
    else:
        raise ValueError(kind)
    return dict(pos=pos, wnormal=nrm * area[:, None], area=area, lam=float(lam), n=n, bbox_min=bbox_min, cell=cell)


@pytest.mark.parametrize("kind,n", [("sheets:0.5", 128), ("sheets:1", 128), ("sheets:2", 256), ("sheets:4", 128), ("cloud", 256), ("cloud", 128), ("areas", 128), ("areas", 256), ("shell", 256)])
def test_tier_budget_on_adversarial_inputs(shm, kind, n):
    """The tiered Step 1 against the all-fp64 arithmetic (shm_opts.step1_arith = EXACT_F64) on seeded inputs built to break the tiers: cancelling sheets, exponents in the
    hundreds, weights over four decades.  The budget on Y holds where the all-fp64 field is finite, the non-finite sets agree, two default runs agree bit for bit; the margin
    is printed (profiles/r06_gpu_tests.txt)."""
    worst = 0.0
    for seed in (1, 2):
        d = _adversarial_sources(kind, n, seed)
        s = make_solver(shm, d)
        s.run_conv()
        Yt = np.stack([s.get_field(k) for k in (0, 1, 2)], axis=1)
        s.run_conv()
        Yt2 = np.stack([s.get_field(k) for k in (0, 1, 2)], axis=1)
        assert np.array_equal(Yt, Yt2, equal_nan=True)
        s.run_conv(step1="exact_f64")
        Ye = np.stack([s.get_field(k) for k in (0, 1, 2)], axis=1)
        s.close()
        ok = np.isfinite(Ye).all(axis=1)
        assert ok.mean() > 0.2, ok.mean()
        assert (np.isfinite(Yt).all(axis=1) == ok).all()
        # Where |X|^2 is a SUBNORMAL double the reference's own X /= X.norm() (signed_heat_grid_solver.cpp:61) has lost most of its bits -- |X| < 1.5e-154, i.e. lambda r beyond
        # ~350: 2.5e-8 off the exact direction at lambda r = 356 on this very input, numpy's long double against the C oracle and against either kernel, which both follow the
        # reference there, each with the rounding of its own summation order; at lambda r ~ 372 it turns into 0/0.  The budget is a statement about the zone where the
        # reference's arithmetic itself is accurate: nodes nearer than lambda r = 335 to the nearest source (profiles/NOTES_r06.md).
        ijk = np.stack(np.meshgrid(np.arange(n), np.arange(n), np.arange(n), indexing="ij"), -1).reshape(-1, 3)[:, ::-1]   # (x fastest)
        xyz = d["bbox_min"] + ijk * d["cell"]
        cen = d["pos"].mean(axis=0)
        r_lb = np.maximum(0.0, np.linalg.norm(xyz - cen, axis=1) - np.linalg.norm(d["pos"] - cen, axis=1).max())   # lower bound of the distance to the nearest source
        ok &= d["lam"] * r_lb < 335.0
        assert ok.mean() > 0.05, ok.mean()
        err = float(np.abs(Yt[ok] - Ye[ok]).max())
        worst = max(worst, err)
    print("\nadversarial %-10s n=%3d: max|dY| tiered vs all-fp64 = %.2e (budget %.0e, margin %.1fx)" % (kind, n, worst, Y_BUDGET, Y_BUDGET / max(worst, 1e-300)))
    assert worst < Y_BUDGET, worst


@pytest.mark.parametrize("precision", [64, 32])
def test_step1_block_order_does_not_change_Y(shm, precision, monkeypatch):
    """The tiered kernel's work queues hand the z-layers of blocks out from the grid's centre to its faces (round 5; SHM_TIER_LAYER_ORDER=0: bottom to top).  A block's
    result does not depend on when it is computed: Y is bit-identical either way, on one slab and on three (whose layers are ordered by their GLOBAL planes)."""
    import os
    from conftest import ROOT
    from signed_heat_3d_amd.host_abi import HostSolver
    pre = HostSolver(os.path.join(ROOT, "data", "bunny_small.obj")).preprocess(hCoef=3.0)     # 128^3: 32 layers of four planes
    for slabs in (1, 3):
        Y = {}
        for order in ("1", "0"):
            monkeypatch.setenv("SHM_TIER_LAYER_ORDER", order)
            s = shm.GridSolver(precision=precision, local_slabs=slabs)
            s.set_problem(pre["pos"], pre["wnormal"], pre["area"], pre["lam"], pre["n"], pre["bbox_min"], pre["cell"])
            s.run_conv()
            Y[order] = np.stack([s.get_field(k) for k in (0, 1, 2)], axis=1)
            s.close()
        assert np.array_equal(Y["1"], Y["0"], equal_nan=True)


@pytest.mark.parametrize("path,hcoef", [("bunny_small.obj", 2.0), ("bunny_small.obj", 3.0), ("rocker.obj", 2.0), ("bunny.pc", 2.0)])
def test_exact_f64_step1_two_kernels_agree(shm, path, hcoef, monkeypatch):
    """shm_opts.step1_arith = SHM_STEP1_EXACT_F64 runs the tiered kernel with nothing far and nothing dropped (round 5) where every pair of the grid stays inside a
    block's exponent span, the all-fp64 kernel of rounds 1-4 otherwise and behind SHM_CONV_EXACT_CLASSIC=1: the two agree to rounding, every pair is counted as an
    fp64 pair, and a wide exponent span (lambda x 8) hands over to the classic kernel with the same answer."""
    import os
    from conftest import ROOT
    from signed_heat_3d_amd.host_abi import HostSolver
    pre = HostSolver(os.path.join(ROOT, "data", path)).preprocess(hCoef=hcoef)
    n, S = pre["n"], len(pre["area"])
    out = {}
    for classic in (False, True):
        if classic:
            monkeypatch.setenv("SHM_CONV_EXACT_CLASSIC", "1")
        else:
            monkeypatch.delenv("SHM_CONV_EXACT_CLASSIC", raising=False)
        s = shm.GridSolver()
        s.set_problem(pre["pos"], pre["wnormal"], pre["area"], pre["lam"], n, pre["bbox_min"], pre["cell"])
        st = s.solve(tol=1e-10, scrub=not path.endswith(".pc"), step1="exact_f64")
        out[classic] = (np.stack([s.get_field(k) for k in (0, 1, 2)], axis=1), s.get_phi()[0], st)
        s.close()
    (Yn, pn, stn), (Yc, pc, stc) = out[False], out[True]
    ok = np.isfinite(Yc).all(axis=1)
    assert (np.isfinite(Yn).all(axis=1) == ok).all()
    dY, dphi = np.abs(Yn[ok] - Yc[ok]).max(), np.abs(pn - pc).max()
    print("\nexact f64, tiered body vs all-fp64 kernel: %s n=%d max|dY| %.2e max|dphi| %.2e" % (path, n, dY, dphi))
    assert dY < 1e-11 and dphi < 1e-10        # (rounding: where |X| is small against its terms -- the medial axis -- the direction amplifies the last bits of the sums)
    assert stn.pairs_fp32 == 0 and stn.pairs_redone == 0
    nz = int((np.abs(np.asarray(pre["wnormal"]).reshape(-1, 3)).sum(axis=1) > 0).sum())
    assert stn.pairs_fp64 >= n ** 3 * nz          # every node against every source of non-zero weight (blocks are padded to 8 x 8 x NPT nodes)
    # beyond the exponent span the classic kernel takes over (same entry point, same answer as asking for it)
    monkeypatch.delenv("SHM_CONV_EXACT_CLASSIC", raising=False)
    s = shm.GridSolver()
    s.set_problem(pre["pos"], pre["wnormal"], pre["area"], pre["lam"] * 8.0, n, pre["bbox_min"], pre["cell"])
    s.run_conv(step1="exact_f64")
    Yw = np.stack([s.get_field(k) for k in (0, 1, 2)], axis=1)
    s.close()
    monkeypatch.setenv("SHM_CONV_EXACT_CLASSIC", "1")
    s = shm.GridSolver()
    s.set_problem(pre["pos"], pre["wnormal"], pre["area"], pre["lam"] * 8.0, n, pre["bbox_min"], pre["cell"])
    s.run_conv(step1="exact_f64")
    Yw2 = np.stack([s.get_field(k) for k in (0, 1, 2)], axis=1)
    s.close()
    okw = np.isfinite(Yw2).all(axis=1)
    assert (np.isfinite(Yw).all(axis=1) == okw).all() and np.array_equal(Yw[okw], Yw2[okw])   # (the same kernel ran)


@pytest.mark.parametrize("case,scrub", [("bunny_small_n16", True), ("bunny_small_n32", True), ("bunny_pc_n32", False)])
def test_divergence_matches_golden(shm, case, scrub):
    d = load_golden(case)
    s = make_solver(shm, d)
    s.run_conv()
    s.run_divergence(scrub)
    b = s.get_field(s.FIELD_DIV)
    scale = np.abs(d["b"]).max()
    assert np.abs(b - d["b"]).max() < 1e-10 * scale


@pytest.mark.parametrize("n", [16, 24, 33, 64])
def test_laplacian_matches_oracle(shm, oracle_c, n):
    d = load_golden("bunny_small_n16")
    s = shm.GridSolver()
    cell = 0.37
    s.set_problem(d["pos"], d["wnormal"], d["area"], float(d["lam"]), n, d["bbox_min"], float(d["cell"]) * 15 / (n - 1))
    rng = np.random.default_rng(n)
    u = rng.standard_normal(n ** 3)
    got = s.apply_laplacian(u)
    ref = np.empty_like(u)
    oracle_c.shmo_laplacian_apply(n, float(d["cell"]) * 15 / (n - 1), u, ref)
    assert np.abs(got - ref).max() < 1e-10 * np.abs(ref).max()


@pytest.mark.parametrize("case", ["bunny_small_n16", "bunny_small_n32", "bunny_pc_n32", "polygon_bear_n16"])
def test_constraint_rows_bit_exact(shm, case):
    d = load_golden(case)
    s = make_solver(shm, d)
    nodes, coeffs = s.get_constraints()
    assert nodes.shape[0] == int(d["m"])
    assert np.array_equal(nodes, d["c_nodes"])          # index work: bit exact
    assert np.array_equal(coeffs, d["c_coeffs"])        # same expression order -> bit exact


@pytest.mark.parametrize("case", ["bunny_small_n16", "bunny_small_n32"])
def test_projector(shm, case):
    """P = I - A^T (A A^T)^-1 A: idempotent, annihilates range(A^T), A P v = 0."""
    d = load_golden(case)
    s = make_solver(shm, d)
    n = int(d["n"])
    rng = np.random.default_rng(1)
    v = rng.standard_normal(n ** 3)
    Pv = s.apply_projector(v)
    nodes, coeffs = d["c_nodes"], d["c_coeffs"]
    APv = (coeffs * Pv[nodes]).sum(axis=1)
    assert np.abs(APv).max() < 1e-11
    PPv = s.apply_projector(Pv)
    assert np.abs(PPv - Pv).max() < 1e-11
    # range(A^T) is annihilated
    w = rng.standard_normal(nodes.shape[0])
    Atw = np.zeros(n ** 3)
    np.add.at(Atw, nodes.ravel(), (coeffs * w[:, None]).ravel())
    assert np.abs(s.apply_projector(Atw)).max() < 1e-10 * np.abs(Atw).max()


@pytest.mark.parametrize("case", ["bunny_small_n32", "bunny_small_n64"])
def test_one_launch_gauss_jordan_gives_the_bits_of_the_three_launch_chain(shm, case, monkeypatch):
    """gj_step_kernel (round 4: one launch per pivot block -- the previous block's update beside this block's pivot inversion and panels, the cross of tiles
    updated on the fly, the pivot tile inverted redundantly by every panel workgroup) claims the arithmetic, operand order and results of the three-launch chain
    it replaces (gj_pivot / gj_panels / gj_update kernels, SHM_GJ_CLASSIC=1; its pivot kernel also takes one elimination step per barrier where gj_step_kernel takes two): the projector P = I - A^T (A A^T)^-1 A must come out BIT-identical either way
    (m = 497 / 1129: 8 / 18 pivot blocks)."""
    d = load_golden(case)
    n = int(d["n"])
    v = np.random.default_rng(5).standard_normal(n ** 3)
    out = []
    for classic in (False, True):
        if classic:
            monkeypatch.setenv("SHM_GJ_CLASSIC", "1")
        else:
            monkeypatch.delenv("SHM_GJ_CLASSIC", raising=False)
        s = make_solver(shm, d)
        out.append(s.apply_projector(v))
        s.close()
    assert np.array_equal(out[0], out[1])
    if "c_nodes" in d.files:
        nodes, coeffs = d["c_nodes"], d["c_coeffs"]
        assert np.abs((coeffs * out[0][nodes]).sum(axis=1)).max() < 1e-11


@pytest.mark.parametrize("n", [16, 32, 64, 128, 256, 512, 22, 45, 90, 181, 362])
def test_preconditioner_is_the_dct_pseudo_inverse(shm, n):
    """M^-1 = C^T D C must equal the pseudo-inverse of K = -L (the 3-D DCT-II diagonalises the Neumann Laplacian):
    compared with scipy's orthonormal DCT on the host, and checked through K M^-1 v = v - mean(v)."""
    from scipy.fft import dctn, idctn
    d = load_golden("bunny_small_n16")
    h = float(d["cell"]) * 15 / (n - 1)
    s = shm.GridSolver()
    s.set_problem(d["pos"], d["wnormal"], d["area"], float(d["lam"]), n, d["bbox_min"], h)
    rng = np.random.default_rng(n)
    v = rng.standard_normal(n ** 3)
    got = s.apply_preconditioner(v)
    lam1 = (2 - 2 * np.cos(np.pi * np.arange(n) / n)) / h ** 2
    LAM = lam1[:, None, None] + lam1[None, :, None] + lam1[None, None, :]
    inv = np.where(LAM > 0, 1 / np.where(LAM > 0, LAM, 1), 0.0)
    ref = idctn(dctn(v.reshape(n, n, n), type=2, norm="ortho") * inv, type=2, norm="ortho").reshape(-1)
    assert np.abs(got - ref).max() < 1e-11 * np.abs(ref).max()
    Lg = s.apply_laplacian(got)      # L M^-1 v = -(v - mean v)
    assert np.abs(-Lg - (v - v.mean())).max() < 1e-9 * np.abs(v).max()


MODES = {"primal-plain": dict(solver="primal", precond="none"), "primal-dct": dict(solver="primal", precond="dct"),
         "dual": dict(solver="dual")}


@pytest.mark.parametrize("mode", sorted(MODES))
@pytest.mark.parametrize("case,scrub", [("bunny_small_n16", True), ("bunny_small_n24", True), ("bunny_small_n32", True),
                                        ("polygon_bear_n16", True), ("bunny_pc_n16", False), ("bunny_pc_n32", False)])
def test_phi_matches_lu_golden(shm, case, scrub, mode):
    """All three solvers of the KKT system (plain projected CG, DCT-preconditioned projected CG, dual Schur-complement CG)
    against the reference-equivalent sparse-LU solution."""
    d = load_golden(case)
    kw = MODES[mode]
    s = make_solver(shm, d)                        # (n = 24 is not a power of two: its fast Poisson solve runs as dense DCT products, shm_dct_gemm.hip.h)
    st = s.solve(tol=1e-10, scrub=scrub, **kw)
    assert st.solver == (2 if mode == "dual" else 1)
    assert st.preconditioner == (1 if mode == "primal-plain" else 2)
    phi, (k0, k1) = s.get_phi()
    assert (k0, k1) == (0, int(d["n"]))
    err = np.abs(phi - d["phi"]).max()
    assert st.m == int(d["m"])
    assert abs(st.shift - float(d["shift"])) < 1e-8
    assert err < 1e-7, (err, st.iters, st.rel_residual)   # far inside the 1e-5 gate


def test_phi_default_tolerance_inside_gate(shm):
    d = load_golden("bunny_small_n32")
    s = make_solver(shm, d)
    st = s.solve()
    phi, _ = s.get_phi()
    assert np.abs(phi - d["phi"]).max() < PHI_GATE


def test_preconditioner_cuts_iterations(shm):
    d = load_golden("bunny_small_n32")
    s = make_solver(shm, d)
    plain = s.solve(tol=1e-8, solver="primal", precond="none")
    phi0, _ = s.get_phi()
    pre = s.solve(tol=1e-8, solver="primal", precond="dct")
    phi1, _ = s.get_phi()
    dual = s.solve(tol=1e-8, solver="dual")
    phi2, _ = s.get_phi()
    assert pre.iters * 4 < plain.iters, (pre.iters, plain.iters)
    assert dual.iters <= pre.iters, (dual.iters, pre.iters)
    assert np.abs(phi0 - phi1).max() < 1e-6 and np.abs(phi0 - phi2).max() < 1e-6
    assert np.abs(phi1 - d["phi"]).max() < PHI_GATE and np.abs(phi2 - d["phi"]).max() < PHI_GATE


@pytest.mark.parametrize("mode", sorted(MODES))
def test_phi_64_config_c1(shm, mode):
    """BASELINE.json configs[0]: bunny_small.obj at 64^3 against the committed LU solution."""
    import os
    from conftest import GOLDEN
    if not os.path.exists(os.path.join(GOLDEN, "bunny_small_n64.npz")):
        pytest.skip("64^3 LU fixture not generated")
    d = load_golden("bunny_small_n64")
    s = make_solver(shm, d)
    st = s.solve(tol=1e-9, **MODES[mode])
    phi, _ = s.get_phi()
    assert np.abs(phi - d["phi"]).max() < 1e-6, st.iters


@pytest.mark.parametrize("slabs,n", [(2, 32), (4, 32), (8, 64), (2, 16)])
def test_distributed_preconditioner_matches_single_slab(shm, slabs, n):
    """Multi-slab DCT (packed y sweeps + two all-to-all transposes, loop-back transport) == single-slab DCT."""
    d = load_golden("bunny_small_n16")
    h = float(d["cell"]) * 15 / (n - 1)
    rng = np.random.default_rng(7)
    v = rng.standard_normal(n ** 3)
    s1 = shm.GridSolver()
    s1.set_problem(d["pos"], d["wnormal"], d["area"], float(d["lam"]), n, d["bbox_min"], h)
    ref = s1.apply_preconditioner(v)
    s = shm.GridSolver(local_slabs=slabs)
    s.set_problem(d["pos"], d["wnormal"], d["area"], float(d["lam"]), n, d["bbox_min"], h)
    got = s.apply_preconditioner(v)
    assert np.abs(got - ref).max() < 1e-12 * np.abs(ref).max()


@pytest.mark.parametrize("mode", ["primal-dct", "dual"])
@pytest.mark.parametrize("slabs", [2, 4])
def test_local_slabs_with_preconditioner(shm, slabs, mode):
    d = load_golden("bunny_small_n32")
    s = make_solver(shm, d, local_slabs=slabs)
    st = s.solve(tol=1e-10, **MODES[mode])
    assert st.preconditioner == 2
    phi, _ = s.get_phi()
    assert np.abs(phi - d["phi"]).max() < 1e-7
    s1 = make_solver(shm, d)
    st1 = s1.solve(tol=1e-10, **MODES[mode])
    if st1.cg_form == 2:   # one slab, dual: the direct solve (explicit S^-1) -- one or two passes instead of a CG
        # round 6: several slabs take the same direct solve (S and S^-1 shared, K^+ on the slabs); with S applied through the grid (dual_form) they iterate
        assert st1.iters <= 2 and st.cg_form == 2 and st.iters <= 2, (st.iters, st1.iters, st.cg_form)
        phi1, _ = s1.get_phi()
        assert np.abs(phi1 - phi).max() < 1e-8
        st_it = s.solve(tol=1e-10, dual_form="through_grid", **MODES[mode])
        phi_it, _ = s.get_phi()
        assert st_it.cg_form == 0 and st_it.iters > 4 and np.abs(phi_it - d["phi"]).max() < 1e-7, (st_it.cg_form, st_it.iters)
        st_cg = s.solve(tol=1e-10, dual_form="explicit_s_cg", **MODES[mode])
        phi_cg, _ = s.get_phi()
        assert st_cg.cg_form == 3 and abs(st_cg.iters - st_it.iters) <= 4 and np.abs(phi_cg - d["phi"]).max() < 1e-7, (st_cg.cg_form, st_cg.iters)
    else:
        assert abs(st.iters - st1.iters) <= 4


@pytest.mark.parametrize("slabs", [2, 3, 5])
def test_local_slabs_match_single_slab(shm, slabs):
    """The z-slab code path (ghost planes, halo copies, per-slab partial sums, straddling constraint rows)
    on one GPU with the loop-back transport must reproduce the single-slab result."""
    d = load_golden("bunny_small_n32")
    s1 = make_solver(shm, d)
    s1.solve(tol=1e-10)
    ref, _ = s1.get_phi()
    s = make_solver(shm, d, local_slabs=slabs)
    s.solve(tol=1e-10, precond="none")
    phi, (k0, k1) = s.get_phi()
    assert (k0, k1) == (0, 32)
    assert np.abs(phi - ref).max() < 1e-9
    assert np.abs(phi - d["phi"]).max() < 1e-7


@pytest.mark.parametrize("fast", [False, True])
def test_weighted_slab_plan_matches_single_slab(shm, fast):
    """Unequal z-slabs (shm_config.slab_plan = SHM_SLAB_PLAN_STEP1: planes weighted by the Step-1 work of the tiered fp64 kernel, cut at multiples of 4 planes)
    through the slab code path on one GPU: plain stencil CG / the slab-chained fast integration must reproduce the single-slab result, and the
    slab-distributed transforms, which need equal slabs, must be refused rather than run on the wrong layout."""
    pre = _preprocess("SprayBottle.pc", 2.0)          # 64^3; most sources are dropped per block and the kept share varies along z
    n = pre["n"]
    w = shm.step1_plane_weights(pre["pos"], pre["wnormal"], pre["lam"], n, pre["bbox_min"], pre["cell"], 64)
    plan = [shm.plan_slab_weighted(n, 3, r, w, 4) for r in range(3)]
    assert plan != [shm.plan_slab(n, 3, r) for r in range(3)], plan   # (equal planes: 22 + 21 + 21; weighted: multiples of 4)
    out = {}
    for slabs in (1, 3):
        s = shm.GridSolver(local_slabs=slabs, slab_plan=1)
        s.set_problem(pre["pos"], pre["wnormal"], pre["area"], pre["lam"], n, pre["bbox_min"], pre["cell"])
        st = s.solve(tol=1e-10, fast=fast, scrub=False, precond="none", solver="primal")
        phi, (k0, k1) = s.get_phi()
        assert (k0, k1) == (0, n)
        out[slabs] = phi
        if slabs == 3 and not fast:
            assert st.solver == 1 and st.preconditioner == 1
            with pytest.raises(shm.ShmError):                      # the slab-distributed transforms need equal slabs
                s.solve(tol=1e-10, scrub=False, solver="dual_slabs")
            with pytest.raises(shm.ShmError):
                s.solve(tol=1e-10, scrub=False, solver="primal", precond="dct")
        s.close()
    assert np.isfinite(out[1]).all()
    assert np.abs(out[3] - out[1]).max() < 1e-8 * max(1.0, np.abs(out[1]).max())


@pytest.mark.parametrize("n,slabs", [(48, 1), (33, 1), (33, 3), (70, 1), (11, 1)])
def test_matches_c_oracle_odd_sizes(shm, oracle_c, n, slabs):
    """Same inputs through the HIP path at sizes that are not powers of two (odd sizes without vector loads, several slabs) and the C oracle (serial
    reference loops + projected CG) -- sizes with no LU fixture.  One slab: the default is the dual solver with the fast Poisson solve as dense DCT
    products (shm_dct_gemm.hip.h); several slabs: plain projected stencil CG.  The plain CG is also run on the single slab (a second algorithm)."""
    d = load_golden("bunny_small_n16")
    cell = float(d["cell"]) * 15 / (n - 1)
    s = shm.GridSolver(local_slabs=slabs)
    s.set_problem(d["pos"], d["wnormal"], d["area"], float(d["lam"]), n, d["bbox_min"], cell)
    st = s.solve(tol=1e-10)
    assert (st.solver, st.preconditioner) == ((2, 2) if slabs == 1 else (1, 1))
    phi, _ = s.get_phi()
    if slabs == 1:
        st1 = s.solve(tol=1e-10, solver="primal", precond="none")
        assert st1.solver == 1 and st1.preconditioner == 1
        assert np.abs(s.get_phi()[0] - phi).max() < 1e-7
    ref = np.zeros(n ** 3)
    st = np.zeros(5)
    rc = oracle_c.shmo_compute_distance(n, c_(d["bbox_min"]), cell, len(d["area"]), c_(d["pos"]).reshape(-1), c_(d["wnormal"]).reshape(-1),
                                        c_(d["area"]), float(d["lam"]), 1, 0, 1e-12, 100000, ref, st)
    assert rc == 0
    assert np.abs(phi - ref).max() < 1e-7


@pytest.mark.parametrize("hCoef", [3.0, 4.0])
def test_phi_of_the_default_solve_against_c_oracle_at_full_size(shm, oracle_c, hCoef):
    """BASELINE.json's gate -- L_inf of phi against the CPU reference path on the same inputs -- at 128^3 always and at 256^3 (configs[1]) where the host has the
    threads for it: the C oracle (the reference's serial loops + projected CG converged to 1e-10, all host threads; signed_heat_grid_solver.cpp:46-111) against the
    library's DEFAULT solve (tiered Step 1, direct dual solve), through the C++ host mirror's pre-processing of data/bunny_small.obj.  The north star asks for
    1e-5; the library is held to 1e-7 like every other phi test."""
    import os
    import time
    pre = _preprocess("bunny_small.obj", hCoef)
    n, S = pre["n"], pre["S"]
    cores = os.cpu_count() or 1
    if n > 128 and cores < 64:
        pytest.skip("256^3 oracle solve needs a many-core host (%d threads here)" % cores)
    s = shm.GridSolver()
    s.set_problem(pre["pos"], pre["wnormal"], pre["area"], pre["lam"], n, pre["bbox_min"], pre["cell"])
    st = s.solve(scrub=True)
    phi, _ = s.get_phi()
    ste = s.solve(scrub=True, step1="exact_f64")        # the reference's arithmetic in every pair of Step 1 (round 5: through the tiered kernel's fp64 body)
    phi_exact, _ = s.get_phi()
    # round 6: the curve the budget knob spans (shm_opts.step1_budget; reported, not a new default): phi of the solve at each budget, held below against the same oracle
    curve = {}
    for budget in (1e-7, 1e-6):
        stb = s.solve(scrub=True, step1_budget=budget)
        curve[budget] = (s.get_phi()[0], stb)
    s.close()
    ref = np.zeros(n ** 3)
    sto = np.zeros(5)
    # (the oracle's CG is a chain of short memory-bound parallel loops: beyond ~64 threads the fork / join of each costs more than the loop -- on the 256-thread
    # host of the GPU boxes the 128^3 solve took 1037 s with every thread against seconds with 64)
    threads = min(cores, 64)
    oracle_c.shmo_set_threads(threads)
    t0 = time.time()
    rc = oracle_c.shmo_compute_distance(n, c_(pre["bbox_min"]), pre["cell"], S, c_(pre["pos"]).reshape(-1), c_(pre["wnormal"]).reshape(-1), c_(pre["area"]), pre["lam"],
                                        1, 0, 1e-10, 100000, ref, sto)
    t_or = time.time() - t0
    oracle_c.shmo_set_threads(min(8, cores))
    assert rc == 0
    err = float(np.abs(phi - ref).max())
    print("\nphi bunny_small.obj n=%d: default solve (solver %d, %d iterations, rel %.1e) against the C oracle (%d CG iterations, rel %.1e, %.1f s on %d threads): L_inf %.3e, phi in [%.4f, %.4f]"
          % (n, st.solver, st.iters, st.rel_residual, int(sto[1]), sto[2], t_or, threads, err, phi.min(), phi.max()))
    assert err < 1e-7, err
    err_exact = float(np.abs(phi_exact - ref).max())
    print("phi bunny_small.obj n=%d: step1_arith = EXACT_F64 (pairs fp64 %.3e, fp32 %.0f) against the C oracle: L_inf %.3e" % (n, ste.pairs_fp64, ste.pairs_fp32, err_exact))
    assert err_exact < 1e-7 and ste.pairs_fp32 == 0, err_exact
    for budget, (phib, stb) in sorted(curve.items()):
        errb = float(np.abs(phib - ref).max())
        print("phi bunny_small.obj n=%d: step1_budget %.0e (Step 1 %.2f ms against %.2f at 1e-8; packed-fp32 share %.3f): L_inf against the C oracle %.3e"
              % (n, budget, stb.ms_conv, st.ms_conv, stb.pairs_fp32 / max(1.0, stb.pairs_fp32 + stb.pairs_fp64), errb))
        assert errb < 1e-6, (budget, errb)     # (phi inherits a few per cent of the budget on Y; the north star's gate is 1e-5)


ALL_DATA = ["bunny_small.obj", "polygon-bear.obj", "rocker.obj", "chair.obj", "knot.obj", "bunny.pc", "rocker.pc", "chair.pc", "knot.pc", "SprayBottle.pc"]


@pytest.mark.parametrize("fname", ALL_DATA)
def test_every_data_file_matches_c_oracle_32(shm, oracle_c, fname):
    """Every input the reference ships (closed, open, non-manifold and polygonal meshes; point clouds), pre-processed by the C++
    host mirror, through the default HIP path at 32^3 against the C oracle on the same inputs (mesh overload: with the divYt scrub;
    point overload: without).  Where the reference's own arithmetic breaks down (|X|^2 underflows far from a finely sampled point
    cloud -> 0/0 -> NaN right-hand side, no scrub in the point overload) the oracle's phi is non-finite and the HIP path must report
    SHM_ERR_BREAKDOWN instead of returning numbers."""
    import os
    from conftest import ROOT
    from signed_heat_3d_amd.host_abi import HostSolver
    pre = HostSolver(os.path.join(ROOT, "data", fname)).preprocess(hCoef=1.0)
    n, S = pre["n"], pre["S"]
    assert n == 32
    scrub = not fname.endswith(".pc")
    ref = np.zeros(n ** 3)
    st = np.zeros(5)
    oracle_c.shmo_set_threads(min(32, os.cpu_count() or 1))
    rc = oracle_c.shmo_compute_distance(n, c_(pre["bbox_min"]), pre["cell"], S, c_(pre["pos"]).reshape(-1), c_(pre["wnormal"]).reshape(-1),
                                        c_(pre["area"]), pre["lam"], int(scrub), 0, 1e-12, 100000, ref, st)
    oracle_c.shmo_set_threads(min(8, os.cpu_count() or 1))
    s = shm.GridSolver()
    s.set_problem(pre["pos"], pre["wnormal"], pre["area"], pre["lam"], n, pre["bbox_min"], pre["cell"])
    if rc != 0 or not np.isfinite(ref).all():
        with pytest.raises(shm.ShmError) as e:
            s.solve(tol=1e-10, scrub=scrub)
        assert "BREAKDOWN" in str(e.value)
        return
    stg = s.solve(tol=1e-10, scrub=scrub)
    assert stg.solver == 2
    phi, _ = s.get_phi()
    assert np.isfinite(phi).all()
    assert np.abs(phi - ref).max() < 1e-7 * max(1.0, np.abs(ref).max())


@pytest.mark.parametrize("seed", list(range(int(__import__("os").environ.get("SHM_FUZZ_SEEDS", "12")))))
def test_fuzz_random_point_sets_match_c_oracle(shm, oracle_c, seed):
    """Seeded random inputs (a noisy, partly open sphere-like point set with random positive weights) on random grid sizes (powers of
    two and not), slab counts and solver choices, against the C oracle on the same inputs."""
    rng = np.random.default_rng(1000 + seed)
    n = int(rng.choice([16, 20, 24, 32, 33, 40, 64]))
    slabs = int(rng.choice([1, 1, 2, 3]))
    S = int(rng.integers(40, 400))
    d = rng.standard_normal((S, 3))
    d /= np.linalg.norm(d, axis=1)[:, None]
    keep = d[:, 2] > rng.uniform(-1.0, -0.2)                     # open cap
    d = d[keep]
    S = len(d)
    rad = 1.0 + 0.15 * rng.standard_normal(S)
    pos = d * rad[:, None] * np.array([1.0, rng.uniform(0.5, 1.0), rng.uniform(0.5, 1.0)])
    nrm = d + 0.2 * rng.standard_normal((S, 3))
    nrm /= np.linalg.norm(nrm, axis=1)[:, None]
    area = rng.uniform(0.2, 1.0, S) * (4 * np.pi / S)
    wn = nrm * area[:, None]
    c = pos.mean(0)
    r = np.linalg.norm(pos - c, axis=1).max()
    s_ = 2.0 * r
    bbox_min = c - s_
    cell = 2 * s_ / (n - 1)
    lam = 1.0 / rng.uniform(0.15, 0.5)
    mode = ["auto", "primal", "dual"][int(rng.integers(0, 3))] if (n & (n - 1)) == 0 and n % slabs == 0 and (slabs & (slabs - 1)) == 0 else "auto"
    g = shm.GridSolver(local_slabs=slabs)
    g.set_problem(pos, wn, area, lam, n, bbox_min, cell)
    st = g.solve(tol=1e-10, solver=mode)
    phi, _ = g.get_phi()
    ref = np.zeros(n ** 3)
    sto = np.zeros(5)
    rc = oracle_c.shmo_compute_distance(n, c_(bbox_min), cell, S, c_(pos).reshape(-1), c_(wn).reshape(-1), c_(area), lam, 1, 0, 1e-12, 200000, ref, sto)
    assert rc == 0
    assert np.isfinite(phi).all()
    assert np.abs(phi - ref).max() < 1e-7 * max(1.0, np.abs(ref).max()), (n, slabs, S, mode, st.iters)


def test_rebuild_false_keeps_the_previous_grid(shm):
    """signed_heat_grid_solver.cpp:8 / src/main.cpp:113,146-148: the mesh overload rebuilds the grid iff `options.rebuild` or nothing was
    built yet; a later call with rebuild=false and another hCoef must keep nx, the bounding box and the cell size of the first call
    (the time step, areas and constraint rows are recomputed every call).  The point overload rebuilds always (:119)."""
    import os
    from conftest import ROOT
    from signed_heat_3d_amd.host_abi import HostSolver
    h = HostSolver(os.path.join(ROOT, "data", "bunny_small.obj"), tol=1e-10)
    assert h.grid_info()["n"] == 0
    phi16, _ = h.compute_distance(hCoef=0.0, rebuild=False)            # nothing built yet: builds although rebuild=false
    g16 = h.grid_info()
    assert g16["n"] == 16 and phi16.size == 16 ** 3
    assert np.abs(phi16 - load_golden("bunny_small_n16")["phi"]).max() < 1e-7
    phi_keep, st = h.compute_distance(hCoef=1.0, rebuild=False)       # hCoef changed, rebuild=false: the 16^3 grid stays
    g = h.grid_info()
    assert g["n"] == 16 and st.n == 16 and phi_keep.size == 16 ** 3
    assert np.array_equal(g["bbox_min"], g16["bbox_min"]) and g["cell"] == g16["cell"]
    assert np.abs(phi_keep - phi16).max() < 1e-9                       # same grid, same sources -> same answer
    phi_scaled, _ = h.compute_distance(hCoef=1.0, scale=3.0, rebuild=False)   # `scale` is a grid parameter too (:15): ignored without rebuild
    assert np.abs(phi_scaled - phi16).max() < 1e-9
    phi_t, _ = h.compute_distance(hCoef=1.0, tCoef=2.0, rebuild=False)        # tCoef is NOT cached (:43): the answer changes, the grid does not
    assert h.grid_info()["n"] == 16 and np.abs(phi_t - phi16).max() > 1e-4
    phi32, st = h.compute_distance(hCoef=1.0, rebuild=True)            # rebuild=true: new grid
    assert h.grid_info()["n"] == 32 and st.n == 32
    assert np.abs(phi32 - load_golden("bunny_small_n32")["phi"]).max() < 1e-7
    # point overload: rebuilds on every call whatever the flag says
    d = load_golden("bunny_pc_n16")
    hp = HostSolver(os.path.join(ROOT, "data", "bunny.pc"), tol=1e-10)
    hp.set_point_areas(d["area"], float(d["h_in"]))
    p16, _ = hp.compute_distance(hCoef=0.0, rebuild=False)
    assert np.abs(p16 - d["phi"]).max() < 1e-7
    p32, st = hp.compute_distance(hCoef=1.0, rebuild=False)
    assert hp.grid_info()["n"] == 32 and st.n == 32 and p32.size == 32 ** 3


@pytest.mark.parametrize("mode", sorted(MODES))
@pytest.mark.parametrize("kind", ["single-source", "one-cell"])
def test_degenerate_constraint_sets_do_not_break_down(shm, oracle_c, kind, mode):
    """m == 1 (one source, or every source in the same grid cell): the projected residual of the multiplier system is exactly zero, so the
    CG scalars hit 0/0 unless guarded; the solve must return the oracle's phi, not SHM_ERR_BREAKDOWN."""
    rng = np.random.default_rng(5)
    n = 16
    S = 1 if kind == "single-source" else 6
    bbox_min = np.array([-1.0, -1.0, -1.0])
    cell = 2.0 / (n - 1)
    base = bbox_min + cell * (np.array([7, 8, 6]) + 0.1)
    pos = base + 0.8 * cell * rng.random((S, 3))                      # all inside cell (7, 8, 6)
    nrm = np.array([0.3, -0.5, 0.8]) + 0.1 * rng.standard_normal((S, 3))
    nrm /= np.linalg.norm(nrm, axis=1)[:, None]
    area = rng.uniform(0.5, 1.0, S)
    wn = nrm * area[:, None]
    lam = 3.0
    g = shm.GridSolver()
    g.set_problem(pos, wn, area, lam, n, bbox_min, cell)
    st = g.solve(tol=1e-10, **MODES[mode])
    assert st.m == 1
    phi, _ = g.get_phi()
    ref = np.zeros(n ** 3)
    sto = np.zeros(5)
    rc = oracle_c.shmo_compute_distance(n, c_(bbox_min), cell, S, c_(pos).reshape(-1), c_(wn).reshape(-1), c_(area), lam, 1, 0, 1e-13, 100000, ref, sto)
    assert rc == 0
    assert np.isfinite(phi).all()
    assert np.abs(phi - ref).max() < 1e-7, (kind, mode, st.iters)


@pytest.mark.parametrize("box", [4, 8])
@pytest.mark.parametrize("case,scrub", [("bunny_small_n32", True), ("bunny_pc_n32", False), ("bunny_small_n64", True)])
def test_two_level_inverse_of_AAT_matches_lu_golden(case, scrub, box, tmp_path):
    """The two-level (boxes + separator) inverse of A A^T that replaces the dense one for large constraint sets, forced onto the small
    fixtures (SHM_TL_MIN_M=32, boxes of 4^3 / 8^3 cells): it is exact, so the projector (primal solvers) and the dual preconditioner are the
    same operators -- same LU-golden phi, same iteration counts as the dense path -- and A P v = 0 to rounding."""
    import os
    import subprocess
    import sys
    from conftest import GOLDEN, ROOT
    if not os.path.exists(os.path.join(GOLDEN, case + ".npz")):
        pytest.skip("fixture not generated")
    code = r"""
import sys, numpy as np
sys.path.insert(0, %r)
import shm_import
shm = shm_import.load()
d = np.load(%r)
out = {}
for mode, kw in (("dual", dict(solver="dual")), ("primal-dct", dict(solver="primal", precond="dct")), ("primal-plain", dict(solver="primal", precond="none"))):
    s = shm.GridSolver()
    s.set_problem(d["pos"], d["wnormal"], d["area"], float(d["lam"]), int(d["n"]), d["bbox_min"], float(d["cell"]))
    st = s.solve(tol=1e-10, scrub=%r, **kw)
    phi, _ = s.get_phi()
    out[mode] = (float(np.abs(phi - d["phi"]).max()), int(st.iters))
    if mode == "dual":
        rng = np.random.default_rng(1)
        v = rng.standard_normal(int(d["n"]) ** 3)
        Pv = s.apply_projector(v)
        nodes, coeffs = s.get_constraints()
        out["APv"] = float(np.abs((coeffs * Pv[nodes]).sum(axis=1)).max())
print(repr(out))
""" % (ROOT, os.path.join(GOLDEN, case + ".npz"), scrub)
    res = {}
    for name, env in (("dense", {}), ("two-level", {"SHM_TL_MIN_M": "32", "SHM_TL_BOX": str(box)})):
        p = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **env), capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, p.stderr
        res[name] = eval(p.stdout.strip().splitlines()[-1])
    for mode in ("dual", "primal-dct", "primal-plain"):
        err, iters = res["two-level"][mode]
        assert err < (1e-6 if case.endswith("n64") else 1e-7), (mode, res)
        assert abs(iters - res["dense"][mode][1]) <= 4, (mode, res)       # same operator -> same iteration count (check granularity: 2-4)
    assert res["two-level"]["APv"] < 1e-10, res


@pytest.mark.parametrize("seed,n", [(1, 256), (2, 256), (3, 512)])
def test_dual_z_recursion_against_primal_dct_with_gapped_planes(shm, seed, n):
    """From n = 256 on the dual solver's sparse fast Poisson solve does its z step by a per-line recursion over the ACTIVE planes only
    (zsolve_sparse_kernel).  A few dozen scattered points leave large gaps between the active planes -- the r^gap / (1/r)^gap branches, planes near
    both ends of the grid, low and high (kx, ky) lines -- which the dense surface inputs never produce.  No oracle finishes at this size; the
    check is against the independent primal solver (projected stencil CG with the dense DCT preconditioner, no recursion anywhere)."""
    rng = np.random.default_rng(seed)
    S = 48
    pos = rng.uniform(-0.93, 0.93, (S, 3))
    pos[:4, 2] = [-0.98, -0.975, 0.97, 0.985]                 # cells next to the bottom / top planes
    d = pos / np.linalg.norm(pos, axis=1)[:, None]
    nrm = d + 0.3 * rng.standard_normal((S, 3))
    nrm /= np.linalg.norm(nrm, axis=1)[:, None]
    area = rng.uniform(0.5, 1.5, S) * 0.05
    bbox_min = np.array([-1.0, -1.0, -1.0])
    cell = 2.0 / (n - 1)
    lam = 6.0
    g = shm.GridSolver()
    g.set_problem(pos, nrm * area[:, None], area, lam, n, bbox_min, cell)
    st_d = g.solve(tol=1e-11, solver="dual")
    phi_d, _ = g.get_phi()
    st_p = g.solve(tol=1e-11, solver="primal", precond="dct")
    phi_p, _ = g.get_phi()
    assert st_d.solver == 2 and st_p.solver == 1 and st_d.m == S
    assert np.isfinite(phi_d).all()
    assert np.abs(phi_d - phi_p).max() < 1e-7 * max(1.0, np.abs(phi_p).max()), (st_d.iters, st_p.iters)


def test_errors_are_reported(shm):
    d = load_golden("bunny_small_n16")
    s = shm.GridSolver()
    with pytest.raises(shm.ShmError) as e:
        s.solve()
    assert e.value.status == 7  # SHM_ERR_STATE
    with pytest.raises(shm.ShmError):
        s.set_problem(d["pos"], d["wnormal"], d["area"], float(d["lam"]), 16, d["bbox_min"] + 100.0, float(d["cell"]))  # sources outside
    with pytest.raises(shm.ShmError):
        s.set_problem(d["pos"], d["wnormal"], d["area"], -1.0, 16, d["bbox_min"], float(d["cell"]))


# ---- fp32 path (BASELINE.json configs[2], configs[4]): "report only" in the north star; here: sanity bounds -----------
@pytest.mark.parametrize("case", ["bunny_small_n16", "bunny_small_n32", "bunny_pc_n32"])
@pytest.mark.parametrize("mode", sorted(MODES))
def test_fp32_path_tracks_fp64_oracle(shm, case, mode):
    """fp32 storage/arithmetic with fp64 dot accumulators and the per-node exponent offset in Step 1 (SURVEY trap #4).
    Tolerance 2e-3 absolute on phi (range ~[-0.5, 4.6]): fp32 CG at kappa ~1e4..1e5 cannot do better."""
    d = load_golden(case)
    s = make_solver(shm, d, precision=shm.SHM_F32)
    st = s.solve(scrub="pc" not in case, allow_noconv=True, **MODES[mode])
    phi, _ = s.get_phi()
    assert np.isfinite(phi).all()
    err = np.abs(phi - d["phi"]).max()
    assert err < 2e-3, (err, st.iters, st.rel_residual)


@pytest.mark.parametrize("coarse", [False, True])
def test_fp32_conv_exponent_offset_prevents_underflow(shm, coarse):
    """lambda*d ~ 400 at the grid corners: exp() underflows in fp32 (it flushes beyond ~87) while fp64 still holds
    ~1e-170.  The per-tile offset keeps every fp32 direction finite and equal to the fp64 one."""
    d = load_golden("bunny_small_n16")
    n = 16 if coarse else 128          # coarse: lambda*cell = 33 -> per-node offsets; fine: lambda*cell = 4 -> per-tile offset
    lam = float(d["lam"]) * 4.0
    c = d["bbox_min"] + 7.5 * float(d["cell"])
    bbox = c - 2.0 * (c - d["bbox_min"])
    cell = float(d["cell"]) * 2.0 * 15 / (n - 1)
    Y = {}
    for prec in (shm.SHM_F64, shm.SHM_F32):
        s = shm.GridSolver(precision=prec)
        s.set_problem(d["pos"], d["wnormal"], d["area"], lam, n, bbox, cell)
        s.run_conv()
        Y[prec] = np.stack([s.get_field(k) for k in (0, 1, 2)], axis=1)
    assert np.isfinite(Y[shm.SHM_F32]).all()
    ok = np.isfinite(Y[shm.SHM_F64]).all(axis=1)
    assert ok.mean() > 0.9
    # directions agree wherever fp64 itself is well conditioned (away from the medial axis the agreement is ~1e-4)
    dots = (Y[shm.SHM_F64][ok] * Y[shm.SHM_F32][ok]).sum(axis=1)
    assert np.median(1 - dots) < 1e-6 and np.quantile(1 - dots, 0.99) < 1e-3


# ---- BASELINE.json configs[2] (rocker.obj, 512^3, fp32) and configs[4] (SprayBottle, 1024^3, fp32; the .obj is missing upstream, the
# ---- .pc of the same model stands in): the fp32 path against the fp64 C oracle at sizes the oracle finishes in seconds, against an
# ---- fp64 GPU run at intermediate sizes, and through size-independent properties at the full sizes.  The north star gates fp64 only
# ---- ("fp32 configs: report only"); the bounds asserted here are what the fp32 path measurably holds, printed per test.
FP32_CASES = {"rocker.obj": True, "SprayBottle.pc": False}     # file -> divYt scrub (mesh overload only, :70-74)


def _preprocess(fname, hCoef):
    import os
    from conftest import ROOT
    from signed_heat_3d_amd.host_abi import HostSolver
    return HostSolver(os.path.join(ROOT, "data", fname)).preprocess(hCoef=hCoef)


def _gpu_phi(shm, pre, precision, scrub, **kw):
    s = shm.GridSolver(precision=precision)
    s.set_problem(pre["pos"], pre["wnormal"], pre["area"], pre["lam"], pre["n"], pre["bbox_min"], pre["cell"])
    st = s.solve(scrub=scrub, allow_noconv=True, **kw)
    phi, _ = s.get_phi()
    return s, st, phi


@pytest.mark.parametrize("fname", sorted(FP32_CASES))
def test_fp32_configs_match_c_oracle_64(shm, oracle_c, fname):
    """rocker.obj / SprayBottle.pc at 64^3: fp32 HIP path (and the fp64 one beside it) against the fp64 C oracle on the same inputs."""
    import os
    scrub = FP32_CASES[fname]
    pre = _preprocess(fname, 2.0)
    n, S = pre["n"], pre["S"]
    assert n == 64
    ref = np.zeros(n ** 3)
    st = np.zeros(5)
    oracle_c.shmo_set_threads(min(64, os.cpu_count() or 1))
    rc = oracle_c.shmo_compute_distance(n, c_(pre["bbox_min"]), pre["cell"], S, c_(pre["pos"]).reshape(-1), c_(pre["wnormal"]).reshape(-1),
                                        c_(pre["area"]), pre["lam"], int(scrub), 0, 1e-12, 100000, ref, st)
    oracle_c.shmo_set_threads(min(8, os.cpu_count() or 1))
    assert rc == 0 and np.isfinite(ref).all()
    _, st64, phi64 = _gpu_phi(shm, pre, shm.SHM_F64, scrub, tol=1e-10)
    _, st32, phi32 = _gpu_phi(shm, pre, shm.SHM_F32, scrub)
    span = np.abs(ref).max()
    e64, e32 = np.abs(phi64 - ref).max(), np.abs(phi32 - ref).max()
    print("\n%s 64^3: L_inf(fp64 - oracle) = %.3e, L_inf(fp32 - oracle) = %.3e (max|phi| = %.3f, fp32 iters %d)" % (fname, e64, e32, span, st32.iters))
    assert np.isfinite(phi32).all()
    assert e64 < 1e-7 * max(1.0, span)
    assert e32 < 2e-4 * span, (e32, span)


@pytest.mark.parametrize("fname,hCoef", [("rocker.obj", 3.0), ("SprayBottle.pc", 3.0), ("SprayBottle.pc", 4.0)])
def test_fp32_configs_track_fp64_gpu(shm, fname, hCoef):
    """128^3 / 256^3 (the C oracle needs minutes there): fp32 against the fp64 HIP path, itself held to the oracle at <= 64^3 above."""
    scrub = FP32_CASES[fname]
    pre = _preprocess(fname, hCoef)
    _, st64, phi64 = _gpu_phi(shm, pre, shm.SHM_F64, scrub)
    _, st32, phi32 = _gpu_phi(shm, pre, shm.SHM_F32, scrub)
    span = np.abs(phi64).max()
    e = np.abs(phi32 - phi64).max()
    print("\n%s %d^3: L_inf(fp32 - fp64) = %.3e (max|phi| = %.3f; iters fp64 %d, fp32 %d)" % (fname, pre["n"], e, span, st64.iters, st32.iters))
    assert np.isfinite(phi32).all() and np.isfinite(phi64).all()
    assert e < 2e-4 * span, (e, span)


def _fullsize_fp32_properties(shm, pre, scrub, with_fp64):
    """Size-independent checks of an fp32 solve: finite everywhere incl. the bbox corners (SURVEY trap #4: lambda*d ~ 114-172 on rocker
    underflows a plain fp32 exp), KKT stationarity away from the constraint stencils (independent laplacian_kernel), equal rows of
    A phi, zero area-weighted source mean; optionally L_inf against an fp64 run of the same configuration."""
    n = pre["n"]
    s, st, phi = _gpu_phi(shm, pre, shm.SHM_F32, scrub)
    assert np.isfinite(phi).all()
    for corner in (0, n - 1, n * (n - 1), n * n * (n - 1), n ** 3 - 1):
        assert np.isfinite(phi[corner]) and phi[corner] > 0          # far outside the surface: positive distance
    s.run_conv()
    s.run_divergence(scrub)
    b = s.get_field(s.FIELD_DIV)
    assert np.isfinite(b).all()
    g = s.apply_laplacian(phi) + b
    nodes, coeffs = s.get_constraints()
    assert nodes.shape[0] == st.m
    touched = np.zeros(phi.size, dtype=bool)
    touched[nodes.ravel()] = True
    scale = np.abs(b).max()
    res = np.abs(g[~touched]).max() / scale
    rows = (coeffs * phi[nodes]).sum(axis=1)
    spread = rows.max() - rows.min()
    span = np.abs(phi).max()
    del g, b, touched
    out = dict(n=n, m=int(st.m), iters=int(st.iters), kkt_residual_rel=float(res), row_spread=float(spread), max_abs_phi=float(span))
    # fp32: phi carries ~6e-8 * span of rounding per entry, the stencil amplifies it by 12 / h^2
    h = pre["cell"]
    assert res < 5e-3 + 12.0 * 6e-8 * span / (h * h) / scale, out
    assert spread < 2e-4 * span, out
    if with_fp64:
        _, st64, phi64 = _gpu_phi(shm, pre, shm.SHM_F64, scrub)
        e = float(np.abs(phi - phi64).max())
        out.update(linf_fp32_vs_fp64=e, iters_fp64=int(st64.iters))
        assert e < 2e-4 * span, out
    print("\nfull-size fp32 properties:", out)
    return out


def test_config2_rocker_512_fp32_full_size(shm):
    """BASELINE.json configs[2] at its full size: rocker.obj, 512^3, fp32 (S = 13 819 faces, m = 12 612)."""
    pre = _preprocess("rocker.obj", 5.0)
    assert pre["n"] == 512 and pre["S"] == 13819
    out = _fullsize_fp32_properties(shm, pre, True, with_fp64=True)
    assert out["m"] == 12612


@pytest.mark.skipif(bool(__import__("os").environ.get("SHM_SKIP_1024")), reason="SHM_SKIP_1024 set")
def test_config4_spraybottle_1024_fp32_full_size(shm):
    """BASELINE.json configs[4] on one GPU: SprayBottle.pc, 1024^3, fp32 (S = 52 290 points, m = 48 893; ~60 GB of HBM, ~10 s solve).
    No fp64 run beside it (another 110 GB of HBM and 6 s: profiles/r03_bench_spraybottle_pc_1024_f64.json); the fp32-vs-fp64 error of this input is asserted at 128^3 / 256^3."""
    import psutil
    if psutil.virtual_memory().available < 80 * 2 ** 30:
        pytest.skip("needs ~50 GB of host memory for the float64 copies of phi, b and L phi")
    pre = _preprocess("SprayBottle.pc", 6.0)
    assert pre["n"] == 1024 and pre["S"] == 52290
    out = _fullsize_fp32_properties(shm, pre, False, with_fp64=False)
    assert out["m"] == 48893


# ---- fastIntegration (--f): integrateGreedily, signed_heat_grid_solver.cpp:224-275 ---------------------------------------
@pytest.mark.parametrize("n", [16, 32])
@pytest.mark.parametrize("slabs", [1, 2, 5])
def test_fast_integration_matches_bfs_golden(shm, n, slabs):
    """The device path (three families of prefix scans) must reproduce the reference's order-dependent FIFO BFS."""
    d = load_golden("bunny_small_fast_n%d" % n)
    s = make_solver(shm, d, local_slabs=slabs)
    st = s.solve(fast=True)
    phi, _ = s.get_phi()
    assert abs(st.shift - float(d["shift"])) < 1e-9
    assert np.abs(phi - d["phi"]).max() < 1e-9


def test_fast_integration_matches_c_oracle_odd_size(shm, oracle_c):
    d = load_golden("bunny_small_n16")
    n = 23
    cell = float(d["cell"]) * 15 / (n - 1)
    s = shm.GridSolver()
    s.set_problem(d["pos"], d["wnormal"], d["area"], float(d["lam"]), n, d["bbox_min"], cell)
    s.solve(fast=True)
    phi, _ = s.get_phi()
    ref = np.zeros(n ** 3)
    st = np.zeros(5)
    oracle_c.shmo_compute_distance(n, c_(d["bbox_min"]), cell, len(d["area"]), c_(d["pos"]).reshape(-1), c_(d["wnormal"]).reshape(-1),
                                   c_(d["area"]), float(d["lam"]), 1, 1, 0.0, 0, ref, st)
    assert np.abs(phi - ref).max() < 1e-9


def test_rccl_single_rank_communicator(shm):
    """world=1 with an ncclUniqueId: the library loads librccl, creates a 1-rank communicator and routes its reductions
    through ncclAllReduce on the solver's stream (the multi-GPU code path, minus the peers)."""
    d = load_golden("bunny_small_n16")
    uid = shm.comm_unique_id()
    assert len(uid) == 128
    s = make_solver(shm, d, world=1, rank=0, rccl_unique_id=uid)
    st = s.solve(tol=1e-10)
    phi, _ = s.get_phi()
    assert np.abs(phi - d["phi"]).max() < 1e-7


# ---- BASELINE.json's full size (configs[1]: bunny_small.obj, 256^3, fp64): size-independent properties -------------------
@pytest.fixture(scope="module")
def full_size(shm):
    import os
    from conftest import ROOT
    from signed_heat_3d_amd.host_abi import HostSolver
    pre = HostSolver(os.path.join(ROOT, "data", "bunny_small.obj")).preprocess(hCoef=4.0)
    s = shm.GridSolver()
    s.set_problem(pre["pos"], pre["wnormal"], pre["area"], pre["lam"], pre["n"], pre["bbox_min"], pre["cell"])
    s.run_conv()
    s.run_divergence(True)
    b = s.get_field(s.FIELD_DIV)
    st = s.solve()                      # library defaults: dual solver, tol 1e-8
    phi, _ = s.get_phi()
    return dict(pre=pre, solver=s, b=b, phi=phi, stats=st)


def test_full_size_kkt_stationarity(full_size):
    """L x + A^T mu = b with x = -phi (:101-108): away from the <= 8m nodes touched by constraint rows the residual
    L phi + b must vanish; on the touched nodes it is A^T mu.  Uses the independent laplacian_kernel."""
    f = full_size
    s, phi, b = f["solver"], f["phi"], f["b"]
    g = s.apply_laplacian(phi) + b
    nodes, coeffs = s.get_constraints()
    touched = np.zeros(phi.size, dtype=bool)
    touched[nodes.ravel()] = True
    assert f["stats"].m == nodes.shape[0] == 2842
    scale = np.abs(b).max()
    assert np.abs(g[~touched]).max() < 1e-6 * scale
    assert np.abs(g[touched]).max() > 1e-3 * scale          # the multipliers are not trivially zero


def test_full_size_constraints_and_shift(full_size):
    """A x = 0 up to the common constant absorbed by the shift: all rows of A phi agree; and the area-weighted mean of the
    trilinear interpolant of phi over ALL sources is zero (:110-111, :466-481)."""
    f = full_size
    s, phi, pre = f["solver"], f["phi"], f["pre"]
    nodes, coeffs = s.get_constraints()
    rows = (coeffs * phi[nodes]).sum(axis=1)
    assert rows.max() - rows.min() < 1e-8
    n, h, b0 = pre["n"], pre["cell"], pre["bbox_min"]
    t = (pre["pos"] - b0) / h
    ijk = np.floor(t).astype(np.int64)
    tx, ty, tz = ((pre["pos"][:, a] - (ijk[:, a] * h + b0[a])) / h for a in range(3))
    at = lambda di, dj, dk: phi[(ijk[:, 0] + di) + (ijk[:, 1] + dj) * n + (ijk[:, 2] + dk) * n * n]  # noqa: E731
    v00 = at(0, 0, 0) * (1 - tx) + at(1, 0, 0) * tx
    v01 = at(0, 0, 1) * (1 - tx) + at(1, 0, 1) * tx
    v10 = at(0, 1, 0) * (1 - tx) + at(1, 1, 0) * tx
    v11 = at(0, 1, 1) * (1 - tx) + at(1, 1, 1) * tx
    v = (v00 * (1 - ty) + v10 * ty) * (1 - tz) + (v01 * (1 - ty) + v11 * ty) * tz
    assert abs((pre["area"] * v).sum() / pre["area"].sum()) < 1e-10


def test_full_size_solvers_agree_and_are_deterministic(full_size, monkeypatch):
    f = full_size
    s, phi = f["solver"], f["phi"]
    assert f["stats"].cg_form == 2 and f["stats"].iters <= 2 and f["stats"].rel_residual < 1e-10   # the direct dual solve (explicit S^-1), one pass
    st2 = s.solve()
    phi2, _ = s.get_phi()
    assert np.array_equal(phi, phi2)                        # fixed-order reductions: bit-identical reruns
    monkeypatch.setenv("SHM_DUAL_NO_DIRECT", "1")           # the iterative dual solver (CG on the explicit S), read per solve
    st_it = s.solve()
    phi_it, _ = s.get_phi()
    monkeypatch.delenv("SHM_DUAL_NO_DIRECT")
    assert st_it.cg_form == 3 and st_it.iters >= 30 and np.abs(phi_it - phi).max() < 1e-8, (st_it.cg_form, st_it.iters)
    st3 = s.solve(solver="primal", precond="dct")
    phi3, _ = s.get_phi()
    assert np.abs(phi3 - phi).max() < 1e-7, (st2.iters, st3.iters)
    assert phi.argmax() == 0 and abs(phi.max() - 4.474) < 2e-3 and abs(phi.min() + 0.5996) < 2e-3   # profiles/r01_parity_*.json


@pytest.mark.parametrize("case", ["bunny_small_n32", "bunny_small_n64"])
def test_direct_dual_solve_accepts_the_rounding_floor(shm, case):
    """A tolerance below eps * cond(S) cannot be met; the direct dual solve (explicit S^-1 + iterative refinement, <= 6 passes) must notice that its passes
    have stopped gaining and return the converged field with the residual it reached, not SHM_ERR_NOCONV (include/shm_grid.h, shm_opts.tol)."""
    d = load_golden(case)
    s = make_solver(shm, d)
    st = s.solve(tol=1e-17)             # would raise on SHM_ERR_NOCONV
    assert st.cg_form == 2 and 2 <= st.iters <= 6 and st.rel_residual < 1e-9, (st.cg_form, st.iters, st.rel_residual)
    phi, _ = s.get_phi()
    assert np.abs(phi - d["phi"]).max() < 1e-7


@pytest.mark.parametrize("fname,hc,precision,form", [("rocker.obj", 3.0, 64, 2), ("chair.pc", 4.0, 64, 2), ("rocker.obj", 4.0, 64, 0), ("chair.obj", 4.0, 32, 0)])
def test_dual_form_choice_and_agreement_of_the_forms(shm, fname, hc, precision, form, monkeypatch):
    """The dual solver picks its form per problem (round 4 re-measured the choices): the direct solve (explicit S^-1) for mid-size
    constraint sets where a sampled estimate of Step 1 says the inversion hides behind it (rocker 128^3: m = 4 169, chair.pc 256^3: 4 535), the CG with S applied
    through the grid where it does not (rocker 256^3: 9 110) and in the fp32 solve (chair 256^3).  Whatever is picked, forcing the iterative form on the same
    solver (SHM_DUAL_NO_DIRECT, read per solve) must give the same field to the solvers' tolerance: they solve the same KKT system."""
    pre = _preprocess(fname, hc)
    scrub = not fname.endswith(".pc")
    s, st, phi = _gpu_phi(shm, pre, precision, scrub)
    assert st.cg_form == form, (st.m, st.cg_form)
    assert np.isfinite(phi).all()
    monkeypatch.setenv("SHM_DUAL_NO_DIRECT", "1")
    st2 = s.solve(scrub=scrub, allow_noconv=True)
    phi2, _ = s.get_phi()
    assert st2.cg_form in (0, 3) and st2.iters > 4
    tol = (2e-7 if precision == 64 else 2e-4) * max(1.0, float(np.abs(phi).max()))
    assert np.abs(phi2 - phi).max() < tol, (np.abs(phi2 - phi).max(), st.iters, st2.iters)
    s.close()


def test_config3_bunny_pc_512_fp64_full_size(shm):
    """BASELINE.json configs[3] on one GPU at its full size: data/bunny.pc (point overload: no divYt scrub, signed_heat_grid_solver.cpp:116-222, :179-180),
    hCoef 5 = 512^3, fp64; areas / h from the build's estimator (inputs of the ABI).  Size-independent properties: KKT stationarity off the
    constraint stencils (independent laplacian_kernel), equal rows of A phi, zero area-weighted source mean, bit-identical reruns, and L_inf against the
    primal + DCT solver (a different algorithm on the same KKT system) <= 1e-7."""
    import psutil
    if psutil.virtual_memory().available < 12 * 2 ** 30:
        pytest.skip("needs ~8 GB of host memory for the float64 copies of phi, b and L phi at 512^3")
    pre = _preprocess("bunny.pc", 5.0)
    n = pre["n"]
    assert n == 512 and pre["S"] == 1430
    s = shm.GridSolver()
    s.set_problem(pre["pos"], pre["wnormal"], pre["area"], pre["lam"], n, pre["bbox_min"], pre["cell"])
    s.run_conv()
    s.run_divergence(False)                                   # point overload: non-finite entries are NOT zeroed (:180)
    b = s.get_field(s.FIELD_DIV)
    assert np.isfinite(b).all()
    st = s.solve(scrub=False)                                 # library defaults: dual solver, tol 1e-8
    phi, _ = s.get_phi()
    assert np.isfinite(phi).all()
    assert st.m == 1430 and st.cg_form == 2 and st.iters <= 2 and st.rel_residual < 1e-10
    nodes, coeffs = s.get_constraints()
    assert nodes.shape[0] == st.m
    g = s.apply_laplacian(phi) + b
    touched = np.zeros(phi.size, dtype=bool)
    touched[nodes.ravel()] = True
    scale = np.abs(b).max()
    assert np.abs(g[~touched]).max() < 1e-6 * scale
    assert np.abs(g[touched]).max() > 1e-3 * scale           # the multipliers are not trivially zero
    del g, touched
    rows = (coeffs * phi[nodes]).sum(axis=1)
    assert rows.max() - rows.min() < 1e-8
    h, b0 = pre["cell"], pre["bbox_min"]
    ijk = np.floor((pre["pos"] - b0) / h).astype(np.int64)
    tx, ty, tz = ((pre["pos"][:, a] - (ijk[:, a] * h + b0[a])) / h for a in range(3))
    at = lambda di, dj, dk: phi[(ijk[:, 0] + di) + (ijk[:, 1] + dj) * n + (ijk[:, 2] + dk) * n * n]  # noqa: E731
    v00 = at(0, 0, 0) * (1 - tx) + at(1, 0, 0) * tx
    v01 = at(0, 0, 1) * (1 - tx) + at(1, 0, 1) * tx
    v10 = at(0, 1, 0) * (1 - tx) + at(1, 1, 0) * tx
    v11 = at(0, 1, 1) * (1 - tx) + at(1, 1, 1) * tx
    v = (v00 * (1 - ty) + v10 * ty) * (1 - tz) + (v01 * (1 - ty) + v11 * ty) * tz
    assert abs((pre["area"] * v).sum() / pre["area"].sum()) < 1e-10
    s.solve(scrub=False)
    phi2, _ = s.get_phi()
    assert np.array_equal(phi, phi2)                          # fixed-order reductions and a deterministic work queue result: bit-identical reruns
    del phi2
    st3 = s.solve(scrub=False, solver="primal", precond="dct", tol=1e-10)
    phi3, _ = s.get_phi()
    e = float(np.abs(phi3 - phi).max())
    print("\nconfigs[3] bunny.pc 512^3 fp64: m %d, direct dual passes %d, primal+dct iters %d, L_inf(dual - primal) = %.3e, max|phi| %.3f" % (st.m, st.iters, st3.iters, e, np.abs(phi).max()))
    assert e < 1e-7, e
    assert phi.argmax() == 0 and phi.min() < 0 < phi.max()


def test_headless_cli_reproduces_golden(tmp_path):
    """shm_grid_cli = headless solve() (src/main.cpp:68-114): same flags, `min/max` line, phi written as raw float64."""
    import os
    import subprocess
    from conftest import ROOT
    exe = os.path.join(ROOT, "signed-heat-3d_amd", "bin", "shm_grid_cli")
    out = str(tmp_path / "phi.f64")
    p = subprocess.run([exe, os.path.join(ROOT, "data", "bunny_small.obj"), "--g", "--V", "--h", "1", "--tol", "1e-10", "--out", out],
                       capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    d = load_golden("bunny_small_n32")
    phi = np.fromfile(out, dtype=np.float64)
    assert phi.size == 32 ** 3 and np.abs(phi - d["phi"]).max() < 1e-7
    assert "min: -0.455887" in p.stderr and "max: 4.53795" in p.stderr     # BASELINE.md spot values, printed like src/main.cpp:101
    # --iso / --export (headless contour + "Export isosurface", src/main.cpp:116-128,167-191)
    obj = str(tmp_path / "isosurface.obj")
    p = subprocess.run([exe, os.path.join(ROOT, "data", "bunny_small.obj"), "--g", "--h", "1", "--iso", "0.1", "--export", obj], capture_output=True, text=True)
    assert p.returncode == 0 and "Isosurface written to" in p.stderr, p.stderr
    lines = open(obj).read().split("\n")
    nv, nf = sum(l.startswith("v ") for l in lines), sum(l.startswith("f ") for l in lines)
    assert nv > 500 and abs(nf - (2 * nv - 4)) <= 8          # marching cubes: closed surface(s) of genus 0 (F = 2V - 4 per component)
    # --f (fastIntegration)
    p = subprocess.run([exe, os.path.join(ROOT, "data", "bunny_small.obj"), "--g", "--f", "--h", "1", "--out", out], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    phi = np.fromfile(out, dtype=np.float64)
    assert np.abs(phi - load_golden("bunny_small_fast_n32")["phi"]).max() < 1e-9


# ---- isosurface extraction ------------------------------------------------------------------------
@pytest.mark.parametrize("method", ["marching_cubes", "marching_tets"])
@pytest.mark.parametrize("slabs", [1, 3])
def test_isosurface_matches_python_restatement(shm, slabs, method):
    """iso_mc_kernel (the demo's contour: Polyscope marching cubes, src/main.cpp:121-124) and iso_kernel (marching tetrahedra) against tests/iso_ref.py:
    same vertices, same triangles (as welded-vertex triples, orientation included), watertight, oriented towards increasing phi."""
    import iso_ref
    d = load_golden("bunny_small_n32")
    s = make_solver(shm, d, local_slabs=slabs)
    s.solve(tol=1e-10, solver="primal", precond="none" if slabs == 3 else "auto")
    phi, _ = s.get_phi()
    V, F = s.isosurface(0.0, method=method)
    pts, tris = getattr(iso_ref, method)(phi, 32, d["bbox_min"], float(d["cell"]), 0.0)
    if method == "marching_cubes":
        Vd, Fd = s.isosurface(0.0)
        assert np.array_equal(Vd, V) and np.array_equal(Fd, F)      # the default entry point is marching cubes
        # triangle by triangle: positions of the three corners, rotated to start at the smallest, as a sorted list
        def canon(T):   # noqa: E306
            out = []
            for t in T:
                k = min(range(3), key=lambda a: tuple(t[a]))
                out.append(np.concatenate([t[(k + a) % 3] for a in range(3)]))
            return np.array(sorted(map(tuple, np.round(np.array(out), 11))))
        got = canon(V[F])
        ref = canon(np.array([[pts[k] for k in t] for t in tris]))
        assert got.shape == ref.shape and np.abs(got - ref).max() < 1e-10
    assert len(V) == len(pts) and len(F) == len(tris)
    ref = np.array(sorted(map(tuple, np.round(np.array(list(pts.values())), 12))))
    got = np.array(sorted(map(tuple, np.round(V, 12))))
    assert np.abs(ref - got).max() < 1e-10
    # watertight and consistently oriented: every edge is used exactly once in each direction
    from collections import Counter
    e = Counter()
    for a, b, c in F:
        for u, v in ((a, b), (b, c), (c, a)):
            e[(int(u), int(v))] += 1
    assert all(cnt == 1 and e.get((v, u), 0) == 1 for (u, v), cnt in e.items())
    # the zero set of the signed distance hugs the input surface: enclosed volume > 0 with outward normals
    p0, p1, p2 = V[F[:, 0]], V[F[:, 1]], V[F[:, 2]]
    vol = np.einsum("ij,ij->i", p0, np.cross(p1, p2)).sum() / 6.0
    area = 0.5 * np.linalg.norm(np.cross(p1 - p0, p2 - p0), axis=1).sum()
    # (at 32^3 the level set of the coarse phi is a blobby bunny: its area is only loosely the mesh area)
    assert vol > 0 and 0.8 * d["area"].sum() < area < 1.6 * d["area"].sum()


def test_isosurface_of_a_sphere_known_answer(shm):
    """phi = |x| - 0.9 sampled on the grid through the solver's phi buffer is not injectable from outside, so use the SHM
    result for sphere point samples: the 0-level set is a closed surface of area ~ 4 pi r^2 and radius ~ r."""
    import math
    P = 3000
    i = np.arange(P) + 0.5
    z = 1 - 2 * i / P
    th = math.pi * (1 + 5 ** 0.5) * i
    pts = np.stack([np.sqrt(1 - z * z) * np.cos(th), np.sqrt(1 - z * z) * np.sin(th), z], axis=1)
    areas = np.full(P, 4 * math.pi / P)
    h = math.sqrt(4 * math.pi / P)
    n = 64
    s = shm.GridSolver()
    s.set_problem(pts, pts * areas[:, None], areas, 1.0 / h, n, np.array([-2.0, -2.0, -2.0]), 4.0 / (n - 1))
    s.solve(scrub=False)
    V, F = s.isosurface(0.0)
    r = np.linalg.norm(V, axis=1)
    assert abs(r.mean() - 1.0) < 0.03 and r.std() < 0.03
    p0, p1, p2 = V[F[:, 0]], V[F[:, 1]], V[F[:, 2]]
    area = 0.5 * np.linalg.norm(np.cross(p1 - p0, p2 - p0), axis=1).sum()
    assert abs(area - 4 * math.pi) < 0.08 * 4 * math.pi
    assert len(F) == 2 * len(V) - 4                                         # one closed genus-0 surface
    Vt, Ft = s.isosurface(0.0, method="marching_tets")
    assert len(Ft) == 2 * len(Vt) - 4 and len(F) < 0.6 * len(Ft)            # the same surface with fewer triangles
    import ctypes as C
    nv, nt = C.c_int64(), C.c_int64()
    assert s._lib.shm_grid_isosurface_ex(s._h, 0.0, 7, C.byref(nv), C.byref(nt)) == 1   # SHM_ERR_INVALID: unknown method


@pytest.mark.skipif(not __import__("os").environ.get("SHM_BIG_TESTS"), reason="1024^3 (~100 GB of HBM, minutes): set SHM_BIG_TESTS=1")
def test_preconditioner_1024_property(shm):
    """n = 1024 uses 8-line tiles and three Stockham passes: K M^-1 v = v - mean(v), checked with the independent stencil."""
    d = load_golden("bunny_small_n16")
    n = 1024
    h = float(d["cell"]) * 15 / (n - 1)
    s = shm.GridSolver()
    s.set_problem(d["pos"], d["wnormal"], d["area"], float(d["lam"]), n, d["bbox_min"], h)
    rng = np.random.default_rng(3)
    v = rng.standard_normal(n ** 3)
    got = s.apply_preconditioner(v)
    Lg = s.apply_laplacian(got)
    assert np.abs(-Lg - (v - v.mean())).max() < 1e-8 * np.abs(v).max()


# ---- several PROCESSES (ranks) on one GPU through a shared-memory double of librccl --------------------------------------
def _build_rccl_mock(tmp_path):
    import os
    import subprocess
    from conftest import ROOT
    so = str(tmp_path / "librccl_mock.so")
    subprocess.check_call(["gcc", "-O2", "-shared", "-fPIC", os.path.join(ROOT, "tests", "native", "rccl_mock.c"), "-o", so, "-I/opt/rocm/include",
                           "-D__HIP_PLATFORM_AMD__", "-L/opt/rocm/lib", "-lamdhip64", "-lrt", "-lpthread"])
    return so


def _run_ranks(tmp_path, so, world, case, mode, tag, extra_env=None):
    """One process per rank on GPU 0 through the shared-memory double of librccl; returns phi (rank-major concatenation) and the per-rank metadata."""
    import os
    import subprocess
    import sys
    from conftest import ROOT
    uid = ("/shmmock_%d_%s_%d" % (os.getpid(), tag, world)).encode().ljust(128, b"\x00")
    env = dict(os.environ, SHM_RCCL_LIB=so)
    env.update(extra_env or {})
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "multiproc_worker.py"), str(r), str(world), uid.hex(), case, mode, str(tmp_path)],
                              env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(world)]
    outs = [p.communicate(timeout=600)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    parts, metas, covered = [], [], 0
    for r in range(world):
        k0, k1, iters, shift = np.load(tmp_path / ("meta_%d.npy" % r))
        assert int(k0) == covered
        covered = int(k1)
        parts.append(np.load(tmp_path / ("phi_%d.npy" % r)))
        metas.append((int(k0), int(k1), int(iters), float(shift)))
    return np.concatenate(parts), metas, covered


# ---- several PROCESSES (ranks) on one GPU through a shared-memory double of librccl --------------------------------------
@pytest.mark.parametrize("world,mode,case", [(2, "dual", "bunny_small_n32"), (4, "dual", "bunny_small_n32"), (8, "dual", "bunny_small_n32"),
                                             (2, "dual-slabs", "bunny_small_n32"), (4, "dual-slabs", "bunny_small_n32"),
                                             # round 6: the slab-distributed explicit-S forms -- S and S^-1 replicated beside every rank's Step 1, K^+ on the slabs (two all-to-alls
                                             # per application), the dual system on m-vectors every rank holds: the direct solve, and CG on the explicit S
                                             (2, "dual-direct-slabs", "bunny_small_n32"), (4, "dual-direct-slabs", "bunny_small_n32"), (8, "dual-direct-slabs", "bunny_small_n32"),
                                             (2, "dual-direct-slabs", "bunny_pc_n32"), (8, "dual-direct-slabs", "bunny_pc_n32"),
                                             (2, "dual-scg-slabs", "bunny_small_n32"), (4, "dual-scg-slabs", "bunny_pc_n32"),
                                             (2, "primal-plain", "bunny_small_n32"), (8, "primal-plain", "bunny_small_n32"),
                                             (2, "primal-dct", "bunny_small_n32"), (8, "primal-dct", "bunny_small_n32"),
                                             (4, "fast", "bunny_small_fast_n32"), (8, "fast", "bunny_small_fast_n32"),
                                             (2, "primal-plain+overlap", "bunny_small_n32"), (4, "primal-dct+overlap", "bunny_small_n32"),
                                             # the point overload (configs[3]: no divYt scrub, signed_heat_grid_solver.cpp:116-222) through the rank path
                                             (2, "dual", "bunny_pc_n32"), (8, "dual", "bunny_pc_n32"), (2, "primal-dct", "bunny_pc_n32"), (8, "primal-dct", "bunny_pc_n32"),
                                             # a grid side that is not a power of two (24): Steps 1-2 on slabs, D^T Y gathered, whole-grid dual solve with the dense DCT products
                                             (2, "dual", "bunny_small_n24"), (3, "dual", "bunny_small_n24"), (2, "primal-plain", "bunny_small_n24")])
def test_multiprocess_ranks_on_one_gpu(shm, tmp_path, world, mode, case):
    """The real multi-rank code path (rank-major z-slabs, halo send/recv, all-reduces, the gather of D^T Y in front of the whole-grid
    dual solve ("dual"), the all-to-all transposes of the distributed DCT ("dual-slabs", "primal-dct"), the slab-chained fast
    integration) with one process per rank.  RCCL refuses two ranks on one device, so its nine entry
    points are replaced by tests/native/rccl_mock.c (SHM_RCCL_LIB) -- everything above the transport is the product code."""
    so = _build_rccl_mock(tmp_path)
    extra = {}
    if mode.endswith("+overlap"):
        # z chunks of 2 planes: every slab (16 / 8 planes) has interior chunks, so the ghost planes of z travel on the second stream while the
        # interior chunks of the DIR sweep run, and the first / last chunk follow (the default chunking leaves < 3 chunks at 32^3: no split)
        extra["SHM_FUSED_ZC"] = "2"
        extra["SHM_HALO_OVERLAP"] = "1"
        mode = mode[:-len("+overlap")]
    phi, metas, covered = _run_ranks(tmp_path, so, world, case, mode, mode.replace("-", "") + case[-6:].replace("_", ""), extra)
    d = load_golden(case)
    for (_, _, _, shift) in metas:
        assert abs(shift - float(d["shift"])) < 1e-7
    assert covered == int(d["n"])
    assert np.abs(phi - d["phi"]).max() < (1e-9 if mode == "fast" else 1e-7)
    if mode.startswith("primal"):
        # against the SAME solver on one rank: the per-slab partial sums are added in rank order (deterministic, but a different order than
        # the one-slab sum), so the results agree to rounding amplified by the CG, not bit for bit
        s1 = make_solver(shm, d)
        s1.solve(tol=1e-10, scrub="_pc_" not in case, **MODES[mode])
        ref, _ = s1.get_phi()
        assert np.abs(phi - ref).max() < 1e-9


@pytest.mark.parametrize("world,fname,precision", [(4, "bunny_small.obj", 64), (4, "bunny.pc", 64), (4, "rocker.obj", 32)])   # (four ranks: the all-to-all blocks of two would not fit the mock's 8 MB mailboxes)
def test_multiprocess_auto_takes_the_slab_distributed_forms(shm, tmp_path, world, fname, precision):
    """Round 6: at the sizes BASELINE.json names (256^3 here) the DEFAULT multi-rank solve no longer gathers D^T Y and solves the whole grid on every rank: S and
    its inverse are replicated beside every rank's Step 1, K^+ runs on the z-slabs (two all-to-alls per application) and the dual system is solved on m-vectors
    every rank holds -- directly (bunny: cg_form 2) or by CG on the explicit S (rocker.obj, m = 9110: cg_form 2 or 3, whichever the rank's Step 1 hides).  Every
    rank must report that path and produce its planes of the single-rank answer."""
    so = _build_rccl_mock(tmp_path)
    case = "file:%s:4" % fname
    phi, metas, covered = _run_ranks(tmp_path, so, world, case, "auto", "auto%d%s" % (precision, fname[:4]),
                                     {"SHM_WORKER_PRECISION": str(precision), "SHM_WORKER_EXPECT_SOLVER": "3:2" if fname != "rocker.obj" else "3:-1"})
    pre = _preprocess(fname, 4.0)
    assert covered == pre["n"] == 256 and np.isfinite(phi).all()
    _, st1, phi1 = _gpu_phi(shm, pre, shm.SHM_F64 if precision == 64 else shm.SHM_F32, not fname.endswith(".pc"), tol=1e-10 if precision == 64 else 0.0)
    span = np.abs(phi1).max()
    err = np.abs(phi - phi1).max()
    print("\n%s 256^3 fp%d, %d ranks, AUTO: slab-distributed explicit-S solve against one rank: L_inf %.3e (max|phi| %.2f)" % (fname, precision, world, err, span))
    assert err < (1e-8 if precision == 64 else 3e-5) * max(1.0, span), (err, span)


@pytest.mark.parametrize("mode,plan", [("dual", 0), ("primal-dct", 0), ("dual", 1), ("primal-plain", 1)])
def test_multiprocess_ranks_culled_fp32(shm, oracle_c, tmp_path, mode, plan):
    """A culled fp32 workload through the rank path (configs[4] in small: SprayBottle.pc, 64^3, four ranks): Step 1 skips most source clusters per
    tile and the slabs differ in how many they keep; every rank must still produce its planes of the single-rank answer.  Against the fp64 C oracle
    at fp32 tolerance, and against the same fp32 solver on one rank much tighter."""
    import os
    so = _build_rccl_mock(tmp_path)
    world, case = (3 if plan else 4), "file:SprayBottle.pc:2"   # (three ranks with the weighted plan: 64 planes in multiples of 8 cannot be three equal slabs)
    phi, metas, covered = _run_ranks(tmp_path, so, world, case, mode, "spray%d" % plan + mode.replace("-", ""), {"SHM_WORKER_PRECISION": "32", "SHM_WORKER_SLAB_PLAN": str(plan)})
    pre = _preprocess("SprayBottle.pc", 2.0)
    n, S = pre["n"], pre["S"]
    assert n == 64 and covered == n and np.isfinite(phi).all()
    heights = [k1 - k0 for (k0, k1, _, _) in metas]
    if plan:   # the source-aware plan (shm_config.slab_plan = SHM_SLAB_PLAN_STEP1): unequal slabs, cut at multiples of 8 planes, the same on every rank
        assert len(set(heights)) > 1 and all(h % 8 == 0 and h >= 8 for h in heights), heights
        w = shm.step1_plane_weights(pre["pos"], pre["wnormal"], pre["lam"], n, pre["bbox_min"], pre["cell"], 32)
        assert [shm.plan_slab_weighted(n, world, r, w, 8) for r in range(world)] == [(k0, k1) for (k0, k1, _, _) in metas]
    else:
        assert heights == [16] * 4
    _, st1, phi1 = _gpu_phi(shm, pre, shm.SHM_F32, False, **MODES[mode])
    span = np.abs(phi1).max()
    assert np.abs(phi - phi1).max() < 3e-5 * span, (np.abs(phi - phi1).max(), span)
    ref = np.zeros(n ** 3)
    st = np.zeros(5)
    oracle_c.shmo_set_threads(min(64, os.cpu_count() or 1))
    rc = oracle_c.shmo_compute_distance(n, c_(pre["bbox_min"]), pre["cell"], S, c_(pre["pos"]).reshape(-1), c_(pre["wnormal"]).reshape(-1),
                                        c_(pre["area"]), pre["lam"], 0, 0, 1e-12, 100000, ref, st)
    oracle_c.shmo_set_threads(min(8, os.cpu_count() or 1))
    assert rc == 0 and np.isfinite(ref).all()
    assert np.abs(phi - ref).max() < 2e-4 * span, (np.abs(phi - ref).max(), span)


def test_multiprocess_noconv_still_returns_phi(shm, tmp_path, monkeypatch):
    """Two ranks, default (gathered dual) solver stopped after 4 iterations: every rank gets SHM_ERR_NOCONV *and* its planes of the
    unconverged phi plus filled statistics (the single-rank contract of include/shm_grid.h), not SHM_ERR_STATE from get_phi.
    (SHM_DUAL_NO_DIRECT: the iterative dual solver -- the direct one cannot run out of iterations.)"""
    monkeypatch.setenv("SHM_DUAL_NO_DIRECT", "1")
    import os
    import subprocess
    import sys
    from conftest import ROOT
    so = str(tmp_path / "librccl_mock.so")
    subprocess.check_call(["gcc", "-O2", "-shared", "-fPIC", os.path.join(ROOT, "tests", "native", "rccl_mock.c"), "-o", so, "-I/opt/rocm/include",
                           "-D__HIP_PLATFORM_AMD__", "-L/opt/rocm/lib", "-lamdhip64", "-lrt", "-lpthread"])
    world, case = 2, "bunny_small_n32"
    uid = ("/shmmock_%d_noconv" % os.getpid()).encode().ljust(128, b"\x00")
    env = dict(os.environ, SHM_RCCL_LIB=so, SHM_WORKER_MAX_ITERS="4")
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "multiproc_worker.py"), str(r), str(world), uid.hex(), case, "dual", str(tmp_path)],
                              env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(world)]
    outs = [p.communicate(timeout=300)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    d = load_golden(case)
    phi = np.concatenate([np.load(tmp_path / ("phi_%d.npy" % r)) for r in range(world)])
    err = np.abs(phi - d["phi"]).max()
    assert np.isfinite(phi).all() and 1e-7 < err < 1e-1, err          # a usable, visibly unconverged field
    s1 = make_solver(shm, d)
    s1.solve(tol=1e-10, max_iters=4, allow_noconv=True, solver="dual")
    ref, _ = s1.get_phi()
    assert np.abs(phi - ref).max() < 1e-9                              # the same four iterations as on one rank


def _device_count():
    try:
        import torch
        return torch.cuda.device_count()
    except Exception:
        return 0


@pytest.mark.parametrize("mode", ["dual", "dual-slabs", "primal-plain", "primal-dct", "fast", "primal-plain+overlap", "primal-dct+overlap"])
def test_multiprocess_ranks_real_rccl(shm, tmp_path, mode):
    """Two ranks on two GPUs through the REAL librccl (no test double): the grouped send/recv of the halo planes and of the gather of
    D^T Y, the all-to-alls of the distributed DCT, the all-reduces and the slab-chained fast integration.  Skipped on boxes with one GPU
    (the development pool has only those: until this test has run somewhere, the RCCL transport itself is verified only through the
    prototypes of <rccl/rccl.h> at build time and through a one-rank communicator)."""
    import os
    import subprocess
    import sys
    from conftest import ROOT
    if _device_count() < 2:
        pytest.skip("needs 2 GPUs")
    world = 2
    case = "bunny_small_fast_n32" if mode == "fast" else "bunny_small_n32"
    env = dict(os.environ, SHM_WORKER_DEVICE_PER_RANK="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("SHM_RCCL_LIB", None)
    if mode.endswith("+overlap"):   # the opt-in halo / compute overlap (second stream): 2-plane z chunks so that a 16-plane slab has interior chunks
        env.update(SHM_FUSED_ZC="2", SHM_HALO_OVERLAP="1")
        mode = mode[:-len("+overlap")]
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "multiproc_worker.py"), str(r), str(world), "file", case, mode, str(tmp_path)],
                              env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(world)]
    outs = [p.communicate(timeout=600)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    d = load_golden(case)
    parts = [np.load(tmp_path / ("phi_%d.npy" % r)) for r in range(world)]
    assert np.abs(np.concatenate(parts) - d["phi"]).max() < (1e-9 if mode == "fast" else 1e-7)


def test_bench_py_multi_rank_flow_on_one_gpu(tmp_path):
    """bench.py launched the way the driver launches it for N>1 (torch.distributed.run, one process per rank), with the ranks sharing
    the one GPU of the test box: gloo for the bootstrap collectives and the shared-memory double for the solver's RCCL calls."""
    import json
    import os
    import subprocess
    import sys
    from conftest import ROOT
    so = str(tmp_path / "librccl_mock.so")
    subprocess.check_call(["gcc", "-O2", "-shared", "-fPIC", os.path.join(ROOT, "tests", "native", "rccl_mock.c"), "-o", so, "-I/opt/rocm/include",
                           "-D__HIP_PLATFORM_AMD__", "-L/opt/rocm/lib", "-lamdhip64", "-lrt", "-lpthread"])
    env = dict(os.environ, SHM_RCCL_LIB=so, SHM_BENCH_ONE_DEVICE="1", SHM_BENCH_MULTI_HCOEF="1")   # (BASELINE's multi-GPU configurations at a 32^3 stand-in size)
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(29600 + os.getpid() % 300), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1",
                        "--workload", "bunny_small_64_f64", "--dist-backend", "gloo"], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout + p.stderr
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["grid"] == "64^3" and d["value"] > 0 and d["scaling"] == "strong"
    assert "roofline" in d and "cpu_baseline" not in d     # the CPU baseline is an N=1 leg
    # the legs of the split the north star names (z-slab stencil PCG: halo per sweep, all-reduce per dot product) ride in the same multi-rank run
    m = d["also_multi"]
    assert m["primal_pcg"]["value"] > 0 and m["primal_pcg"]["rel_residual"] < 1e-7 and m["primal_pcg"]["preconditioner"].startswith("dct")
    assert m["primal_plain_cg_200"]["cg_iters"] == 200 and "cg_fused_kernel<DIR>" in m["primal_plain_cg_200"]["kernels"]
    assert m["gathered_dual"]["value"] > 0 and m["gathered_dual"]["solver"] == 2     # the default of rounds 1-5 beside the round-6 one, on the same ranks
    # BASELINE.json configs[3] / configs[4] end to end on the same ranks (here at 32^3): value, phases and every rank's Step-1 pairs
    for wl, dtype in (("bunny_pc_512_f64_end_to_end", "f64"), ("spraybottle_pc_1024_f32_end_to_end", "f32")):
        leg = m[wl]
        assert "failed" not in leg, leg
        assert leg["value"] > 0 and leg["grid"] == "32^3" and leg["dtype"].startswith(dtype) and leg["phases_ms"]["ms_conv"] > 0
        assert [r["rank"] for r in leg["per_rank"]] == [0, 1] and all(r["pairs_fp64"] + r["pairs_fp32"] > 0 for r in leg["per_rank"])
    assert "weighted" in m["spraybottle_pc_1024_f32_end_to_end"]["partition"]
    # the legs run after the headline record is complete and under a watchdog: if they do not finish in time (here: at once) the line is printed ONCE without
    # them -- an extra can never cost the timed result --
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(29900 + os.getpid() % 90), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1",
                        "--workload", "bunny_small_64_f64", "--dist-backend", "gloo"], env=dict(env, SHM_BENCH_LEGS_TIMEOUT="0.001"), capture_output=True, text=True, timeout=600)
    assert p.returncode != 0, p.stdout + p.stderr          # ... but a hang is not a green run: the launcher sees a failing status
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d2 = json.loads(lines[0])
    assert d2["n_gpus"] == 2 and d2["value"] > 0 and "roofline" in d2 and "failed" in d2["also_multi"]


@pytest.mark.gpu
@pytest.mark.parametrize("case,scrub", [("bunny_small_n16", True), ("bunny_small_n32", True), ("bunny_pc_n32", False), ("bunny_small_n64", True)])
def test_explicit_schur_complement_is_A_Kplus_AT(case, scrub, tmp_path):
    """The dual solver's explicit S (image-sum Green's table + trilinear stencils, csrc/shm_schur.hip.h) against the operator it replaces, applied through
    the public entry points: column j of A K^+ A^T = gather(apply_preconditioner(scatter(e_j))).  Then the solve itself three ways (fresh processes: the
    knobs are read once): direct (S inverted beside Step 1), CG on the explicit S (SHM_DUAL_NO_DIRECT=1), CG with S applied through the grid
    (SHM_DUAL_NO_DENSE_S=1): same LU-golden phi; the two CGs take the same iterations, the direct solve one or two passes."""
    import os
    import subprocess
    import sys
    from conftest import GOLDEN, ROOT
    if not os.path.exists(os.path.join(GOLDEN, case + ".npz")):
        pytest.skip("fixture not generated")
    code = r"""
import sys, numpy as np
sys.path.insert(0, %r)
import shm_import
shm = shm_import.load()
d = np.load(%r)
s = shm.GridSolver()
n = int(d["n"])
s.set_problem(d["pos"], d["wnormal"], d["area"], float(d["lam"]), n, d["bbox_min"], float(d["cell"]))
out = {}
try:
    S = s.get_schur()
    nodes, coeffs = s.get_constraints()
    m = len(nodes)
    rng = np.random.default_rng(3)
    worst = 0.0
    for j in list(rng.integers(0, m, 6)) + [0, m - 1]:
        v = np.zeros(n ** 3)
        np.add.at(v, nodes[j], coeffs[j])
        z = s.apply_preconditioner(v)
        col = (coeffs * z[nodes]).sum(axis=1)
        worst = max(worst, float(np.abs(col - S[:, j]).max() / np.abs(S[:, j]).max()))
    out["S_rel_err"] = worst
    out["S_sym"] = float(np.abs(S - S.T).max())
except shm.ShmError as e:
    out["no_S"] = str(e)
st = s.solve(tol=1e-10, scrub=%r, solver="dual")
phi, _ = s.get_phi()
out["phi_err"] = float(np.abs(phi - d["phi"]).max())
out["iters"] = int(st.iters)
print(repr(out))
""" % (ROOT, os.path.join(GOLDEN, case + ".npz"), scrub)
    res = {}
    for name, knobs in (("direct", {}), ("cg_dense", {"SHM_DUAL_NO_DIRECT": "1"}), ("cg_sweeps", {"SHM_DUAL_NO_DENSE_S": "1"})):
        env = dict(os.environ, SHM_DUAL_DENSE_S_ALWAYS="1")   # (by default grids this small apply S through the grid: their Step 1 is too short to hide the assembly)
        for k in ("SHM_DUAL_NO_DENSE_S", "SHM_DUAL_NO_DIRECT"):
            env.pop(k, None)
        env.update(knobs)
        p = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, p.stdout + p.stderr
        res[name] = eval(p.stdout.strip().splitlines()[-1])
    d, a, b = res["direct"], res["cg_dense"], res["cg_sweeps"]
    assert "no_S" in b and "no_S" not in a and "no_S" not in d, res
    for r in (d, a):
        assert r["S_rel_err"] < 1e-10 and r["S_sym"] == 0.0, r          # the transforms are fp64: agreement to their rounding
    assert d["phi_err"] < 1e-7 and a["phi_err"] < 1e-7 and b["phi_err"] < 1e-7, res
    assert abs(a["iters"] - b["iters"]) <= 2, res                       # same operator: same CG
    assert d["iters"] <= 2, d                                           # the direct solve: one pass, at most one of refinement


@pytest.mark.gpu
@pytest.mark.parametrize("waves", ["16"])   # (round 6: the 4-wave shapes are A/B shapes, compiled only into -DSHM_AB_SHAPES builds)
def test_fused_sweeps_other_workgroup_shapes_match_lu_golden(waves):
    """The fused stencil-CG sweeps ship with 8 waves per workgroup where a grid row needs one or two waves and with 16 where it needs four or more
    (512^3 fp64, 1024^3 fp32) -- sizes the small fixtures never reach.  SHM_FUSED_WAVES forces the 16-wave kernels onto the fixtures
    (a fresh process: the knob is read once): same LU-golden phi from the plain and the DCT-preconditioned primal solvers, fp64 and fp32 transforms."""
    import os
    import subprocess
    import sys
    from conftest import GOLDEN, ROOT
    code = r"""
import sys, numpy as np
sys.path.insert(0, %r)
import shm_import
shm = shm_import.load()
out = {}
for case in ("bunny_small_n32", "bunny_small_n64", "bunny_pc_n32"):
    d = np.load(%r + "/" + case + ".npz")
    for mode, kw in (("plain", dict(solver="primal", precond="none")), ("dct", dict(solver="primal", precond="dct"))):
        s = shm.GridSolver()
        s.set_problem(d["pos"], d["wnormal"], d["area"], float(d["lam"]), int(d["n"]), d["bbox_min"], float(d["cell"]))
        st = s.solve(tol=1e-10, scrub=not case.startswith("bunny_pc"), **kw)
        phi, _ = s.get_phi()
        out[case + "/" + mode] = (float(np.abs(phi - d["phi"]).max()), int(st.iters), int(st.cg_form))
        s.close()
print(repr(out))
""" % (ROOT, GOLDEN)
    env = dict(os.environ, SHM_FUSED_WAVES=waves)
    p = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout + p.stderr
    res = eval(p.stdout.strip().splitlines()[-1])
    for k, (err, iters, form) in res.items():
        assert form in (1, 4) and err < 1e-7, (k, err, iters, form)   # (4: plain CG on one GPU -- x updated on half the grid in every iteration)

"""CPU tests (no GPU, no compute calls): the C-ABI library loads and exports every symbol of include/shm_grid.h,
fails loudly without a device, and the C++ host mirror's pre-processing reproduces the oracle's golden values."""
import ctypes
import os
import sys
import re

import numpy as np
import pytest

from conftest import ROOT, load_golden


def test_library_exports_every_declared_symbol(shm):
    lib = shm.load_library()
    header = open(os.path.join(ROOT, "include", "shm_grid.h")).read()
    declared = set(re.findall(r"\b(shm_(?:grid|comm|plan|step1)_\w+)\s*\(", header))
    from signed_heat_3d_amd.grid_abi import ABI_SYMBOLS
    assert declared == set(ABI_SYMBOLS), declared ^ set(ABI_SYMBOLS)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.shm_grid_abi_version() == 5


def test_no_cpu_fallback(shm):
    """Without a HIP device the product path must fail loudly (this container has no GPU; on the GPU box the test is
    vacuous and skipped)."""
    try:
        s = shm.GridSolver()
    except shm.ShmError as e:
        assert e.status == 2 and "no CPU fallback" in str(e)
    else:
        s.close()
        pytest.skip("a HIP device is present")


def test_stats_struct_layout_matches_header(shm):
    """ctypes mirror of shm_stats must have the same size as the C struct (checked through the C++ host library, which
    copies an shm_stats by value)."""
    header = open(os.path.join(ROOT, "include", "shm_grid.h")).read()
    body = header[header.index("typedef struct {\n    int32_t n, m;"):header.index("} shm_stats;")]
    fields = re.findall(r"\b(?:int32_t|int64_t|double)\s+([\w, ]+);", body)
    names = [n.strip() for f in fields for n in f.split(",")]
    assert names == [n for n, _ in shm.ShmStats._fields_]


def test_opts_struct_layout_matches_header(shm):
    """ctypes mirror of shm_opts: same fields, same order, same types as the C struct (ABI 5: dual_form, step1_budget at the end)."""
    import ctypes as C
    from signed_heat_3d_amd.grid_abi import _Opts
    header = open(os.path.join(ROOT, "include", "shm_grid.h")).read()
    body = header[header.index("typedef struct {\n    int32_t fast_integration;"):header.index("} shm_opts;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    decl = re.findall(r"\b(int32_t|int64_t|double)\s+(\w+);", body)
    ctype = {"int32_t": C.c_int32, "int64_t": C.c_int64, "double": C.c_double}
    assert [(n, ctype[t]) for t, n in decl] == list(_Opts._fields_)


@pytest.mark.parametrize("n", [1, 2, 3, 7, 8])
def test_slab_plan_is_a_partition(shm, n):
    for planes in (16, 24, 33, 512):
        if planes < n:
            continue
        ranges = [shm.plan_slab(planes, n, s) for s in range(n)]
        assert ranges[0][0] == 0 and ranges[-1][1] == planes
        for a, b in zip(ranges, ranges[1:]):
            assert a[1] == b[0]
        sizes = [b - a for a, b in ranges]
        assert max(sizes) - min(sizes) <= 1 and min(sizes) >= 1


@pytest.mark.parametrize("case,path,hc", [("bunny_small_n16", "data/bunny_small.obj", 0.0), ("bunny_small_n24", "data/bunny_small.obj", None),
                                          ("bunny_small_n32", "data/bunny_small.obj", 1.0), ("polygon_bear_n16", "data/polygon-bear.obj", 0.0)])
def test_host_mirror_preprocessing_matches_golden(shm, case, path, hc):
    """centroid / radius / meanEdgeLength / setFaceVectorAreas / barycenter of the C++ host layer vs the oracle."""
    from signed_heat_3d_amd.host_abi import HostSolver
    d = load_golden(case)
    if hc is None:
        hc = float(d["hCoef"])
    h = HostSolver(os.path.join(ROOT, path))
    r = h.preprocess(hCoef=hc)
    assert r["n"] == int(d["n"])
    assert np.abs(r["centroid"] - d["centroid"]).max() < 1e-14
    assert abs(r["radius"] - float(d["radius"])) < 1e-14
    assert abs(r["h"] - float(d["h_mesh"])) < 1e-14
    assert abs(r["lam"] - float(d["lam"])) < 1e-12
    assert np.abs(r["bbox_min"] - d["bbox_min"]).max() < 1e-14 and abs(r["cell"] - float(d["cell"])) < 1e-15
    scale = np.abs(d["pos"]).max()
    assert np.abs(r["pos"] - d["pos"]).max() < 1e-15 * scale
    assert np.abs(r["wnormal"] - d["wnormal"]).max() < 1e-13 * np.abs(d["wnormal"]).max()
    assert np.abs(r["area"] - d["area"]).max() < 1e-13 * np.abs(d["area"]).max()


def test_host_mirror_strips_unreferenced_vertices(shm):
    from signed_heat_3d_amd.host_abi import HostSolver
    g = load_golden("rocker_preproc")
    h = HostSolver(os.path.join(ROOT, "data", "rocker.obj"))
    assert h.counts() == (int(g["nV"]), int(g["nF"])) == (8742, 13819)
    r = h.preprocess(hCoef=2.0)
    assert np.abs(r["centroid"] - g["centroid"]).max() < 1e-13
    assert abs(r["radius"] - float(g["radius"])) < 1e-13 and abs(r["h"] - float(g["h_mesh"])) < 1e-13
    assert abs(r["area"].sum() - float(g["area_sum"])) < 1e-10
    assert np.abs(r["pos"][:64] - g["pos_head"]).max() < 1e-13 and np.abs(r["wnormal"][:64] - g["wnormal_head"]).max() < 1e-13


def test_host_mirror_point_cloud(shm):
    """.pc loader (src/main.cpp:196-225) and explicit area/h inputs (tufted-triangulation values in the demo)."""
    from signed_heat_3d_amd.host_abi import HostSolver
    d = load_golden("bunny_pc_n16")
    h = HostSolver(os.path.join(ROOT, "data", "bunny.pc"))
    assert h.counts() == (1430, 0)
    h.set_point_areas(d["area"], float(d["h_in"]))
    r = h.preprocess(hCoef=0.0)
    assert np.abs(r["pos"] - d["pos"]).max() < 1e-14 and np.abs(r["wnormal"] - d["wnormal"]).max() < 1e-14
    assert abs(r["lam"] - float(d["lam"])) < 1e-12 and np.abs(r["bbox_min"] - d["bbox_min"]).max() < 1e-14


def test_point_cloud_estimator_against_the_matching_meshes(shm):
    """Headless estimator of the per-point dual areas and h (tangent-plane local Delaunay triangulations + intrinsic Delaunay flips, SURVEY 8(f)
    rank 3).  bunny.pc and rocker.pc are exactly the (stripped) vertices of bunny_small.obj / rocker.obj.  Anchors: the meshes' barycentric dual
    areas (total, per-point correlation), and for h the mean edge length of the mesh surface after INTRINSIC DELAUNAY flips -- that, not the
    mean edge length of the mesh as modelled, is what geometry-central's tufted intrinsic Delaunay triangulation measures on these points
    (signed_heat_grid_solver.cpp:149-151); tools/delaunay_anchor.py computes it (flips the mesh itself): 0.091045 / 0.105880."""
    from signed_heat_3d_amd.host_abi import HostSolver
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import shm_oracle as o
    import delaunay_anchor as da
    from scipy.spatial import cKDTree
    for pc, obj, tol_sum, h_delaunay, tol_h, min_corr in (("bunny.pc", "bunny_small.obj", 0.02, 0.091045, 0.01, 0.75), ("rocker.pc", "rocker.obj", 0.03, 0.105880, 0.025, 0.70)):
        V, F = o.read_obj(os.path.join(ROOT, "data", obj))
        dual = np.zeros(len(V))
        for f in F:
            p = V[list(f)]
            a = 0.5 * np.linalg.norm(np.cross(p[1] - p[0], p[2] - p[0]))
            dual[list(f)] += a / 3.0
        if pc == "bunny.pc":   # the anchor itself, recomputed (rocker: 1900 flips in Python take a while; its value is pinned above)
            Ff, L, G, _ = da.build(V, [f for f in F if len(f) == 3])
            assert abs(da.flip_to_delaunay(Ff, L, G) - h_delaunay) < 1e-6
        r = HostSolver(os.path.join(ROOT, "data", pc)).preprocess()
        dist, idx = cKDTree(V).query(r["pos"])
        assert dist.max() < 1e-4
        ref = dual[idx]
        assert abs(r["area"].sum() - dual.sum()) < tol_sum * dual.sum(), (pc, r["area"].sum(), dual.sum())
        assert abs(r["h"] - h_delaunay) < tol_h * h_delaunay, (pc, r["h"], h_delaunay)
        assert abs(r["h"] - o.mean_edge_length(V, F)) < 0.05 * o.mean_edge_length(V, F)     # and within 5 % of the mesh as modelled
        assert np.corrcoef(r["area"], ref)[0, 1] > min_corr
        assert (r["area"] > 0).all()


def test_adapter_syntax_only_against_stubs(tmp_path):
    """TYPO GUARD, not parity evidence: host/adapter_geometrycentral.h (the file a maintainer of the reference drops in) has no compiler
    in this image because geometry-central / Polyscope are absent.  It is type-checked (g++ -fsyntax-only, nothing is linked or run)
    against minimal stub declarations of the names it touches (tests/native/adapter_stubs/) and against the real include/shm_grid.h,
    used the way src/main.cpp:52,88-100,290-292 uses the class."""
    import shutil
    import subprocess
    shutil.copy(os.path.join(ROOT, "signed-heat-3d_amd", "host", "adapter_geometrycentral.h"), tmp_path)   # away from the host mirror's own signed_heat_3d.h
    p = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-Wall", "-Wextra", "-I", str(tmp_path), "-I", os.path.join(ROOT, "tests", "native", "adapter_stubs"),
                        "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "native", "adapter_syntax_check.cpp")], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr


def test_cli_reports_missing_device_or_runs():
    import subprocess
    exe = os.path.join(ROOT, "signed-heat-3d_amd", "bin", "shm_grid_cli")
    p = subprocess.run([exe, "--help"], capture_output=True, text=True)
    assert p.returncode == 0 and "--h <hCoef>" in p.stdout and "--export <file>" in p.stdout
    p = subprocess.run([exe], capture_output=True, text=True)
    assert p.returncode != 0 and "Please specify a mesh file" in p.stderr


def test_fft_core_on_host(tmp_path):
    """The Stockham/DCT building blocks the device kernels use (csrc/shm_fft_core.h) are plain C++: run them on the
    host against naive O(n^2) transforms for every supported length."""
    import subprocess
    exe = str(tmp_path / "test_fft_core")
    for flags in ([], ["-DTEST_POW=true"], ["-DTEST_LC=4", "-DTEST_POW=true"]):
        subprocess.check_call(["g++", "-O2", "-std=c++17"] + flags + [os.path.join(ROOT, "tests", "native", "test_fft_core.cpp"), "-o", exe])
        out = subprocess.run([exe], capture_output=True, text=True)
        assert out.returncode == 0 and out.stdout.strip().endswith("OK"), (flags, out.stdout)


def test_kernel_register_schedules():
    """The compiler's resource report of the library build (csrc/Makefile keeps it beside the .so).  Two things have cost measured time silently
    and are pinned here: a register spill in a transform kernel (n = 512 packed layout: 9 % of the sweep), and the fp64 Step-1 kernel falling out
    of its 241-register schedule into a tighter, slower one on an unrelated source change (165 registers: 44.0 instead of 40.8 ms at 256^3)."""
    import subprocess
    path = os.path.join(ROOT, "signed-heat-3d_amd", "lib", "kernel_resources.txt")
    if not os.path.exists(path):
        pytest.skip("library was built without the resource report")
    res, cur = {}, None
    for line in open(path):
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = m.group(1)
            res[cur] = {}
        m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", line)
        if m and cur:
            res[cur][m.group(1)] = int(m.group(2))
    assert len(res) > 200
    names = list(res)
    dem = subprocess.run(["c++filt"] + names, capture_output=True, text=True).stdout.splitlines()
    by_name = {re.sub(r"\(.*", "", d): res[n] for n, d in zip(names, dem)}
    spilled = {k: v["VGPRs Spill"] for k, v in by_name.items() if v.get("VGPRs Spill", 0)}   # (scalar registers spill into vector lanes: harmless)
    # gj_step_kernel runs under a 128-register cap (it has to fit beside Step 1's waves): the compiler parks a few address registers and 12 operand registers
    # of the last product in scratch, each stored and reloaded ONCE per workgroup, outside every loop (checked in the ISA; a capped build with 74 registers
    # spilled inside the inversion loop measured 1.5x slower and is what this bound keeps out)
    step_spill = spilled.pop("shm::gj_step_kernel", 0)
    assert step_spill == 0, step_spill      # (round 5: 144 registers at three waves per SIMD -- the 128-register cap of round 4 spilled 21)
    assert not spilled, spilled
    conv = by_name["void shm::conv_normalize_kernel<double, 4>"]
    assert conv["VGPRs"] >= 200 and conv["Occupancy"] == 2, conv
    # the tiered fp64 Step 1 must leave room on every SIMD for a wave of the set-up kernels (two of its waves + one of theirs <= 512 registers, LDS likewise):
    # that is what lets the constraint set-up run WHILE Step 1 runs instead of in the gaps between its launches
    tier = by_name["void shm::conv_tiered_kernel<4, double, true>"]
    assert tier["VGPRs"] <= 184 and tier["LDS Size"] <= 52 * 1024, tier      # (two of its workgroups per CU + a set-up workgroup of <= 42 KB: 146 of 160 KB)
    tier32 = by_name["void shm::conv_tiered_kernel<4, float, false>"]        # round 5: Step 1 of the fp32 solve -- the same budget, so that its set-up runs beside it too
    assert tier32["VGPRs"] <= 184 and tier32["LDS Size"] <= 52 * 1024, tier32
    room = 512 - 2 * ((tier["VGPRs"] + 7) // 8 * 8)
    for k in ("void shm::dgemm_rm_kernel<1>", "shm::gj_pivot_block4_kernel", "shm::gj_panels_kernel", "void shm::gj_update_kernel<0>", "shm::schur_assemble_kernel", "shm::green_symbol_kernel",
              "shm::gj_step_kernel"):
        v = by_name[k]
        waves_per_simd = 4 if "gj_pivot_kernel<2>" in k else 1      # (1024-thread workgroup: four waves on every SIMD)
        # (round 4, measured: a 144-register build of gj_step_kernel does NOT run beside a Step 1 of 2 x 184 registers although 2 x 184 + 144 = 512; the 137 of
        # gj_pivot_block4_kernel do: 16 registers of slack.  Round 5: Step 1 holds 2 x 176 -- room 160 -- and the 144-register gj_step_kernel runs beside it, measured)
        assert waves_per_simd * (v["VGPRs"] + v.get("AGPRs", 0)) <= room - 16 and v["LDS Size"] <= 42 * 1024, (k, v, room)
    conv32 = by_name["void shm::conv_normalize_kernel<float, 8>"]   # two sources in flight, three waves per SIMD (measured best with the tile queue)
    assert conv32["VGPRs"] <= 168 and conv32["Occupancy"] >= 3, conv32
    for k, v in by_name.items():   # the shipped shape of the fused stencil-CG sweeps: two rows per lane, four waves per SIMD
        if re.match(r"void shm::cg_fused_kernel<(double, 2|float, 4), 2, ", k):
            assert v["Occupancy"] >= 4, (k, v)


def test_step1_hot_loops_keep_their_schedule():
    """Round 6: the far loop of the Step-1 kernel once came out of the compiler with 24 s_nop (trans-use hazards) -- 142 instructions where 116 were possible, +8 % on the fp32
    solve -- after unrelated code around it had changed.  The loops are now written so that their schedule does not depend on the scheduler finding the interleaving; this test
    holds them to it from the compiler's own assembly of the kernel (tools/step1_isa_check.py; device-only compile, no GPU, three seconds)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("step1_isa_check", os.path.join(ROOT, "tools", "step1_isa_check.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    r = mod.loops()
    assert set(r) == {"fp64 solve near loop", "fp64 solve far loop", "fp32 solve near loop", "fp32 solve far loop"}, r
    for name, v in r.items():
        assert v["s_nop"] <= 2, (name, v)
    assert r["fp64 solve near loop"]["instructions"] <= 108 and r["fp32 solve near loop"]["instructions"] <= 108, r     # one near source x 4 nodes (measured: 104)
    assert r["fp64 solve far loop"]["instructions"] <= 128, r      # four far sources x 4 nodes incl. the L1 sums of the a-posteriori test (measured: 123)
    assert r["fp32 solve far loop"]["instructions"] <= 120, r      # ... without them (measured: 115)


def test_weighted_slab_plan_is_a_partition_and_balances_its_own_weights(shm):
    """shm_step1_plane_weights + shm_plan_slab_weighted (pure host logic, include/shm_grid.h): contiguous cover of the planes, boundaries on the granule,
    at least one granule per slab, identical for every caller (the plan is derived independently on every rank), and better balanced than equal planes
    on the culled fp32 configuration it exists for (configs[4] in small: SprayBottle.pc at 256^3)."""
    from signed_heat_3d_amd.host_abi import HostSolver
    pre = HostSolver(os.path.join(ROOT, "data", "SprayBottle.pc")).preprocess(hCoef=4.0)
    n = pre["n"]
    w = shm.step1_plane_weights(pre["pos"], pre["wnormal"], pre["lam"], n, pre["bbox_min"], pre["cell"], 32)
    w2 = shm.step1_plane_weights(pre["pos"], pre["wnormal"], pre["lam"], n, pre["bbox_min"], pre["cell"], 32)
    assert w.shape == (n,) and (w > 0).all() and np.array_equal(w, w2)
    for P, g in ((2, 8), (4, 8), (8, 8), (8, 4), (5, 8)):
        plan = [shm.plan_slab_weighted(n, P, s, w, g) for s in range(P)]
        assert plan[0][0] == 0 and plan[-1][1] == n
        for (a0, a1), (b0, b1) in zip(plan[:-1], plan[1:]):
            assert a1 == b0
        for k0, k1 in plan:
            assert k1 - k0 >= g and k0 % g == 0
        eq = [shm.plan_slab(n, P, s) for s in range(P)]
        imb = lambda pl: max(w[a:b].sum() for a, b in pl) / (w.sum() / P)   # noqa: E731
        assert imb(plan) <= imb(eq) + 1e-12
        if P == 8 and g == 8:   # 32-plane slabs cut at multiples of 8 planes: a quarter of a slab per step
            assert imb(eq) > 1.15 and imb(plan) < imb(eq) - 0.08, (imb(eq), imb(plan))
        if P == 4 and g == 8:
            assert imb(plan) < 1.04, (imb(eq), imb(plan))
    # degenerate weights: the equal-plane plan (up to the granule)
    assert [shm.plan_slab_weighted(64, 4, s, np.ones(64), 4) for s in range(4)] == [(0, 16), (16, 32), (32, 48), (48, 64)]
    assert [shm.plan_slab_weighted(64, 4, s, np.zeros(64), 4) for s in range(4)] == [(0, 16), (16, 32), (32, 48), (48, 64)]


# ---- marching-cubes case table (device kernel: iso_mc_kernel, checked against tests/iso_ref.py in test_gpu_parity.py) ----------------------
def _mesh_is_closed_and_oriented(tris):
    from collections import Counter
    e = Counter()
    for a, b, c in tris:
        for u, v in ((a, b), (b, c), (c, a)):
            e[(u, v)] += 1
    return all(cnt == 1 and e.get((v, u), 0) == 1 for (u, v), cnt in e.items())


def test_marching_cubes_table_is_the_generators_and_watertight():
    """csrc/shm_mc_table.h is the output of tools/gen_mc_table.py (regenerated here and compared byte for byte), and the construction does what it says:
    on random sign patterns -- every one of the 256 cases occurs, ambiguous faces and all -- the surface of a field that is positive on the boundary is closed,
    every edge used exactly once in each direction (no cracks between cells, consistent orientation), and it encloses the inside nodes."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("gen_mc_table", os.path.join(ROOT, "tools", "gen_mc_table.py"))
    g = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(g)
    T = g.build_table()
    assert g.check_table(T) == 5 and sum(len(t) for t in T) == 820
    assert open(os.path.join(ROOT, "signed-heat-3d_amd", "csrc", "shm_mc_table.h")).read() == g.header(T)
    for c in range(256):   # complementary cases cut the same edges; a case and its mirror image in x have the same number of triangles
        assert {e for t in T[c] for e in t} == {e for t in T[255 - c] for e in t}
        mirrored = sum(((c >> q) & 1) << (q ^ 1) for q in range(8))
        assert len(T[c]) == len(T[mirrored])
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from iso_ref import marching_cubes
    rng = np.random.default_rng(5)
    seen = set()
    for n in (7, 12, 14):
        phi = rng.standard_normal((n, n, n))
        phi[0, :, :] = phi[-1, :, :] = phi[:, 0, :] = phi[:, -1, :] = phi[:, :, 0] = phi[:, :, -1] = 1.0 + rng.random()
        pts, tris = marching_cubes(phi.ravel(), n, np.zeros(3), 1.0, 0.0)
        assert len(tris) > 50 and _mesh_is_closed_and_oriented(tris)
        inside = phi < 0
        case = np.zeros((n - 1, n - 1, n - 1), dtype=np.int32)
        for q in range(8):
            case |= inside[(q >> 2) & 1:n - 1 + ((q >> 2) & 1), (q >> 1) & 1:n - 1 + ((q >> 1) & 1), (q & 1):n - 1 + (q & 1)].astype(np.int32) << q
        seen |= set(case.ravel().tolist())
        # normals towards increasing phi: the closed surface around the inside nodes has positive signed volume
        P = np.array([pts[k] for t in tris for k in t]).reshape(-1, 3, 3)
        vol = np.einsum("ij,ij->i", P[:, 0], np.cross(P[:, 1], P[:, 2])).sum() / 6.0
        assert vol > 0
    assert len(seen) >= 250, len(seen)


def test_marching_cubes_reference_on_an_analytic_sphere():
    """The python restatement the device kernel is held to, on phi = |x| - r: closed, oriented outwards, area and volume of the sphere to discretisation accuracy,
    every vertex on the sphere to O(h^2), and fewer triangles than marching tetrahedra on the same field."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from iso_ref import marching_cubes, marching_tets
    n, r = 24, 0.8
    h = 2.0 / (n - 1)
    ax = -1.0 + h * np.arange(n)
    Z, Y, X = np.meshgrid(ax, ax, ax, indexing="ij")
    phi = np.sqrt(X * X + Y * Y + Z * Z) - r
    pts, tris = marching_cubes(phi.ravel(), n, np.array([-1.0, -1.0, -1.0]), h, 0.0)
    assert _mesh_is_closed_and_oriented(tris)
    V = np.array(list(pts.values()))
    assert np.abs(np.linalg.norm(V, axis=1) - r).max() < 0.6 * h * h / r
    P = np.array([pts[k] for t in tris for k in t]).reshape(-1, 3, 3)
    vol = np.einsum("ij,ij->i", P[:, 0], np.cross(P[:, 1], P[:, 2])).sum() / 6.0
    area = 0.5 * np.linalg.norm(np.cross(P[:, 1] - P[:, 0], P[:, 2] - P[:, 0]), axis=1).sum()
    assert abs(vol - 4.0 / 3.0 * np.pi * r ** 3) < 0.02 * vol and abs(area - 4 * np.pi * r * r) < 0.02 * area
    nrm = np.cross(P[:, 1] - P[:, 0], P[:, 2] - P[:, 0])
    assert (np.einsum("ij,ij->i", nrm, P.mean(axis=1)) > 0).all()      # outwards = towards increasing phi
    _, tets = marching_tets(phi.ravel(), n, np.array([-1.0, -1.0, -1.0]), h, 0.0)
    assert len(tris) < 0.6 * len(tets)

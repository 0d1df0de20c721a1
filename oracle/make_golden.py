"""Generate the golden fixtures under tests/golden/ from the numpy/scipy oracle (literal KKT + SuperLU).

Run in the build container only (needs scipy and tens of seconds .. 20 min for --with-64):
    python oracle/make_golden.py            # 16^3 / 24^3 / 32^3 cases, ~1 min
    python oracle/make_golden.py --with-64  # additionally bunny_small at 64^3 (config C1): ~20 min, ~21 GB RSS

The fixtures are DATA (inputs + expected outputs); the reference's sources are not involved.
"""
import argparse
import math
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
import shm_oracle as o  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")


def mesh_dual_areas(V, faces):
    """Barycentric dual areas (area/3 per incident triangle) -- plausibility stand-in for
    geometry-central's tufted vertexDualAreas; an INPUT of the solver."""
    A = np.zeros(len(V))
    for f in faces:
        assert len(f) == 3
        a = 0.5 * np.linalg.norm(np.cross(V[f[1]] - V[f[0]], V[f[2]] - V[f[0]]))
        for v in f:
            A[v] += a / 3.0
    return A


def save_case(name, g, src, phi, info, *, extra=None, full=True):
    d = dict(n=np.int64(g.n), bbox_min=g.bbox_min, cell=np.float64(g.cell), lam=np.float64(src.lam),
             pos=src.pos, wnormal=src.wnormal, area=src.area, phi=phi, shift=np.float64(info["shift"]))
    if "m" in info:
        d["m"] = np.int64(info["m"])
    if full:
        d["Y"] = info["Y"]
        d["b"] = info["b"]
    if extra:
        d.update(extra)
    path = os.path.join(GOLD, name + ".npz")
    np.savez_compressed(path, **d)
    print("wrote", path, "%.1f KB" % (os.path.getsize(path) / 1024), {k: v for k, v in info.items() if k not in ("Y", "b")})


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--with-64", action="store_true")
    ap.add_argument("--only-64", action="store_true")
    args = ap.parse_args()
    os.makedirs(GOLD, exist_ok=True)

    V, F = o.read_obj(os.path.join(ROOT, "data", "bunny_small.obj"))

    if not args.only_64:
        # ---- mesh overload, triangle mesh ------------------------------------------------------
        for n in (16, 24, 32):
            hc = math.log2(n / 16.0)
            g, src, phi, info = o.compute_distance_mesh(V, F, hCoef=hc, detail=True)
            assert g.n == n, (g.n, n)
            nodes, coeffs = o.constraint_rows(g, src.pos)
            save_case("bunny_small_n%d" % n, g, src, phi, info,
                      extra=dict(hCoef=np.float64(hc), c_nodes=nodes, c_coeffs=coeffs,
                                 centroid=o.centroid(V), radius=np.float64(o.radius(V, o.centroid(V))),
                                 h_mesh=np.float64(o.mean_edge_length(V, F))))

        # ---- mesh overload, fast integration (BFS), signed_heat_grid_solver.cpp:224-275 ----------
        for n in (16, 32):
            g, src, phi, info = o.compute_distance_mesh(V, F, hCoef=math.log2(n / 16.0), fast=True, detail=True)
            save_case("bunny_small_fast_n%d" % n, g, src, phi, info, full=False)

        # ---- mesh overload, polygon mesh (shoelace areas on n-gons) ----------------------------
        Vp, Fp = o.read_obj(os.path.join(ROOT, "data", "polygon-bear.obj"))
        g, src, phi, info = o.compute_distance_mesh(Vp, Fp, hCoef=0.0, detail=True)
        nodes, coeffs = o.constraint_rows(g, src.pos)
        save_case("polygon_bear_n16", g, src, phi, info,
                  extra=dict(c_nodes=nodes, c_coeffs=coeffs, centroid=o.centroid(Vp),
                             radius=np.float64(o.radius(Vp, o.centroid(Vp))),
                             h_mesh=np.float64(o.mean_edge_length(Vp, Fp))))

        # ---- point overload (no divergence scrub; areas / h are inputs) -------------------------
        P, Nn = o.read_pc(os.path.join(ROOT, "data", "bunny.pc"))
        assert np.allclose(P, V, atol=2e-6), "bunny.pc positions are the bunny_small.obj vertices"
        areas = mesh_dual_areas(V, F)
        h = o.mean_edge_length(V, F)
        for n in (16, 32):
            g, src, phi, info = o.compute_distance_points(P, Nn, areas, h, hCoef=math.log2(n / 16.0), detail=True)
            nodes, coeffs = o.constraint_rows(g, src.pos)
            save_case("bunny_pc_n%d" % n, g, src, phi, info,
                      extra=dict(c_nodes=nodes, c_coeffs=coeffs, normals=Nn, h_in=np.float64(h)))

        # ---- pre-processing only: rocker.obj has 142 unreferenced vertices ----------
        Vr, Fr = o.read_obj(os.path.join(ROOT, "data", "rocker.obj"))
        cr = o.centroid(Vr)
        srcr = o.mesh_sources(Vr, Fr)
        gr = o.grid_setup(Vr, 2.0, 2.0)
        nodes, _ = o.constraint_rows(gr, srcr.pos)
        np.savez_compressed(os.path.join(GOLD, "rocker_preproc.npz"), nV=np.int64(len(Vr)), nF=np.int64(len(Fr)),
                            centroid=cr, radius=np.float64(o.radius(Vr, cr)),
                            h_mesh=np.float64(o.mean_edge_length(Vr, Fr)), lam=np.float64(srcr.lam),
                            area_sum=np.float64(srcr.area.sum()), pos_head=srcr.pos[:64], wnormal_head=srcr.wnormal[:64],
                            n=np.int64(gr.n), cell=np.float64(gr.cell), bbox_min=gr.bbox_min, m_n64=np.int64(len(nodes)))
        print("rocker: V", len(Vr), "F", len(Fr), "c", cr, "r", o.radius(Vr, cr), "lam", srcr.lam, "m@64", len(nodes))

    if args.with_64 or args.only_64:
        t = time.time()
        g, src, phi, info = o.compute_distance_mesh(V, F, hCoef=2.0, detail=True)
        assert g.n == 64
        # phi only (2 MB) + the Step-1/2 intermediates as float32-free float64 would be 8 MB: keep b, drop Y
        info_small = dict(info)
        save_case("bunny_small_n64", g, src, phi, info_small, full=False,
                  extra=dict(b=info["b"], hCoef=np.float64(2.0)))
        print("64^3 LU took %.1f s" % (time.time() - t))


if __name__ == "__main__":
    main()

"""Plain-python marching tetrahedra (Kuhn split) and marching cubes, the test-side restatements of csrc iso_kernel / iso_mc_kernel.  Test infrastructure."""
import numpy as np

KUHN = [(0, 1, 3, 7), (0, 1, 5, 7), (0, 2, 3, 7), (0, 2, 6, 7), (0, 4, 5, 7), (0, 4, 6, 7)]


def marching_tets(phi, n, bbox_min, cell, iso):
    """Returns (edge_points: dict edge_key -> xyz, triangles: list of 3 edge keys) ; edge key = (node_lo, node_hi)."""
    P = phi.reshape(n, n, n)  # [k,j,i]
    pts, tris = {}, []
    inside = P < iso
    cand = np.zeros((n - 1, n - 1, n - 1), dtype=np.int32)
    for q in range(8):
        cand += inside[(q >> 2) & 1:n - 1 + ((q >> 2) & 1), (q >> 1) & 1:n - 1 + ((q >> 1) & 1), (q & 1):n - 1 + (q & 1)]
    for k, j, i in zip(*np.nonzero((cand > 0) & (cand < 8))):
        node = lambda q: (i + (q & 1)) + (j + ((q >> 1) & 1)) * n + (k + ((q >> 2) & 1)) * n * n  # noqa: E731
        pos = lambda q: np.array([(i + (q & 1)) * cell + bbox_min[0], (j + ((q >> 1) & 1)) * cell + bbox_min[1],  # noqa: E731
                                  (k + ((q >> 2) & 1)) * cell + bbox_min[2]])
        val = lambda q: P[k + ((q >> 2) & 1), j + ((q >> 1) & 1), i + (q & 1)]  # noqa: E731
        for tet in KUHN:
            ins = [q for q in tet if val(q) < iso]
            outs = [q for q in tet if not val(q) < iso]
            if not ins or not outs:
                continue

            def ep(a, b):
                key = (min(node(a), node(b)), max(node(a), node(b)))
                t = (iso - val(a)) / (val(b) - val(a))
                pts[key] = pos(a) + t * (pos(b) - pos(a))
                return key
            if len(ins) == 1:
                loop = [ep(ins[0], o) for o in outs]
            elif len(ins) == 3:
                loop = [ep(a, outs[0]) for a in ins]
            else:
                loop = [ep(ins[0], outs[0]), ep(ins[0], outs[1]), ep(ins[1], outs[1]), ep(ins[1], outs[0])]
            tris.append(tuple(loop[:3]))
            if len(loop) == 4:
                tris.append((loop[0], loop[2], loop[3]))
    return pts, tris


def _mc_table():
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("gen_mc_table", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "gen_mc_table.py"))
    g = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(g)
    return g.build_table(), g.EDGES


def marching_cubes(phi, n, bbox_min, cell, iso, nz=None):
    """Same return convention as marching_tets.  The case table comes from the generator (tools/gen_mc_table.py), not from the C header: the header is checked
    against the generator in tests/test_abi_and_host.py, the kernel against this function.  nz: number of z planes when the field is not a cube."""
    table, edges = _mc_table()
    nz = n if nz is None else nz
    P = phi.reshape(nz, n, n)  # [k,j,i]
    pts, tris = {}, []
    inside = P < iso
    case = np.zeros((nz - 1, n - 1, n - 1), dtype=np.int32)
    for q in range(8):
        case |= inside[(q >> 2) & 1:nz - 1 + ((q >> 2) & 1), (q >> 1) & 1:n - 1 + ((q >> 1) & 1), (q & 1):n - 1 + (q & 1)].astype(np.int32) << q
    for k, j, i in zip(*np.nonzero((case > 0) & (case < 255))):
        node = lambda q: (i + (q & 1)) + (j + ((q >> 1) & 1)) * n + (k + ((q >> 2) & 1)) * n * n  # noqa: E731
        pos = lambda q: np.array([(i + (q & 1)) * cell + bbox_min[0], (j + ((q >> 1) & 1)) * cell + bbox_min[1],  # noqa: E731
                                  (k + ((q >> 2) & 1)) * cell + bbox_min[2]])
        val = lambda q: P[k + ((q >> 2) & 1), j + ((q >> 1) & 1), i + (q & 1)]  # noqa: E731
        for tri in table[case[k, j, i]]:
            keys = []
            for e in tri:
                a, b = edges[e]
                key = (node(a), node(b))
                t = (iso - val(a)) / (val(b) - val(a))
                pts[key] = pos(a) + t * (pos(b) - pos(a))
                keys.append(key)
            tris.append(tuple(keys))
    return pts, tris

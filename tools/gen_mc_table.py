"""Generator of the marching-cubes case table of csrc/shm_mc_table.h (the demo contours the grid phi with Polyscope's marching cubes,
/root/reference/src/main.cpp:121-124 -- a third-party submodule that is empty here, so the table is derived, not transcribed).

Construction (watertight by construction, unlike the classic 15-case table with complement symmetry):
  * corner q of a cell sits at (q & 1, q >> 1 & 1, q >> 2 & 1); bit q of the case index is set when that corner is INSIDE (phi < isovalue);
  * a cube edge whose end points differ carries one surface vertex;
  * on every cube face the surface vertices are joined by segments that depend on that face's four corner flags only: two vertices -> one segment; four
    vertices (the ambiguous face: two diagonal corners inside) -> two segments, each cutting off ONE INSIDE corner.  Two cells sharing a face see the same
    four flags, so they draw the same segments on it: the triangulated cells fit together without cracks or holes;
  * every surface vertex lies on two faces, so the segments close into loops; a loop of L vertices becomes L - 2 triangles, none of whose interior edges
    lies in a face of the cube (see triangulate());
  * segments are directed so that, seen from outside the cell, the inside corner lies to the RIGHT of the direction of travel, which makes every loop run
    counter-clockwise around the normal that points towards increasing phi (checked for every loop of every case): the orientation of iso_kernel's
    marching tetrahedra.
Output: per case the number of triangles and 3 edge ids per triangle; edge e joins corners kMcEdge[e][0] < kMcEdge[e][1].

    python tools/gen_mc_table.py            # prints the header to stdout
    python tools/gen_mc_table.py --write    # rewrites signed-heat-3d_amd/csrc/shm_mc_table.h
"""
import os
import sys

import numpy as np

CORNER = [np.array([q & 1, (q >> 1) & 1, (q >> 2) & 1], dtype=float) for q in range(8)]
EDGES = [(a, b) for a in range(8) for b in range(a + 1, 8) if bin(a ^ b).count("1") == 1]   # 12 edges, lexicographic: (0,1) (0,2) (0,4) (1,3) ...
EDGE_ID = {e: i for i, e in enumerate(EDGES)}
# the six faces: axis, side -> corners in cyclic order, outward normal
FACES = []
for axis in range(3):
    for side in (0, 1):
        u, v = [a for a in range(3) if a != axis]
        cyc = []
        for (cu, cv) in ((0, 0), (1, 0), (1, 1), (0, 1)):
            q = (side << axis) | (cu << u) | (cv << v)
            cyc.append(q)
        nrm = np.zeros(3)
        nrm[axis] = 1.0 if side else -1.0
        FACES.append((cyc, nrm))


def mid(e):
    a, b = EDGES[e]
    return 0.5 * (CORNER[a] + CORNER[b])


def edge_of(a, b):
    return EDGE_ID[(min(a, b), max(a, b))]


def face_segments(case, cyc, nrm):
    """Directed segments (e_from, e_to) on one face."""
    ins = [(case >> q) & 1 for q in cyc]
    cut = [k for k in range(4) if ins[k] != ins[(k + 1) % 4]]   # face edge k joins cyc[k], cyc[k+1]
    segs = []
    if len(cut) == 2:
        pairs = [(cut[0], cut[1])]
        # an inside corner next to the segment: the inside end point of the first cut edge
        anchors = [cyc[cut[0]] if ins[cut[0]] else cyc[(cut[0] + 1) % 4]]
    elif len(cut) == 4:
        pairs, anchors = [], []
        for k in range(4):   # every inside corner is cut off on its own: its two face edges are k-1 and k
            if ins[k]:
                pairs.append(((k - 1) % 4, k))
                anchors.append(cyc[k])
    else:
        return segs
    for (ka, kb), anchor in zip(pairs, anchors):
        ea, eb = edge_of(cyc[ka], cyc[(ka + 1) % 4]), edge_of(cyc[kb], cyc[(kb + 1) % 4])
        pa, pb = mid(ea), mid(eb)
        # seen from outside the cell, the inside corner to the RIGHT of the direction of travel:  nrm . (d x (s - pa)) < 0
        s = float(np.dot(nrm, np.cross(pb - pa, CORNER[anchor] - pa)))
        assert abs(s) > 1e-9
        segs.append((ea, eb) if s < 0 else (eb, ea))
    return segs


def case_loops(case):
    nxt = {}
    for cyc, nrm in FACES:
        for a, b in face_segments(case, cyc, nrm):
            assert a not in nxt, (case, a)
            nxt[a] = b
    loops, seen = [], set()
    for start in sorted(nxt):
        if start in seen:
            continue
        loop, e = [], start
        while e not in seen:
            seen.add(e)
            loop.append(e)
            e = nxt[e]
        assert e == start
        loops.append(loop)
    return loops


def same_face(ea, eb):
    """Do the two cube edges lie in one face of the cube?"""
    qs = set(EDGES[ea]) | set(EDGES[eb])
    return any(qs <= set(cyc) for cyc, _ in FACES)


def triangulations(idx):
    """All triangulations of the polygon idx[0..L-1] (lists of index triples, orientation kept)."""
    if len(idx) < 3:
        return [[]]
    if len(idx) == 3:
        return [[tuple(idx)]]
    out = []
    for k in range(1, len(idx) - 1):   # the triangle on the edge idx[0] -> idx[-1]... here: (idx[0], idx[k], idx[-1])
        for left in triangulations(idx[:k + 1]):
            for right in triangulations(idx[k:]):
                out.append(left + [(idx[0], idx[k], idx[-1])] + right)
    return out


def triangulate(loop):
    """Triangles of one loop.  A diagonal whose two surface vertices lie in the same cube face would lie IN that face, where the neighbouring cell may draw the
    same diagonal: four triangles on one edge (closed, but not a manifold).  Of all triangulations of the polygon take the first, in enumeration order, with no
    such diagonal (one exists for every loop of every case: asserted)."""
    best, best_bad = None, None
    for tri in triangulations(list(range(len(loop)))):
        bad = 0
        for t in tri:
            for a, b in ((t[0], t[1]), (t[1], t[2]), (t[2], t[0])):
                if (b - a) % len(loop) not in (1, len(loop) - 1) and same_face(loop[a], loop[b]):
                    bad += 1
        if best is None or bad < best_bad:
            best, best_bad = tri, bad
    assert best_bad == 0, (loop, best_bad)
    return [(loop[a], loop[b], loop[c]) for a, b, c in best]


def build_table():
    table = []
    for case in range(256):
        tris = []
        for loop in case_loops(case):
            assert len(loop) >= 3
            tris += triangulate(loop)
        table.append(tris)
    return table


def check_table(table):
    """Every cut edge is used; every loop is oriented towards the outside (Newell normal against the inside -> outside direction of its own cut edges)."""
    for case, tris in enumerate(table):
        cut = {e for e, (a, b) in enumerate(EDGES) if ((case >> a) & 1) != ((case >> b) & 1)}
        used = {e for t in tris for e in t}
        assert used == cut, (case, used, cut)
        for loop in case_loops(case):
            pts = [mid(e) for e in loop]
            newell = sum(np.cross(pts[k], pts[(k + 1) % len(pts)]) for k in range(len(pts)))
            d = np.zeros(3)
            for e in loop:
                a, b = EDGES[e]
                d += (CORNER[b] - CORNER[a]) * (1.0 if (case >> a) & 1 else -1.0)   # inside -> outside
            assert float(np.dot(newell, d)) > 1e-9, (case, loop)
    # directed edges of the triangles of a case: interior fan diagonals cancel, the rest are the face segments -- each once
    return max(len(t) for t in table)


def header(table):
    mx = max(len(t) for t in table)
    out = []
    out.append("// GENERATED by tools/gen_mc_table.py -- do not edit (tests/test_abi_and_host.py regenerates and compares).")
    out.append("// Marching-cubes case table: bit q of the case = corner q (at (q & 1, q >> 1 & 1, q >> 2 & 1)) is inside (phi < isovalue).  The ambiguous faces are")
    out.append("// resolved by the face's own four flags (every inside corner cut off on its own), so neighbouring cells agree on every shared face; triangles are")
    out.append("// oriented towards increasing phi.  See the generator for the construction.")
    out.append("#pragma once")
    out.append("namespace shm {")
    out.append("constexpr int kMcMaxTris = %d;" % mx)
    out.append("__device__ __constant__ unsigned char kMcEdge[12][2] = {%s};" % ", ".join("{%d, %d}" % e for e in EDGES))
    out.append("__device__ __constant__ unsigned char kMcCount[256] = {")
    for r in range(0, 256, 32):
        out.append("    " + ", ".join(str(len(table[c])) for c in range(r, r + 32)) + ",")
    out.append("};")
    out.append("__device__ __constant__ unsigned char kMcTris[256][3 * kMcMaxTris] = {")
    for c in range(256):
        flat = [e for t in table[c] for e in t]
        flat += [255] * (3 * mx - len(flat))
        out.append("    {%s}," % ", ".join(str(x) for x in flat))
    out.append("};")
    out.append("}  // namespace shm")
    return "\n".join(out) + "\n"


if __name__ == "__main__":
    T = build_table()
    mx = check_table(T)
    text = header(T)
    if "--write" in sys.argv:
        p = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "signed-heat-3d_amd", "csrc", "shm_mc_table.h")
        open(p, "w").write(text)
        print("wrote", p, "max triangles per cell", mx, "total", sum(len(t) for t in T))
    else:
        sys.stdout.write(text)

#include "signed_heat_3d.h"

#include <algorithm>
#include <cstdint>
#include <unordered_set>

namespace shm_host {

Vector3 centroid(const VertexPositionGeometry& geometry) {
    Vector3 c;
    for (const Vector3& p : geometry.vertexPositions) c += p;
    c /= (double)geometry.mesh.nVertices();
    return c;
}

Vector3 centroid(const PointPositionNormalGeometry& pointGeom) {
    Vector3 c;
    for (const Vector3& p : pointGeom.positions) c += p;
    c /= (double)pointGeom.positions.size();
    return c;
}

static double max_dist(const std::vector<Vector3>& P, const Vector3& c) {
    double r = 0;
    for (const Vector3& p : P) r = std::max(r, (c - p).norm());
    return r;
}
double radius(const VertexPositionGeometry& geometry, const Vector3& c) { return max_dist(geometry.vertexPositions, c); }
double radius(const PointPositionNormalGeometry& pointGeom, const Vector3& c) { return max_dist(pointGeom.positions, c); }

double yukawaPotential(const Vector3& x, const Vector3& y, const double& lambda) {
    const double r = (x - y).norm();
    return std::exp(-lambda * r) / r;
}

// Mean length over the UNIQUE undirected edges, visited in first-appearance order while walking the faces.
double meanEdgeLength(const VertexPositionGeometry& geometry) {
    std::unordered_set<uint64_t> seen;
    const uint64_t nv = geometry.mesh.nVertices();
    double h = 0;
    size_t count = 0;
    for (const auto& f : geometry.mesh.faces) {
        const size_t d = f.size();
        for (size_t a = 0; a < d; a++) {
            size_t u = f[a], v = f[(a + 1) % d];
            if (u > v) std::swap(u, v);
            if (!seen.insert((uint64_t)u * nv + v).second) continue;
            h += (geometry.vertexPositions[u] - geometry.vertexPositions[v]).norm();
            count++;
        }
    }
    return h / (double)count;
}

// Shoelace vector area for arbitrary polygons (the triangular shortcut of the reference is overwritten, SURVEY trap #3).
void setFaceVectorAreas(const VertexPositionGeometry& geometry, std::vector<double>& areas, std::vector<Vector3>& normals) {
    const size_t F = geometry.mesh.nFaces();
    areas.assign(F, 0.);
    normals.assign(F, Vector3());
    for (size_t fi = 0; fi < F; fi++) {
        const auto& f = geometry.mesh.faces[fi];
        Vector3 N;
        for (size_t a = 0; a < f.size(); a++) N += cross(geometry.vertexPositions[f[a]], geometry.vertexPositions[f[(a + 1) % f.size()]]);
        N = N * 0.5;
        areas[fi] = N.norm();
        normals[fi] = N / areas[fi];
    }
}

Vector3 barycenter(const VertexPositionGeometry& geometry, size_t fi) {
    Vector3 c;
    const auto& f = geometry.mesh.faces[fi];
    for (size_t v : f) c += geometry.vertexPositions[v];
    c /= (double)f.size();
    return c;
}

// k-NN disk estimate: r_k = distance to the k-th neighbour; a disk of radius r_k holds ~k+1 samples, so each
// sample owns pi r_k^2 / (k+1).  h = mean distance to the 6 nearest neighbours (a triangle mesh has ~6 edges per
// vertex).  O(P^2) brute force on the host: P <= 52 290 in the shipped data (2.7e9 distance evaluations).
void estimatePointAreas(PointPositionNormalGeometry& g, int k) {
    const size_t P = g.positions.size();
    g.dualAreas.assign(P, 0.);
    double hsum = 0;
    size_t hcount = 0;
    const int kk = std::max(k, 6);
    std::vector<double> best(kk);
#pragma omp parallel for firstprivate(best) reduction(+ : hsum, hcount) schedule(dynamic, 64)
    for (size_t a = 0; a < P; a++) {
        std::fill(best.begin(), best.end(), 1e300);
        for (size_t b = 0; b < P; b++) {
            if (a == b) continue;
            const Vector3 d = g.positions[a] - g.positions[b];
            const double d2 = d.x * d.x + d.y * d.y + d.z * d.z;
            if (d2 >= best[kk - 1]) continue;
            int pos = kk - 1;
            while (pos > 0 && best[pos - 1] > d2) {
                best[pos] = best[pos - 1];
                pos--;
            }
            best[pos] = d2;
        }
        const int kuse = (int)std::min<size_t>(k, P - 1);
        g.dualAreas[a] = M_PI * best[kuse - 1] / (double)(kuse + 1);
        const int ne = (int)std::min<size_t>(6, P - 1);
        for (int e = 0; e < ne; e++) hsum += std::sqrt(best[e]);
        hcount += ne;
    }
    g.meanEdgeLength = hsum / (double)hcount;
}

}  // namespace shm_host

// Flat C wrapper around the C++ host layer so that Python tests / bench.py can drive the same code path the
// headless CLI uses (ctypes cannot call C++ methods).  Not part of the drop-in boundary.
#include <chrono>
#include <cstring>
#include <string>

#include "mesh_io.h"
#include "signed_heat_grid_solver.h"

using namespace shm_host;

namespace {
thread_local std::string g_err;
struct Host {
    SignedHeatGridSolver solver;
    VertexPositionGeometry mesh;
    PointPositionNormalGeometry cloud;
    bool is_cloud = false;
    explicit Host(const GridBackendOptions& b) : solver(b) {}
};
template <typename F> int guard(F&& f) {
    try {
        f();
        g_err.clear();
        return 0;
    } catch (const std::exception& e) {
        g_err = e.what();
        return 1;
    }
}
}  // namespace

extern "C" {

const char* shmh_last_error(void) { return g_err.c_str(); }

void* shmh_new(int device, int precision, double tol, int max_iters, int local_slabs, int verbose) {
    GridBackendOptions b;
    b.device = device;
    b.precision = precision;
    b.tol = tol;
    b.maxIters = max_iters;
    b.localSlabs = local_slabs;
    Host* h = new Host(b);
    h->solver.VERBOSE = verbose != 0;
    return h;
}
void shmh_delete(void* h) { delete (Host*)h; }

int shmh_load(void* hv, const char* path) {
    Host* h = (Host*)hv;
    return guard([&] {
        const std::string p(path);
        const std::string ext = p.substr(p.find_last_of(".") + 1);
        h->is_cloud = (ext == "pc");  // src/main.cpp:267-268
        if (h->is_cloud) h->cloud = readPointCloud(p);
        else h->mesh = readSurfaceMesh(p);
    });
}

// counts: [0]=vertices/points [1]=faces
void shmh_counts(void* hv, int64_t* counts) {
    Host* h = (Host*)hv;
    counts[0] = h->is_cloud ? (int64_t)h->cloud.positions.size() : (int64_t)h->mesh.vertexPositions.size();
    counts[1] = h->is_cloud ? 0 : (int64_t)h->mesh.mesh.nFaces();
}

int shmh_set_point_areas(void* hv, const double* areas, double h_len) {
    Host* h = (Host*)hv;
    return guard([&] {
        h->cloud.dualAreas.assign(areas, areas + h->cloud.positions.size());
        h->cloud.meanEdgeLength = h_len;
    });
}

// Host pre-processing only (no GPU): centroid[3], radius, h, lambda, n, bbox_min[3], cell -> out[11];
// pos/wnormal (3S) and area (S) may be NULL.  S is returned through *S_out.
int shmh_preprocess(void* hv, double tCoef, double hCoef, double scale, double* out, double* pos, double* wn, double* area, int64_t* S_out) {
    Host* h = (Host*)hv;
    return guard([&] {
        Vector3 c;
        double r, hh;
        size_t S;
        if (h->is_cloud) {
            c = centroid(h->cloud);
            r = radius(h->cloud, c);
            if (h->cloud.dualAreas.size() != h->cloud.positions.size() || !(h->cloud.meanEdgeLength > 0.)) estimatePointAreas(h->cloud);
            hh = h->cloud.meanEdgeLength;
            S = h->cloud.positions.size();
        } else {
            c = centroid(h->mesh);
            r = radius(h->mesh, c);
            hh = meanEdgeLength(h->mesh);
            S = h->mesh.mesh.nFaces();
        }
        const double s = r * scale;
        const size_t n = (size_t)(2 * std::pow(2, hCoef + 3));
        for (int a = 0; a < 3; a++) out[a] = c[a];
        out[3] = r;
        out[4] = hh;
        out[5] = std::sqrt(1. / (tCoef * hh * hh));
        out[6] = (double)n;
        for (int a = 0; a < 3; a++) out[7 + a] = c[a] - s;
        out[10] = 2. * s / (n - 1);
        *S_out = (int64_t)S;
        if (!pos) return;
        if (h->is_cloud) {
            for (size_t p = 0; p < S; p++) {
                const Vector3 w = h->cloud.normals[p] * h->cloud.dualAreas[p];
                for (int a = 0; a < 3; a++) {
                    pos[3 * p + a] = h->cloud.positions[p][a];
                    wn[3 * p + a] = w[a];
                }
                area[p] = h->cloud.dualAreas[p];
            }
        } else {
            std::vector<double> areas;
            std::vector<Vector3> normals;
            setFaceVectorAreas(h->mesh, areas, normals);
            for (size_t f = 0; f < S; f++) {
                const Vector3 b = barycenter(h->mesh, f), w = normals[f] * areas[f];
                for (int a = 0; a < 3; a++) {
                    pos[3 * f + a] = b[a];
                    wn[3 * f + a] = w[a];
                }
                area[f] = areas[f];
            }
        }
    });
}

// Grid block the solver object currently holds (the reference keeps it across calls with rebuild=false, signed_heat_grid_solver.cpp:8):
// out[0] = nodes per side (0 before the first build), out[1..3] = bboxMin, out[4..6] = bboxMax, out[7] = cellSize.
void shmh_grid_info(void* hv, double* out) {
    Host* h = (Host*)hv;
    out[0] = (double)h->solver.gridSize();
    const Vector3 lo = h->solver.gridMin(), hi = h->solver.gridMax();
    for (int a = 0; a < 3; a++) {
        out[1 + a] = lo[a];
        out[4 + a] = hi[a];
    }
    out[7] = h->solver.gridCell();
}

// computeDistance through the C++ class; phi_out holds max(n_requested, n_current)^3 doubles (with rebuild=false a mesh solve keeps the
// grid of the previous call, so the result has shmh_grid_info()[0]^3 entries whatever hCoef says).
int shmh_compute_distance(void* hv, double tCoef, double hCoef, double scale, int rebuild, int fast, double* phi_out, shm_stats* stats) {
    Host* h = (Host*)hv;
    return guard([&] {
        SignedHeat3DOptions o;
        o.tCoef = tCoef;
        o.hCoef = hCoef;
        o.scale = scale;
        o.rebuild = rebuild != 0;
        o.fastIntegration = fast != 0;
        VectorXd phi = h->is_cloud ? h->solver.computeDistance(h->cloud, o) : h->solver.computeDistance(h->mesh, o);
        std::memcpy(phi_out, phi.data(), phi.size() * sizeof(double));
        if (stats) *stats = h->solver.lastStats();
    });
}

// Wall time of the C++ drop-in call alone -- computeDistance() returning its VectorXd, as the reference's main.cpp:90-91 consumes it --
// without this flat wrapper's extra copy into a caller buffer (tools/pcie_inclusive.py).
int shmh_time_compute_distance(void* hv, double tCoef, double hCoef, double scale, int rebuild, int fast, double* seconds_out, shm_stats* stats) {
    Host* h = (Host*)hv;
    return guard([&] {
        SignedHeat3DOptions o;
        o.tCoef = tCoef;
        o.hCoef = hCoef;
        o.scale = scale;
        o.rebuild = rebuild != 0;
        o.fastIntegration = fast != 0;
        const auto t0 = std::chrono::steady_clock::now();
        VectorXd phi = h->is_cloud ? h->solver.computeDistance(h->cloud, o) : h->solver.computeDistance(h->mesh, o);
        const auto t1 = std::chrono::steady_clock::now();
        if (seconds_out) *seconds_out = std::chrono::duration<double>(t1 - t0).count() + 0. * phi[0];
        if (stats) *stats = h->solver.lastStats();
    });
}

}  // extern "C"

#include "signed_heat_grid_solver.h"

#include <chrono>
#include <cmath>
#include <iostream>
#include <stdexcept>
#include <string>

namespace shm_host {

SignedHeatGridSolver::SignedHeatGridSolver() {}
SignedHeatGridSolver::SignedHeatGridSolver(const GridBackendOptions& b) : backend(b) {}
SignedHeatGridSolver::~SignedHeatGridSolver() {
    if (handle) shm_grid_destroy(handle);
}

// The reference reports failures as C++ exceptions (thrown by geometry-central / Eigen); so does the mirror.
void SignedHeatGridSolver::ensureHandle() {
    if (handle) return;
    shm_config cfg{};
    cfg.device = backend.device;
    cfg.precision = backend.precision == 32 ? SHM_F32 : SHM_F64;
    cfg.local_slabs = backend.localSlabs;
    cfg.rank = 0;
    cfg.world = 1;
    cfg.verbose = 0;
    const shm_status rc = shm_grid_create(&cfg, &handle);
    if (rc != SHM_OK) throw std::runtime_error(std::string("shm_grid_create: ") + shm_grid_last_error(nullptr));
}

// Grid block, signed_heat_grid_solver.cpp:13-26 / :124-137.
void SignedHeatGridSolver::buildGrid(const Vector3& c, double r, const SignedHeat3DOptions& options) {
    if (VERBOSE) std::cerr << "Building grid..." << std::endl;
    const auto t1 = std::chrono::high_resolution_clock::now();
    const double s = r * options.scale;
    bboxMin = Vector3{-s, -s, -s} + c;
    bboxMax = Vector3{s, s, s} + c;
    nx = (size_t)(2 * std::pow(2, options.hCoef + 3));
    ny = nx;
    nz = nx;
    cellSize = 2. * s / (nx - 1);
    // No Laplacian is assembled or factorised: the device operator is matrix-free.
    const auto t2 = std::chrono::high_resolution_clock::now();
    if (VERBOSE) std::cerr << "Pre-compute time (s): " << std::chrono::duration<double>(t2 - t1).count() << std::endl;
}

VectorXd SignedHeatGridSolver::solveOnDevice(bool scrub, const SignedHeat3DOptions& options) {
    ensureHandle();
    shm_sources src{};
    src.S = (int64_t)srcArea.size();
    src.pos = srcPos.data();
    src.wnormal = srcWn.data();
    src.area = srcArea.data();
    src.lambda = lambda;
    shm_grid grid{};
    grid.n = (int32_t)nx;
    for (int a = 0; a < 3; a++) grid.bbox_min[a] = bboxMin[a];
    grid.cell = cellSize;
    shm_opts opts{};
    opts.fast_integration = options.fastIntegration ? 1 : 0;
    opts.scrub_nonfinite = scrub ? 1 : 0;
    opts.tol = backend.tol;
    opts.max_iters = backend.maxIters;
    opts.step1_arith = backend.exactStep1 ? SHM_STEP1_EXACT_F64 : SHM_STEP1_AUTO;
    VectorXd phi(nx * ny * nz);
    if (VERBOSE) std::cerr << "Steps 1 & 2..." << std::endl;
    const shm_status rc = shm_grid_compute_distance(handle, &src, &grid, &opts, phi.data(), &stats);
    if (rc != SHM_OK) throw std::runtime_error(std::string("shm_grid_compute_distance: ") + shm_grid_last_error(handle));
    if (VERBOSE) {
        std::cerr << "\tCompleted." << std::endl << "Step 3..." << std::endl << "\tCompleted." << std::endl;
        std::cerr << "\t[gfx950] conv " << stats.ms_conv << " ms, div " << stats.ms_div << " ms, setup " << stats.ms_setup << " ms, pcg "
                  << stats.ms_pcg << " ms (" << stats.iters << " its, rel.res " << stats.rel_residual << "), m=" << stats.m << std::endl;
    }
    return phi;
}

void SignedHeatGridSolver::isosurface(double isoval, std::vector<Vector3>& vertices, std::vector<std::array<size_t, 3>>& faces) {
    if (!handle) throw std::runtime_error("isosurface: computeDistance has not been called");
    int64_t nv = 0, nt = 0;
    if (shm_grid_isosurface(handle, isoval, &nv, &nt) != SHM_OK) throw std::runtime_error(std::string("shm_grid_isosurface: ") + shm_grid_last_error(handle));
    std::vector<double> v((size_t)3 * nv);
    std::vector<int64_t> f((size_t)3 * nt);
    if (shm_grid_get_isosurface(handle, v.data(), f.data()) != SHM_OK) throw std::runtime_error(shm_grid_last_error(handle));
    vertices.resize((size_t)nv);
    faces.resize((size_t)nt);
    for (int64_t a = 0; a < nv; a++) vertices[(size_t)a] = Vector3{v[3 * a], v[3 * a + 1], v[3 * a + 2]};
    for (int64_t a = 0; a < nt; a++) faces[(size_t)a] = {(size_t)f[3 * a], (size_t)f[3 * a + 1], (size_t)f[3 * a + 2]};
}

VectorXd SignedHeatGridSolver::computeDistance(VertexPositionGeometry& geometry, const SignedHeat3DOptions& options) {
    if (options.rebuild || !gridBuilt) {
        const Vector3 c = centroid(geometry);
        buildGrid(c, radius(geometry, c), options);
        gridBuilt = true;
    }
    // time step from the input mesh, :42-44
    const double h = meanEdgeLength(geometry);
    shortTime = options.tCoef * h * h;
    lambda = std::sqrt(1. / shortTime);
    std::vector<double> areas;
    std::vector<Vector3> normals;
    setFaceVectorAreas(geometry, areas, normals);
    const size_t F = geometry.mesh.nFaces();
    srcPos.resize(3 * F);
    srcWn.resize(3 * F);
    srcArea = areas;
    for (size_t f = 0; f < F; f++) {
        const Vector3 b = barycenter(geometry, f);  // once per face, not once per (node, face) pair (:55)
        const Vector3 wn = normals[f] * areas[f];   // N * A first, then * yukawa (:57)
        for (int a = 0; a < 3; a++) {
            srcPos[3 * f + a] = b[a];
            srcWn[3 * f + a] = wn[a];
        }
    }
    return solveOnDevice(/*scrub=*/true, options);
}

VectorXd SignedHeatGridSolver::computeDistance(PointPositionNormalGeometry& pointGeom, const SignedHeat3DOptions& options) {
    // The point overload of the reference never sets poissonSolver, so it rebuilds the grid on every call (:119).
    {
        const Vector3 c = centroid(pointGeom);
        buildGrid(c, radius(pointGeom, c), options);
    }
    const size_t P = pointGeom.positions.size();
    if (pointGeom.dualAreas.size() != P || !(pointGeom.meanEdgeLength > 0.)) estimatePointAreas(pointGeom);
    const double h = pointGeom.meanEdgeLength;  // :151
    shortTime = options.tCoef * h * h;
    lambda = std::sqrt(1. / shortTime);
    srcPos.resize(3 * P);
    srcWn.resize(3 * P);
    srcArea = pointGeom.dualAreas;
    for (size_t p = 0; p < P; p++) {
        const Vector3 wn = pointGeom.normals[p] * pointGeom.dualAreas[p];  // n * A (:166)
        for (int a = 0; a < 3; a++) {
            srcPos[3 * p + a] = pointGeom.positions[p][a];
            srcWn[3 * p + a] = wn[a];
        }
    }
    return solveOnDevice(/*scrub=*/false, options);
}

}  // namespace shm_host

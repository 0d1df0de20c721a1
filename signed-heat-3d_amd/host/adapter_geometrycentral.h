// Drop-in adapter for the Polyscope demo of nzfeng/signed-heat-3d.  NOT compiled in this repository's build
// (geometry-central and Polyscope are absent from the image); it is the file a maintainer of the reference adds
// next to include/signed_heat_grid_solver.h, replacing src/signed_heat_grid_solver.cpp, and links with
// libshm_grid.so.  src/main.cpp is untouched: same class name, constructor, computeDistance overloads, VERBOSE
// member (include/signed_heat_grid_solver.h:11-22) and the same `registerVolumeGrid("domain", ...)` side effect
// (src/signed_heat_grid_solver.cpp:35,143) that src/main.cpp:95 relies on.  See INTEGRATION.md.
#pragma once

#include "geometrycentral/pointcloud/point_position_normal_geometry.h"
#include "geometrycentral/surface/vertex_position_geometry.h"
#include "polyscope/volume_grid.h"

#include "shm_grid.h"        // this repository's include/shm_grid.h
#include "signed_heat_3d.h"  // the reference's own header: SignedHeat3DOptions, centroid, radius, meanEdgeLength, setFaceVectorAreas

#include <cmath>
#include <iostream>
#include <stdexcept>
#include <string>
#include <vector>

using namespace geometrycentral;
using namespace geometrycentral::surface;

class SignedHeatGridSolver {
  public:
    SignedHeatGridSolver() {}
    ~SignedHeatGridSolver() {
        if (handle) shm_grid_destroy(handle);
    }

    Vector<double> computeDistance(VertexPositionGeometry& geometry, const SignedHeat3DOptions& options = SignedHeat3DOptions()) {
        if (options.rebuild || !gridBuilt) {
            Vector3 c = centroid(geometry);
            buildGrid(c, radius(geometry, c), options);
            gridBuilt = true;
        }
        SurfaceMesh& mesh = geometry.mesh;
        double h = meanEdgeLength(geometry);
        double lambda = std::sqrt(1. / (options.tCoef * h * h));
        FaceData<double> faceAreas;
        FaceData<Vector3> faceNormals;
        setFaceVectorAreas(geometry, faceAreas, faceNormals);
        std::vector<double> pos, wn, area;
        for (Face f : mesh.faces()) {
            Vector3 b = {0, 0, 0};
            for (Vertex v : f.adjacentVertices()) b += geometry.vertexPositions[v];
            b /= f.degree();
            Vector3 w = faceNormals[f] * faceAreas[f];
            for (int a = 0; a < 3; a++) {
                pos.push_back(b[a]);
                wn.push_back(w[a]);
            }
            area.push_back(faceAreas[f]);
        }
        return solve(pos, wn, area, lambda, /*scrub=*/1, options);
    }

    Vector<double> computeDistance(pointcloud::PointPositionNormalGeometry& pointGeom,
                                   const SignedHeat3DOptions& options = SignedHeat3DOptions()) {
        {   // the reference's point overload rebuilds on every call (poissonSolver is never set, :119)
            Vector3 c = centroid(pointGeom);
            buildGrid(c, radius(pointGeom, c), options);
        }
        pointGeom.requireTuftedTriangulation();
        pointGeom.tuftedGeom->requireVertexDualAreas();
        double h = meanEdgeLength(*(pointGeom.tuftedGeom));
        double lambda = std::sqrt(1. / (options.tCoef * h * h));
        size_t P = pointGeom.cloud.nPoints();
        std::vector<double> pos, wn, area;
        for (size_t p = 0; p < P; p++) {
            double A = pointGeom.tuftedGeom->vertexDualAreas[p];
            Vector3 w = pointGeom.normals[p] * A;
            for (int a = 0; a < 3; a++) {
                pos.push_back(pointGeom.positions[p][a]);
                wn.push_back(w[a]);
            }
            area.push_back(A);
        }
        Vector<double> phi = solve(pos, wn, area, lambda, /*scrub=*/0, options);
        pointGeom.unrequireTuftedTriangulation();
        pointGeom.tuftedGeom->unrequireVertexDualAreas();
        return phi;
    }

    bool VERBOSE = true;
    bool exactStep1 = false;   // backend knob (no counterpart in the reference): Step 1 entirely in fp64, shm_opts.step1_arith

  private:
    shm_solver* handle = nullptr;
    bool gridBuilt = false;
    size_t nx = 0, ny = 0, nz = 0;
    Vector3 bboxMin, bboxMax;
    double cellSize = 0.;

    void buildGrid(const Vector3& c, double r, const SignedHeat3DOptions& options) {
        if (VERBOSE) std::cerr << "Building grid..." << std::endl;
        double s = r * options.scale;
        bboxMin = Vector3{-s, -s, -s} + c;
        bboxMax = Vector3{s, s, s} + c;
        nx = 2 * std::pow(2, options.hCoef + 3);
        ny = nx;
        nz = nx;
        cellSize = 2. * s / (nx - 1);
        glm::vec3 boundMin, boundMax;
        for (int i = 0; i < 3; i++) {
            boundMin[i] = bboxMin[i];
            boundMax[i] = bboxMax[i];
        }
        polyscope::registerVolumeGrid("domain", {nx, ny, nz}, boundMin, boundMax);
    }

    Vector<double> solve(const std::vector<double>& pos, const std::vector<double>& wn, const std::vector<double>& area, double lambda,
                         int scrub, const SignedHeat3DOptions& options) {
        if (!handle) {
            shm_config cfg{};
            cfg.precision = SHM_F64;
            cfg.local_slabs = 1;
            cfg.world = 1;
            if (shm_grid_create(&cfg, &handle) != SHM_OK) throw std::runtime_error(shm_grid_last_error(nullptr));
        }
        shm_sources src{(int64_t)area.size(), pos.data(), wn.data(), area.data(), lambda};
        shm_grid grid{};
        grid.n = (int32_t)nx;
        for (int a = 0; a < 3; a++) grid.bbox_min[a] = bboxMin[a];
        grid.cell = cellSize;
        shm_opts opts{};
        opts.fast_integration = options.fastIntegration;
        opts.scrub_nonfinite = scrub;
        opts.step1_arith = exactStep1 ? SHM_STEP1_EXACT_F64 : SHM_STEP1_AUTO;
        Vector<double> phi(nx * ny * nz);
        if (VERBOSE) std::cerr << "Steps 1 & 2... Step 3..." << std::endl;
        if (shm_grid_compute_distance(handle, &src, &grid, &opts, phi.data(), nullptr) != SHM_OK)
            throw std::runtime_error(shm_grid_last_error(handle));
        if (VERBOSE) std::cerr << "\tCompleted." << std::endl;
        return phi;
    }
};

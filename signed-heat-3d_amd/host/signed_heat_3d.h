// Host-side helpers shared by the SHM solvers: same names, argument meaning and defaults as the
// reference's include/signed_heat_3d.h:20-35, on the headless containers of geometry.h.
#pragma once
#include "geometry.h"

namespace shm_host {

enum class LevelSetConstraint { None = 0, ZeroSet, Multiple };  // geometry-central's enum; the grid solver ignores it

struct SignedHeat3DOptions {  // signed_heat_3d.h:20-28 of the reference -- field names and defaults kept verbatim
    LevelSetConstraint levelSetConstraint = LevelSetConstraint::ZeroSet;
    double tCoef = 1.0;
    double hCoef = 0.0;
    bool rebuild = true;
    double scale = 2.;
    bool useCrouzeixRaviart = true;
    bool fastIntegration = false;
};

// Backend knobs that have no counterpart in the reference live in a separate struct so that reference call
// sites (`computeDistance(geometry, SHM_OPTIONS)`) compile unchanged.
struct GridBackendOptions {
    int device = 0;
    int precision = 64;   // 64 | 32
    double tol = 0.;      // <=0: library default
    int maxIters = 0;
    int localSlabs = 1;
    bool exactStep1 = false;   // shm_opts.step1_arith = SHM_STEP1_EXACT_F64: every (node, source) pair of Step 1 in the reference's fp64 arithmetic
};

Vector3 centroid(const VertexPositionGeometry& geometry);                       // signed_heat_3d.cpp:3-12
Vector3 centroid(const PointPositionNormalGeometry& pointGeom);                  // :24-33
double radius(const VertexPositionGeometry& geometry, const Vector3& c);         // :14-22
double radius(const PointPositionNormalGeometry& pointGeom, const Vector3& c);   // :35-43
double yukawaPotential(const Vector3& x, const Vector3& y, const double& lambda);  // :45-49
double meanEdgeLength(const VertexPositionGeometry& geometry);                   // :51-60 (unique edges)
void setFaceVectorAreas(const VertexPositionGeometry& geometry, std::vector<double>& areas, std::vector<Vector3>& normals);  // :62-89
Vector3 barycenter(const VertexPositionGeometry& geometry, size_t f);            // signed_heat_grid_solver.cpp:498-503

// Headless replacement of geometry-central's tufted-triangulation dual areas / edge length: k nearest
// neighbours, tangent-plane local Delaunay triangulations, union of the local triangles (geometry-central's construction without
// the intrinsic flips).  Parity with geometry-central is unpinned; both quantities are solver INPUTS.
void estimatePointAreas(PointPositionNormalGeometry& pointGeom, int k = 30);

}  // namespace shm_host

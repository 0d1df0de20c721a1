// STUB for the -fsyntax-only typo guard of host/adapter_geometrycentral.h (see README.md in this directory).  It stands in for the reference's
// include/signed_heat_3d.h, which a real build of the adapter uses instead; nothing here is compiled into any product, oracle or numerical test.
// Only what the grid adapter touches is declared: the five option fields SignedHeatGridSolver reads (reference: signed_heat_grid_solver.cpp:8,15,24,43,77)
// and the four host helpers it calls.
#pragma once
#include "geometrycentral/pointcloud/point_position_normal_geometry.h"
#include "geometrycentral/surface/vertex_position_geometry.h"

using namespace geometrycentral;            // the reference's header opens these two namespaces for its includers; the adapter relies on it
using namespace geometrycentral::surface;

struct SignedHeat3DOptions {
    bool rebuild = true, fastIntegration = false;
    double scale = 2., hCoef = 0., tCoef = 1.;
};

Vector3 centroid(VertexPositionGeometry&);
Vector3 centroid(pointcloud::PointPositionGeometry&);
double radius(VertexPositionGeometry&, const Vector3&);
double radius(pointcloud::PointPositionGeometry&, const Vector3&);
double meanEdgeLength(IntrinsicGeometryInterface&);
void setFaceVectorAreas(VertexPositionGeometry&, FaceData<double>&, FaceData<Vector3>&);

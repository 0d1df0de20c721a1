// STUB (typo guard only, see README.md) of the declarations the adapter needs from the reference's include/signed_heat_3d.h:20-37
// (in a real build the reference's own header is used; nothing here is compiled into any product or oracle).
#pragma once
#include "geometrycentral/pointcloud/point_position_normal_geometry.h"
#include "geometrycentral/surface/vertex_position_geometry.h"

using namespace geometrycentral;
using namespace geometrycentral::surface;

struct SignedHeat3DOptions {
    LevelSetConstraint levelSetConstraint = LevelSetConstraint::ZeroSet;
    double tCoef = 1.0;
    double hCoef = 0.0;
    bool rebuild = true;
    double scale = 2.;
    bool useCrouzeixRaviart = true;
    bool fastIntegration = false;
};
Vector3 centroid(VertexPositionGeometry& geometry);
Vector3 centroid(pointcloud::PointPositionGeometry& pointGeom);
double radius(VertexPositionGeometry& geometry, const Vector3& centroid);
double radius(pointcloud::PointPositionGeometry& pointGeom, const Vector3& c);
double meanEdgeLength(IntrinsicGeometryInterface& geom);
void setFaceVectorAreas(VertexPositionGeometry& geometry, FaceData<double>& areas, FaceData<Vector3>& normals);

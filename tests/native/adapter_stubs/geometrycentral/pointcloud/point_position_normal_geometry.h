// STUB (typo guard only, see ../../README.md)
#pragma once
#include "geometrycentral/surface/vertex_position_geometry.h"

namespace geometrycentral {
namespace pointcloud {
struct PointCloud {
    size_t nPoints() const { return 0; }
};
template <typename T> struct PointData {
    std::vector<T> d;
    T& operator[](size_t i) { return d[i]; }
};
struct PointPositionGeometry {
    PointCloud& cloud;
    PointData<Vector3> positions;
    std::unique_ptr<surface::IntrinsicGeometryInterface> tuftedGeom;
    explicit PointPositionGeometry(PointCloud& c) : cloud(c) {}
    void requireTuftedTriangulation() {}
    void unrequireTuftedTriangulation() {}
};
struct PointPositionNormalGeometry : PointPositionGeometry {
    PointData<Vector3> normals;
    using PointPositionGeometry::PointPositionGeometry;
};
}  // namespace pointcloud
}  // namespace geometrycentral

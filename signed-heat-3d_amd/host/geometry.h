// Minimal host-side geometry containers for the headless build of the grid solver.
//
// In the Polyscope demo the solver receives geometry-central objects (VertexPositionGeometry,
// pointcloud::PointPositionNormalGeometry); geometry-central is not available in this image, so the
// headless host layer carries its own plain containers with the same *roles*.  The adapter for the real
// geometry-central types is host/adapter_geometrycentral.h (compiled only where that library exists).
#pragma once
#include <array>
#include <cmath>
#include <cstddef>
#include <string>
#include <vector>

namespace shm_host {

struct Vector3 {
    double x = 0, y = 0, z = 0;
    double& operator[](int i) { return i == 0 ? x : (i == 1 ? y : z); }
    double operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
    Vector3 operator+(const Vector3& o) const { return {x + o.x, y + o.y, z + o.z}; }
    Vector3 operator-(const Vector3& o) const { return {x - o.x, y - o.y, z - o.z}; }
    Vector3 operator*(double s) const { return {x * s, y * s, z * s}; }
    Vector3 operator/(double s) const { return {x / s, y / s, z / s}; }
    Vector3& operator+=(const Vector3& o) { x += o.x; y += o.y; z += o.z; return *this; }
    Vector3& operator/=(double s) { x /= s; y /= s; z /= s; return *this; }
    double norm() const { return std::sqrt(x * x + y * y + z * z); }
};
inline Vector3 cross(const Vector3& a, const Vector3& b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }

// Polygon mesh: faces are index lists into the (already compacted) vertex array.
struct SurfaceMesh {
    std::vector<std::vector<size_t>> faces;
    size_t nVerts = 0;
    size_t nVertices() const { return nVerts; }
    size_t nFaces() const { return faces.size(); }
    bool isTriangular() const {
        for (const auto& f : faces)
            if (f.size() != 3) return false;
        return true;
    }
};

struct VertexPositionGeometry {
    SurfaceMesh mesh;
    std::vector<Vector3> vertexPositions;
};

// Point cloud with normals.  The reference takes the per-point area and the length scale h from
// geometry-central's tufted intrinsic triangulation (signed_heat_grid_solver.cpp:149-151,165); here they
// are explicit inputs (filled by estimatePointAreas() in the headless build, or by the adapter).
struct PointPositionNormalGeometry {
    std::vector<Vector3> positions, normals;
    std::vector<double> dualAreas;  // one per point
    double meanEdgeLength = 0.;     // h
};

}  // namespace shm_host

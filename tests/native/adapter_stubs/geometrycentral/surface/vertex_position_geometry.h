// STUB (typo guard only, see ../../README.md): minimal declarations of the geometry-central names adapter_geometrycentral.h uses.
#pragma once
#include <cstddef>
#include <memory>
#include <vector>

namespace geometrycentral {
struct Vector3 {
    double x, y, z;
    double& operator[](int i) { return (&x)[i]; }
    double operator[](int i) const { return (&x)[i]; }
    Vector3 operator+(const Vector3& o) const { return {x + o.x, y + o.y, z + o.z}; }
    Vector3 operator*(double s) const { return {x * s, y * s, z * s}; }
    Vector3& operator+=(const Vector3& o) { x += o.x; y += o.y; z += o.z; return *this; }
    Vector3& operator/=(double s) { x /= s; y /= s; z /= s; return *this; }
};
template <typename T> struct Vector {  // stands for Eigen::Matrix<T, Dynamic, 1>
    std::vector<T> v;
    Vector() {}
    explicit Vector(size_t n) : v(n) {}
    T* data() { return v.data(); }
    T& operator[](size_t i) { return v[i]; }
    size_t size() const { return v.size(); }
};
namespace surface {
struct Vertex { size_t i; };
struct Face {
    size_t i;
    std::vector<Vertex> adjacentVertices() const { return {}; }
    size_t degree() const { return 3; }
};
struct SurfaceMesh {
    std::vector<Face> faces() const { return {}; }
};
template <typename T> struct FaceData {
    std::vector<T> d;
    T& operator[](Face f) { return d[f.i]; }
};
template <typename T> struct VertexData {
    std::vector<T> d;
    T& operator[](Vertex v) { return d[v.i]; }
    T& operator[](size_t i) { return d[i]; }
};
struct IntrinsicGeometryInterface {
    VertexData<double> vertexDualAreas;
    void requireVertexDualAreas() {}
    void unrequireVertexDualAreas() {}
};
struct VertexPositionGeometry : IntrinsicGeometryInterface {
    SurfaceMesh& mesh;
    VertexData<Vector3> vertexPositions;
    explicit VertexPositionGeometry(SurfaceMesh& m) : mesh(m) {}
};
enum class LevelSetConstraint { None, ZeroSet, Multiple };
}  // namespace surface
}  // namespace geometrycentral

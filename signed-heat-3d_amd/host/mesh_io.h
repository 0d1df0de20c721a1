// OBJ / .pc loaders for the headless build (the demo uses geometry-central's readSurfaceMesh, src/main.cpp:269,
// and its own readPointCloud, src/main.cpp:196-225).
#pragma once
#include <string>

#include "geometry.h"

namespace shm_host {
// Polygonal OBJ: `v`, and `f` with v, v/vt, v//vn, v/vt/vn tokens (negative indices allowed).  Vertices no face
// references are dropped, file order preserved (the reference's loader does the same).
VertexPositionGeometry readSurfaceMesh(const std::string& path);
// `v x y z` -> position, `vn x y z` -> normal, anything else ignored (src/main.cpp:196-225).
PointPositionNormalGeometry readPointCloud(const std::string& path);
// Triangle mesh -> Wavefront OBJ (the demo writes ../export/isosurface.obj through geometry-central, src/main.cpp:188-190).
void writeSurfaceMesh(const std::vector<Vector3>& vertices, const std::vector<std::array<size_t, 3>>& faces, const std::string& path);
}  // namespace shm_host

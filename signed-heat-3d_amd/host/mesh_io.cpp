#include "mesh_io.h"

#include <fstream>
#include <sstream>
#include <stdexcept>

namespace shm_host {

VertexPositionGeometry readSurfaceMesh(const std::string& path) {
    std::ifstream in(path);
    if (!in.is_open()) throw std::runtime_error("Could not open file <" + path + ">.");
    std::vector<Vector3> verts;
    std::vector<std::vector<size_t>> faces;
    std::string line, tag, tok;
    while (std::getline(in, line)) {
        std::istringstream iss(line);
        if (!(iss >> tag)) continue;
        if (tag == "v") {
            Vector3 p;
            iss >> p.x >> p.y >> p.z;
            verts.push_back(p);
        } else if (tag == "f") {
            std::vector<size_t> f;
            while (iss >> tok) {
                const long vi = std::stol(tok.substr(0, tok.find('/')));
                f.push_back(vi > 0 ? (size_t)(vi - 1) : (size_t)((long)verts.size() + vi));
            }
            if (f.size() >= 3) faces.push_back(f);
        }
    }
    std::vector<char> used(verts.size(), 0);
    for (const auto& f : faces)
        for (size_t v : f) {
            if (v >= verts.size()) throw std::runtime_error("face index out of range in " + path);
            used[v] = 1;
        }
    std::vector<size_t> remap(verts.size(), 0);
    VertexPositionGeometry g;
    for (size_t v = 0; v < verts.size(); v++)
        if (used[v]) {
            remap[v] = g.vertexPositions.size();
            g.vertexPositions.push_back(verts[v]);
        }
    for (auto& f : faces)
        for (size_t& v : f) v = remap[v];
    g.mesh.faces = std::move(faces);
    g.mesh.nVerts = g.vertexPositions.size();
    return g;
}

PointPositionNormalGeometry readPointCloud(const std::string& path) {
    std::ifstream in(path);
    if (!in.is_open()) throw std::runtime_error("Could not open file <" + path + ">.");
    PointPositionNormalGeometry g;
    std::string line, tag;
    while (std::getline(in, line)) {
        std::istringstream iss(line);
        if (!(iss >> tag)) continue;
        Vector3 p;
        if (tag == "v") {
            iss >> p.x >> p.y >> p.z;
            g.positions.push_back(p);
        } else if (tag == "vn") {
            iss >> p.x >> p.y >> p.z;
            g.normals.push_back(p);
        }
    }
    if (g.positions.size() != g.normals.size()) throw std::runtime_error("point cloud needs one vn per v: " + path);
    return g;
}

void writeSurfaceMesh(const std::vector<Vector3>& vertices, const std::vector<std::array<size_t, 3>>& faces, const std::string& path) {
    std::ofstream out(path);
    if (!out.is_open()) throw std::runtime_error("Could not open file <" + path + "> for writing.");
    out.precision(17);
    for (const Vector3& p : vertices) out << "v " << p.x << " " << p.y << " " << p.z << "\n";
    for (const auto& f : faces) out << "f " << f[0] + 1 << " " << f[1] + 1 << " " << f[2] + 1 << "\n";
}

}  // namespace shm_host

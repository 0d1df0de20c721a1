// STUB (typo guard only, see ../README.md)
#pragma once
#include <array>
#include <cstddef>
#include <string>

namespace glm {
struct vec3 {
    float v[3];
    float& operator[](int i) { return v[i]; }
};
struct uvec3 {
    unsigned x, y, z;
    template <typename A, typename B, typename C> uvec3(A a, B b, C c) : x((unsigned)a), y((unsigned)b), z((unsigned)c) {}
};
}  // namespace glm
namespace polyscope {
struct VolumeGrid {};
inline VolumeGrid* registerVolumeGrid(std::string, glm::uvec3, glm::vec3, glm::vec3) { return nullptr; }
}  // namespace polyscope

#pragma once
#include <cstdint>
#include <vector>

#include "layout.h"

namespace spc {

struct HostMesh {  // triangle soup after the quad-light triangles were appended
    const float* vertices = nullptr;
    const float* texcoords = nullptr;  // may be null
    const uint32_t* indices = nullptr;
    const int32_t* tri_material = nullptr;
    const uint8_t* tri_emitter = nullptr;
    int n_vertices = 0, n_triangles = 0;
};

struct Lbvh {
    std::vector<float> nodes;      // 16 dwords per quantised 4-wide node (layout.h)
    std::vector<float> tris;       // 16 floats per triangle, BVH order
    std::vector<int32_t> tri_orig; // BVH order -> input triangle index
    int depth = 0;         // depth of the 4-wide tree
    int binary_depth = 0;  // depth of the binary radix tree it was folded from
};

void build_lbvh(const HostMesh& mesh, Lbvh& out);

}  // namespace spc

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
    // Round 6: what the pooled traversal pass of the eye megakernel tests triangles from -- 16 floats per triangle slot, BVH order.
    // Slot i of a FAN PAIR (triangles i and i + 1 of one leaf with B.P0 == A.P0 and B.P1 == A.P2 bit for bit: the two halves of a
    // quad as every mesher emits them) holds (A.P0, A.P1 - A.P0, A.P2 - A.P0, B.P2 - A.P0) + flags -- a corner and the edges the test starts
    // from, subtracted once on the host -- so that ONE step of a lane tests both with A's and B's own arithmetic; any other slot holds its
    // triangle's corner and two edges + flags.  flags (w of the fourth quad): bit 0 = pair, bit 31 / 30 =
    // back-face culling of A / B (single-sided emitters).  `tris` stays what every other consumer reads (tails, one-ray-per-lane loop, shading).
    std::vector<float> pairs;
    int n_paired = 0;              // triangles that are half of a pair
    int depth = 0;         // depth of the 4-wide tree
    int binary_depth = 0;  // depth of the binary radix tree it was folded from
};

void build_lbvh(const HostMesh& mesh, Lbvh& out);

}  // namespace spc

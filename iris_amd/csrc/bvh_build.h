// Host-side BVH construction for libiris_hip.so: binned-SAH binary build, collapse to a W-wide tree.
// (The reference has no counterpart: it calls mitsuba.load_dict -> OptiX GAS build, bake_shading.py:55-61.)
#pragma once
#include <cstdint>
#include <vector>

namespace iris {

constexpr int kMaxWidth = 8;

struct WideNode {
    int n = 0;                       // children in use (slots [0,n))
    float lo[kMaxWidth][3];          // child boxes (already padded)
    float hi[kMaxWidth][3];
    int32_t child[kMaxWidth];        // >=0: index of an internal wide node; -1: leaf
    int32_t leaf_start[kMaxWidth];   // leaf: first triangle in tri_order
    int32_t leaf_count[kMaxWidth];   // leaf: number of triangles
    uint8_t order[8][kMaxWidth];     // per ray octant (bit 0: d.x < 0, bit 1: d.y < 0, bit 2: d.z < 0): the slots in front-to-back order as the binary splits the
                                     // node was collapsed from give it (at every split the side the ray enters first; the left child holds the lower centroids)
};

struct WideBvh {
    int width = 0;
    std::vector<WideNode> nodes;     // nodes[0] is the root; the internal children of a node are consecutive
    std::vector<int32_t> tri_order;  // leaf order -> original triangle index; a node's leaf triangles are consecutive.  Longer than the mesh
                                     // when long triangles were split into several references (presplit): those appear once per reference
    float root_lo[3], root_hi[3];
    int depth = 0;
    float sah_cost = 0.f;
    float pad = 0.f;
};

// verts: (nv,3) f32, faces: (nf,3) i32.  leaf_tris: max triangles per leaf (1..7).  tri_cost: cost of one triangle test relative to
// one wide-node visit in the SAH the collapse minimises (measured on the traversal kernels: ~70 against ~110 instructions).
// Boxes are padded by `pad_rel * max(|coordinate|, extent)` so that the f32 slab test is conservative with respect to the
// Moeller-Trumbore test (see DESIGN.md "closest-hit semantics").  presplit: early split clipping of triangles whose box is longer than
// presplit x the median triangle's (0 = off), see bvh_build.cpp.
WideBvh build_wide_bvh(const float* verts, int64_t nv, const int32_t* faces, int64_t nf, int width, int leaf_tris,
                       float pad_rel = 2e-5f, float tri_cost = 0.7f, float presplit = 8.f);

}  // namespace iris

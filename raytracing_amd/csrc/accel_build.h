// Host-side construction of the GPU-oriented acceleration structure used by
// RT_HIP_ACCEL_REBUILT (see rt_traverse_fast.h for the layout rationale).
#pragma once

#include <cstdint>
#include <vector>

#include "rt_device_scene.h"
#include "rt_types.h"

namespace rtd {

struct built_accel {
    std::vector<wide_node> nodes;         // TLAS and every BLAS, most visited first
    std::vector<tri_pos> tris;            // leaf order; ref_index -> uploaded triangles[]
    std::vector<instance_rec> instances;  // TLAS leaves (an instance or a part of one), most visited first
    int32_t tlas_root = 0;                // node index, or an encoded leaf
    uint32_t max_stack_bound = 0;         // TLAS depth + 1 + deepest BLAS depth
    // Triangles the reference's intersection test rejects before looking at the ray
    // (triangle_never_hit) are left out of the leaves, so the traversal need not repeat that
    // test per ray.  A mesh made of nothing else keeps its triangles (a BLAS cannot be empty);
    // then this is true and the traversal keeps the test.
    bool lists_degenerate_triangles = false;
};

// shaders/raytracer/ray.glsl:124-128 (and :52-56): a triangle whose edge cross product is
// shorter than 1e-8 is never hit, whatever the ray.  Same float operations, in the same
// order, as the device's hit_triangle, so both always agree.
bool triangle_never_hit(rt_triangle const& tri);

// Most TLAS leaves that opening instances may add (build_wide_accel splits the crowded
// instances with the largest world boxes into subtrees of their BLAS; how many leaves it adds
// depends on how many triangles those instances have).
constexpr uint32_t kOpenedEntries = 16384;

// Builds binned-SAH binary trees over the uploaded triangles (one BLAS per
// mesh, shared by its instances) and over the world boxes of the instances
// or, where instances were opened, of their parts.
// Throws std::runtime_error on malformed input.
built_accel build_wide_accel(rt_scene_desc const& scene, uint32_t max_opened_entries = kOpenedEntries);

}  // namespace rtd

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
};

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

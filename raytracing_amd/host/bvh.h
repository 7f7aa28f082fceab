// Host-side two-level BVH build: the producer of the tlas / blas / mesh /
// instance / triangle buffers the compute pass traverses.
// Reference: src/renderer/bvh.h:59-67, src/renderer/bvh.cpp:16-272,
// src/renderer/aabb.cpp.
#pragma once

#include <vector>

#include "scene.h"

struct bvh {
    std::vector<rt_bvh_node> tlas;       // over instances, world-space boxes
    std::vector<rt_bvh_node> blas;       // all meshes back to back, object space
    std::vector<rt_mesh> meshes;
    std::vector<rt_instance> instances;  // in TLAS leaf order
    std::vector<rt_triangle> triangles;  // per mesh, in BLAS leaf order
};

bvh create_bvh(scene const& scene);

// Longest root-to-leaf path (in nodes) of the subtree starting at `root`.
uint32_t bvh_depth(std::vector<rt_bvh_node> const& nodes, uint32_t root);

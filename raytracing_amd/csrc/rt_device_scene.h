// Device-resident scene: what the kernels receive by value.
#pragma once

#include "rt_math.h"
#include "rt_types.h"

namespace rtd {

// One entry per texture; pixels live in one device allocation each.
struct texture_dev {
    const void* data;
    uint32_t width;
    uint32_t height;
    uint32_t format;  // RT_TEX_*
    uint32_t _pad;
};

// 64-byte node of the rebuilt BVH: both children's boxes live in the parent,
// so one aligned 64-byte fetch decides the whole step.
//   child >= 0 : index of an inner node
//   child <  0 : leaf; ~child = (first << 3) | (count - 1)   (count 1..8)
// "first" indexes tri_pos (BLAS) or instance records (TLAS).
struct wide_node {
    float lo0x, lo0y, lo0z, hi0x;
    float hi0y, hi0z, lo1x, lo1y;
    float lo1z, hi1x, hi1y, hi1z;
    int32_t child0, child1;
    uint32_t _pad0, _pad1;
};
static_assert(sizeof(wide_node) == 64, "wide_node must be one 64-byte line");

// Positions-only triangle of the rebuilt BLAS: 48 bytes, three 16-byte loads.
// w0 carries the triangle's index in the uploaded (reference-ordered) array.
struct tri_pos {
    float ax, ay, az;
    uint32_t ref_index;
    float bx, by, bz;
    uint32_t _p0;
    float cx, cy, cz;
    uint32_t _p1;
};
static_assert(sizeof(tri_pos) == 48, "tri_pos is three float4");

// Per-instance record of the rebuilt TLAS (64 bytes): the three rows used of
// the world->object matrix, the BLAS root and the reference instance index.
struct instance_rec {
    float m[12];         // column-major 3 rows x 4 columns: m[col*3 + row]
    int32_t blas_root;   // index into wide nodes (>=0) or encoded leaf (<0)
    uint32_t ref_index;  // index in the uploaded instances[] array
    int32_t has_transform;
    uint32_t _pad;
};
static_assert(sizeof(instance_rec) == 64, "instance_rec is one 64-byte line");

struct scene_dev {
    const rt_bvh_node* tlas;
    const rt_bvh_node* blas;
    const rt_mesh* meshes;
    const rt_instance* instances;
    const rt_triangle* triangles;
    const rt_mat4* transforms;
    const rt_mat4* inverse_transforms;
    const rt_material* materials;
    const rt_light* lights;
    const texture_dev* textures;
    // rebuilt acceleration structure (RT_HIP_ACCEL_REBUILT only)
    const wide_node* wide_nodes;
    const tri_pos* wide_tris;
    const instance_rec* wide_instances;
    int32_t wide_tlas_root;
    uint32_t light_count;
    int32_t sky_light;
    uint32_t _pad;
};

}  // namespace rtd

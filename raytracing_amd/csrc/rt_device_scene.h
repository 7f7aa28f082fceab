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

// 128-byte node of the rebuilt BVH: up to FOUR children with their boxes (full floats), laid
// out per plane so that child c is component c of each float4.  The walk runs at the latency
// of its dependent fetches; seven 16-byte loads issued together cost one latency, like the
// four of a two-child node, and the walk has two thirds as many steps.
//   child >= 0            : index of an inner node
//   child <  0            : leaf; ~child = (first << 3) | (count - 1)   (count 1..8)
//   child == kNoChild     : unused slot (its box is inverted: never hit)
// "first" indexes tri_pos (BLAS) or instance records (TLAS).
constexpr int32_t kNoChild = 0x7FFFFFFE;
constexpr uint32_t kWideNodeQuads = 8;  // float4s per node (the last one is padding)

struct wide_node {
    float lo_x[4], lo_y[4], lo_z[4];
    float hi_x[4], hi_y[4], hi_z[4];
    int32_t child[4];
    uint32_t _pad[4];
};
static_assert(sizeof(wide_node) == 128, "wide_node is eight float4");

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

struct normal_matrix {
    float4 c0, c1, c2;
};

struct scene_dev {
    const rt_bvh_node* tlas;
    const rt_bvh_node* blas;
    const rt_mesh* meshes;
    const rt_instance* instances;
    const rt_triangle* triangles;
    const rt_mat4* transforms;
    const rt_mat4* inverse_transforms;
    // per transform: the inverse of its upper 3x3 as three columns (w unused), which carries
    // normals to world space - computed once per scene with the shading code's own inverse33
    // instead of once per hit (megakernel_raytracer.comp:299 and :474 do it per hit)
    const normal_matrix* normal_matrices;
    const rt_material* materials;
    const rt_light* lights;
    const texture_dev* textures;
    // rebuilt acceleration structure (RT_HIP_ACCEL_REBUILT only)
    const wide_node* wide_nodes;
    const tri_pos* wide_tris;
    const instance_rec* wide_instances;
    int32_t wide_tlas_root;
    uint32_t wide_lists_degenerate;  // 0: no leaf lists a triangle that hit_triangle's first test rejects
    uint32_t light_count;
    int32_t sky_light;
    // LDS plan of the cached traversal kernel: per-lane stack entries, and how many of the
    // first nodes / instance records each workgroup copies into LDS
    uint32_t stack_entries;
    uint32_t cached_nodes;
    uint32_t cached_instances;
    // Opt-in importance sampling of the environment map (null / 0 = the reference's sampling):
    // density[y * w + x] = probability density of texel (x, y) over the unit (u, v) square,
    // conditional[y * (w + 1) + x] = normalised CDF along row y, marginal[y] = CDF over rows.
    const float* sky_density;
    const float* sky_conditional;
    const float* sky_marginal;
    uint32_t sky_w, sky_h;
    // sizes of the small tables the shading kernel may stage in LDS
    uint32_t instance_count, transform_count, material_count, mesh_count, texture_count;
};

// Which of the scene's small tables the shading kernel copies into LDS, and where.  The
// tables are read with per-lane indices (the hit instance, its material, a random light):
// from LDS such a read costs a few cycles, from memory one cache-line lookup per lane.
enum : uint32_t {
    kTableLights = 0,
    kTableMaterials,
    kTableMeshes,
    kTableTextures,
    kTableInstances,
    kTableTransforms,
    kTableNormalMatrices,
    kTableCount,
};
constexpr uint32_t kNotStaged = 0xFFFFFFFFu;

struct shade_plan {
    uint32_t offset[kTableCount];  // byte offset in LDS, or kNotStaged
    uint32_t bytes[kTableCount];
    uint32_t total;
};

// Greedy in the order of the enum (smallest and most used first); host and kernel run the
// same function, so they agree on the layout without passing it around.
__host__ __device__ inline shade_plan plan_shade_tables(scene_dev const& sc, uint32_t budget) {
    shade_plan plan;
    plan.bytes[kTableLights] = sc.light_count * (uint32_t) sizeof(rt_light);
    plan.bytes[kTableMaterials] = sc.material_count * (uint32_t) sizeof(rt_material);
    plan.bytes[kTableMeshes] = sc.mesh_count * (uint32_t) sizeof(rt_mesh);
    plan.bytes[kTableTextures] = sc.texture_count * (uint32_t) sizeof(texture_dev);
    plan.bytes[kTableInstances] = sc.instance_count * (uint32_t) sizeof(rt_instance);
    plan.bytes[kTableTransforms] = sc.transform_count * (uint32_t) sizeof(rt_mat4);
    plan.bytes[kTableNormalMatrices] = sc.transform_count * (uint32_t) sizeof(normal_matrix);
    plan.total = 0;
    for (uint32_t t = 0; t < kTableCount; ++t) {
        uint32_t const padded = (plan.bytes[t] + 15u) & ~15u;
        if (plan.bytes[t] != 0u && plan.total + padded <= budget) {
            plan.offset[t] = plan.total;
            plan.total += padded;
        } else {
            plan.offset[t] = kNotStaged;
        }
    }
    return plan;
}

}  // namespace rtd

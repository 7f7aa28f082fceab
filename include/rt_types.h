/*
 * Plain-old-data layouts shared by the host layer, the HIP library and the
 * CPU oracle.  Every struct is byte-identical to the buffer element the
 * reference host uploads to its compute shader, so a reference-side wrapper
 * can pass `std::vector<...>::data()` straight through.
 *
 * Reference definitions (relative to the upstream repository root):
 *   rt_bvh_node   src/renderer/bvh.h:30-37    shaders/raytracer/bvh.glsl:26-32
 *   rt_mesh       src/renderer/bvh.h:39-43    shaders/raytracer/bvh.glsl:1-5
 *   rt_instance   src/renderer/bvh.h:45-51    shaders/raytracer/bvh.glsl:7-13
 *   rt_vertex     src/asset/mesh.h:13-16      shaders/common/geometry.glsl:1-10
 *   rt_triangle   src/renderer/bvh.h:53-57    shaders/raytracer/megakernel_raytracer.comp:116-124
 *   rt_material   src/asset/material.h:8-32   shaders/common/material.glsl:1-24
 *   rt_medium     src/asset/material.h:40-47  shaders/common/material.glsl:30-35
 *   rt_light      src/asset/light.h:17-27     shaders/common/light.glsl:6-13
 *   rt_camera     src/asset/camera.h:11-16    shaders/raytracer/camera.glsl:8-22
 */
#ifndef RT_TYPES_H
#define RT_TYPES_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 40 bytes.  Left child = own index + 1 (depth-first order); leaf iff obj_count > 0. */
typedef struct rt_bvh_node {
    float x_min, x_max, y_min, y_max, z_min, z_max;
    uint32_t right;
    uint32_t first_obj;
    uint32_t obj_count;
    int32_t split_axis; /* -1 none, 0 x, 1 y, 2 z */
} rt_bvh_node;

/* 12 bytes. bvh_start = index of the mesh's BLAS root inside the shared blas[] array. */
typedef struct rt_mesh {
    uint32_t triangle_offset;
    uint32_t triangle_count;
    uint32_t bvh_start;
} rt_mesh;

/* 20 bytes. Area lights are instances with material = -1 and light >= 0. */
typedef struct rt_instance {
    uint32_t mesh;
    int32_t transform;
    int32_t material;
    int32_t medium;
    int32_t light;
} rt_instance;

/* 32 bytes: (position.xyz, u), (normal.xyz, v). */
typedef struct rt_vertex {
    float position_texu[4];
    float normal_texv[4];
} rt_vertex;

/* 96 bytes. */
typedef struct rt_triangle {
    rt_vertex a, b, c;
} rt_triangle;

/* 64 bytes, column-major: m[col][row]. */
typedef struct rt_mat4 {
    float m[4][4];
} rt_mat4;

/* 96 bytes. */
typedef struct rt_material {
    float albedo[3];
    int32_t albedo_tex;
    float emission[3];
    int32_t emission_tex;
    float metallic, spec_trans, ior, subsurface;
    float roughness, specular_tint, anisotropic, sheen;
    float sheen_tint, clearcoat, clearcoat_gloss;
    int32_t normal_tex;
    int32_t metallic_roughness_tex;
    float _p0[3];
} rt_material;

/* 32 bytes.  Bound by the reference shader but never read. */
typedef struct rt_medium {
    float color[3];
    float density;
    float anisotropic;
    int32_t type;
    float _p0[2];
} rt_medium;

enum {
    RT_LIGHT_DISTANT = 0,
    RT_LIGHT_AREA_SINGLE_SIDED = 1,
    RT_LIGHT_AREA_DOUBLE_SIDED = 2,
    RT_LIGHT_SKY = 3
};

/* 48 bytes.  direction.x = total area (area lights);
 * direction = (total_luminance, tex_w, tex_h) for the sky light. */
typedef struct rt_light {
    float intensity[3];
    int32_t emission_tex;
    float direction[3];
    int32_t type;
    uint32_t mesh;
    int32_t transform;
    float _p0[2];
} rt_light;

enum { RT_TEX_UNORM8X4 = 0, RT_TEX_F32X4 = 1 };

/* One texture: 4 channels, row-major, top row first, no mip chain.
 * Sampled bilinear with REPEAT addressing and no sRGB decode
 * (src/vulkan/vulkan_image.cpp:309-330). */
typedef struct rt_texture {
    uint32_t width;
    uint32_t height;
    uint32_t format; /* RT_TEX_* */
    uint32_t _pad;
    const void* data; /* width*height*4 bytes (unorm8) or *16 bytes (f32) */
} rt_texture;

/* 48 bytes: the 12 packed floats of the push constant. */
typedef struct rt_camera {
    float pixel_delta_u[3];
    float pixel_delta_v[3];
    float upper_left_pixel[3];
    float position[3];
} rt_camera;

/* Everything the compute pass binds (megakernel_raytracer.comp:31-73 and the
 * non-camera push-constant words), as host pointers + element counts. */
typedef struct rt_scene_desc {
    const rt_bvh_node* tlas;
    const rt_bvh_node* blas;
    const rt_mesh* meshes;
    const rt_instance* instances;
    const rt_triangle* triangles;
    const rt_mat4* transforms;
    const rt_mat4* inverse_transforms;
    const rt_material* materials;
    const rt_medium* mediums;
    const rt_light* lights;
    const rt_texture* textures;
    uint32_t tlas_count;
    uint32_t blas_count;
    uint32_t mesh_count;
    uint32_t instance_count;
    uint32_t triangle_count;
    uint32_t transform_count;
    uint32_t material_count;
    uint32_t medium_count;
    uint32_t light_count;  /* push constant light_count */
    uint32_t texture_count;
    int32_t sky_light;     /* push constant sky_light: index of the sky light or -1 */
    uint32_t _pad;
} rt_scene_desc;

/* Work counters. */
typedef struct rt_stats {
    uint64_t samples;       /* pixel-samples traced */
    uint64_t closest_rays;  /* closest_hit() calls */
    uint64_t shadow_rays;   /* any_hit() calls */
    uint64_t closest_hits;  /* closest_hit() calls that returned a hit */
    uint64_t nodes_visited; /* BVH nodes fetched (TLAS + BLAS) */
    uint64_t tris_tested;   /* ray/triangle tests */
    uint64_t max_stack;     /* deepest traversal stack seen */
    uint64_t shadow_skipped;/* occlusion rays the shader traces although the light sample is
                               then discarded (BSDF pdf <= 0); not traced, not in shadow_rays */
    double   seconds;       /* device time of the counted launches */
} rt_stats;

#ifdef __cplusplus
}
#endif

#endif /* RT_TYPES_H */

/*
 * C interface over the host layer (scene JSON loader, BVH builder, camera),
 * for callers that cannot include the C++ headers (the Python tests and
 * benchmark).  It exposes the same data the reference host hands to its
 * renderer back end:
 *   load_scene()                 src/asset/scene.cpp:34
 *   create_bvh()                 src/renderer/bvh.cpp:16
 *   glm::inverse per transform   src/renderer/megakernel_raytracer.cpp:277-282
 *   light_count / sky_light_idx  src/renderer/megakernel_raytracer.cpp:272-275
 *   get_glsl_raytracer_camera()  src/asset/camera.cpp:66-88
 *
 * All functions return 0 on success and a non-zero code on failure;
 * rt_host_last_error() then describes the failure (thread-local).
 */
#ifndef RT_HOST_H
#define RT_HOST_H

#include <stddef.h>
#include <stdint.h>

#include "rt_types.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rt_host_scene rt_host_scene;

typedef struct rt_host_info {
    uint32_t resolution_x, resolution_y; /* scene JSON renderer.resolution */
    uint32_t max_depth;                  /* renderer.max_depth */
    uint32_t tile_width, tile_height;    /* renderer.tile */
    uint32_t tlas_depth;                 /* longest root-to-leaf path, in nodes */
    uint32_t blas_depth;                 /* maximum over all meshes */
    uint32_t vertex_count;
    float cam_position[3], cam_front[3], cam_right[3], cam_up[3];
    float cam_fov;
} rt_host_info;

/* Load a scene JSON plus the meshes/textures it names, and build its BVH. */
int rt_host_load_scene(const char* json_path, rt_host_scene** out);

/* Load a scene snapshot (see rt_host_save_snapshot) from memory or a file. */
int rt_host_load_snapshot(const void* bytes, size_t size, rt_host_scene** out);
int rt_host_load_snapshot_file(const char* path, rt_host_scene** out);

/* Write (render options, camera, scene) as a compressed snapshot. */
int rt_host_save_snapshot(const rt_host_scene* scene, const char* path);

/* Fill `out` with pointers into the scene; valid until rt_host_free_scene(). */
int rt_host_scene_desc(const rt_host_scene* scene, rt_scene_desc* out);

int rt_host_scene_info(const rt_host_scene* scene, rt_host_info* out);

/* Packed camera for a frame of the given size, from the scene's own camera. */
int rt_host_camera(const rt_host_scene* scene, uint32_t width, uint32_t height, rt_camera* out);

/* Packed camera from explicit look-from / look-at / vertical fov (degrees). */
int rt_host_make_camera(const float lookfrom[3], const float lookat[3], float fov,
    uint32_t width, uint32_t height, rt_camera* out);

/* Fixed-seed stand-in for the reference's per-dispatch random_seed
 * (src/renderer/megakernel_raytracer.cpp:80-86 draws from
 * mt19937(random_device) for every tile dispatch, which is not reproducible).
 * Writes outputs [first, first+count) of std::mt19937(seed): the dispatch of
 * pass p, tile k (tiles numbered row-major, x fastest, as next_tile() walks
 * them, :58-72) uses output p*tile_count + k. */
int rt_host_seed_schedule(uint32_t seed, uint64_t first, uint64_t count, uint32_t* out);

void rt_host_free_scene(rt_host_scene* scene);

const char* rt_host_last_error(void);

#ifdef __cplusplus
}
#endif

#endif /* RT_HOST_H */

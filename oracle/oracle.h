/*
 * CPU ORACLE - TEST INFRASTRUCTURE ONLY.
 *
 * A scalar C restatement of the reference's path-tracing compute pass
 * (shaders/raytracer/megakernel_raytracer.comp and the .glsl files it
 * includes).  It exists to check the HIP kernels and to give a CPU baseline;
 * nothing in the product path may include, link or call it.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg use it.
 *
 * PARITY UNPINNED: the reference ships no tests, golden vectors or fixed-seed
 * images, and its shader cannot be compiled or run offline (no GLSL compiler,
 * no Vulkan driver), so this restatement cannot be checked against the
 * reference's own output.  It follows the GLSL statement by statement; every
 * function cites the lines it restates.  Choices the GLSL leaves to the
 * implementation (operation order inside dot/mat*vec/normalize/inverse,
 * contraction, transcendental precision, bilinear weights) are pinned in
 * oracle.c and documented there.
 */
#ifndef RT_ORACLE_H
#define RT_ORACLE_H

#include <stdint.h>

#include "rt_types.h"

#ifdef __cplusplus
extern "C" {
#endif

#define ORACLE_MISS 0xFFFFFFFFu

typedef struct oracle_params {
    uint32_t width, height;        /* frame size; flat pixel index = width*y + x */
    uint32_t max_depth;            /* push constant max_depth */
    uint32_t tile_width, tile_height; /* one random_seed per (pass, tile) */
    uint32_t first_pass;           /* global index of the first pass rendered */
    uint32_t pass_count;           /* passes to add to the accumulator */
    uint32_t preview;              /* 1: store the colour instead of accumulating */
    const uint32_t* seeds;         /* pass_count * tiles_x * tiles_y words:
                                      seeds[(pass - first_pass) * tiles + ty * tiles_x + tx] */
    uint32_t rect_x0, rect_y0, rect_x1, rect_y1; /* pixel rectangle [x0,x1) x [y0,y1);
                                                     all zero = whole frame */
    uint32_t threads;              /* OpenMP threads; 0 = runtime default */
    uint32_t sky_importance;       /* opt-in extension, not in the reference: light samples on an
                                      environment-mapped sky are drawn in proportion to the map
                                      (luminance x sin theta per texel) and the sky pdf is that
                                      density; 0 = the shader's uniform sampling */
} oracle_params;

/* megakernel_raytracer.comp:85-114 for every pixel of the rectangle and every
 * pass.  accum is width*height RGBA32F, read-modify-written like out_img[0]. */
int oracle_render(const rt_scene_desc* scene, const rt_camera* camera,
    const oracle_params* params, float* accum, rt_stats* stats);

/* closest_hit (megakernel_raytracer.comp:184-323) of the jittered primary ray
 * of pass params->first_pass.  Outputs are width*height; misses get
 * ORACLE_MISS ids and t = 1e14.  bary may be NULL (else 3 floats / pixel). */
int oracle_primary_ids(const rt_scene_desc* scene, const rt_camera* camera,
    const oracle_params* params, uint32_t* instance_ids, uint32_t* triangle_ids,
    float* hit_t, float* bary);

/* Sensitivity probe: while on, one transcendental result in eight (sin, cos, acos, atan2, pow,
 * log) is replaced by a neighbouring float, as a different conforming math library might
 * return.  Process-wide; not for golden or baseline runs. */
void oracle_set_ulp_noise(int on);

/* --- function-level probes for known-answer tests --- */

/* random.glsl:4-24.  Starting from `seed`, writes n rand_01() values and the
 * state after each draw. */
void oracle_rand01_sequence(uint32_t seed, uint32_t n, float* values, uint32_t* states);
/* random.glsl:8-10 applied to `seed` with pixel index v. */
uint32_t oracle_hash_combine(uint32_t seed, uint32_t v);
/* random.glsl:12-17: returns value, updates *seed. */
uint32_t oracle_rand_uint(uint32_t* seed, uint32_t lo, uint32_t hi);

/* ray.glsl:11-31.  box = {x_min,x_max,y_min,y_max,z_min,z_max}. */
int oracle_hit_aabb(const float box[6], const float origin[3], const float direction[3],
    float t_min, float t_max);
/* ray.glsl:119-183.  tri = 9 floats a,b,c.  out = {b0,b1,b2,t}; returns hit flag. */
int oracle_hit_triangle(const float tri[9], const float origin[3], const float direction[3],
    float t_min, float t_max, float out[4]);
/* ray.glsl:47-105. */
int oracle_hit_triangle_quick(const float tri[9], const float origin[3],
    const float direction[3], float t_min, float t_max);

/* disney.glsl:170-248 / 250-315 on a surface built from `material` the way
 * get_surface_info does for an untextured front-face hit (eta = 1/ior).
 * normal, v_in (incoming ray direction, pointing at the surface) and l are
 * world-space unit vectors.  eval writes {f.x,f.y,f.z,pdf}. */
void oracle_eval_disney(const rt_material* material, int front_face, const float normal[3],
    const float v_in[3], const float l[3], float out[4]);
/* Draws from *seed (3 or 4 rand_01) and writes the sampled direction. */
void oracle_sample_disney(const rt_material* material, int front_face, const float normal[3],
    const float v_in[3], uint32_t* seed, float out_l[3]);

/* Bilinear REPEAT fetch used for texture(): out = rgba. */
void oracle_sample_texture(const rt_texture* texture, float u, float v, float out[4]);
/* eval_sky_light (megakernel_raytracer.comp:413-438) of the scene's sky: radiance rgb, pdf. */
void oracle_eval_sky_light(const rt_scene_desc* scene, const float direction[3], float out[4]);
/* The same with the opt-in sky distribution (oracle_params.sky_importance): its pdf. */
void oracle_eval_sky_light_importance(const rt_scene_desc* scene, const float direction[3],
    float out[4]);
void oracle_eval_sky_light_many(const rt_scene_desc* scene, int importance, uint32_t n,
    const float* directions, float* out);
void oracle_sample_sky_importance(const rt_scene_desc* scene, uint32_t* seed, float out[3]);
void oracle_sample_sky_importance_many(const rt_scene_desc* scene, uint32_t seed, uint32_t n,
    float* directions, float* pdfs);

#ifdef __cplusplus
}
#endif

#endif /* RT_ORACLE_H */

/*
 * C ABI of the MI355X (gfx950) path-tracing library, librt_hip.so.
 *
 * This is the drop-in boundary for the reference's ray-tracing back end: the
 * calls below are what the six `renderer` callbacks
 * (src/renderer/renderer.h:5-17, implemented for Vulkan in
 * src/renderer/megakernel_raytracer.cpp:525-822) need from a device layer.
 * Only plain pointers, sizes and PODs cross it; every function returns 0 on
 * success and a non-zero code on failure, with rt_hip_last_error() holding a
 * message for the calling thread.  A renderer handle is used from one host
 * thread at a time (the reference's callbacks are never re-entered), and its
 * calls are ordered on the device: a call on another stream than the handle's
 * previous call first waits for that call's work, because every call shares
 * the handle's path state and counters.  Calls leave the calling thread's
 * current HIP device as they found it.
 *
 * Buffers named *_dev are device pointers on the renderer's GPU; everything
 * else is host memory that the call has finished reading when it returns.
 */
#ifndef RT_HIP_H
#define RT_HIP_H

#include <stddef.h>
#include <stdint.h>

#include "rt_types.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rt_hip_renderer rt_hip_renderer;

enum {
    /* Walk the uploaded TLAS/BLAS nodes in the reference shader's order
     * (megakernel_raytracer.comp:184-276): hit ids, t and barycentrics are
     * bit-identical to the reference algorithm, ties included. */
    RT_HIP_ACCEL_REFERENCE = 0,
    /* Rebuild the acceleration structure at upload into a layout made for the
     * GPU (see rt_hip_upload_scene).  Same triangles and the same
     * intersection arithmetic, different visiting order: results differ only
     * where two hits tie at the same t or a box test grazes. */
    RT_HIP_ACCEL_REBUILT = 1
};

enum {
    RT_HIP_FLAG_COUNT_NODES = 1u, /* also count BVH nodes fetched / triangles tested */
    /* rt_hip_render normally runs as a pipeline of small kernels (trace / shade per
     * bounce, paths handed over through queues in device memory).  With this flag it
     * runs as ONE kernel that keeps a path in registers from the first ray to the last,
     * like the reference's shader; rt_hip_render_tile and rt_hip_render_preview always
     * do.  Both give bit-identical images. */
    RT_HIP_FLAG_FUSED = 2u,
    /* Also time the pipeline's trace and shade launches one by one (two HIP events per
     * launch); see rt_hip_kernel_time_of. */
    RT_HIP_FLAG_TIME_KERNELS = 4u,
    /* Ray binning in the pipeline: before each bounce's trace launch, regroup the ray queues
     * so that a wave traces rays that start in the same part of the scene and point the same
     * way.  It pays only where the trace launch works through its queue in fixed chunks of 64
     * rays and those rays are expensive and unlike each other; a trace launch that refills its
     * finished lanes (rt_hip_trace_refill) gains nothing from it.  Images are bit-identical
     * either way.  Default (neither flag): decided per uploaded scene and per bounce from the
     * first two large render calls - the first runs with fixed chunks and unbinned queues and
     * is timed bounce by bounce, the second refills; a bounce is binned when fixed chunks won
     * it and its rays cost enough, see
     * rt_hip_ray_binning.  The flags force it on (every bounce) or off. */
    RT_HIP_FLAG_BIN_RAYS = 8u,
    RT_HIP_FLAG_NO_BIN_RAYS = 16u,
    /* Opt-in, changes the image's noise (not what it converges to being the point): light
     * samples on an environment-mapped sky are drawn in proportion to the map's brightness
     * (a luminance x sin(theta) distribution over its texels, built at upload) instead of the
     * reference's uniform direction (megakernel_raytracer.comp:447-457), and the sky's pdf
     * (:431-436) becomes the density of that distribution, for the light sample and for the
     * MIS weight of BSDF-sampled rays that reach the sky alike.  Without the flag, and for a
     * sky without a map, sampling is the reference's. */
    RT_HIP_FLAG_SKY_IMPORTANCE = 32u,
    /* How the pipeline's trace launches hand out rays (see rt_hip_trace_refill): every bounce
     * with refilled lanes, or every bounce in fixed chunks of 64 rays.  Default (neither flag):
     * measured per uploaded scene and bounce.  Images are bit-identical either way; with both
     * policies forced (these and the binning flags) a renderer's throughput no longer depends
     * on a measurement. */
    RT_HIP_FLAG_TRACE_REFILL = 64u,
    RT_HIP_FLAG_TRACE_CHUNKS = 128u
};

enum {
    RT_HIP_BINNING_UNDECIDED = 0, /* no render call large enough to measure yet */
    RT_HIP_BINNING_OFF = 1,
    RT_HIP_BINNING_ON = 2
};

enum {
    RT_HIP_REFILL_UNDECIDED = 0, /* not yet measured on this scene: refilling is used meanwhile */
    RT_HIP_REFILL_OFF = 1,
    RT_HIP_REFILL_ON = 2
};

enum {
    RT_HIP_KERNEL_RENDER = 0, /* everything one rt_hip_render* call enqueues */
    RT_HIP_KERNEL_EXTEND = 1, /* one trace launch of the pipeline (one bounce of a batch) */
    RT_HIP_KERNEL_SHADE = 2   /* one shade launch of the pipeline */
};

#define RT_HIP_MISS 0xFFFFFFFFu

typedef struct rt_hip_config {
    uint32_t width, height;           /* render_options.resolution_x/y */
    uint32_t max_depth;               /* render_options.max_depth */
    uint32_t tile_width, tile_height; /* render_options.tile_width/height */
    int32_t device;                   /* HIP device ordinal; -1 = current device */
    uint32_t accel;                   /* RT_HIP_ACCEL_* */
    uint32_t flags;                   /* RT_HIP_FLAG_* */
} rt_hip_config;

/* Replaces megakernel_raytracer_initialize (megakernel_raytracer.cpp:525-553)
 * plus the accumulation / preview image creation of :314-324.  Allocates the
 * W x H RGBA32F accumulator and the (W/10) x (H/10) preview image. */
int rt_hip_create(const rt_hip_config* config, rt_hip_renderer** out);

/* Replaces prepare_megakernel_raytracer_resources (:269-467): copies every
 * buffer the compute pass binds to the device.  May be called again to
 * replace the scene.  With RT_HIP_ACCEL_REBUILT the TLAS/BLAS are rebuilt
 * from the uploaded instances and triangles; reported ids still index the
 * uploaded (reference-ordered) arrays. */
int rt_hip_upload_scene(rt_hip_renderer* r, const rt_scene_desc* scene);

/* Makes the per-dispatch random_seed words (push constant, :650-658) of passes
 * [first_pass, first_pass + pass_count) resident: seeds[(p - first_pass) *
 * tiles + tile], tiles numbered row-major. */
int rt_hip_upload_seeds(rt_hip_renderer* r, const uint32_t* seeds, uint32_t first_pass,
    uint32_t pass_count);

/* Replaces clearColorImage of the accumulation image (:628-649) and resets
 * the pass counter.  accum_dev = NULL means the renderer's own accumulator. */
int rt_hip_clear(rt_hip_renderer* r, float* accum_dev, void* stream);

/* The hot path.  Replaces pass_count x tiles dispatchBase calls (:650-669):
 * traces passes [first_pass, first_pass + pass_count) of every pixel in
 * rect = {x0, y0, x1, y1} (NULL = whole frame) and adds them to the
 * accumulator in pass order, exactly as that many sequential dispatches
 * would.  Seeds must have been uploaded for those passes.  Asynchronous on
 * `stream` (a hipStream_t; NULL = the default stream).
 * accum_dev: W*H*4 floats on the device, or NULL for the renderer's own.
 * A call of at least 2^24 paths (pixels x passes) is traced as two or more batches of
 * passes, two of them in flight at a time: one on `stream`, the other on a stream of the
 * renderer's own that starts after everything queued on `stream` before the call and that
 * `stream` waits for before the call's work counts as done - towards the caller the call
 * behaves like any other work queued on `stream`.  The second batch fills the GPU while the
 * first one's late bounces (few paths, dependent launches) leave it mostly idle. */
int rt_hip_render(rt_hip_renderer* r, const rt_camera* camera, uint32_t first_pass,
    uint32_t pass_count, const uint32_t* rect, float* accum_dev, void* stream);

/* Optional: allocates now what rt_hip_render would otherwise allocate on its first call (and
 * grow, with a device synchronisation, whenever a larger call arrives): the pipeline's path
 * state for calls of up to pass_count passes over rectangles of up to width x height pixels
 * (160 B per path in flight, at most 2^28 paths per batch - less when the device is short of
 * memory or the environment variable RT_HIP_MAX_BATCH_PATHS says so -, two batches - two sets
 * of state - when the call is large enough to be traced as two, see rt_hip_render; plus the ray-binning
 * buffers unless binning is forced off).  After it, render calls within that size allocate nothing.  No counterpart in
 * the reference, whose images and buffers are all created in prepare_data (:269-467). */
int rt_hip_reserve(rt_hip_renderer* r, uint32_t width, uint32_t height, uint32_t pass_count);

/* (diagnostics, no counterpart) What the uploaded scene's materials and lights can make the
 * shading code do - bit 0: a material with spec_trans != 0 (glass lobe), 1: clearcoat != 0,
 * 2: metallic != 0 or a metallic-roughness texture, 3: any texture on a material or light,
 * 4: a sky light - and the feature set of the pipeline's shading kernels chosen for it (the
 * smallest compiled set that covers those bits; code a scene cannot reach is compiled out,
 * which cannot change a result). */
int rt_hip_shade_features(rt_hip_renderer* r, uint32_t* scene_features, uint32_t* kernel_features);

/* Optional: settles, before the first render that counts, what rt_hip_render otherwise
 * measures on its first two large calls (at least 2^22 paths each) after every scene upload:
 * per bounce, whether the trace launch hands out rays in fixed chunks or refills finished
 * lanes, and whether the ray queues are regrouped first.  Those two calls run 5-40 % below
 * the speed of every later one (the first in fixed chunks throughout, the second refilling
 * throughout); images never depend on the outcome.  This entry makes the same two calls -
 * pass_count passes of the whole frame from `camera`, into a scratch image that is freed again,
 * with a constant seed - and waits for each, so a render call that follows runs at the settled
 * speed.  Accumulator and pass counter are untouched; the pipeline state is allocated as by
 * rt_hip_reserve(width, height, pass_count).  pass_count x frame below 2^22 paths measures
 * nothing (such calls always refill).  No counterpart in the reference. */
int rt_hip_tune(rt_hip_renderer* r, const rt_camera* camera, uint32_t pass_count, void* stream);

/* rt_hip_render for callers that cannot allocate device memory while they render: code under
 * HIP graph capture, tight serving loops, harnesses that forbid hipMalloc.  No counterpart in
 * the reference.  Same arguments, same result bit for bit, but the call allocates nothing,
 * creates no events and records no timings: the pipeline's path state and queues live in a
 * fixed device array inside the library (about 41 MB per GPU the library runs on, held until
 * the process ends) that holds RT_HIP_NOALLOC_MAX_PATHS paths.  One pass of the rectangle,
 * rounded up to whole 8x8 pixel blocks, must fit (e.g. 512 x 512 pixels, or the reference's
 * 128 x 72 tiles many passes at a time); larger rectangles and max_depth above
 * RT_HIP_NOALLOC_MAX_DEPTH are refused with an error.  Always runs the pipeline of kernels
 * (RT_HIP_FLAG_FUSED and ray binning do not apply).  All renderers on one GPU share the
 * array: calls that overlap on one GPU (different handles on different streams) race. */
#define RT_HIP_NOALLOC_MAX_PATHS (1u << 18)
#define RT_HIP_NOALLOC_MAX_DEPTH 63u
int rt_hip_render_noalloc(rt_hip_renderer* r, const rt_camera* camera, uint32_t first_pass,
    uint32_t pass_count, const uint32_t* rect, float* accum_dev, void* stream);

/* One reference dispatch: one pass of one tile with an explicit random_seed
 * (:650-666), into the renderer's own accumulator.  When the tile is the last
 * of the frame the pass counter advances (:667-669). */
int rt_hip_render_tile(rt_hip_renderer* r, const rt_camera* camera, uint32_t random_seed,
    uint32_t tile_x, uint32_t tile_y, void* stream);

/* The camera.dirty branch (:600-619): one sample per pixel of the
 * (W/10) x (H/10) preview image, stored (not accumulated); resets the
 * accumulation state like the reference does. */
int rt_hip_render_preview(rt_hip_renderer* r, const rt_camera* preview_camera,
    uint32_t random_seed, void* stream);

/* closest_hit() of every pixel's primary ray of pass `pass` (seeds must be
 * resident).  Host outputs, W*H each (bary: 3 floats per pixel); any may be
 * NULL.  Misses report RT_HIP_MISS ids and t = 1e14. */
int rt_hip_primary_ids(rt_hip_renderer* r, const rt_camera* camera, uint32_t pass,
    uint32_t* instance_ids, uint32_t* triangle_ids, float* hit_t, float* bary);

/* Known-answer probes (no counterpart in the reference): evaluates ONE device function of the
 * path - the same code the render kernels call - on each of `count` caller-supplied records
 * and returns the results, so that each function can be checked against the reference shader's
 * definition on its own.  `in` and `out` are host arrays.  Kinds and their record layout:
 *
 *   RAND01          u[0] = seed                      -> f[k] = k-th rand_01, u[k] = state after it (k < 16)
 *   HASH_COMBINE    u[0] = seed, u[1] = value        -> u[0]
 *   RAND_UINT       u[0] = seed, u[1] = lo, u[2] = hi -> u[0] = value, u[1] = state after
 *   HIT_AABB        f[0..5] = x_min x_max y_min y_max z_min z_max, f[6..8] = origin,
 *                   f[9..11] = direction, f[12] = t_min, f[13] = t_max
 *                                                    -> u[0] = hit_aabb (ray.glsl:11-31),
 *                                                       u[1] = the rebuilt structure's widened test (t_min 0)
 *   HIT_TRIANGLE    f[0..8] = a b c, f[9..11] = origin, f[12..14] = direction, f[15] = t_max
 *                                                    -> u[0] = hit, f[0..3] = b0 b1 b2 t
 *   EVAL_DISNEY     material (no textures), u[0] = front_face, f[0..2] = normal,
 *                   f[3..5] = incoming ray direction, f[6..8] = L  -> f[0..3] = f * |cos|, pdf
 *   SAMPLE_DISNEY   material, u[0] = front_face, u[1] = seed, f[0..2] = normal,
 *                   f[3..5] = incoming ray direction -> f[0..2] = L, u[0] = state after
 *   SAMPLE_TEXTURE  u[0] = texture index of the uploaded scene, f[0] = u, f[1] = v -> f[0..3] = texel
 *   SKY             f[0..2] = direction              -> f[0..3] = radiance, pdf (uploaded scene's sky)
 *
 *   SKY_SAMPLE      u[0] = seed                      -> f[0..2] = direction drawn from the opt-in sky
 *                                                       distribution, f[3] = its pdf, u[0] = state after
 *   TRIANGLE_DEGENERATE  f[0..8] = a b c             -> u[0] = hit_triangle's first test (ray.glsl:124-128:
 *                                                       edge cross product shorter than 1e-8), the one
 *                                                       rt_hip_triangle_never_hit evaluates on the host
 *
 * SAMPLE_TEXTURE, SKY and SKY_SAMPLE need an uploaded scene (SKY_SAMPLE one uploaded with
 * RT_HIP_FLAG_SKY_IMPORTANCE and an environment map); the others work on any renderer. */
enum {
    RT_HIP_PROBE_RAND01 = 0,
    RT_HIP_PROBE_HASH_COMBINE = 1,
    RT_HIP_PROBE_RAND_UINT = 2,
    RT_HIP_PROBE_HIT_AABB = 3,
    RT_HIP_PROBE_HIT_TRIANGLE = 4,
    RT_HIP_PROBE_EVAL_DISNEY = 5,
    RT_HIP_PROBE_SAMPLE_DISNEY = 6,
    RT_HIP_PROBE_SAMPLE_TEXTURE = 7,
    RT_HIP_PROBE_SKY = 8,
    RT_HIP_PROBE_SKY_SAMPLE = 9,
    RT_HIP_PROBE_TRIANGLE_DEGENERATE = 10
};
typedef struct rt_hip_probe_in {
    uint32_t u[4];
    float f[16];
    rt_material material;
} rt_hip_probe_in;
typedef struct rt_hip_probe_out {
    uint32_t u[16];
    float f[16];
} rt_hip_probe_out;
int rt_hip_probe(rt_hip_renderer* r, uint32_t kind, uint32_t count, const rt_hip_probe_in* in,
    rt_hip_probe_out* out);

/* Parity probe, host side (no counterpart in the reference): the test by which the rebuilt
 * acceleration structure leaves triangles out of its leaves - hit_triangle's degenerate-triangle
 * rejection (ray.glsl:124-128), evaluated once per triangle at upload with the device's float
 * operations.  corners: 9 floats (a, b, c) per triangle; out[i] = 1 when triangle i can never be
 * hit.  Must agree with RT_HIP_PROBE_TRIANGLE_DEGENERATE bit for bit: a disagreement would make
 * the rebuilt mode hit a triangle the shader rejects, or lose one it hits. */
int rt_hip_triangle_never_hit(const float* corners, uint32_t count, uint32_t* out);

/* Synchronises `stream`, then copies the renderer's own accumulator (W*H*4 floats)
 * / preview image ((W/10)*(H/10)*4 floats) to the host.  *passes = number
 * of complete passes accumulated so far (the reference's accumulation_counter). */
int rt_hip_read_accum(rt_hip_renderer* r, float* rgba, uint32_t* passes, void* stream);
int rt_hip_read_preview(rt_hip_renderer* r, float* rgba, void* stream);

/* Device address of the renderer's own accumulator (for a collective). */
int rt_hip_accum_ptr(rt_hip_renderer* r, void** accum_dev);

/* Multi-GPU (no counterpart in the reference, which drives one device): the scene is uploaded
 * to one renderer per GPU (rt_hip_config.device), each traces its own range of passes of the
 * whole frame into its own accumulator, and this call sums the accumulators onto
 * renderers[0]'s: renderers that share a GPU are added by a kernel, the per-GPU sums are
 * combined by ONE RCCL sum-reduce (W*H*4 floats; xGMI between the GPUs of a node).  The other
 * renderers are left cleared with a pass count of zero, so accumulation can go on;
 * renderers[0]'s pass count becomes the total and its alpha channel stays 1.  Waits for all
 * outstanding work of every listed renderer and returns when the sum is complete.  librccl is
 * loaded on first use and only when two GPUs are involved. */
int rt_hip_reduce(rt_hip_renderer* const* renderers, uint32_t count);

/* Complete passes accumulated so far in the renderer's own accumulator (the
 * reference's accumulation_counter, megakernel_raytracer.cpp:109, :667-669).
 * Does not synchronise. */
int rt_hip_passes(rt_hip_renderer* r, uint32_t* passes);

/* Tell the renderer how many complete passes its accumulator holds (used
 * after an external reduce into it). */
int rt_hip_set_passes(rt_hip_renderer* r, uint32_t passes);

/* Work counters since creation or the last reset; synchronises the device.
 * seconds = device time of the rt_hip_render* launches, from HIP events
 * recorded on the launch stream. */
int rt_hip_stats(rt_hip_renderer* r, rt_stats* out, int reset);

/* Average device time, in milliseconds, of what one rt_hip_render* call enqueues
 * (from HIP events recorded on the launch stream) since the last reset of the
 * statistics, and how many calls that covers. */
int rt_hip_kernel_time(rt_hip_renderer* r, double* avg_ms, uint32_t* launches);

/* The same for one kind of launch (RT_HIP_KERNEL_*).  The per-launch kinds are only
 * recorded with RT_HIP_FLAG_TIME_KERNELS. */
int rt_hip_kernel_time_of(rt_hip_renderer* r, uint32_t kernel, double* avg_ms,
    uint32_t* launches);

/* A render call of two or more batches of passes runs two batches at a time, side by side on
 * two streams (rt_hip_render), so launches of one kind overlap and their durations add up to
 * more than the time the GPU spent on them.  busy_ms = the time, since the last reset of the
 * statistics, during which AT LEAST ONE launch of that kind was running (the union of the
 * launches' intervals).  Needs RT_HIP_FLAG_TIME_KERNELS for the per-launch kinds. */
int rt_hip_kernel_busy_of(rt_hip_renderer* r, uint32_t kernel, double* busy_ms);

/* Diagnostics of the pipeline: in the most recent batch of passes - in the most recent TWO,
 * added up, when the last render call ran its batches two at a time - how many paths were
 * alive in bounce `trip` (counts[0]) and how many closest-hit (counts[1]) and occlusion
 * (counts[2]) rays it traced.  Synchronises the device.  No counterpart in the reference. */
int rt_hip_trip_counts(rt_hip_renderer* r, uint32_t trip, uint32_t* counts);

/* Diagnostics of the pipeline with RT_HIP_FLAG_COUNT_NODES and the rebuilt acceleration
 * structure: what the traversal fetched since the statistics were last reset, and how it was
 * served.  nodes / tris repeat rt_stats.nodes_visited / tris_tested; the trace launches read
 * nodes_from_lds of the nodes and instance_records_from_lds of the instance records from the
 * workgroup's LDS copy of the most visited ones (node_bytes_cached of node_bytes), the rest
 * through the cache hierarchy; finish_* is the share of the kernel that runs the last few
 * paths of a batch to their end.  *_bytes: size of the structure as traversed (the reference
 * layout in RT_HIP_ACCEL_REFERENCE mode).  Synchronises the device.  No counterpart in the
 * reference. */
typedef struct rt_hip_trace_work_counts {
    uint64_t nodes, tris;
    uint64_t nodes_from_lds;
    uint64_t instance_records, instance_records_from_lds;
    uint64_t finish_nodes, finish_tris;
    uint64_t node_bytes, node_bytes_cached, triangle_bytes, instance_bytes;
    /* trace launches: how often a wave executed a node step / a triangle test / an instance
     * entry for any of its lanes; (nodes - finish_nodes) / (64 * node_wave_steps) is the share
     * of lanes in use in the node steps, and so on */
    uint64_t node_wave_steps, triangle_wave_steps, instance_wave_steps;
} rt_hip_trace_work_counts;
int rt_hip_trace_work(rt_hip_renderer* r, rt_hip_trace_work_counts* out);

/* Diagnostics of the pipeline: whether ray binning is in use for the uploaded scene
 * (RT_HIP_BINNING_*: ON = for at least one bounce; UNDECIDED until the second large render
 * call has been measured) and, when it was decided by measurement, the measured cost of
 * tracing one ray of the third bounce in fixed chunks without binning, in nanoseconds of
 * whole-GPU time (0 when forced by a flag or not yet measured).  No counterpart in the
 * reference. */
int rt_hip_ray_binning(rt_hip_renderer* r, uint32_t* state, double* probe_ns_per_ray);

/* Diagnostics of the pipeline: how its trace launches hand out rays on the uploaded scene
 * (RT_HIP_REFILL_*).  A wave either takes fixed chunks of 64 rays or gives each lane that has
 * finished its ray the next one (once half its lanes wait); which is faster depends on how
 * unequal the rays' walks are, so the first two large render calls of a scene are timed, one
 * each way, and the faster is kept bounce by bounce (ON = some bounce refills).  ms_off /
 * ms_on: the summed trace launches of those two calls, -1 until measured.  trip_mask: bit b set
 * = the trace launch of bounce b refills (bit 31 stands for every later bounce).  Out pointers
 * other than `state` may be null.  Images are bit-identical either way.  No counterpart in the
 * reference. */
int rt_hip_trace_refill(rt_hip_renderer* r, uint32_t* state, double* ms_off, double* ms_on,
    uint32_t* trip_mask);

/* Replaces megakernel_raytracer_destroy (:793-813). */
int rt_hip_destroy(rt_hip_renderer* r);

const char* rt_hip_last_error(void);

#ifdef __cplusplus
}
#endif

#endif /* RT_HIP_H */

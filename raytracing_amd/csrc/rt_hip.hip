// librt_hip.so: kernels and the C ABI declared in include/rt_hip.h.
//
// Built for gfx950 only, with -ffp-contract=off: every fused multiply-add in
// the kernels is written explicitly (rt_traverse_fast.h), everything else is
// plain IEEE binary32 so results can be compared with the CPU oracle.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <rccl/rccl.h>  // types and prototypes only: the library is loaded on first use

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <vector>

#include "accel_build.h"
#include "rt_hip.h"
#include "rt_path.h"
#include "rt_probe.h"
#include "rt_wavefront.h"

using namespace rtd;

// The shading kernels of the pipeline, one instantiation per feature set: each set is compiled
// in its own translation unit (shade_variant.hip).  RT_EXP_F (diagnostic builds, e.g. with
// RT_SHADE_REGION_TIMING, whose cycle counters are per translation unit): one set only, compiled
// here, used for every scene.
#ifdef RT_EXP_F
template void rtd::launch_shade_stage<(uint32_t) (RT_EXP_F)>(wavefront_params const&, uint32_t,
    dim3, dim3, uint32_t, hipStream_t);
#else
#define RT_EXTERN_SHADE(F)                                                                    \
    extern template void rtd::launch_shade_stage<F>(wavefront_params const&, uint32_t, dim3,  \
        dim3, uint32_t, hipStream_t);
RT_SHADE_VARIANTS(RT_EXTERN_SHADE)
#undef RT_EXTERN_SHADE
#endif

// ---------------------------------------------------------------------------
// kernels

namespace {

// samples, closest, shadow, hits, nodes, tris, max_stack, shadow_skipped
#ifdef RT_PHASE_TIMING
// Diagnostic build: words 8.. hold per-wave cycle sums of the kernel's phases
// (occlusion walks, closest-hit walks, everything else, whole wave).
constexpr uint32_t kCounterWords = 16;
#else
constexpr uint32_t kCounterWords = 8;
#endif
constexpr uint32_t kPreviewRatio = 10; // src/renderer/megakernel_raytracer.cpp:54

// Pixel owned by this lane: a wave covers one 8x8 pixel block of the rectangle.
// Workgroups are dealt round-robin over the 8 XCDs (ids equal mod 8 share an
// XCD and its L2).  Block rows are interleaved over the XCDs: ids 8j..8j+7 are
// the same block column in 8 consecutive block rows, so every XCD walks the
// whole picture in 8-pixel-high strips.  Each L2 still sees spatially coherent
// rays, and an expensive region of the image is shared by all XCDs instead of
// leaving seven of them idle (contiguous per-XCD bands measured ~1.4 resident
// waves per SIMD on average because of that tail).
//
// When the previous launch of the same frame left per-block costs behind
// (render_params::block_order), blocks are instead taken most-expensive first:
// the hardware hands the next workgroup to whichever slot frees up, so a
// longest-first order keeps all slots busy until the very end (the cost of a
// block barely changes from one launch to the next: same camera, same scene,
// different random numbers).  `block` receives the linear block index.
__device__ __forceinline__ bool lane_pixel(render_params const& p, uint32_t& x, uint32_t& y,
    uint32_t& block) {
    uint32_t const blocks_x = (p.x1 - p.x0 + 7u) / 8u;
    uint32_t const blocks_y = (p.y1 - p.y0 + 7u) / 8u;
    uint32_t const i = blockIdx.x;
    uint32_t bx, by;
    if (p.block_order != nullptr) {
        uint32_t const b = p.block_order[i];
        bx = b % blocks_x;
        by = b / blocks_x;
    } else {
        uint32_t const full_rows = blocks_y & ~7u;  // rows that interleave evenly
        uint32_t const full_blocks = full_rows * blocks_x;
        if (i < full_blocks) {
            uint32_t const j = i / 8u;
            bx = j % blocks_x;
            by = (j / blocks_x) * 8u + (i % 8u);
        } else {
            uint32_t const t = i - full_blocks;
            bx = t % blocks_x;
            by = full_rows + t / blocks_x;
        }
    }
    block = by * blocks_x + bx;
    uint32_t const lane = threadIdx.x;
    x = p.x0 + bx * 8u + (lane & 7u);
    y = p.y0 + by * 8u + (lane >> 3);
    return x < p.x1 && y < p.y1;
}

// Turns the per-block costs of the launch that just finished into the block
// order of the next one (descending cost), and clears the costs.  One
// workgroup: a 256-bucket counting sort on cost relative to the maximum; the
// order inside a bucket does not matter.
constexpr uint32_t kOrderThreads = 1024;
constexpr uint32_t kOrderBuckets = 256;

__global__ void __launch_bounds__(kOrderThreads) order_blocks_kernel(uint32_t* cost,
    uint32_t* order, uint32_t n) {
    __shared__ uint32_t s_max;
    __shared__ uint32_t s_count[kOrderBuckets];
    __shared__ uint32_t s_start[kOrderBuckets];
    uint32_t const tid = threadIdx.x;
    if (tid == 0) {
        s_max = 1u;
    }
    if (tid < kOrderBuckets) {
        s_count[tid] = 0u;
    }
    __syncthreads();
    uint32_t local_max = 0u;
    for (uint32_t i = tid; i < n; i += kOrderThreads) {
        local_max = max(local_max, cost[i]);
    }
    atomicMax(&s_max, local_max);
    __syncthreads();
    uint32_t const top = s_max;
    auto bucket_of = [top](uint32_t c) -> uint32_t {
        // bucket 0 = most expensive
        uint32_t const q = (uint32_t) (((unsigned long long) c * (kOrderBuckets - 1u)) / top);
        return (kOrderBuckets - 1u) - min(q, kOrderBuckets - 1u);
    };
    for (uint32_t i = tid; i < n; i += kOrderThreads) {
        atomicAdd(&s_count[bucket_of(cost[i])], 1u);
    }
    __syncthreads();
    if (tid == 0) {
        uint32_t running = 0u;
        for (uint32_t b = 0; b < kOrderBuckets; ++b) {
            s_start[b] = running;
            running += s_count[b];
        }
    }
    __syncthreads();
    for (uint32_t i = tid; i < n; i += kOrderThreads) {
        uint32_t const slot = atomicAdd(&s_start[bucket_of(cost[i])], 1u);
        order[slot] = i;
    }
    __syncthreads();
    for (uint32_t i = tid; i < n; i += kOrderThreads) {
        cost[i] = 0u;
    }
}

// Second launch-bound argument = minimum waves per SIMD the register allocation must allow.
#ifndef RT_WAVES_PER_SIMD
#define RT_WAVES_PER_SIMD 2
#endif

template <int ACCEL, bool COUNT>
__global__ void __launch_bounds__(64, RT_WAVES_PER_SIMD) path_trace_kernel(render_params const p) {
    __shared__ uint32_t lds_stack[ACCEL == RT_HIP_ACCEL_REBUILT ? kFastLdsEntries * 64u : 64u];
    fast_stack stk;
    stk.lds = lds_stack + threadIdx.x;

    uint32_t x, y, block;
    if (!lane_pixel(p, x, y, block)) {
        return;
    }
    unsigned long long const started = __builtin_readcyclecounter();
    scene_dev const& sc = p.sc;
    float* const px = p.accum + 4u * ((size_t) y * p.frame_width + x);
    f3 acc = splat3(0.0f);
    if (!p.preview) {
        float4 const old = *reinterpret_cast<const float4*>(px);
        acc = mk3(old.x, old.y, old.z);
    }

    trace_counters cnt;
    uint32_t closest_rays = 0, shadow_rays = 0, closest_hits = 0, shadow_skipped = 0;

    // Each trip of the loop traces one ray pair - the previous bounce's occlusion ray, then
    // this bounce's closest-hit ray - and shades the hit (shade_trip).
    uint32_t pass = 0;
    ray_pair rp;
    rp.has_shadow = false;
    rp.has_closest = true;
    path_state ps;
    ps.nee = splat3(0.0f);
    ps.radiance = splat3(0.0f);
    ps.throughput = splat3(1.0f);
    ps.last_bsdf_pdf = 0.0f;
    ps.depth = 0;
    primary_ray(p, x, y, pass_seed(p, 0u, x, y), ps.seed, rp.origin, rp.direction);

    for (;;) {
        pair_result pr;
        closest_rays += rp.has_closest ? 1u : 0u;
        shadow_rays += rp.has_shadow ? 1u : 0u;
#ifdef RT_PHASE_TIMING
        {
            unsigned long long const t0 = __builtin_readcyclecounter();
            hit_result unused;
            pr.occluded = trace_fast<true, COUNT>(sc, stk, rp.has_shadow, rp.shadow_origin,
                rp.shadow_direction, rp.shadow_t_max, unused, cnt);
            unsigned long long const t1 = __builtin_readcyclecounter();
            pr.hit = trace_fast<false, COUNT>(
                sc, stk, rp.has_closest, rp.origin, rp.direction, kInfinity, pr.best, cnt);
            unsigned long long const t2 = __builtin_readcyclecounter();
            phase_shadow += t1 - t0;
            phase_closest += t2 - t1;
            phase_trips += 1;
            phase_shadow_lanes += __builtin_popcountll(__ballot(rp.has_shadow));
            phase_closest_lanes += __builtin_popcountll(__ballot(rp.has_closest));
        }
#else
        trace_pair<ACCEL, COUNT>(sc, stk, rp, pr, cnt);
#endif
        shade_trip(sc, p.max_depth, rp, pr, ps, closest_hits, shadow_skipped);
        if (!rp.has_closest && !rp.has_shadow) {
            // the path is over and its last occlusion ray is resolved
            acc = p.preview ? ps.radiance : acc + ps.radiance;
            ++pass;
            if (pass == p.pass_count) {
                break;
            }
            ps.radiance = splat3(0.0f);
            ps.throughput = splat3(1.0f);
            ps.depth = 0;
            rp.has_closest = true;
            primary_ray(p, x, y, pass_seed(p, pass, x, y), ps.seed, rp.origin, rp.direction);
        }
    }
    *reinterpret_cast<float4*>(px) = make_float4(acc.x, acc.y, acc.z, 1.0f);

#ifdef RT_PHASE_TIMING
    if (p.counters && threadIdx.x == 0) {
        atomicAdd(&p.counters[8], phase_shadow);
        atomicAdd(&p.counters[9], phase_closest);
        atomicAdd(&p.counters[10], (unsigned long long) (__builtin_readcyclecounter() - started));
        atomicAdd(&p.counters[11], phase_trips);
        atomicAdd(&p.counters[12], phase_shadow_lanes);
        atomicAdd(&p.counters[13], phase_closest_lanes);
    }
#endif
    if (p.block_cost) {
        // how long this block's wave took: it orders the next launch (see lane_pixel)
        unsigned long long const spent = __builtin_readcyclecounter() - started;
        atomicMax(&p.block_cost[block], (uint32_t) min(spent, 0xFFFFFFFFull));
    }
    if (p.counters) {
        // one atomic per lane and counter is plenty: this runs once per pixel per launch
        atomicAdd(&p.counters[0], (unsigned long long) p.pass_count);
        atomicAdd(&p.counters[1], (unsigned long long) closest_rays);
        atomicAdd(&p.counters[2], (unsigned long long) shadow_rays);
        atomicAdd(&p.counters[3], (unsigned long long) closest_hits);
        atomicAdd(&p.counters[7], (unsigned long long) shadow_skipped);
        if (COUNT) {
            atomicAdd(&p.counters[4], (unsigned long long) cnt.nodes);
            atomicAdd(&p.counters[5], (unsigned long long) cnt.tris);
            atomicMax(&p.counters[6], (unsigned long long) cnt.max_stack);
        }
    }
}

template <int ACCEL>
__global__ void __launch_bounds__(64) primary_ids_kernel(render_params const p) {
    __shared__ uint32_t lds_stack[ACCEL == RT_HIP_ACCEL_REBUILT ? kFastLdsEntries * 64u : 64u];
    fast_stack stk;
    stk.lds = lds_stack + threadIdx.x;
    uint32_t x, y, block;
    if (!lane_pixel(p, x, y, block)) {
        return;
    }
    uint32_t seed;
    ray_pair rp;
    rp.has_shadow = false;
    rp.has_closest = true;
    primary_ray(p, x, y, pass_seed(p, 0u, x, y), seed, rp.origin, rp.direction);
    pair_result pr;
    trace_counters cnt;
    trace_pair<ACCEL, false>(p.sc, stk, rp, pr, cnt);
    bool const hit = pr.hit;
    hit_result const& hr = pr.best;
    size_t const i = (size_t) y * p.frame_width + x;
    if (p.out_instance) {
        p.out_instance[i] = hit ? hr.instance : RT_HIP_MISS;
    }
    if (p.out_triangle) {
        p.out_triangle[i] = hit ? hr.triangle : RT_HIP_MISS;
    }
    if (p.out_t) {
        p.out_t[i] = hit ? hr.t : kInfinity;
    }
    if (p.out_bary) {
        p.out_bary[3 * i + 0] = hit ? hr.b0 : 0.0f;
        p.out_bary[3 * i + 1] = hit ? hr.b1 : 0.0f;
        p.out_bary[3 * i + 2] = hit ? hr.b2 : 0.0f;
    }
}

// The shader's inverse(mat3(transform)) (megakernel_raytracer.comp:299, :474), once per
// transform instead of once per hit: the same inverse33 on the same three columns, compiled
// without contraction like the shading code, so the table holds the bits a hit would compute.
__global__ void normal_matrix_kernel(const rt_mat4* transforms, uint32_t count,
    rtd::normal_matrix* out) {
    uint32_t const i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) {
        return;
    }
    rt_mat4 const& m = transforms[i];
    rtd::m33 m3;
    m3.c0 = rtd::mk3(m.m[0][0], m.m[0][1], m.m[0][2]);
    m3.c1 = rtd::mk3(m.m[1][0], m.m[1][1], m.m[1][2]);
    m3.c2 = rtd::mk3(m.m[2][0], m.m[2][1], m.m[2][2]);
    rtd::m33 const inv = rtd::inverse33(m3);
    out[i].c0 = make_float4(inv.c0.x, inv.c0.y, inv.c0.z, 0.0f);
    out[i].c1 = make_float4(inv.c1.x, inv.c1.y, inv.c1.z, 0.0f);
    out[i].c2 = make_float4(inv.c2.x, inv.c2.y, inv.c2.z, 0.0f);
}

// Fills the derived fields of the device's material table (derive_material).
__global__ void derive_materials_kernel(rt_material* materials, uint32_t count) {
    uint32_t const i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) {
        rt_material m = materials[i];
        rtd::derive_material(m);
        materials[i] = m;
    }
}

__global__ void clear_rgba_kernel(float4* img, size_t count) {
    size_t const i = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) {
        img[i] = make_float4(0.0f, 0.0f, 0.0f, 1.0f);
    }
}

// dst += src on the colour channels; alpha stays 1 (both hold vec4(colour, 1) sums).
__global__ void add_rgb_kernel(float4* dst, const float4* src, size_t count) {
    size_t const i = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) {
        float4 a = dst[i];
        float4 const b = src[i];
        a.x += b.x, a.y += b.y, a.z += b.z;
        a.w = 1.0f;
        dst[i] = a;
    }
}

__global__ void set_alpha_kernel(float4* img, size_t count) {
    size_t const i = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) {
        img[i].w = 1.0f;
    }
}

thread_local std::string g_last_error;

// Makes `device` current for the duration of one C ABI call and puts the caller's device
// back afterwards (a caller that also drives PyTorch or another renderer must not find its
// current device changed).
class device_scope {
public:
    explicit device_scope(int device) {
        if (hipGetDevice(&previous_) != hipSuccess) {
            previous_ = -1;
        }
        if (previous_ != device) {
            hipError_t const e = hipSetDevice(device);
            if (e != hipSuccess) {
                throw std::runtime_error(std::string{"hipSetDevice: "} + hipGetErrorString(e));
            }
            changed_ = true;
        }
    }
    ~device_scope() {
        if (changed_ && previous_ >= 0) {
            (void) hipSetDevice(previous_);
        }
    }
    device_scope(device_scope const&) = delete;
    device_scope& operator=(device_scope const&) = delete;

private:
    int previous_ = -1;
    bool changed_ = false;
};

struct hip_failure : std::runtime_error {
    using std::runtime_error::runtime_error;
};

void hip_check(hipError_t e, char const* what) {
    if (e != hipSuccess) {
        throw hip_failure(std::string{what} + ": " + hipGetErrorString(e));
    }
}
#define HIP_CHECK(expr) hip_check((expr), #expr)

template <typename F>
int guarded(F&& body) {
    try {
        body();
        return 0;
    } catch (std::exception const& e) {
        g_last_error = e.what();
    } catch (...) {
        g_last_error = "unknown error";
    }
    return 1;
}

}  // namespace

// ---------------------------------------------------------------------------
// renderer object

struct rt_hip_renderer {
    rt_hip_config cfg{};
    int device = 0;
    uint32_t tiles_x = 0, tiles_y = 0;
    uint32_t preview_w = 0, preview_h = 0;

    float* accum = nullptr;
    float* preview = nullptr;
    unsigned long long* counters = nullptr;

    // cost feedback for full-frame launches (see lane_pixel / order_blocks_kernel)
    uint32_t* block_cost = nullptr;
    uint32_t* block_order = nullptr;
    uint32_t block_total = 0;
    bool order_valid = false;
    rt_camera order_camera{};

    std::vector<void*> scene_allocs;
    scene_dev sc{};
    bool scene_ready = false;
    uint64_t accel_node_bytes = 0, accel_tri_bytes = 0, accel_instance_bytes = 0;  // as traversed
    // what the scene's materials and lights can make the shading code do (kFeat* bits) and the
    // compiled feature set of the pipeline's shading kernels chosen for it
    uint32_t shade_features = kShadeAll, shade_variant = kShadeAll;

    uint32_t* seeds = nullptr;
    size_t seeds_capacity = 0;
    uint32_t seeds_first_pass = 0, seeds_pass_count = 0;

    uint32_t passes_done = 0;
    uint32_t tile_cursor = 0;  // tiles dispatched in the current pass (render_tile cadence)

    // wavefront pipeline state (allocated on first use, see launch_wavefront): two sets, so that
    // two batches of one render call can be in flight side by side (set 1 only exists once a
    // call had two batches)
    std::vector<void*> wf_allocs[2];
    wavefront_params wf_set[2] = {};
    wavefront_params& wf = wf_set[0];
    size_t wf_paths[2] = {0, 0};
    uint32_t wf_trips[2] = {0, 0};
    uint32_t wf_sets_in_last_call = 1;  // rt_hip_trip_counts adds up their last batches
    hipStream_t side_stream = nullptr;  // batches on set 1 run here
    hipEvent_t fork_event = nullptr, join_event = nullptr, resolved_event[2] = {nullptr, nullptr};
    uint32_t wf_extend_grid = 0, wf_extend_waves = 0, wf_shade_grid = 0;  // workgroups, waves
    uint32_t wf_settle_grid = 0;
    uint32_t wf_finish_grid = 0;
    uint32_t wf_stat_waves = 0;       // slots of wf.wave_stats
    int handover_setting = -1;        // RT_HIP_HANDOVER_BELOW at creation (tuning knob), -1 = default
    wavefront_params noalloc{};       // the same arrays inside this device's scratch array
    uint32_t wf_bin_groups = 0;       // workgroups of the ray binning launches
    bool wf_binned[2] = {false, false};  // the binning buffers exist
    float scene_lo[3] = {0, 0, 0}, scene_hi[3] = {1, 1, 1};  // world bounds of the scene
    // Ray binning policy for the uploaded scene (RT_HIP_BINNING_*).  Undecided: the next
    // large enough render call runs without binning and times the trace launch of every
    // trip; the call after that reads the result and decides trip by trip.
    uint32_t binning = RT_HIP_BINNING_UNDECIDED;
    std::vector<char> bin_trip;        // per trip: regroup its ray queues (when binning is ON)
    // How the trace launches hand out rays (RT_HIP_REFILL_*): in fixed chunks of 64, or with a
    // wave refilling its finished lanes.  Refilling wins where rays need very different numbers
    // of steps (the 10 000-instance scene: +37 %) and costs a few percent where binned chunks
    // are coherent (an interior), so it is measured like the binning: once the binning is
    // settled, one large call is timed in each mode and the faster one kept.
    uint32_t refill = RT_HIP_REFILL_UNDECIDED;
    double refill_ms[2] = {-1.0, -1.0};  // summed trace launches of the timed call: off, on
    std::vector<float> refill_trip_ms[2];  // the same per trip
    std::vector<uint64_t> refill_trip_rays[2];  // rays those launches traced, per trip
    size_t refill_base_paths = 0;          // paths per batch of the call that took the base line
    std::vector<uint8_t> refill_trip;      // the verdict per trip (trips beyond it: the last one's)
    uint32_t probe_kind = 0;           // what the probe in flight measures: 1 binning, 2 / 3 refill off / on
    bool bin_candidates = false;       // bin_trip holds the bounces whose rays are dear enough
    bool probe_pending = false;
    uint32_t probe_trips = 0;          // trips timed by the probe in flight
    std::vector<hipEvent_t> probe_events;  // 2 per trip (start, stop), created on demand
    hipEvent_t probe_done = nullptr;
    uint32_t* probe_counts = nullptr;  // pinned copy of the control block: [trip][kCtlWords]
    uint32_t probe_counts_trips = 0;
    double probe_ns_per_ray = 0.0;     // of trip kProbeTrip (reported by rt_hip_ray_binning)

    struct timed_launch {
        hipEvent_t start, stop;
        uint32_t kind;  // RT_HIP_KERNEL_*
    };
    std::vector<timed_launch> launches;        // not yet folded into the totals
    std::vector<timed_launch> event_pool;      // reusable events
    double seconds_total[3] = {0.0, 0.0, 0.0};  // per RT_HIP_KERNEL_*
    uint32_t launches_total[3] = {0, 0, 0};
    // Two batches of a call run side by side, so launches of one kind overlap: per kind, also
    // the time during which AT LEAST ONE such launch was running (the union of the intervals),
    // on a time axis that starts at `time_base`.
    double busy_seconds[3] = {0.0, 0.0, 0.0};
    double busy_until_ms[3] = {0.0, 0.0, 0.0};  // end of the union so far
    hipEvent_t time_base = nullptr;

    // One handle is driven through one stream at a time: a call on another stream than the
    // previous one first waits (on the device) for that call's work, because all calls share
    // the handle's path state, queues and counters.
    hipEvent_t last_done = nullptr;
    hipStream_t last_stream = nullptr;
    bool last_recorded = false;

    void follow(hipStream_t stream) {
        if (last_recorded && stream != last_stream) {
            HIP_CHECK(hipStreamWaitEvent(stream, last_done, 0));
        }
    }
    void mark(hipStream_t stream) {
        HIP_CHECK(hipEventRecord(last_done, stream));
        last_stream = stream;
        last_recorded = true;
    }

    template <typename T>
    T* upload(T const* host, size_t count) {
        if (count == 0) {
            // keep a valid, never-dereferenced pointer
            count = 1;
            void* d = nullptr;
            HIP_CHECK(hipMalloc(&d, sizeof(T)));
            HIP_CHECK(hipMemset(d, 0, sizeof(T)));
            scene_allocs.push_back(d);
            return static_cast<T*>(d);
        }
        void* d = nullptr;
        HIP_CHECK(hipMalloc(&d, count * sizeof(T)));
        scene_allocs.push_back(d);
        HIP_CHECK(hipMemcpy(d, host, count * sizeof(T), hipMemcpyHostToDevice));
        return static_cast<T*>(d);
    }

    void free_scene() {
        for (void* p : scene_allocs) {
            (void) hipFree(p);
        }
        scene_allocs.clear();
        scene_ready = false;
    }

    void free_wavefront(int set) {
        for (void* p : wf_allocs[set]) {
            (void) hipFree(p);
        }
        wf_allocs[set].clear();
        wf_paths[set] = 0;
        wf_trips[set] = 0;
        wf_binned[set] = false;
    }
    void free_wavefront() {
        free_wavefront(0);
        free_wavefront(1);
    }

    timed_launch begin_timing(hipStream_t stream, uint32_t kind = RT_HIP_KERNEL_RENDER) {
        timed_launch t{};
        if (!event_pool.empty()) {
            t = event_pool.back();
            event_pool.pop_back();
        } else {
            HIP_CHECK(hipEventCreate(&t.start));
            HIP_CHECK(hipEventCreate(&t.stop));
        }
        t.kind = kind;
        if (time_base == nullptr) {
            HIP_CHECK(hipEventCreate(&time_base));
            HIP_CHECK(hipEventRecord(time_base, stream));
        }
        HIP_CHECK(hipEventRecord(t.start, stream));
        return t;
    }
    void end_timing(timed_launch t, hipStream_t stream) {
        HIP_CHECK(hipEventRecord(t.stop, stream));
        launches.push_back(t);
        if (launches.size() >= 256) {
            fold_timings();
        }
    }
    void fold_timings() {
        struct interval {
            double from, to;
            uint32_t kind;
        };
        std::vector<interval> spans;
        spans.reserve(launches.size());
        for (timed_launch& t : launches) {
            HIP_CHECK(hipEventSynchronize(t.stop));
            float ms = 0.0f, from = 0.0f;
            HIP_CHECK(hipEventElapsedTime(&ms, t.start, t.stop));
            HIP_CHECK(hipEventElapsedTime(&from, time_base, t.start));
            seconds_total[t.kind] += (double) ms * 1e-3;
            ++launches_total[t.kind];
            spans.push_back({(double) from, (double) from + (double) ms, t.kind});
            event_pool.push_back(t);
        }
        launches.clear();
        // union per kind: intervals in order of their start, each counted from where the union
        // so far ends
        std::sort(spans.begin(), spans.end(),
            [](interval const& a, interval const& b) { return a.from < b.from; });
        for (interval const& v : spans) {
            double const from = std::max(v.from, busy_until_ms[v.kind]);
            if (v.to > from) {
                busy_seconds[v.kind] += (v.to - from) * 1e-3;
                busy_until_ms[v.kind] = v.to;
            }
        }
    }
};

namespace {

void validate_scene(rt_scene_desc const& s) {
    auto need = [](bool ok, char const* what) {
        if (!ok) {
            throw std::runtime_error(std::string{"upload_scene: "} + what);
        }
    };
    need(s.tlas && s.tlas_count > 0, "empty TLAS");
    need(s.blas && s.blas_count > 0, "empty BLAS");
    need(s.meshes && s.mesh_count > 0, "no meshes");
    need(s.instances && s.instance_count > 0, "no instances");
    need(s.triangles && s.triangle_count > 0, "no triangles");
    need(s.lights && s.light_count > 0,
        "no lights (the reference host dereferences lights.back() and the shader divides by "
        "light_count)");
    need(s.sky_light < (int32_t) s.light_count, "sky_light out of range");
    need(s.transform_count == 0 || (s.transforms && s.inverse_transforms), "missing transforms");
    need(s.material_count == 0 || s.materials, "missing materials");
    need(s.texture_count == 0 || s.textures, "missing textures");
    // Every index a kernel will follow is checked here, on the host, once.
    for (uint32_t i = 0; i < s.tlas_count; ++i) {
        rt_bvh_node const& n = s.tlas[i];
        if (n.obj_count > 0) {
            need((uint64_t) n.first_obj + n.obj_count <= s.instance_count, "TLAS leaf out of range");
        } else {
            need(i + 1 < s.tlas_count && n.right < s.tlas_count && n.right > i,
                "TLAS child out of range");
            need(n.split_axis >= 0 && n.split_axis <= 2, "TLAS split axis");
        }
    }
    for (uint32_t i = 0; i < s.blas_count; ++i) {
        rt_bvh_node const& n = s.blas[i];
        if (n.obj_count > 0) {
            need((uint64_t) n.first_obj + n.obj_count <= s.triangle_count, "BLAS leaf out of range");
        } else {
            need(i + 1 < s.blas_count && n.right < s.blas_count && n.right > i,
                "BLAS child out of range");
            need(n.split_axis >= 0 && n.split_axis <= 2, "BLAS split axis");
        }
    }
    for (uint32_t i = 0; i < s.mesh_count; ++i) {
        need(s.meshes[i].bvh_start < s.blas_count, "mesh BLAS root out of range");
        need(s.meshes[i].triangle_count > 0 &&
                 (uint64_t) s.meshes[i].triangle_offset + s.meshes[i].triangle_count <=
                     s.triangle_count,
            "mesh triangle range");
    }
    for (uint32_t i = 0; i < s.instance_count; ++i) {
        rt_instance const& in = s.instances[i];
        need(in.mesh < s.mesh_count, "instance mesh");
        need(in.transform < (int32_t) s.transform_count, "instance transform");
        need(in.material < (int32_t) s.material_count, "instance material");
        need(in.light < (int32_t) s.light_count, "instance light");
    }
    auto tex_ok = [&](int32_t t) { return t < (int32_t) s.texture_count; };
    for (uint32_t i = 0; i < s.material_count; ++i) {
        rt_material const& m = s.materials[i];
        need(tex_ok(m.albedo_tex) && tex_ok(m.emission_tex) && tex_ok(m.normal_tex) &&
                 tex_ok(m.metallic_roughness_tex),
            "material texture index");
    }
    for (uint32_t i = 0; i < s.light_count; ++i) {
        rt_light const& l = s.lights[i];
        need(tex_ok(l.emission_tex), "light texture index");
        if (l.type == RT_LIGHT_AREA_SINGLE_SIDED || l.type == RT_LIGHT_AREA_DOUBLE_SIDED) {
            need(l.mesh < s.mesh_count, "light mesh");
            need(l.transform < (int32_t) s.transform_count, "light transform");
        }
    }
    for (uint32_t i = 0; i < s.texture_count; ++i) {
        need(s.textures[i].data && s.textures[i].width > 0 && s.textures[i].height > 0 &&
                 s.textures[i].format <= RT_TEX_F32X4,
            "texture");
    }
}

// What the scene's materials and lights can make the shading code do (rt_shading.h: kFeat*).
// A feature the scene cannot use is compiled out of the pipeline's shading kernels; every test
// here is on the uploaded values themselves, so no path of the scene can reach removed code.
uint32_t scene_features(rt_scene_desc const& s, uint32_t flags) {
    uint32_t f = 0;
    for (uint32_t i = 0; i < s.material_count; ++i) {
        rt_material const& m = s.materials[i];
        // (anything but an exact zero counts, NaN included)
        if (!(m.spec_trans == 0.0f)) {
            f |= kFeatGlass;
        }
        if (!(m.clearcoat == 0.0f)) {
            f |= kFeatClearcoat;
        }
        if (!(m.metallic == 0.0f) || m.metallic_roughness_tex >= 0) {
            f |= kFeatMetal;
        }
        if (m.albedo_tex >= 0 || m.emission_tex >= 0 || m.normal_tex >= 0 ||
            m.metallic_roughness_tex >= 0) {
            f |= kFeatTextures;
        }
    }
    for (uint32_t i = 0; i < s.light_count; ++i) {
        if (s.lights[i].emission_tex >= 0) {
            f |= kFeatTextures;
        }
        if (s.lights[i].type == RT_LIGHT_SKY) {
            f |= kFeatSky;
        }
    }
    if (s.sky_light >= 0) {
        f |= kFeatSky;
    }
    return f;
}

// Tables for the opt-in importance sampling of the environment map (rt_shading.h:
// sample_sky_importance): a piecewise-constant distribution over the map's texels, weighted by
// the luminance of what eval_sky_light returns for the texel (intensity x Reinhard-mapped
// texel, megakernel_raytracer.comp:426-430) times sin(theta) of the texel's row (equal areas
// of the (u, v) square are not equal solid angles).  Sums run in double, tables are float.
struct sky_tables {
    uint32_t width = 0, height = 0;
    std::vector<float> density;      // [h][w]: pdf over the unit (u, v) square
    std::vector<float> conditional;  // [h][w + 1]: normalised CDF along each row
    std::vector<float> marginal;     // [h + 1]: CDF over rows
};

sky_tables build_sky_tables(rt_light const& sky, rt_texture const& map) {
    sky_tables t;
    uint32_t const w = map.width, h = map.height;
    t.width = w;
    t.height = h;
    t.density.assign((size_t) w * h, 0.0f);
    t.conditional.assign((size_t) h * (w + 1u), 0.0f);
    t.marginal.assign((size_t) h + 1u, 0.0f);
    std::vector<double> row_sum(h, 0.0);
    auto channel = [&](size_t texel, int c) -> float {
        if (map.format == RT_TEX_F32X4) {
            return static_cast<const float*>(map.data)[texel * 4 + c];
        }
        return (float) static_cast<const uint8_t*>(map.data)[texel * 4 + c] / 255.0f;
    };
    double total = 0.0;
    for (uint32_t y = 0; y < h; ++y) {
        float const sin_theta = sinf(kPi * ((float) y + 0.5f) / (float) h);
        double running = 0.0;
        for (uint32_t x = 0; x < w; ++x) {
            size_t const i = (size_t) y * w + x;
            float rgb[3];
            for (int c = 0; c < 3; ++c) {
                float const v = channel(i, c);
                rgb[c] = sky.intensity[c] * (v / (v + 1.0f));
            }
            float const lum = (0.212671f * rgb[0] + 0.715160f * rgb[1]) + 0.072169f * rgb[2];
            float const f = fmaxf(lum, 0.0f) * sin_theta;
            t.density[i] = f;
            t.conditional[(size_t) y * (w + 1u) + x] = (float) running;
            running += (double) f / (double) w;
        }
        row_sum[y] = running;
        float* const row = &t.conditional[(size_t) y * (w + 1u)];
        for (uint32_t x = 0; x <= w; ++x) {
            row[x] = running > 0.0 ? (float) ((double) row[x] / running) : (float) x / (float) w;
        }
        row[w] = 1.0f;
        t.marginal[y] = (float) total;
        total += running / (double) h;
    }
    if (!(total > 0.0)) {
        throw std::runtime_error("upload_scene: the environment map is black; it cannot be "
                                 "importance sampled");
    }
    for (uint32_t y = 0; y < h; ++y) {
        t.marginal[y] = (float) ((double) t.marginal[y] / total);
    }
    t.marginal[h] = 1.0f;
    for (float& f : t.density) {
        f = (float) ((double) f / total);
    }
    return t;
}

uint32_t ref_depth(rt_bvh_node const* nodes, uint32_t root) {
    uint32_t deepest = 0;
    std::vector<std::pair<uint32_t, uint32_t>> todo{{root, 1u}};
    while (!todo.empty()) {
        auto const [idx, depth] = todo.back();
        todo.pop_back();
        deepest = std::max(deepest, depth);
        if (nodes[idx].obj_count == 0) {
            todo.push_back({idx + 1, depth + 1});
            todo.push_back({nodes[idx].right, depth + 1});
        }
    }
    return deepest;
}

template <typename K>
void launch_rect(K kernel, render_params const& p, hipStream_t stream) {
    uint32_t const bx = (p.x1 - p.x0 + 7u) / 8u;
    uint32_t const by = (p.y1 - p.y0 + 7u) / 8u;
    hipLaunchKernelGGL(kernel, dim3(bx * by), dim3(64), 0, stream, p);
    HIP_CHECK(hipGetLastError());
}

void launch_path_trace(rt_hip_renderer& r, render_params& p, hipStream_t stream) {
    bool const count = (r.cfg.flags & RT_HIP_FLAG_COUNT_NODES) != 0;
    // Full-frame accumulation launches feed their per-block costs to the next one.
    bool const whole_frame = !p.preview && p.x0 == 0 && p.y0 == 0 && p.x1 == r.cfg.width &&
                             p.y1 == r.cfg.height && p.frame_width == r.cfg.width;
    if (whole_frame) {
        if (std::memcmp(&r.order_camera, &p.cam, sizeof(rt_camera)) != 0) {
            r.order_valid = false;  // another view: yesterday's costs say nothing
            r.order_camera = p.cam;
        }
        p.block_cost = r.block_cost;
        p.block_order = r.order_valid ? r.block_order : nullptr;
    }
    auto t = r.begin_timing(stream);
    if (r.cfg.accel == RT_HIP_ACCEL_REFERENCE) {
        if (count) {
            launch_rect(path_trace_kernel<RT_HIP_ACCEL_REFERENCE, true>, p, stream);
        } else {
            launch_rect(path_trace_kernel<RT_HIP_ACCEL_REFERENCE, false>, p, stream);
        }
    } else {
        if (count) {
            launch_rect(path_trace_kernel<RT_HIP_ACCEL_REBUILT, true>, p, stream);
        } else {
            launch_rect(path_trace_kernel<RT_HIP_ACCEL_REBUILT, false>, p, stream);
        }
    }
    r.end_timing(t, stream);
    if (whole_frame) {
        hipLaunchKernelGGL(order_blocks_kernel, dim3(1), dim3(kOrderThreads), 0, stream,
            r.block_cost, r.block_order, r.block_total);
        HIP_CHECK(hipGetLastError());
        r.order_valid = true;
    }
}

// Paths in flight per batch of the wavefront pipeline: at most 2^28 (160 B of state, lists and
// counters per path, 172 B with the ray binning's buffers: 43-46 GB per state set, two sets for
// two batches in flight, of 288 GB).  Large batches amortise launch tails and give the late,
// sparse bounces launches worth starting: per sample, 1280x720 renders 1.15x faster at 64 passes
// per batch than at 16, and 1.04x / 1.10x faster again at 128 / 256 (two batches in flight each
// time).  The state arrays are allocated for the batches a caller actually asks for, and a
// batch never takes more than its share of HALF the device memory that is free (batch_limit).
#ifndef RT_MAX_PATHS_LOG2
#define RT_MAX_PATHS_LOG2 28
#endif
constexpr size_t kWavefrontMaxPaths = (size_t) 1 << RT_MAX_PATHS_LOG2;
constexpr size_t kStateBytesPerPath = 176;

// Regrouping a trip's ray queues is worth its launches when tracing one of its rays costs more
// than kBinningPaysAboveNs nanoseconds of whole-GPU time and the trip has enough rays to pay for
// the fixed part (four small launches).  Measured on MI355X at 64 passes of 1280x720, unbinned
// cost per ray -> change of that trip's trace launch with binning:
//   staircase interior   bounce 1: 0.56 ns -> -8 %, bounces 2-7: 0.72-0.8 ns -> -40 %
//   mustang              bounce 1: 0.25 ns -> -23 %, bounce 2: 0.46 ns -> -33 %
//   hyperion             bounce 1: 0.09 ns -> +12 % (slower), bounces 2-3: 0.29 ns, 6.5 and 4.2 M
//                        rays -> -12 % / +10 %, a wash after the sort's own time
//   Cornell box          0.08-0.11 ns -> slower
constexpr double kBinningPaysAboveNs = 0.2;
constexpr uint32_t kBinningMinRays = 8u << 20;  // per trip
constexpr uint32_t kProbeTrip = 2;              // the trip whose cost rt_hip_ray_binning reports
constexpr uint32_t kProbeMinPaths = 4u << 20;   // smaller batches time the launches, not the rays

// The binning verdict, once the way the trace launches hand out rays is known: a bounce whose
// trace launch refills its lanes is not binned.  Refilling takes the cost of unequal rays out
// of the wave, which is what the regrouped queues were for, and leaves the regrouping's own
// launches as pure cost (measured with both on: staircase 309 against 342 Msamples/s without
// the binning, mustang 1551 against 1656, the 10 000-instance scene 835 against 865).
void settle_binning(rt_hip_renderer& r) {
    bool any = false;
    for (size_t trip = 0; trip < r.bin_trip.size(); ++trip) {
        bool refills = r.refill == RT_HIP_REFILL_ON;
        if (refills && !r.refill_trip.empty()) {
            refills = r.refill_trip[std::min(trip, r.refill_trip.size() - 1)] != 0;
        }
        if (refills) {
            r.bin_trip[trip] = 0;
        }
        any = any || r.bin_trip[trip] != 0;
    }
    r.binning = any ? RT_HIP_BINNING_ON : RT_HIP_BINNING_OFF;
}

// Rays the timed trace launch of bounce `trip` had in its two queues (from the copy of the
// control block that the probe left in pinned memory).
uint64_t probe_rays(rt_hip_renderer const& r, uint32_t trip) {
    const uint32_t* const ctl = r.probe_counts + trip * kCtlWords;
    return (uint64_t) ctl[kCtlClosest] + ctl[kCtlShadow];
}

// Settles an outstanding probe if its events have completed (never waits).
void settle_probe(rt_hip_renderer& r) {
    if (!r.probe_pending || hipEventQuery(r.probe_done) != hipSuccess) {
        return;
    }
    r.probe_pending = false;
    if (r.probe_kind != 1u) {
        int const mode = r.probe_kind == 3u ? 1 : 0;
        double total = 0.0;
        r.refill_trip_ms[mode].assign(r.probe_trips, 0.0f);
        r.refill_trip_rays[mode].assign(r.probe_trips, 0);
        for (uint32_t trip = 0; trip < r.probe_trips; ++trip) {
            float ms = 0.0f;
            HIP_CHECK(hipEventElapsedTime(&ms, r.probe_events[2 * trip], r.probe_events[2 * trip + 1]));
            r.refill_trip_ms[mode][trip] = ms;
            r.refill_trip_rays[mode][trip] = probe_rays(r, trip);
            total += (double) ms;
        }
        r.refill_ms[mode] = total;
        if (r.refill_ms[0] >= 0.0 && r.refill_ms[1] >= 0.0) {
            // Trip by trip (camera rays of one chunk walk alike, the rays of later bounces do
            // not), and per RAY: the two measurements come from two render calls, which may
            // differ in passes, rectangle or camera, so the launch times as such do not compare.
            size_t const n = std::min(r.refill_trip_ms[0].size(), r.refill_trip_ms[1].size());
            r.refill_trip.assign(n, 0);
            bool any = false;
            for (size_t trip = 0; trip < n; ++trip) {
                uint64_t const rays_off = r.refill_trip_rays[0][trip];
                uint64_t const rays_on = r.refill_trip_rays[1][trip];
                if (rays_off == 0 || rays_on == 0) {
                    // (a bounce that one of the calls handed to the finishing kernel: its launch
                    // returned at once; follow the bounce before it)
                    r.refill_trip[trip] = trip > 0 ? r.refill_trip[trip - 1] : 1;
                } else {
                    double const ns_off = (double) r.refill_trip_ms[0][trip] * 1e6 / (double) rays_off;
                    double const ns_on = (double) r.refill_trip_ms[1][trip] * 1e6 / (double) rays_on;
                    r.refill_trip[trip] = ns_on < ns_off ? 1 : 0;
                }
                any = any || r.refill_trip[trip] != 0;
            }
            r.refill = any ? RT_HIP_REFILL_ON : RT_HIP_REFILL_OFF;
            if (r.binning == RT_HIP_BINNING_UNDECIDED && r.bin_candidates) {
                settle_binning(r);
            }
        }
        return;
    }
    // kind 1: fixed chunks, queues as compacted - the base line of both decisions
    bool const refill_open = r.refill == RT_HIP_REFILL_UNDECIDED &&
                             r.cfg.accel == RT_HIP_ACCEL_REBUILT;
    r.bin_trip.assign(r.probe_trips, 0);
    if (refill_open) {
        r.refill_trip_ms[0].assign(r.probe_trips, 0.0f);
        r.refill_trip_rays[0].assign(r.probe_trips, 0);
        r.refill_ms[0] = 0.0;
    }
    for (uint32_t trip = 0; trip < r.probe_trips; ++trip) {
        float ms = 0.0f;
        HIP_CHECK(hipEventElapsedTime(&ms, r.probe_events[2 * trip], r.probe_events[2 * trip + 1]));
        if (refill_open) {
            r.refill_trip_ms[0][trip] = ms;
            r.refill_trip_rays[0][trip] = probe_rays(r, trip);
            r.refill_ms[0] += (double) ms;
        }
        if (trip == 0u) {
            continue;  // camera rays are traced in pixel order
        }
        const uint32_t* const ctl = r.probe_counts + trip * kCtlWords;
        uint64_t const rays = (uint64_t) ctl[kCtlClosest] + ctl[kCtlShadow];
        double const ns = rays > 0 ? (double) ms * 1e6 / (double) rays : 0.0;
        if (trip == kProbeTrip) {
            r.probe_ns_per_ray = ns;
        }
        r.bin_trip[trip] = rays >= kBinningMinRays && ns > kBinningPaysAboveNs ? 1 : 0;
    }
    r.bin_candidates = true;
    if (!refill_open) {
        settle_binning(r);
    }
}

// Launch geometry of the pipeline's kernels on this renderer's device, their LDS allowances
// and the per-wave work counters: set up once, when the renderer is created, so that no render
// call has to (the allocation-free entry point may not).
void init_launch_plan(rt_hip_renderer& r) {
    hipDeviceProp_t prop{};
    HIP_CHECK(hipGetDeviceProperties(&prop, r.device));
    uint32_t const cus = (uint32_t) std::max(1, prop.multiProcessorCount);
    auto resident = [&](auto kernel) {
        int per_cu = 0;
        HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, 64, 0));
        return cus * (uint32_t) std::max(1, per_cu);
    };
    bool const count = (r.cfg.flags & RT_HIP_FLAG_COUNT_NODES) != 0;
    if (r.cfg.accel == RT_HIP_ACCEL_REFERENCE) {
        r.wf_extend_grid = count ? resident(wf_extend_kernel<RT_HIP_ACCEL_REFERENCE, true, false>)
                                 : resident(wf_extend_kernel<RT_HIP_ACCEL_REFERENCE, false, false>);
        r.wf_extend_waves = r.wf_extend_grid;
    } else {
        // one workgroup per CU: it takes (nearly) all of the CU's LDS
        r.wf_extend_grid = cus;
        r.wf_extend_waves = cus * (kExtendThreads / 64u);
        auto allow = [&](auto kernel) {
            HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                hipFuncAttributeMaxDynamicSharedMemorySize, (int) kLdsBytesPerCu));
        };
        allow(wf_extend_kernel<RT_HIP_ACCEL_REBUILT, true, false>);
        allow(wf_extend_kernel<RT_HIP_ACCEL_REBUILT, false, false>);
        allow(wf_extend_kernel<RT_HIP_ACCEL_REBUILT, true, true>);
        allow(wf_extend_kernel<RT_HIP_ACCEL_REBUILT, false, true>);
    }
    // two workgroups of four waves per CU (the kernel's registers allow two waves per
    // SIMD), each with its own LDS copy of the scene's small tables
    r.wf_shade_grid = cus * kShadeGroupsPerCu;
    r.wf_settle_grid = cus * kSettleGroupsPerCu;
    r.wf_finish_grid = cus * 8u;  // one-wave workgroups, two per SIMD
    r.wf_bin_groups = cus;
    HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(wf_bin_count_kernel),
        hipFuncAttributeMaxDynamicSharedMemorySize, (int) (kBinCount * 4u)));
    HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(wf_bin_place_kernel),
        hipFuncAttributeMaxDynamicSharedMemorySize, (int) (kBinCount * 4u)));
    // per-wave work counters: they outlive every regrowth of the state arrays
    r.wf_stat_waves = std::max({r.wf_extend_waves, r.wf_shade_grid * (kShadeThreads / 64u),
        r.wf_settle_grid * (kSettleThreads / 64u), r.wf_finish_grid});
    size_t const stat_bytes = (size_t) r.wf_stat_waves * kStatWords * sizeof(unsigned long long);
    // (one array, one half per state set: batches in flight side by side count separately)
    HIP_CHECK(hipMalloc((void**) &r.wf.wave_stats, 2u * stat_bytes));
    HIP_CHECK(hipMemset(r.wf.wave_stats, 0, 2u * stat_bytes));
    r.wf_set[1].wave_stats = r.wf.wave_stats + (size_t) r.wf_stat_waves * kStatWords;
}

// A render call of at least this many paths is cut into two batches even when one would hold
// it (see "two batches in flight" in launch_wavefront): 1280x720 x 64 passes as 2 x 32 is 2 %
// faster than as one batch although the halves' launches are smaller.
constexpr size_t kSplitMinPaths = (size_t) 1 << 24;

// The multiplier and shift that divide() turns into floor(n / d) for n < 2^31 (rt_wavefront.h),
// checked against the division itself at the ends of the range and around multiples of d.
divider make_divider(uint32_t d) {
    if (d == 0u) {
        throw std::runtime_error("make_divider: zero");
    }
    uint32_t shift = 0;
    while ((2u << shift) <= d && shift < 31u) {
        ++shift;  // floor(log2 d)
    }
    divider v{};
    v.shift = shift;
    if ((d & (d - 1u)) == 0u) {
        v.mul = 0u, v.keep = ~0u;
    } else {
        unsigned long long const two_k = 1ull << (32u + shift);
        v.mul = (uint32_t) ((two_k + d - 1u) / d);
        v.keep = 0u;
    }
    auto quotient = [&](uint32_t n) {
        return (uint32_t) ((((unsigned long long) n * v.mul) >> 32) + (n & v.keep)) >> v.shift;
    };
    uint32_t const top = 0x7FFFFFFFu;
    uint32_t const probes[] = {0u, 1u, d - 1u, d, d + 1u, 2u * d - 1u, 2u * d, top, top - 1u,
        top / d * d, top / d * d - 1u, (top / 2u) / d * d, (top / 2u) / d * d - 1u, 0x0FFFFFFFu};
    for (uint32_t n : probes) {
        if (n <= top && quotient(n) != n / d) {
            throw std::logic_error("make_divider: wrong quotient for " + std::to_string(n) + " / " +
                                   std::to_string(d));
        }
    }
    return v;
}

// Points `w` at the state arrays, lists and counters of `set`.
void take_state_arrays(wavefront_params& w, wavefront_params const& set) {
    w.ray_o = set.ray_o, w.ray_d = set.ray_d, w.sh_o = set.sh_o, w.sh_d = set.sh_d;
    w.nee = set.nee, w.thr = set.thr, w.rad = set.rad, w.hit0 = set.hit0, w.hit1 = set.hit1;
    w.live = set.live, w.q_closest = set.q_closest;
    w.q_shadow = set.q_shadow, w.raw = set.raw, w.going = set.going;
    w.going_count = set.going_count, w.chunk_counts = set.chunk_counts;
    w.tile_sums = set.tile_sums, w.control = set.control;
    w.binned_closest = set.binned_closest, w.binned_shadow = set.binned_shadow;
    w.bin_keys_closest = set.bin_keys_closest, w.bin_keys_shadow = set.bin_keys_shadow;
    w.bin_hist = set.bin_hist, w.bin_total = set.bin_total;
    w.wave_stats = set.wave_stats;
}

void ensure_wavefront(rt_hip_renderer& r, int set, size_t paths, uint32_t trips, bool binned) {
    auto alloc = [&](size_t bytes) {
        void* d = nullptr;
        HIP_CHECK(hipMalloc(&d, bytes));
        r.wf_allocs[set].push_back(d);
        return d;
    };
    wavefront_params& w = r.wf_set[set];
    auto alloc_binning = [&](size_t for_paths) {
        w.binned_closest = (uint32_t*) alloc(for_paths * 4);
        w.binned_shadow = (uint32_t*) alloc(for_paths * 4);
        w.bin_keys_closest = (uint16_t*) alloc(for_paths * 2);
        w.bin_keys_shadow = (uint16_t*) alloc(for_paths * 2);
        w.bin_hist = (uint32_t*) alloc((size_t) 2u * kBinCount * r.wf_bin_groups * 4);
        w.bin_total = (uint32_t*) alloc((size_t) 2u * kBinCount * 4);
        r.wf_binned[set] = true;
    };
    if (paths <= r.wf_paths[set] && trips <= r.wf_trips[set]) {
        if (binned && !r.wf_binned[set]) {
            alloc_binning(r.wf_paths[set]);  // binning was switched on: add its buffers, keep the rest
        }
        return;
    }
    paths = std::max(paths, r.wf_paths[set]);
    trips = std::max(trips, r.wf_trips[set]);
    binned = binned || r.wf_binned[set];
    HIP_CHECK(hipDeviceSynchronize());
    unsigned long long* const wave_stats = w.wave_stats;  // outlives the regrowth
    r.free_wavefront(set);
    w = wavefront_params{};
    w.wave_stats = wave_stats;
    w.ray_o = (float4*) alloc(paths * 16);
    w.ray_d = (float4*) alloc(paths * 16);
    w.sh_o = (float4*) alloc(paths * 16);
    w.sh_d = (float4*) alloc(paths * 16);
    w.nee = (float4*) alloc(paths * 16);
    w.thr = (float4*) alloc(paths * 16);
    w.rad = (float4*) alloc(paths * 16);
    w.hit0 = (float4*) alloc(paths * 16);
    w.hit1 = (float2*) alloc(paths * 8);
    w.live = (uint32_t*) alloc(paths * 4);
    w.q_closest = (uint32_t*) alloc(paths * 4);
    w.q_shadow = (uint32_t*) alloc(paths * 4);
    w.raw = (uint32_t*) alloc(paths * 4);
    w.going = (uint2*) alloc(paths * 8);
    w.going_count = (uint32_t*) alloc(((paths + kRegionSlots - 1u) / kRegionSlots) * 4);
    size_t const chunks = (paths + 63u) / 64u;
    w.chunk_counts = (uint32_t*) alloc(chunks * 4);
    w.tile_sums = (uint32_t*) alloc(((chunks + kTileChunks - 1u) / kTileChunks) * 12);
    w.control = (uint32_t*) alloc((size_t) (trips + 1u) * kCtlWords * 4);
    r.wf_paths[set] = paths;
    r.wf_trips[set] = trips;
    if (binned) {
        alloc_binning(paths);
    }
}

// The hard bound on a batch: kWavefrontMaxPaths, or less when the environment variable
// RT_HIP_MAX_BATCH_PATHS asks for it (a process that shares the device and wants this renderer's
// state arrays small; read on every call, so it can be changed between calls).
size_t batch_cap() {
    size_t cap = kWavefrontMaxPaths;
    if (const char* text = std::getenv("RT_HIP_MAX_BATCH_PATHS")) {
        char* end = nullptr;
        unsigned long long const asked = std::strtoull(text, &end, 0);
        if (end != text && *end == '\0' && asked >= 64u) {
            cap = std::min<size_t>(cap, (size_t) asked);
        }
    }
    return cap;
}

// Most paths one batch of a call of `pass_count` passes may hold right now: batch_cap(), or less
// when the device is short of memory - both state sets together (two batches in flight) stay
// within half of what is free, counting what the renderer holds already as free.  One pass of
// the rectangle is the smallest batch there is, so up to the cap `per_pass` is always allowed
// (its allocation then succeeds or fails as such); a pass beyond the cap gets the cap, and the
// caller refuses it.  The device is asked only when the call's batches would make the state
// arrays grow.
size_t batch_limit(rt_hip_renderer const& r, size_t per_pass, uint32_t pass_count) {
    size_t const cap = batch_cap();
    if (per_pass >= cap) {
        return cap;
    }
    // (two batches at least once the call is large enough to be split, see launch_wavefront)
    bool const split = pass_count >= 2u && per_pass * pass_count >= kSplitMinPaths;
    size_t const wanted = std::min(cap, per_pass * (split ? (pass_count + 1u) / 2u : pass_count));
    if (wanted <= (split ? std::min(r.wf_paths[0], r.wf_paths[1]) : r.wf_paths[0])) {
        return cap;
    }
    size_t free_bytes = 0, total_bytes = 0;
    if (hipMemGetInfo(&free_bytes, &total_bytes) != hipSuccess) {
        (void) hipGetLastError();
        return cap;
    }
    size_t const held = (r.wf_paths[0] + r.wf_paths[1]) * kStateBytesPerPath;
    size_t const affordable = (free_bytes + held) / 2u / (2u * kStateBytesPerPath);
    return std::min(cap, std::max(per_pass, affordable));
}

// rt_hip_render through the wavefront pipeline (rt_wavefront.h).
//
// `fixed` = nullptr: the pipeline state lives in the renderer's own device allocations, grown on
// demand (ensure_wavefront).  Otherwise `fixed` describes state arrays for `fixed_paths` paths
// that already exist (the allocation-free entry point): nothing is allocated, created or timed,
// the ray queues are never regrouped, and a rectangle that does not fit is refused.
void launch_wavefront(rt_hip_renderer& r, render_params const& p, hipStream_t stream,
    const wavefront_params* fixed = nullptr, size_t fixed_paths = 0) {
    uint32_t const blocks_x = (p.x1 - p.x0 + 7u) / 8u;
    uint32_t const blocks_y = (p.y1 - p.y0 + 7u) / 8u;
    size_t const per_pass = (size_t) blocks_x * blocks_y * 64u;
    static_assert(kWavefrontMaxPaths / 64u / kTileChunks <= kScanThreads * kScanTilesPerThread,
        "one workgroup must be able to scan the tile sums");
    static_assert(kWavefrontMaxPaths <= (size_t) kLivePidMask + 1u, "a pid must fit a list entry");
    size_t const limit = fixed ? fixed_paths : batch_limit(r, per_pass, p.pass_count);
    if (per_pass > limit) {
        // one pass of the rectangle is the smallest batch; the compaction (one workgroup scans
        // the tile sums) and the state arrays are sized for `limit` paths
        throw std::runtime_error("render: one pass of the rectangle is " +
                                 std::to_string(per_pass) + " paths, more than the " +
                                 std::to_string(limit) +
                                 " one batch can hold; render it as smaller rectangles");
    }
    // Batches: as few as the state arrays allow - but at least two when the call is large
    // enough, because two batches run side by side on two streams (below).
    size_t const most = std::max<size_t>(1, limit / per_pass);  // passes one batch can hold
    uint32_t batches = (uint32_t) ((p.pass_count + most - 1u) / most);
    if (!fixed && p.pass_count >= 2u && per_pass * p.pass_count >= kSplitMinPaths) {
        batches = std::max(batches, 2u);
    }
    uint32_t const batch = (p.pass_count + batches - 1u) / batches;
    uint32_t const trips = p.max_depth + 1u;
    if (!fixed) {
        settle_probe(r);
    }
    bool const binned = !fixed && r.binning == RT_HIP_BINNING_ON;
    // trips beyond those measured (a forced ON has measured none) are binned
    auto trip_binned = [&](uint32_t trip) {
        return binned && trip >= 1u && (trip >= r.bin_trip.size() || r.bin_trip[trip] != 0);
    };
    // measure this call?  (one batch is enough; a probe in flight is left alone)
    bool const large = per_pass * batch >= kProbeMinPaths;
    // The first large call runs with fixed chunks and the queues as compacted and is timed bounce
    // by bounce (kind 1): the base line.  The second refills (kind 3); then both policies are
    // settled.  (Kind 2: the base line again, when a flag has fixed the binning beforehand.)
    bool probe = !fixed && r.binning == RT_HIP_BINNING_UNDECIDED && !r.bin_candidates &&
                 !r.probe_pending && trips > 1u && large;
    uint32_t probe_kind = probe ? 1u : 0u;
    if (!fixed && !probe && !r.probe_pending && large &&
        (r.binning != RT_HIP_BINNING_UNDECIDED || r.bin_candidates || trips <= 1u) &&
        r.refill == RT_HIP_REFILL_UNDECIDED && r.cfg.accel == RT_HIP_ACCEL_REBUILT) {
        probe = true;
        // the refilling call is compared with the base line per ray, but the two should still be
        // of one order of size (launch tails, cache footprints): if this call's batch is not
        // within a factor of two of the base line's, it becomes the base line instead
        size_t const paths = per_pass * batch;
        bool const comparable = r.refill_ms[0] >= 0.0 && 2u * paths >= r.refill_base_paths &&
                                paths <= 2u * r.refill_base_paths;
        probe_kind = comparable ? 3u : 2u;
    }
    if (probe && probe_kind != 3u) {
        r.refill_base_paths = per_pass * batch;
        if (probe_kind == 2u) {
            r.refill_ms[0] = -1.0;
        }
    }
    // undecided (small calls, the allocation-free entry point before any large call): refill
    bool const refill = probe_kind == 3u ||
                        (probe_kind == 0u && r.refill != RT_HIP_REFILL_OFF);
    // trips beyond those measured (a forced ON has measured none) follow the last measured one
    auto trip_refills = [&](uint32_t trip) {
        if (!refill || probe_kind == 3u || r.refill_trip.empty()) {
            return refill;
        }
        return r.refill_trip[std::min<size_t>(trip, r.refill_trip.size() - 1)] != 0;
    };
    if (probe) {
        while (r.probe_events.size() < 2u * trips) {
            hipEvent_t e = nullptr;
            HIP_CHECK(hipEventCreate(&e));
            r.probe_events.push_back(e);
        }
        if (r.probe_counts_trips < trips) {
            if (r.probe_counts) {
                HIP_CHECK(hipHostFree(r.probe_counts));
                r.probe_counts = nullptr;
            }
            HIP_CHECK(hipHostMalloc((void**) &r.probe_counts, (size_t) trips * kCtlWords * 4));
            r.probe_counts_trips = trips;
        }
    }
    // Two batches in flight side by side.  Late bounces leave most of the GPU idle: few paths
    // are alive, every launch still waits for its slowest wave, and the launches of one batch
    // depend on each other.  A second batch with its own path state on a second stream fills
    // those gaps with its early, full launches (measured: two independent 64-pass renders of
    // the 1280x720 frame issued together finish 12 % sooner than one after the other; three or
    // four in flight are slower again).  Batch b runs on set b % 2; set 0 on the caller's
    // stream, set 1 on a stream of the renderer's own that starts after everything the caller
    // has queued so far and that the caller's stream waits for at the end.  A pixel's passes are
    // still added in pass order: each batch's resolve launch waits for the previous batch's.
    // A measured call (probe) keeps to one stream, so that its launch times mean something.
    uint32_t const lanes = (!fixed && !probe && p.pass_count > batch) ? 2u : 1u;
    if (!fixed) {
        ensure_wavefront(r, 0, per_pass * batch, trips, binned);
        if (lanes == 2u) {
            ensure_wavefront(r, 1, per_pass * batch, trips, binned);
            if (r.side_stream == nullptr) {
                HIP_CHECK(hipStreamCreateWithFlags(&r.side_stream, hipStreamNonBlocking));
                HIP_CHECK(hipEventCreateWithFlags(&r.fork_event, hipEventDisableTiming));
                HIP_CHECK(hipEventCreateWithFlags(&r.join_event, hipEventDisableTiming));
                for (hipEvent_t& e : r.resolved_event) {
                    HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
                }
            }
        }
        r.wf_sets_in_last_call = lanes;
    }
    hipStream_t lane_stream = stream;  // the stream of the batch being issued

    wavefront_params w = fixed ? *fixed : r.wf;
    w.sort_rays = 0u;  // set per trip below
    w.refill = 0u;     // likewise
    {
        // the grid of origin cells: kBinCellBits halvings, each of the axis whose cells are
        // longest, so that the cells come out roughly cubic whatever the scene's proportions
        float extent[3];
        uint32_t bits[3] = {0, 0, 0};
        for (int a = 0; a < 3; ++a) {
            extent[a] = std::max(r.scene_hi[a] - r.scene_lo[a], 0.0f);
        }
        for (uint32_t k = 0; k < kBinCellBits; ++k) {
            int longest = 0;
            for (int a = 1; a < 3; ++a) {
                if (extent[a] / (float) (1u << bits[a]) > extent[longest] / (float) (1u << bits[longest])) {
                    longest = a;
                }
            }
            ++bits[longest];
        }
        uint32_t shift = 0;
        for (int a = 0; a < 3; ++a) {
            w.sort_lo[a] = r.scene_lo[a];
            w.sort_scale[a] = extent[a] > 0.0f ? (float) (1u << bits[a]) / extent[a] : 0.0f;
            w.sort_top[a] = (float) ((1u << bits[a]) - 1u);
            w.sort_shift[a] = shift;
            shift += bits[a];
        }
    }
    w.sc = p.sc;
    w.cam = p.cam;
    w.accum = p.accum;
    w.scalar_seed = p.scalar_seed;
    w.frame_width = p.frame_width;
    w.x0 = p.x0;
    w.y0 = p.y0;
    w.x1 = p.x1;
    w.y1 = p.y1;
    w.blocks_x = blocks_x;
    w.block_count = blocks_x * blocks_y;
    w.padded = ((p.x1 - p.x0) % 8u != 0u || (p.y1 - p.y0) % 8u != 0u) ? 1u : 0u;
    w.tile_width = p.tile_width;
    w.tile_height = p.tile_height;
    w.tiles_x = p.tiles_x;
    w.tile_count = p.tile_count;
    w.by_per_pass = make_divider(w.block_count * 64u);
    w.by_blocks_x = make_divider(w.blocks_x);
    w.by_tile_width = make_divider(w.tile_width);
    w.by_tile_height = make_divider(w.tile_height);
    w.max_depth = p.max_depth;
    w.preview = p.preview;
    w.counters = p.counters;
    bool const count = (r.cfg.flags & RT_HIP_FLAG_COUNT_NODES) != 0;
    bool const each = !fixed && (r.cfg.flags & RT_HIP_FLAG_TIME_KERNELS) != 0;

    uint32_t const extend_lds =
        w.sc.cached_nodes * (uint32_t) sizeof(wide_node) + w.sc.cached_instances * 64u +
        (kExtendThreads / 64u) * w.sc.stack_entries * 256u;
    uint32_t const shade_lds = plan_shade_tables(w.sc, kShadeLdsBudget).total;
    auto extend = [&](uint32_t trip, uint32_t chunk_bound) {
        bool const reference = r.cfg.accel == RT_HIP_ACCEL_REFERENCE;
        uint32_t const per_group = reference ? 1u : kExtendThreads / 64u;
        dim3 const grid(std::min(r.wf_extend_grid, (chunk_bound + per_group - 1u) / per_group));
        dim3 const block(reference ? 64u : kExtendThreads);
        uint32_t const lds = reference ? 0u : extend_lds;
        auto go = [&](auto kernel) {
            hipLaunchKernelGGL(kernel, grid, block, lds, lane_stream, w, trip);
        };
        bool const first = trip == 0u;
        if (reference) {
            if (count) {
                first ? go(wf_extend_kernel<RT_HIP_ACCEL_REFERENCE, true, true>)
                      : go(wf_extend_kernel<RT_HIP_ACCEL_REFERENCE, true, false>);
            } else {
                first ? go(wf_extend_kernel<RT_HIP_ACCEL_REFERENCE, false, true>)
                      : go(wf_extend_kernel<RT_HIP_ACCEL_REFERENCE, false, false>);
            }
        } else {
            if (count) {
                first ? go(wf_extend_kernel<RT_HIP_ACCEL_REBUILT, true, true>)
                      : go(wf_extend_kernel<RT_HIP_ACCEL_REBUILT, true, false>);
            } else {
                first ? go(wf_extend_kernel<RT_HIP_ACCEL_REBUILT, false, true>)
                      : go(wf_extend_kernel<RT_HIP_ACCEL_REBUILT, false, false>);
            }
        }
        HIP_CHECK(hipGetLastError());
    };
    // Paths alive below which a batch is handed to the finishing kernel (decided on the
    // device after every compaction): 0.4 % of the batch, at least a GPU's worth of lanes
    // (1 % before the trace launches refilled their lanes: mustang -2 %, 4K scene -1.7 %).
    int const handover_setting = r.handover_setting;
    auto finish = [&](uint32_t trip, uint32_t chunk_bound) {
        dim3 const grid(std::max(1u, std::min(r.wf_finish_grid, chunk_bound)));
        if (r.cfg.accel == RT_HIP_ACCEL_REFERENCE) {
            if (count) {
                hipLaunchKernelGGL((wf_finish_kernel<RT_HIP_ACCEL_REFERENCE, true>), grid, dim3(64),
                    0, lane_stream, w, trip);
            } else {
                hipLaunchKernelGGL((wf_finish_kernel<RT_HIP_ACCEL_REFERENCE, false>), grid,
                    dim3(64), 0, lane_stream, w, trip);
            }
        } else {
            if (count) {
                hipLaunchKernelGGL((wf_finish_kernel<RT_HIP_ACCEL_REBUILT, true>), grid, dim3(64),
                    0, lane_stream, w, trip);
            } else {
                hipLaunchKernelGGL((wf_finish_kernel<RT_HIP_ACCEL_REBUILT, false>), grid, dim3(64),
                    0, lane_stream, w, trip);
            }
        }
        HIP_CHECK(hipGetLastError());
    };
    // Builds trip `trip`'s lists from the raw slots its producer left behind.
    auto compact = [&](uint32_t trip, const uint32_t* in_count_ptr, uint32_t in_count,
                       uint32_t chunk_bound, uint32_t handover_below) {
        dim3 const tiles(std::max(1u, (chunk_bound + kTileChunks - 1u) / kTileChunks));
        hipLaunchKernelGGL(wf_tile_sums_kernel, tiles, dim3(kTileThreads), 0, lane_stream, w,
            in_count_ptr, in_count);
        HIP_CHECK(hipGetLastError());
        hipLaunchKernelGGL(wf_scan_tiles_kernel, dim3(1), dim3(kScanThreads), 0, lane_stream, w, trip,
            in_count_ptr, in_count, handover_below);
        HIP_CHECK(hipGetLastError());
        hipLaunchKernelGGL(wf_scatter_kernel, tiles, dim3(kTileThreads), 0, lane_stream, w,
            in_count_ptr, in_count);
        HIP_CHECK(hipGetLastError());
        if (trip_binned(trip)) {
            dim3 const shares(r.wf_bin_groups, 2u);
            hipLaunchKernelGGL(wf_bin_count_kernel, shares, dim3(kBinThreads), kBinCount * 4u, lane_stream, w, trip);
            HIP_CHECK(hipGetLastError());
            hipLaunchKernelGGL(wf_bin_columns_kernel, dim3(kBinCount / kBinColumnThreads, 2u),
                dim3(kBinColumnThreads), 0, lane_stream, w, trip, r.wf_bin_groups);
            HIP_CHECK(hipGetLastError());
            hipLaunchKernelGGL(wf_bin_bases_kernel, dim3(2u), dim3(kBinThreads), 0, lane_stream, w, trip);
            HIP_CHECK(hipGetLastError());
            hipLaunchKernelGGL(wf_bin_place_kernel, shares, dim3(kBinThreads), kBinCount * 4u, lane_stream, w, trip);
            HIP_CHECK(hipGetLastError());
        }
    };

    rt_hip_renderer::timed_launch whole{};
    if (!fixed) {
        whole = r.begin_timing(stream);
    }
    if (lanes == 2u) {
        HIP_CHECK(hipEventRecord(r.fork_event, stream));
        HIP_CHECK(hipStreamWaitEvent(r.side_stream, r.fork_event, 0));
    }
    wavefront_params const frame = w;  // everything but the state arrays of a set
    uint32_t issued = 0;
    for (uint32_t first = 0; first < p.pass_count; first += batch, ++issued) {
        uint32_t const set = issued % lanes;
        lane_stream = set == 0u ? stream : r.side_stream;
        if (!fixed) {
            w = frame;
            take_state_arrays(w, r.wf_set[set]);
        }
        w.pass_count = std::min(batch, p.pass_count - first);
        w.seeds = p.seeds ? p.seeds + (size_t) first * p.tile_count : nullptr;
        size_t const paths = per_pass * w.pass_count;
        uint32_t const chunk_bound = (uint32_t) (paths / 64u);
        HIP_CHECK(hipMemsetAsync(w.control, 0, (size_t) (trips + 1u) * kCtlWords * 4, lane_stream));
        uint32_t const handover_below =
            handover_setting >= 0 ? (uint32_t) handover_setting
                                  : (uint32_t) std::max<size_t>(paths / 256u, 65536u);
        for (uint32_t trip = 0; trip < trips; ++trip) {
            if (trip > 0u) {
                finish(trip, chunk_bound);  // a no-op unless the compaction handed over
            }
            // a trip has at most one closest-hit and one occlusion ray per path
            uint32_t const region_bound =
                (uint32_t) ((paths + kRegionSlots - 1u) / kRegionSlots);  // regions of the live list
            dim3 const settle_grid(std::min(r.wf_settle_grid,
                (region_bound + kSettleThreads / 64u - 1u) / (kSettleThreads / 64u)));
            dim3 const shade_grid(std::min(r.wf_shade_grid,
                (region_bound + kShadeThreads / 64u - 1u) / (kShadeThreads / 64u)));
            rt_hip_renderer::timed_launch t{};
            if (each) {
                t = r.begin_timing(lane_stream, RT_HIP_KERNEL_EXTEND);
            }
            bool const probe_here = probe;
            if (probe_here) {
                HIP_CHECK(hipEventRecord(r.probe_events[2 * trip], lane_stream));
            }
            w.sort_rays = trip_binned(trip) ? 1u : 0u;
            w.refill = trip_refills(trip) ? 1u : 0u;
            extend(trip, trip == 0u ? chunk_bound : 2u * chunk_bound);
            if (probe_here) {
                HIP_CHECK(hipEventRecord(r.probe_events[2 * trip + 1], lane_stream));
                if (trip + 1u == trips) {
                    HIP_CHECK(hipMemcpyAsync(r.probe_counts, w.control,
                        (size_t) trips * kCtlWords * 4, hipMemcpyDeviceToHost, lane_stream));
                    HIP_CHECK(hipEventRecord(r.probe_done, lane_stream));
                    r.probe_pending = true;
                    r.probe_trips = trips;
                    r.probe_kind = probe_kind;
                    probe = false;
                }
            }
            if (each) {
                r.end_timing(t, lane_stream);
                t = r.begin_timing(lane_stream, RT_HIP_KERNEL_SHADE);
            }
#ifdef RT_EXP_F
            launch_shade_stage<(uint32_t) (RT_EXP_F)>(w, trip, settle_grid, shade_grid, shade_lds,
                lane_stream);
#else
            switch (r.shade_variant) {
#define RT_LAUNCH_SHADE(F)                                                                 \
    case F:                                                                                \
        launch_shade_stage<F>(w, trip, settle_grid, shade_grid, shade_lds, lane_stream);        \
        break;
                RT_SHADE_VARIANTS(RT_LAUNCH_SHADE)
#undef RT_LAUNCH_SHADE
            default:
                throw std::logic_error("no shading kernels for the scene's feature set");
            }
#endif
            HIP_CHECK(hipGetLastError());
            if (each) {
                r.end_timing(t, lane_stream);
            }
            if (trip + 1u < trips) {
                // nothing survives the last trip: its paths are at max_depth
                // trip 0 publishes one raw slot per path, later trips one per live path
                compact(trip + 1u, trip == 0u ? nullptr : w.control + trip * kCtlWords + kCtlLive,
                    (uint32_t) paths, chunk_bound, handover_below);
            }
        }
        if (lanes == 2u && issued > 0u) {
            // the pixel's earlier passes first (the previous batch ran on the other set)
            HIP_CHECK(hipStreamWaitEvent(lane_stream, r.resolved_event[(issued - 1u) % 2u], 0));
        }
        hipLaunchKernelGGL(wf_resolve_kernel, dim3((uint32_t) ((per_pass + 255u) / 256u)),
            dim3(256), 0, lane_stream, w);
        HIP_CHECK(hipGetLastError());
        if (lanes == 2u) {
            HIP_CHECK(hipEventRecord(r.resolved_event[issued % 2u], lane_stream));
        }
    }
    if (lanes == 2u) {
        HIP_CHECK(hipEventRecord(r.join_event, r.side_stream));
        HIP_CHECK(hipStreamWaitEvent(stream, r.join_event, 0));
    }
    if (!fixed) {
        r.end_timing(whole, stream);
    }
}

render_params base_params(rt_hip_renderer const& r, rt_camera const& cam) {
    render_params p{};
    p.sc = r.sc;
    p.cam = cam;
    p.accum = r.accum;
    p.frame_width = r.cfg.width;
    p.x0 = 0;
    p.y0 = 0;
    p.x1 = r.cfg.width;
    p.y1 = r.cfg.height;
    p.tile_width = r.cfg.tile_width;
    p.tile_height = r.cfg.tile_height;
    p.tiles_x = r.tiles_x;
    p.tile_count = r.tiles_x * r.tiles_y;
    p.pass_count = 1;
    p.max_depth = r.cfg.max_depth;
    p.counters = r.counters;
    return p;
}

void require_scene(rt_hip_renderer const* r) {
    if (!r) {
        throw std::runtime_error("null renderer");
    }
    if (!r->scene_ready) {
        throw std::runtime_error("no scene uploaded");
    }
}

}  // namespace

// ---------------------------------------------------------------------------
// State of the allocation-free entry point (rt_hip_render_noalloc): the pipeline's path
// state, lists and control block for at most RT_HIP_NOALLOC_MAX_PATHS paths in flight, in a
// device array that is part of the loaded code object.  Every GPU the library runs on holds
// its own copy at its own address for as long as the process lives; rt_hip_create looks that
// address up once per device (scratch_base) and host code uses only that pointer, never the
// array's name, which on the host is a stand-in and not a device address.
constexpr size_t kNoAllocPaths = RT_HIP_NOALLOC_MAX_PATHS;
constexpr uint32_t kNoAllocTrips = RT_HIP_NOALLOC_MAX_DEPTH + 1u;
constexpr size_t scratch_align(size_t bytes) { return (bytes + 255u) & ~(size_t) 255u; }
constexpr size_t kNoAllocChunks = (kNoAllocPaths + 63u) / 64u;
constexpr size_t kScratchVec4 = scratch_align(kNoAllocPaths * 16u);   // one float4 array
constexpr size_t kScratchVec2 = scratch_align(kNoAllocPaths * 8u);    // the float2 array
constexpr size_t kScratchWord = scratch_align(kNoAllocPaths * 4u);    // one u32 array
constexpr size_t kScratchChunkCounts = scratch_align(kNoAllocChunks * 4u);
constexpr size_t kScratchTileSums =
    scratch_align(((kNoAllocChunks + kTileChunks - 1u) / kTileChunks) * 12u);
constexpr size_t kScratchControl = scratch_align((size_t) (kNoAllocTrips + 1u) * kCtlWords * 4u);
constexpr size_t kScratchGoingCount =
    scratch_align(((kNoAllocPaths + kRegionSlots - 1u) / kRegionSlots) * 4u);
constexpr size_t kScratchBytes = 8u * kScratchVec4 + 2u * kScratchVec2 + 4u * kScratchWord +
                                 kScratchGoingCount + kScratchChunkCounts + kScratchTileSums +
                                 kScratchControl;
static_assert(kScratchBytes <= (size_t) 48u << 20, "keep the resident scratch array modest");

static __device__ __align__(256) unsigned char scratch[kScratchBytes];

constexpr int kMaxDevices = 64;
static unsigned char* g_scratch_base[kMaxDevices] = {};  // per device; set by rt_hip_create

namespace {

// Looks up this device's copy of `scratch` (the device must be current).
unsigned char* scratch_base(int device) {
    if (device < 0 || device >= kMaxDevices) {
        throw std::runtime_error("device ordinal beyond the per-device table");
    }
    if (g_scratch_base[device] == nullptr) {
        void* ptr = nullptr;
        HIP_CHECK(hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch)));
        if (ptr == nullptr) {
            throw std::runtime_error("hipGetSymbolAddress returned no address for the scratch array");
        }
        g_scratch_base[device] = static_cast<unsigned char*>(ptr);
    }
    return g_scratch_base[device];
}

// The pipeline's arrays carved out of the device's scratch array.
wavefront_params carve_scratch(unsigned char* base, unsigned long long* wave_stats) {
    wavefront_params w{};
    size_t at = 0;
    auto take = [&](size_t bytes) {
        unsigned char* const here = base + at;
        at += bytes;
        return here;
    };
    w.ray_o = (float4*) take(kScratchVec4);
    w.ray_d = (float4*) take(kScratchVec4);
    w.sh_o = (float4*) take(kScratchVec4);
    w.sh_d = (float4*) take(kScratchVec4);
    w.nee = (float4*) take(kScratchVec4);
    w.thr = (float4*) take(kScratchVec4);
    w.rad = (float4*) take(kScratchVec4);
    w.hit0 = (float4*) take(kScratchVec4);
    w.hit1 = (float2*) take(kScratchVec2);
    w.live = (uint32_t*) take(kScratchWord);
    w.q_closest = (uint32_t*) take(kScratchWord);
    w.q_shadow = (uint32_t*) take(kScratchWord);
    w.raw = (uint32_t*) take(kScratchWord);
    w.going = (uint2*) take(kScratchVec2);
    w.going_count = (uint32_t*) take(kScratchGoingCount);
    w.chunk_counts = (uint32_t*) take(kScratchChunkCounts);
    w.tile_sums = (uint32_t*) take(kScratchTileSums);
    w.control = (uint32_t*) take(kScratchControl);
    if (at != kScratchBytes) {
        throw std::logic_error("scratch layout does not add up");
    }
    w.wave_stats = wave_stats;
    return w;
}

}  // namespace

namespace {

void render_passes(rt_hip_renderer* r, const rt_camera* camera, uint32_t first_pass,
    uint32_t pass_count, const uint32_t* rect, float* accum_dev, hipStream_t stream,
    bool allocation_free) {
    require_scene(r);
    if (!camera) {
        throw std::runtime_error("render: null camera");
    }
    if (pass_count == 0) {
        return;
    }
    if (!r->seeds || first_pass < r->seeds_first_pass ||
        (uint64_t) first_pass + pass_count > (uint64_t) r->seeds_first_pass + r->seeds_pass_count) {
        throw std::runtime_error("render: seeds for the requested passes are not resident");
    }
    device_scope const on_device(r->device);
    render_params p = base_params(*r, *camera);
    if (rect) {
        if (rect[0] >= rect[2] || rect[1] >= rect[3] || rect[2] > r->cfg.width ||
            rect[3] > r->cfg.height) {
            throw std::runtime_error("render: bad rectangle");
        }
        p.x0 = rect[0];
        p.y0 = rect[1];
        p.x1 = rect[2];
        p.y1 = rect[3];
    }
    if (accum_dev) {
        p.accum = accum_dev;
    }
    p.pass_count = pass_count;
    p.seeds = r->seeds + (size_t) (first_pass - r->seeds_first_pass) * p.tile_count;
    // Under stream capture (the allocation-free entry point may be recorded into a HIP graph)
    // the hand-over between streams is the capturing caller's business: an event recorded
    // inside a capture must not be waited for by work outside it.
    bool capturing = false;
    if (allocation_free) {
        hipStreamCaptureStatus status = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(stream, &status) == hipSuccess) {
            capturing = status == hipStreamCaptureStatusActive;
        }
    }
    if (!capturing) {
        r->follow(stream);
    }
    if (allocation_free) {
        if (p.max_depth > RT_HIP_NOALLOC_MAX_DEPTH) {
            throw std::runtime_error("render_noalloc: max_depth " + std::to_string(p.max_depth) +
                                     " exceeds RT_HIP_NOALLOC_MAX_DEPTH");
        }
        launch_wavefront(*r, p, stream, &r->noalloc, kNoAllocPaths);
    } else if (r->cfg.flags & RT_HIP_FLAG_FUSED) {
        launch_path_trace(*r, p, stream);
    } else {
        launch_wavefront(*r, p, stream);
    }
    if (!capturing) {
        r->mark(stream);
    }
    if (!accum_dev && !rect) {
        r->passes_done += pass_count;
    }
}

}  // namespace


// ---------------------------------------------------------------------------
// Multi-GPU: one sum-reduce of the accumulators over RCCL (xGMI between the GPUs of a node).
namespace {

struct rccl_api {
    decltype(&ncclCommInitAll) comm_init_all = nullptr;
    decltype(&ncclCommDestroy) comm_destroy = nullptr;
    decltype(&ncclReduce) reduce = nullptr;
    decltype(&ncclGroupStart) group_start = nullptr;
    decltype(&ncclGroupEnd) group_end = nullptr;
    decltype(&ncclGetErrorString) error_string = nullptr;
};

// Loaded on the first reduce that needs it, so a single-GPU user never maps RCCL.  A process
// that already holds an RCCL (PyTorch ships one) shares it.
rccl_api const& rccl() {
    static rccl_api api;
    static std::once_flag once;
    static std::string failure;
    std::call_once(once, [] {
        void* lib = dlopen("librccl.so", RTLD_NOW | RTLD_NOLOAD);
        for (char const* name : {"librccl.so.1", "librccl.so"}) {
            if (lib == nullptr) {
                lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            }
        }
        if (lib == nullptr) {
            failure = std::string{"cannot load librccl: "} + dlerror();
            return;
        }
        auto find = [&](auto& slot, char const* symbol) {
            slot = reinterpret_cast<std::remove_reference_t<decltype(slot)>>(dlsym(lib, symbol));
            if (slot == nullptr && failure.empty()) {
                failure = std::string{"librccl lacks "} + symbol;
            }
        };
        find(api.comm_init_all, "ncclCommInitAll");
        find(api.comm_destroy, "ncclCommDestroy");
        find(api.reduce, "ncclReduce");
        find(api.group_start, "ncclGroupStart");
        find(api.group_end, "ncclGroupEnd");
        find(api.error_string, "ncclGetErrorString");
    });
    if (!failure.empty()) {
        throw std::runtime_error(failure);
    }
    return api;
}

void rccl_check(ncclResult_t code, char const* what) {
    if (code != ncclSuccess) {
        throw std::runtime_error(std::string{what} + ": " + rccl().error_string(code));
    }
}

// One communicator set per distinct device list, kept for the life of the process
// (creating one costs tens of milliseconds; a render reduces once).
std::vector<ncclComm_t> const& communicators(std::vector<int> const& devices) {
    static std::map<std::vector<int>, std::vector<ncclComm_t>> cache;
    static std::mutex guard;
    std::lock_guard<std::mutex> const lock(guard);
    auto it = cache.find(devices);
    if (it == cache.end()) {
        std::vector<ncclComm_t> comms(devices.size());
        rccl_check(rccl().comm_init_all(comms.data(), (int) devices.size(), devices.data()),
            "ncclCommInitAll");
        it = cache.emplace(devices, std::move(comms)).first;
    }
    return it->second;
}

void reduce_renderers(rt_hip_renderer* const* renderers, uint32_t count) {
    if (!renderers || count == 0) {
        throw std::runtime_error("reduce: no renderers");
    }
    rt_hip_renderer& root = *renderers[0];
    size_t const pixels = (size_t) root.cfg.width * root.cfg.height;
    for (uint32_t i = 0; i < count; ++i) {
        rt_hip_renderer const* r = renderers[i];
        if (!r) {
            throw std::runtime_error("reduce: null renderer");
        }
        if (r->cfg.width != root.cfg.width || r->cfg.height != root.cfg.height) {
            throw std::runtime_error("reduce: the renderers' frames differ in size");
        }
        for (uint32_t k = 0; k < i; ++k) {
            if (renderers[k] == r) {
                throw std::runtime_error("reduce: the same renderer is listed twice");
            }
        }
    }
    dim3 const grid((uint32_t) ((pixels + 255) / 256)), block(256);
    // 1. wait for every renderer's work; renderers that share a GPU with an earlier one are
    //    added to it by a kernel (no transport needed) and cleared
    std::vector<rt_hip_renderer*> leaders;  // first renderer listed on each device
    for (uint32_t i = 0; i < count; ++i) {
        rt_hip_renderer* r = renderers[i];
        device_scope const on_device(r->device);
        HIP_CHECK(hipDeviceSynchronize());
        rt_hip_renderer* leader = nullptr;
        for (rt_hip_renderer* l : leaders) {
            if (l->device == r->device) {
                leader = l;
            }
        }
        if (leader == nullptr) {
            leaders.push_back(r);
            continue;
        }
        hipLaunchKernelGGL(add_rgb_kernel, grid, block, 0, nullptr,
            reinterpret_cast<float4*>(leader->accum), reinterpret_cast<const float4*>(r->accum),
            pixels);
        HIP_CHECK(hipGetLastError());
        hipLaunchKernelGGL(clear_rgba_kernel, grid, block, 0, nullptr,
            reinterpret_cast<float4*>(r->accum), pixels);
        HIP_CHECK(hipGetLastError());
        HIP_CHECK(hipDeviceSynchronize());
    }
    // 2. one RCCL sum-reduce of the per-GPU accumulators onto the root's GPU
    static bool const always_rccl = [] {
        char const* v = getenv("RT_HIP_REDUCE_ALWAYS_RCCL");  // lets a one-GPU test drive RCCL
        return v && atoi(v) != 0;
    }();
    if (leaders.size() > 1 || always_rccl) {
        std::vector<int> devices;
        for (rt_hip_renderer* l : leaders) {
            devices.push_back(l->device);
        }
        std::vector<ncclComm_t> const& comms = communicators(devices);
        rccl_check(rccl().group_start(), "ncclGroupStart");
        for (size_t k = 0; k < leaders.size(); ++k) {
            device_scope const on_device(leaders[k]->device);
            rccl_check(rccl().reduce(leaders[k]->accum, k == 0 ? root.accum : nullptr, pixels * 4,
                           ncclFloat, ncclSum, 0, comms[k], nullptr),
                "ncclReduce");
        }
        rccl_check(rccl().group_end(), "ncclGroupEnd");
        for (size_t k = 0; k < leaders.size(); ++k) {
            device_scope const on_device(leaders[k]->device);
            HIP_CHECK(hipDeviceSynchronize());
            if (k > 0) {
                hipLaunchKernelGGL(clear_rgba_kernel, grid, block, 0, nullptr,
                    reinterpret_cast<float4*>(leaders[k]->accum), pixels);
                HIP_CHECK(hipGetLastError());
                HIP_CHECK(hipDeviceSynchronize());
            }
        }
    }
    // 3. alpha back to 1 (the sum of N images holds N), pass counters
    {
        device_scope const on_device(root.device);
        hipLaunchKernelGGL(set_alpha_kernel, grid, block, 0, nullptr,
            reinterpret_cast<float4*>(root.accum), pixels);
        HIP_CHECK(hipGetLastError());
        HIP_CHECK(hipDeviceSynchronize());
    }
    for (uint32_t i = 1; i < count; ++i) {
        root.passes_done += renderers[i]->passes_done;
        renderers[i]->passes_done = 0;
        renderers[i]->tile_cursor = 0;
        renderers[i]->last_recorded = false;
    }
    root.last_recorded = false;
}

}  // namespace

// ---------------------------------------------------------------------------
// C ABI

extern "C" {

int rt_hip_create(const rt_hip_config* config, rt_hip_renderer** out) {
    return guarded([&] {
        if (!config || !out) {
            throw std::runtime_error("create: null argument");
        }
        rt_hip_config const& c = *config;
        if (c.width == 0 || c.height == 0 || c.tile_width == 0 || c.tile_height == 0) {
            throw std::runtime_error("create: zero-sized frame or tile");
        }
        if (c.width % c.tile_width != 0) {
            throw std::runtime_error("Window width isn't divisible by tile width");
        }
        if (c.height % c.tile_height != 0) {
            throw std::runtime_error("Window height isn't divisible by tile height");
        }
        if (c.accel > RT_HIP_ACCEL_REBUILT) {
            throw std::runtime_error("create: unknown accel mode");
        }
        if (((config->flags & RT_HIP_FLAG_BIN_RAYS) && (config->flags & RT_HIP_FLAG_NO_BIN_RAYS)) ||
            ((config->flags & RT_HIP_FLAG_TRACE_REFILL) && (config->flags & RT_HIP_FLAG_TRACE_CHUNKS))) {
            throw std::runtime_error("create: flags that force a policy on and off together");
        }
        int device_count = 0;
        HIP_CHECK(hipGetDeviceCount(&device_count));
        if (device_count == 0) {
            throw std::runtime_error("create: no HIP device is visible");
        }
        auto r = std::make_unique<rt_hip_renderer>();
        r->cfg = c;
        if (c.device >= 0) {
            r->device = c.device;
        } else {
            HIP_CHECK(hipGetDevice(&r->device));
        }
        if (r->device >= device_count) {
            throw std::runtime_error("create: device ordinal out of range");
        }
        device_scope const on_device(r->device);
        r->tiles_x = c.width / c.tile_width;
        r->tiles_y = c.height / c.tile_height;
        r->preview_w = std::max(1u, c.width / kPreviewRatio);
        r->preview_h = std::max(1u, c.height / kPreviewRatio);
        size_t const accum_bytes = (size_t) c.width * c.height * 4 * sizeof(float);
        size_t const preview_bytes = (size_t) r->preview_w * r->preview_h * 4 * sizeof(float);
        HIP_CHECK(hipMalloc((void**) &r->accum, accum_bytes));
        HIP_CHECK(hipMalloc((void**) &r->preview, preview_bytes));
        HIP_CHECK(hipMalloc((void**) &r->counters, kCounterWords * sizeof(unsigned long long)));
        HIP_CHECK(hipMemset(r->accum, 0, accum_bytes));
        HIP_CHECK(hipMemset(r->preview, 0, preview_bytes));
        HIP_CHECK(hipMemset(r->counters, 0, kCounterWords * sizeof(unsigned long long)));
        r->block_total = ((c.width + 7u) / 8u) * ((c.height + 7u) / 8u);
        HIP_CHECK(hipMalloc((void**) &r->block_cost, r->block_total * sizeof(uint32_t)));
        HIP_CHECK(hipMalloc((void**) &r->block_order, r->block_total * sizeof(uint32_t)));
        HIP_CHECK(hipMemset(r->block_cost, 0, r->block_total * sizeof(uint32_t)));
        HIP_CHECK(hipEventCreate(&r->probe_done));
        HIP_CHECK(hipEventCreateWithFlags(&r->last_done, hipEventDisableTiming));
        init_launch_plan(*r);
        if (char const* const v = getenv("RT_HIP_HANDOVER_BELOW")) {
            r->handover_setting = atoi(v);  // tuning knob: absolute number of live paths
        }
        r->noalloc = carve_scratch(scratch_base(r->device), r->wf.wave_stats);
        *out = r.release();
    });
}

int rt_hip_upload_scene(rt_hip_renderer* r, const rt_scene_desc* scene) {
    return guarded([&] {
        if (!r || !scene) {
            throw std::runtime_error("upload_scene: null argument");
        }
        validate_scene(*scene);
        rt_scene_desc const& s = *scene;
        device_scope const on_device(r->device);
        HIP_CHECK(hipDeviceSynchronize());
        r->free_scene();
        scene_dev d{};
        d.tlas = r->upload(s.tlas, s.tlas_count);
        d.blas = r->upload(s.blas, s.blas_count);
        d.meshes = r->upload(s.meshes, s.mesh_count);
        d.instances = r->upload(s.instances, s.instance_count);
        d.triangles = r->upload(s.triangles, s.triangle_count);
        d.transforms = r->upload(s.transforms, s.transform_count);
        d.inverse_transforms = r->upload(s.inverse_transforms, s.transform_count);
        {
            normal_matrix* table = nullptr;
            size_t const count = std::max<size_t>(s.transform_count, 1);
            HIP_CHECK(hipMalloc((void**) &table, count * sizeof(normal_matrix)));
            r->scene_allocs.push_back(table);
            if (s.transform_count > 0) {
                hipLaunchKernelGGL(normal_matrix_kernel, dim3((s.transform_count + 63u) / 64u),
                    dim3(64), 0, 0, d.transforms, s.transform_count, table);
                HIP_CHECK(hipGetLastError());
                HIP_CHECK(hipStreamSynchronize(nullptr));  // render calls run on other streams
            }
            d.normal_matrices = table;
        }
        {
            rt_material* const table = r->upload(s.materials, s.material_count);
            if (s.material_count > 0) {
                hipLaunchKernelGGL(derive_materials_kernel, dim3((s.material_count + 63u) / 64u),
                    dim3(64), 0, 0, table, s.material_count);
                HIP_CHECK(hipGetLastError());
                HIP_CHECK(hipStreamSynchronize(nullptr));
            }
            d.materials = table;
        }
        d.lights = r->upload(s.lights, s.light_count);
        std::vector<texture_dev> textures(s.texture_count);
        for (uint32_t i = 0; i < s.texture_count; ++i) {
            rt_texture const& t = s.textures[i];
            size_t const texel = t.format == RT_TEX_F32X4 ? 16 : 4;
            textures[i].data = r->upload(
                static_cast<const uint8_t*>(t.data), (size_t) t.width * t.height * texel);
            textures[i].width = t.width;
            textures[i].height = t.height;
            textures[i].format = t.format;
        }
        d.textures = r->upload(textures.data(), textures.size());
        d.light_count = s.light_count;
        d.sky_light = s.sky_light;
        r->shade_features = scene_features(s, r->cfg.flags);
        r->shade_variant = shade_variant_for(r->shade_features);
        if (char const* const v = getenv("RT_HIP_SHADE_FEATURES")) {  // experiments: a superset
            uint32_t const forced = (uint32_t) atoi(v) & kShadeAll;
            if ((r->shade_features & ~forced) == 0u) {
                r->shade_variant = shade_variant_for(forced);
            }
        }
        d.instance_count = s.instance_count;
        d.transform_count = s.transform_count;
        d.material_count = s.material_count;
        d.mesh_count = s.mesh_count;
        d.texture_count = s.texture_count;
        if ((r->cfg.flags & RT_HIP_FLAG_SKY_IMPORTANCE) && s.sky_light >= 0 &&
            s.lights[s.sky_light].emission_tex >= 0) {
            sky_tables const t = build_sky_tables(s.lights[s.sky_light],
                s.textures[s.lights[s.sky_light].emission_tex]);
            d.sky_density = r->upload(t.density.data(), t.density.size());
            d.sky_conditional = r->upload(t.conditional.data(), t.conditional.size());
            d.sky_marginal = r->upload(t.marginal.data(), t.marginal.size());
            d.sky_w = t.width;
            d.sky_h = t.height;
        }
        if (r->cfg.accel == RT_HIP_ACCEL_REFERENCE) {
            uint32_t blas_depth = 0;
            for (uint32_t m = 0; m < s.mesh_count; ++m) {
                blas_depth = std::max(blas_depth, ref_depth(s.blas, s.meshes[m].bvh_start));
            }
            r->accel_node_bytes = ((uint64_t) s.tlas_count + s.blas_count) * sizeof(rt_bvh_node);
            r->accel_tri_bytes = (uint64_t) s.triangle_count * sizeof(rt_triangle);
            r->accel_instance_bytes = (uint64_t) s.instance_count * sizeof(rt_instance) +
                                      (uint64_t) s.transform_count * sizeof(rt_mat4);
            uint32_t const need = ref_depth(s.tlas, 0) + blas_depth;
            if (need > kRefStackCapacity) {
                throw std::runtime_error("upload_scene: the uploaded BVH is " +
                                         std::to_string(need) +
                                         " levels deep; reference-order traversal supports " +
                                         std::to_string(kRefStackCapacity));
            }
        } else {
            // RT_HIP_OPENED_ENTRIES: tuning knob for experiments (TLAS leaves added by opening)
            char const* const opened = getenv("RT_HIP_OPENED_ENTRIES");
            built_accel const accel =
                build_wide_accel(s, opened ? (uint32_t) atoi(opened) : kOpenedEntries);
            if (accel.max_stack_bound > kFastLdsEntries) {
                throw std::runtime_error("upload_scene: rebuilt BVH exceeds the traversal stack");
            }
            // the traversal addresses these arrays with 32-bit byte offsets
            if (accel.nodes.size() * sizeof(wide_node) >= ((size_t) 1 << 32) ||
                accel.tris.size() * sizeof(tri_pos) >= ((size_t) 1 << 32) ||
                accel.instances.size() * sizeof(instance_rec) >= ((size_t) 1 << 32)) {
                throw std::runtime_error("upload_scene: rebuilt BVH exceeds 4 GiB per array");
            }
            d.wide_nodes = r->upload(accel.nodes.data(), accel.nodes.size());
            d.wide_tris = r->upload(accel.tris.data(), accel.tris.size());
            d.wide_instances = r->upload(accel.instances.data(), accel.instances.size());
            d.wide_tlas_root = accel.tlas_root;
            d.wide_lists_degenerate = accel.lists_degenerate_triangles ? 1u : 0u;
            r->accel_node_bytes = accel.nodes.size() * sizeof(wide_node);
            r->accel_tri_bytes = accel.tris.size() * sizeof(tri_pos);
            r->accel_instance_bytes = accel.instances.size() * sizeof(instance_rec);
            // LDS plan of the cached traversal kernel (wf_extend_kernel): stacks first, the
            // rest goes to the instance records (at most half) and the hottest nodes.
            d.stack_entries = std::max(1u, accel.max_stack_bound);
            uint32_t const stack_bytes = (kExtendThreads / 64u) * d.stack_entries * 256u;
            uint32_t const spare = kLdsBytesPerCu - stack_bytes;
            d.cached_instances =
                (uint32_t) std::min<size_t>(accel.instances.size(), spare / 2u / 64u);
            d.cached_nodes = (uint32_t) std::min<size_t>(accel.nodes.size(),
                (spare - d.cached_instances * 64u) / sizeof(wide_node));
            if (!kTraceCopies) {
                d.cached_nodes = d.cached_instances = 0u;
            }
        }
        if (s.tlas_count > 0) {
            // the root of the uploaded TLAS bounds every instance (bvh_linear_node, bvh.h:30-37)
            rt_bvh_node const& root = s.tlas[0];
            r->scene_lo[0] = root.x_min, r->scene_hi[0] = root.x_max;
            r->scene_lo[1] = root.y_min, r->scene_hi[1] = root.y_max;
            r->scene_lo[2] = root.z_min, r->scene_hi[2] = root.z_max;
        }
        // ray binning: forced by a flag (or RT_HIP_BIN_RAYS=0/1 in the environment, for
        // experiments), otherwise measured on this scene by the next large render call
        r->probe_pending = false;  // the device was synchronised above
        r->probe_ns_per_ray = 0.0;
        r->refill = RT_HIP_REFILL_UNDECIDED;
        r->refill_ms[0] = r->refill_ms[1] = -1.0;
        r->refill_trip.clear();
        r->bin_candidates = false;
        if (r->cfg.flags & RT_HIP_FLAG_TRACE_REFILL) {
            r->refill = RT_HIP_REFILL_ON;
        } else if (r->cfg.flags & RT_HIP_FLAG_TRACE_CHUNKS) {
            r->refill = RT_HIP_REFILL_OFF;
        } else if (char const* const v = getenv("RT_HIP_TRACE_REFILL")) {  // experiments
            r->refill = atoi(v) != 0 ? RT_HIP_REFILL_ON : RT_HIP_REFILL_OFF;
        }
        r->bin_trip.clear();
        char const* const forced = getenv("RT_HIP_BIN_RAYS");
        if (r->cfg.accel != RT_HIP_ACCEL_REBUILT || (r->cfg.flags & RT_HIP_FLAG_NO_BIN_RAYS) ||
            (forced && atoi(forced) == 0)) {
            r->binning = RT_HIP_BINNING_OFF;
        } else if ((r->cfg.flags & RT_HIP_FLAG_BIN_RAYS) || (forced && atoi(forced) != 0)) {
            r->binning = RT_HIP_BINNING_ON;
        } else {
            r->binning = RT_HIP_BINNING_UNDECIDED;
        }
        r->sc = d;
        r->scene_ready = true;
        r->order_valid = false;
    });
}

int rt_hip_upload_seeds(rt_hip_renderer* r, const uint32_t* seeds, uint32_t first_pass,
    uint32_t pass_count) {
    return guarded([&] {
        if (!r || !seeds || pass_count == 0) {
            throw std::runtime_error("upload_seeds: null or empty argument");
        }
        device_scope const on_device(r->device);
        size_t const words = (size_t) pass_count * r->tiles_x * r->tiles_y;
        HIP_CHECK(hipDeviceSynchronize());
        if (words > r->seeds_capacity) {
            if (r->seeds) {
                HIP_CHECK(hipFree(r->seeds));
                r->seeds = nullptr;
                r->seeds_capacity = 0;
            }
            HIP_CHECK(hipMalloc((void**) &r->seeds, words * sizeof(uint32_t)));
            r->seeds_capacity = words;
        }
        HIP_CHECK(hipMemcpy(r->seeds, seeds, words * sizeof(uint32_t), hipMemcpyHostToDevice));
        r->seeds_first_pass = first_pass;
        r->seeds_pass_count = pass_count;
    });
}

int rt_hip_clear(rt_hip_renderer* r, float* accum_dev, void* stream) {
    return guarded([&] {
        if (!r) {
            throw std::runtime_error("null renderer");
        }
        device_scope const on_device(r->device);
        size_t const pixels = (size_t) r->cfg.width * r->cfg.height;
        float* target = accum_dev ? accum_dev : r->accum;
        r->follow((hipStream_t) stream);
        hipLaunchKernelGGL(clear_rgba_kernel, dim3((uint32_t) ((pixels + 255) / 256)), dim3(256),
            0, (hipStream_t) stream, reinterpret_cast<float4*>(target), pixels);
        HIP_CHECK(hipGetLastError());
        r->mark((hipStream_t) stream);
        if (!accum_dev) {
            r->passes_done = 0;
            r->tile_cursor = 0;
        }
    });
}

int rt_hip_render(rt_hip_renderer* r, const rt_camera* camera, uint32_t first_pass,
    uint32_t pass_count, const uint32_t* rect, float* accum_dev, void* stream) {
    return guarded([&] {
        render_passes(r, camera, first_pass, pass_count, rect, accum_dev, (hipStream_t) stream,
            false);
    });
}

int rt_hip_reserve(rt_hip_renderer* r, uint32_t width, uint32_t height, uint32_t pass_count) {
    return guarded([&] {
        if (!r) {
            throw std::runtime_error("null renderer");
        }
        if (width == 0 || height == 0 || pass_count == 0 || width > r->cfg.width ||
            height > r->cfg.height) {
            throw std::runtime_error("reserve: bad size");
        }
        if (r->cfg.flags & RT_HIP_FLAG_FUSED) {
            return;  // the fused kernel keeps its paths in registers
        }
        size_t const per_pass = (size_t) ((width + 7u) / 8u) * ((height + 7u) / 8u) * 64u;
        if (per_pass > batch_cap()) {
            throw std::runtime_error("reserve: one pass of that rectangle exceeds one batch");
        }
        // (the batches launch_wavefront would cut such a call into)
        device_scope const on_device(r->device);
        size_t const most = std::max<size_t>(1, batch_limit(*r, per_pass, pass_count) / per_pass);
        uint32_t batches = (uint32_t) ((pass_count + most - 1u) / most);
        if (pass_count >= 2u && per_pass * pass_count >= kSplitMinPaths) {
            batches = std::max(batches, 2u);
        }
        size_t const batch = (pass_count + batches - 1u) / batches;
        bool const binned = r->binning != RT_HIP_BINNING_OFF;  // decided later: be ready for it
        ensure_wavefront(*r, 0, per_pass * batch, r->cfg.max_depth + 1u, binned);
        if (batches >= 2u) {
            ensure_wavefront(*r, 1, per_pass * batch, r->cfg.max_depth + 1u, binned);
        }
    });
}

int rt_hip_shade_features(rt_hip_renderer* r, uint32_t* scene_features, uint32_t* kernel_features) {
    return guarded([&] {
        require_scene(r);
        if (scene_features) {
            *scene_features = r->shade_features;
        }
        if (kernel_features) {
#ifdef RT_EXP_F
            *kernel_features = (uint32_t) (RT_EXP_F);
#else
            *kernel_features = r->shade_variant;
#endif
        }
    });
}

int rt_hip_tune(rt_hip_renderer* r, const rt_camera* camera, uint32_t pass_count, void* stream) {
    return guarded([&] {
        require_scene(r);
        if (!camera || pass_count == 0) {
            throw std::runtime_error("tune: null camera or no passes");
        }
        if (r->cfg.flags & RT_HIP_FLAG_FUSED) {
            return;  // the fused kernel has no policies to settle
        }
        device_scope const on_device(r->device);
        hipStream_t const s = (hipStream_t) stream;
        float* scratch_accum = nullptr;
        size_t const bytes = (size_t) r->cfg.width * r->cfg.height * 4u * sizeof(float);
        HIP_CHECK(hipMalloc((void**) &scratch_accum, bytes));
        try {
            // the first large call is the fixed-chunk base line, the second refills its lanes;
            // each is waited for, so that the verdict is in before the next one starts
            for (uint32_t call = 0; call < 2u; ++call) {
                render_params p = base_params(*r, *camera);
                p.accum = scratch_accum;
                p.pass_count = pass_count;
                p.seeds = nullptr;
                p.scalar_seed = 0x9E3779B9u + call;
                r->follow(s);
                launch_wavefront(*r, p, s);
                r->mark(s);
                HIP_CHECK(hipStreamSynchronize(s));
                settle_probe(*r);
            }
        } catch (...) {
            (void) hipFree(scratch_accum);
            throw;
        }
        HIP_CHECK(hipFree(scratch_accum));
    });
}

int rt_hip_render_noalloc(rt_hip_renderer* r, const rt_camera* camera, uint32_t first_pass,
    uint32_t pass_count, const uint32_t* rect, float* accum_dev, void* stream) {
    return guarded([&] {
        render_passes(r, camera, first_pass, pass_count, rect, accum_dev, (hipStream_t) stream,
            true);
    });
}

int rt_hip_render_tile(rt_hip_renderer* r, const rt_camera* camera, uint32_t random_seed,
    uint32_t tile_x, uint32_t tile_y, void* stream) {
    return guarded([&] {
        require_scene(r);
        if (!camera) {
            throw std::runtime_error("render_tile: null camera");
        }
        if (tile_x >= r->tiles_x || tile_y >= r->tiles_y) {
            throw std::runtime_error("render_tile: tile index out of range");
        }
        device_scope const on_device(r->device);
        render_params p = base_params(*r, *camera);
        p.x0 = tile_x * r->cfg.tile_width;
        p.y0 = tile_y * r->cfg.tile_height;
        p.x1 = p.x0 + r->cfg.tile_width;
        p.y1 = p.y0 + r->cfg.tile_height;
        p.seeds = nullptr;
        p.scalar_seed = random_seed;
        r->follow((hipStream_t) stream);
        launch_path_trace(*r, p, (hipStream_t) stream);
        r->mark((hipStream_t) stream);
        if (++r->tile_cursor == r->tiles_x * r->tiles_y) {
            r->tile_cursor = 0;
            ++r->passes_done;
        }
    });
}

int rt_hip_render_preview(rt_hip_renderer* r, const rt_camera* preview_camera,
    uint32_t random_seed, void* stream) {
    return guarded([&] {
        require_scene(r);
        if (!preview_camera) {
            throw std::runtime_error("render_preview: null camera");
        }
        device_scope const on_device(r->device);
        render_params p = base_params(*r, *preview_camera);
        p.accum = r->preview;
        p.frame_width = r->preview_w;
        p.x1 = r->preview_w;
        p.y1 = r->preview_h;
        p.preview = 1;
        p.seeds = nullptr;
        p.scalar_seed = random_seed;
        r->follow((hipStream_t) stream);
        launch_path_trace(*r, p, (hipStream_t) stream);
        r->mark((hipStream_t) stream);
        r->passes_done = 0;
        r->tile_cursor = 0;
    });
}

int rt_hip_primary_ids(rt_hip_renderer* r, const rt_camera* camera, uint32_t pass,
    uint32_t* instance_ids, uint32_t* triangle_ids, float* hit_t, float* bary) {
    return guarded([&] {
        require_scene(r);
        if (!camera) {
            throw std::runtime_error("primary_ids: null camera");
        }
        if (!r->seeds || pass < r->seeds_first_pass ||
            pass >= r->seeds_first_pass + r->seeds_pass_count) {
            throw std::runtime_error("primary_ids: seeds for that pass are not resident");
        }
        device_scope const on_device(r->device);
        size_t const pixels = (size_t) r->cfg.width * r->cfg.height;
        render_params p = base_params(*r, *camera);
        p.seeds = r->seeds + (size_t) (pass - r->seeds_first_pass) * p.tile_count;
        p.counters = nullptr;
        std::vector<void*> temps;
        auto temp = [&](size_t bytes) {
            void* d = nullptr;
            HIP_CHECK(hipMalloc(&d, bytes));
            temps.push_back(d);
            return d;
        };
        try {
            p.out_instance = instance_ids ? (uint32_t*) temp(pixels * 4) : nullptr;
            p.out_triangle = triangle_ids ? (uint32_t*) temp(pixels * 4) : nullptr;
            p.out_t = hit_t ? (float*) temp(pixels * 4) : nullptr;
            p.out_bary = bary ? (float*) temp(pixels * 12) : nullptr;
            if (r->cfg.accel == RT_HIP_ACCEL_REFERENCE) {
                launch_rect(primary_ids_kernel<RT_HIP_ACCEL_REFERENCE>, p, nullptr);
            } else {
                launch_rect(primary_ids_kernel<RT_HIP_ACCEL_REBUILT>, p, nullptr);
            }
            HIP_CHECK(hipDeviceSynchronize());
            if (instance_ids) {
                HIP_CHECK(hipMemcpy(instance_ids, p.out_instance, pixels * 4, hipMemcpyDeviceToHost));
            }
            if (triangle_ids) {
                HIP_CHECK(hipMemcpy(triangle_ids, p.out_triangle, pixels * 4, hipMemcpyDeviceToHost));
            }
            if (hit_t) {
                HIP_CHECK(hipMemcpy(hit_t, p.out_t, pixels * 4, hipMemcpyDeviceToHost));
            }
            if (bary) {
                HIP_CHECK(hipMemcpy(bary, p.out_bary, pixels * 12, hipMemcpyDeviceToHost));
            }
        } catch (...) {
            for (void* d : temps) {
                (void) hipFree(d);
            }
            throw;
        }
        for (void* d : temps) {
            (void) hipFree(d);
        }
    });
}

int rt_hip_probe(rt_hip_renderer* r, uint32_t kind, uint32_t count, const rt_hip_probe_in* in,
    rt_hip_probe_out* out) {
    return guarded([&] {
        if (!r || !in || !out) {
            throw std::runtime_error("probe: null argument");
        }
        if (kind > RT_HIP_PROBE_TRIANGLE_DEGENERATE) {
            throw std::runtime_error("probe: unknown kind");
        }
        if (count == 0) {
            return;
        }
        bool const needs_scene = kind == RT_HIP_PROBE_SAMPLE_TEXTURE ||
                                 kind == RT_HIP_PROBE_SKY || kind == RT_HIP_PROBE_SKY_SAMPLE;
        if (needs_scene && !r->scene_ready) {
            throw std::runtime_error("probe: this kind needs an uploaded scene");
        }
        if (kind == RT_HIP_PROBE_SKY_SAMPLE && r->sc.sky_density == nullptr) {
            throw std::runtime_error("probe: no sky distribution (RT_HIP_FLAG_SKY_IMPORTANCE and "
                                     "an environment map are needed)");
        }
        // every index a probe will follow is checked here, on the host
        for (uint32_t i = 0; i < count; ++i) {
            if (kind == RT_HIP_PROBE_SAMPLE_TEXTURE && in[i].u[0] >= r->sc.texture_count) {
                throw std::runtime_error("probe: texture index out of range");
            }
            if (kind == RT_HIP_PROBE_EVAL_DISNEY || kind == RT_HIP_PROBE_SAMPLE_DISNEY) {
                rt_material const& m = in[i].material;
                if (m.albedo_tex >= 0 || m.emission_tex >= 0 || m.normal_tex >= 0 ||
                    m.metallic_roughness_tex >= 0) {
                    throw std::runtime_error("probe: BSDF probes take untextured materials");
                }
            }
            if (kind == RT_HIP_PROBE_RAND_UINT && in[i].u[2] < in[i].u[1]) {
                throw std::runtime_error("probe: rand_uint range");
            }
        }
        device_scope const on_device(r->device);
        rt_hip_probe_in* d_in = nullptr;
        rt_hip_probe_out* d_out = nullptr;
        try {
            HIP_CHECK(hipMalloc((void**) &d_in, (size_t) count * sizeof(rt_hip_probe_in)));
            HIP_CHECK(hipMalloc((void**) &d_out, (size_t) count * sizeof(rt_hip_probe_out)));
            HIP_CHECK(hipMemcpy(d_in, in, (size_t) count * sizeof(rt_hip_probe_in),
                hipMemcpyHostToDevice));
            hipLaunchKernelGGL(probe_kernel, dim3((count + 63u) / 64u), dim3(64), 0, nullptr,
                r->scene_ready ? r->sc : scene_dev{}, kind, count, d_in, d_out);
            HIP_CHECK(hipGetLastError());
            HIP_CHECK(hipDeviceSynchronize());
            HIP_CHECK(hipMemcpy(out, d_out, (size_t) count * sizeof(rt_hip_probe_out),
                hipMemcpyDeviceToHost));
        } catch (...) {
            (void) hipFree(d_in);
            (void) hipFree(d_out);
            throw;
        }
        (void) hipFree(d_in);
        (void) hipFree(d_out);
    });
}

int rt_hip_triangle_never_hit(const float* corners, uint32_t count, uint32_t* out) {
    return guarded([&] {
        if ((!corners || !out) && count > 0) {
            throw std::runtime_error("triangle_never_hit: null argument");
        }
        for (uint32_t i = 0; i < count; ++i) {
            rt_triangle tri{};
            for (int k = 0; k < 3; ++k) {
                tri.a.position_texu[k] = corners[9u * i + k];
                tri.b.position_texu[k] = corners[9u * i + 3 + k];
                tri.c.position_texu[k] = corners[9u * i + 6 + k];
            }
            out[i] = triangle_never_hit(tri) ? 1u : 0u;
        }
    });
}

int rt_hip_read_accum(rt_hip_renderer* r, float* rgba, uint32_t* passes, void* stream) {
    return guarded([&] {
        if (!r || !rgba) {
            throw std::runtime_error("read_accum: null argument");
        }
        device_scope const on_device(r->device);
        HIP_CHECK(hipStreamSynchronize((hipStream_t) stream));
        HIP_CHECK(hipMemcpy(rgba, r->accum, (size_t) r->cfg.width * r->cfg.height * 16,
            hipMemcpyDeviceToHost));
        if (passes) {
            *passes = r->passes_done;
        }
    });
}

int rt_hip_read_preview(rt_hip_renderer* r, float* rgba, void* stream) {
    return guarded([&] {
        if (!r || !rgba) {
            throw std::runtime_error("read_preview: null argument");
        }
        device_scope const on_device(r->device);
        HIP_CHECK(hipStreamSynchronize((hipStream_t) stream));
        HIP_CHECK(hipMemcpy(rgba, r->preview, (size_t) r->preview_w * r->preview_h * 16,
            hipMemcpyDeviceToHost));
    });
}

int rt_hip_accum_ptr(rt_hip_renderer* r, void** accum_dev) {
    return guarded([&] {
        if (!r || !accum_dev) {
            throw std::runtime_error("accum_ptr: null argument");
        }
        *accum_dev = r->accum;
    });
}

int rt_hip_passes(rt_hip_renderer* r, uint32_t* passes) {
    return guarded([&] {
        if (!r || !passes) {
            throw std::runtime_error("passes: null argument");
        }
        *passes = r->passes_done;
    });
}

int rt_hip_set_passes(rt_hip_renderer* r, uint32_t passes) {
    return guarded([&] {
        if (!r) {
            throw std::runtime_error("null renderer");
        }
        r->passes_done = passes;
    });
}

int rt_hip_stats(rt_hip_renderer* r, rt_stats* out, int reset) {
    return guarded([&] {
        if (!r) {
            throw std::runtime_error("null renderer");
        }
        device_scope const on_device(r->device);
        HIP_CHECK(hipDeviceSynchronize());
        r->fold_timings();
        unsigned long long words[kCounterWords];
        HIP_CHECK(hipMemcpy(words, r->counters, sizeof(words), hipMemcpyDeviceToHost));
        if (r->wf.wave_stats != nullptr) {
            // the pipeline keeps its work counters per wave
            size_t const waves = 2u * (size_t) r->wf_stat_waves;  // both state sets' halves
            std::vector<unsigned long long> per_wave(waves * kStatWords);
            HIP_CHECK(hipMemcpy(per_wave.data(), r->wf.wave_stats,
                per_wave.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
            for (size_t wv = 0; wv < waves; ++wv) {
                unsigned long long const* s = &per_wave[wv * kStatWords];
                words[0] += s[kStatSamples];
                words[1] += s[kStatSamples] + s[kStatClosestRays];
                words[2] += s[kStatShadowRays];
                words[3] += s[kStatClosestHits];
                words[7] += s[kStatShadowSkipped];
                words[4] += s[kStatNodes];
                words[5] += s[kStatTris];
                words[6] = std::max(words[6], s[kStatMaxStack]);
            }
            if (reset) {
                HIP_CHECK(hipMemset(r->wf.wave_stats, 0,
                    per_wave.size() * sizeof(unsigned long long)));
            }
        }
        if (out) {
            out->samples = words[0];
            out->closest_rays = words[1];
            out->shadow_rays = words[2];
            out->closest_hits = words[3];
            out->nodes_visited = words[4];
            out->tris_tested = words[5];
            out->max_stack = words[6];
            out->shadow_skipped = words[7];
            out->seconds = r->seconds_total[RT_HIP_KERNEL_RENDER];
        }
#ifdef RT_PHASE_TIMING
        fprintf(stderr,
            "phase cycles: shadow %llu closest %llu total %llu | wave trips %llu, lanes with "
            "shadow ray %llu, with closest ray %llu\n",
            words[8], words[9], words[10], words[11], words[12], words[13]);
#endif
        if (reset) {
            HIP_CHECK(hipMemset(r->counters, 0, sizeof(words)));
            for (int k = 0; k < 3; ++k) {
                r->seconds_total[k] = 0.0;
                r->launches_total[k] = 0;
                r->busy_seconds[k] = 0.0;
            }
        }
    });
}

int rt_hip_kernel_time_of(rt_hip_renderer* r, uint32_t kernel, double* avg_ms,
    uint32_t* launches) {
    return guarded([&] {
        if (!r) {
            throw std::runtime_error("null renderer");
        }
        if (kernel > RT_HIP_KERNEL_SHADE) {
            throw std::runtime_error("kernel_time_of: unknown kernel");
        }
        device_scope const on_device(r->device);
        r->fold_timings();
        if (avg_ms) {
            *avg_ms = r->launches_total[kernel]
                          ? r->seconds_total[kernel] * 1e3 / r->launches_total[kernel]
                          : 0.0;
        }
        if (launches) {
            *launches = r->launches_total[kernel];
        }
    });
}

int rt_hip_kernel_busy_of(rt_hip_renderer* r, uint32_t kernel, double* busy_ms) {
    return guarded([&] {
        if (!r || !busy_ms) {
            throw std::runtime_error("kernel_busy_of: null argument");
        }
        if (kernel > RT_HIP_KERNEL_SHADE) {
            throw std::runtime_error("kernel_busy_of: unknown kernel");
        }
        device_scope const on_device(r->device);
        r->fold_timings();
        *busy_ms = r->busy_seconds[kernel] * 1e3;
    });
}

int rt_hip_trip_counts(rt_hip_renderer* r, uint32_t trip, uint32_t* counts) {
    return guarded([&] {
        if (!r || !counts) {
            throw std::runtime_error("trip_counts: null argument");
        }
        if (r->wf_allocs[0].empty() || trip >= r->wf_trips[0]) {
            throw std::runtime_error("trip_counts: no pipeline batch has run that many trips");
        }
        device_scope const on_device(r->device);
        HIP_CHECK(hipDeviceSynchronize());
        counts[0] = counts[1] = counts[2] = 0;
        // a call of two batches ran one on each state set: their last batches together
        for (uint32_t set = 0; set < r->wf_sets_in_last_call; ++set) {
            uint32_t words[kCtlWords];
            HIP_CHECK(hipMemcpy(words, r->wf_set[set].control + (size_t) trip * kCtlWords,
                sizeof(words), hipMemcpyDeviceToHost));
            counts[0] += words[kCtlLive];
            counts[1] += words[kCtlClosest];
            counts[2] += words[kCtlShadow];
        }
    });
}

int rt_hip_trace_work(rt_hip_renderer* r, rt_hip_trace_work_counts* out) {
    return guarded([&] {
        if (!r || !out) {
            throw std::runtime_error("trace_work: null argument");
        }
        device_scope const on_device(r->device);
        HIP_CHECK(hipDeviceSynchronize());
        *out = rt_hip_trace_work_counts{};
        if (r->wf.wave_stats == nullptr) {
            return;
        }
        size_t const waves = 2u * (size_t) r->wf_stat_waves;  // both state sets' halves
        std::vector<unsigned long long> per_wave(waves * kStatWords);
        HIP_CHECK(hipMemcpy(per_wave.data(), r->wf.wave_stats,
            per_wave.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        for (size_t wv = 0; wv < waves; ++wv) {
            unsigned long long const* s = &per_wave[wv * kStatWords];
            out->nodes += s[kStatNodes];
            out->tris += s[kStatTris];
            out->nodes_from_lds += s[kStatLdsNodes];
            out->instance_records += s[kStatInstances];
            out->instance_records_from_lds += s[kStatLdsInstances];
            out->finish_nodes += s[kStatFinishNodes];
            out->finish_tris += s[kStatFinishTris];
            out->node_wave_steps += s[kStatNodeWaveSteps];
            out->triangle_wave_steps += s[kStatTriWaveSteps];
            out->instance_wave_steps += s[kStatInstanceWaveSteps];
        }
        out->node_bytes_cached = (uint64_t) r->sc.cached_nodes * sizeof(wide_node);
        out->node_bytes = r->accel_node_bytes;
        out->triangle_bytes = r->accel_tri_bytes;
        out->instance_bytes = r->accel_instance_bytes;
    });
}

int rt_hip_kernel_time(rt_hip_renderer* r, double* avg_ms, uint32_t* launches) {
    return rt_hip_kernel_time_of(r, RT_HIP_KERNEL_RENDER, avg_ms, launches);
}

int rt_hip_ray_binning(rt_hip_renderer* r, uint32_t* state, double* probe_ns_per_ray) {
    return guarded([&] {
        if (!r || !state || !probe_ns_per_ray) {
            throw std::runtime_error("ray_binning: null argument");
        }
        device_scope const on_device(r->device);
        settle_probe(*r);
        *state = r->binning;
        *probe_ns_per_ray = r->probe_ns_per_ray;
    });
}

int rt_hip_reduce(rt_hip_renderer* const* renderers, uint32_t count) {
    return guarded([&] { reduce_renderers(renderers, count); });
}

int rt_hip_trace_refill(rt_hip_renderer* r, uint32_t* state, double* ms_off, double* ms_on,
    uint32_t* trip_mask) {
    return guarded([&] {
        if (!r || !state) {
            throw std::runtime_error("trace_refill: null argument");
        }
        device_scope const on_device(r->device);
        settle_probe(*r);
        *state = r->refill;
        if (ms_off) {
            *ms_off = r->refill_ms[0];
        }
        if (ms_on) {
            *ms_on = r->refill_ms[1];
        }
        if (trip_mask) {
            uint32_t mask = r->refill == RT_HIP_REFILL_OFF ? 0u : ~0u;
            if (!r->refill_trip.empty()) {
                mask = 0u;
                for (uint32_t bit = 0; bit < 32u; ++bit) {
                    size_t const trip = std::min<size_t>(bit, r->refill_trip.size() - 1);
                    mask |= r->refill_trip[trip] != 0 ? 1u << bit : 0u;
                }
            }
            *trip_mask = mask;
        }
    });
}

int rt_hip_destroy(rt_hip_renderer* r) {
    return guarded([&] {
        if (!r) {
            return;
        }
        device_scope const on_device(r->device);
        (void) hipDeviceSynchronize();
#ifdef RT_SHADE_REGION_TIMING
        {
            unsigned long long host[2][12] = {};
            (void) hipMemcpyFromSymbol(host, HIP_SYMBOL(g_region_cycles), sizeof(host));
            for (int f = 0; f < 2; ++f) {
                unsigned long long sum = 0;
                for (int k = 0; k < 12; ++k) {
                    sum += host[f][k];
                }
                fprintf(stderr, "shade regions %s:", f == 0 ? "trip 0" : "later trips");
                for (int k = 0; k < 12; ++k) {
                    fprintf(stderr, " [%d] %.1f%%", k, sum ? 100.0 * (double) host[f][k] / (double) sum : 0.0);
                }
                fprintf(stderr, "  total %.3g cycles\n", (double) sum);
            }
            unsigned long long const zero[2][12] = {};
            (void) hipMemcpyToSymbol(HIP_SYMBOL(g_region_cycles), zero, sizeof(zero));
        }
#endif
        r->free_scene();
        r->free_wavefront();
        (void) hipFree(r->wf.wave_stats);
        for (auto& t : r->launches) {
            (void) hipEventDestroy(t.start);
            (void) hipEventDestroy(t.stop);
        }
        for (auto& t : r->event_pool) {
            (void) hipEventDestroy(t.start);
            (void) hipEventDestroy(t.stop);
        }
        for (hipEvent_t e : r->probe_events) {
            (void) hipEventDestroy(e);
        }
        (void) hipEventDestroy(r->probe_done);
        if (r->time_base != nullptr) {
            (void) hipEventDestroy(r->time_base);
        }
        if (r->side_stream != nullptr) {
            (void) hipStreamDestroy(r->side_stream);
            (void) hipEventDestroy(r->fork_event);
            (void) hipEventDestroy(r->join_event);
            (void) hipEventDestroy(r->resolved_event[0]);
            (void) hipEventDestroy(r->resolved_event[1]);
        }
        (void) hipEventDestroy(r->last_done);
        if (r->probe_counts) {
            (void) hipHostFree(r->probe_counts);
        }
        (void) hipFree(r->accum);
        (void) hipFree(r->preview);
        (void) hipFree(r->counters);
        (void) hipFree(r->block_cost);
        (void) hipFree(r->block_order);
        (void) hipFree(r->seeds);
        delete r;
    });
}

const char* rt_hip_last_error(void) {
    return g_last_error.c_str();
}

}  // extern "C"

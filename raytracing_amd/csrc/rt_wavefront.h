// Wavefront form of the path tracer: the bounce loop of megakernel_raytracer.comp:509-565
// cut into small kernels that hand paths to each other through queues in HBM.
//
// Why: the fused kernel needs ~200 VGPRs (the Disney BSDF), which leaves two waves per SIMD,
// and traversal is a chain of dependent loads - it runs at the speed of memory latency
// divided by the number of resident waves.  The traversal kernel alone fits in well under
// 100 VGPRs, so five waves per SIMD hide that latency, and because it pulls rays from a
// compacted queue every lane of a wave has a ray (in the fused kernel only about half the
// lanes have an occlusion ray to trace).
//
// All paths of a batch of passes are in flight at once.  Path `pid` = (pass, 8x8 pixel
// block, lane) owns slot `pid` of every state array for its whole life; queues carry pids.
// One trip = one `extend` launch (trace the previous bounce's occlusion ray and this
// bounce's closest-hit ray of every live path) + one `shade` launch (shade_trip, the same
// function the fused kernel calls).  Finished paths leave their radiance in `rad[pid]`;
// `resolve` adds a pixel's passes in pass order, so the image is bit-identical to the
// reference's one-dispatch-per-pass accumulation.
//
// No queue is filled through an atomic counter: tens of thousands of waves adding to one
// address serialise (measured: ~7.5 ns per atomic, more than the shading itself).  Instead
// a producer writes one word per input slot ("raw": the surviving path with its ray flags,
// or 0) plus one packed count per 64 slots; a scan turns the counts into offsets and a
// scatter kernel writes the compacted lists.  The lists therefore keep pid
// order from trip to trip - neighbouring pixels stay neighbours, which keeps the state
// arrays' accesses coalesced and the rays of a wave coherent - and a render is
// reproducible down to the order of its memory accesses.  (Where late-bounce rays are
// expensive, the ray queues - not the live list - are regrouped before the trace launch,
// see "Ray binning" below; the order inside a bin then depends on the timing of LDS atomics,
// which no result depends on.)
//
// The per-trip kernels run on a fixed grid of resident waves; a wave takes chunks w, w + W, ...
// of a list, or groups of chunks from a counter (list lengths exist only on the device).  Work counters are kept per wave
// (wave_stats) and summed when they are read.
#pragma once

#include "rt_path.h"

namespace rtd {

// Words of the per-trip control block.
enum : uint32_t {
    kCtlLive = 0,     // entries in this trip's live list
    kCtlClosest = 1,  // entries in this trip's closest-hit ray queue
    kCtlShadow = 2,   // entries in this trip's occlusion ray queue
    kCtlFinish = 3,   // entries of the live list handed to the finishing kernel instead
    kCtlRegionCursor = 4,  // next batch of regions of the `going` list to be shaded (wf_shade_kernel)
    kCtlGoing = 5,         // paths on the `going` list
    kCtlClosestCursor = 6, // next group of chunks of the closest-hit queue (dynamic hand-out)
    kCtlShadowCursor = 7,  // likewise for the occlusion queue
    kCtlWords = 8,
};

// Per-wave work counters (wave_stats[slot][k]); slot = wave index in its launch.
enum : uint32_t {
    kStatSamples = 7,      // paths started (trip 0 has no list whose length could be read)
    kStatClosestRays = 0,  // rays traced by the finishing kernel only (the per-trip launches
    kStatShadowRays = 6,   // take their ray counts from the list lengths)
    kStatClosestHits = 1,
    kStatShadowSkipped = 2,
    kStatNodes = 3,
    kStatTris = 4,
    kStatMaxStack = 5,
    // how the rebuilt structure's fetches were served (counted with RT_HIP_FLAG_COUNT_NODES)
    kStatLdsNodes = 8,       // of kStatNodes: read from LDS by the trace launches
    kStatInstances = 9,      // instance records fetched
    kStatLdsInstances = 10,  // of those: read from LDS by the trace launches
    kStatFinishNodes = 11,   // of kStatNodes / kStatTris: the finishing kernel's share
    kStatFinishTris = 12,
    kStatNodeWaveSteps = 13,      // trace launches: executions of each kind of step by a wave
    kStatTriWaveSteps = 14,
    kStatInstanceWaveSteps = 15,
    kStatWords = 16,
};
constexpr uint32_t kScanThreads = 1024;
constexpr uint32_t kScanTilesPerThread = 4;  // of the one-workgroup scan over the tile sums

// Live-list entry = pid | flags.
constexpr uint32_t kLiveHasShadow = 0x40000000u;
constexpr uint32_t kLiveHasClosest = 0x80000000u;
constexpr uint32_t kLivePidMask = 0x3FFFFFFFu;

// n / d for a divisor fixed per launch and n < 2^31, without the two dozen instructions of a
// 32-bit division: floor(n / d) = (mulhi(n, mul) + (n & keep)) >> shift with, for d a power of
// two, mul = 0, keep = ~0, shift = log2 d, and otherwise mul = ceil(2^(32 + shift) / d),
// keep = 0, shift = floor(log2 d) (make_divider checks the pair it returns).
struct divider {
    uint32_t mul, keep, shift;
};
RT_DEV uint32_t divide(uint32_t n, divider const& d) {
    return (__umulhi(n, d.mul) + (n & d.keep)) >> d.shift;
}

struct wavefront_params {
    scene_dev sc;
    rt_camera cam;
    // path state, one slot per pid
    float4* ray_o;     // closest-hit ray origin
    float4* ray_d;     // closest-hit ray direction
    float4* sh_o;      // occlusion ray origin, w = t_max
    float4* sh_d;      // occlusion ray direction
    float4* nee;       // pending next-event contribution, w = last BSDF pdf
    float4* thr;       // throughput, w = random seed bits
    float4* rad;       // radiance so far (final when the path is over), w = depth bits
    float4* hit0;      // instance bits, triangle bits, b0, b1   (instance = RT_HIP_MISS: no hit)
    float2* hit1;      // b2, t
    uint32_t* live;          // compacted: pid | flags of every path with a ray this trip
    uint32_t* q_closest;     // compacted pids with a closest-hit ray
    uint32_t* q_shadow;      // compacted pids with an occlusion ray
    uint32_t* raw;           // per input slot: pid | flags of the surviving path, or 0
    uint2* going;            // per region of kRegionSlots list slots: (slot, entry) of the paths
                             // the shading kernel has to shade, packed at the region's start
    uint32_t* going_count;   // per region: how many
    uint32_t* chunk_counts;  // per 64 raw slots: live | closest << 8 | shadow << 16
    uint32_t* tile_sums;     // per 1024 chunks: 3 words, sums and then offsets of the tile
    uint32_t* control;       // [trip][kCtlWords]
    // ray binning (sort_rays != 0): the queues regrouped by direction octant and origin cell,
    // with the rays copied next to them so that the trace launch reads them in queue order
    uint32_t sort_rays;
    uint32_t refill;  // trace launches: a wave hands its finished lanes new rays (rebuilt structure)
    float sort_lo[3], sort_scale[3];  // cell = (origin - lo) * scale, clamped to sort_top
    float sort_top[3];                // cells along each axis - 1
    uint32_t sort_shift[3];           // position of each axis' cell index in the bin number
    uint32_t* binned_closest;         // pids
    uint32_t* binned_shadow;
    uint16_t* bin_keys_closest;       // [queue position before binning]
    uint16_t* bin_keys_shadow;
    uint32_t* bin_hist;               // [kind][workgroup][bin]: counts, then offsets in the bin
    uint32_t* bin_total;              // [kind][bin]: sizes, then bases of the bins
    unsigned long long* wave_stats;  // [wave][kStatWords]
    unsigned long long* counters;    // closest / occlusion ray totals (one add per launch)
    // frame
    float* accum;
    const uint32_t* seeds;  // [pass][tile] of this batch, or nullptr -> scalar_seed
    uint32_t scalar_seed;
    uint32_t frame_width;
    uint32_t x0, y0, x1, y1;
    uint32_t blocks_x, block_count;  // 8x8 blocks covering the rectangle
    uint32_t padded;                 // the rectangle's sides are not multiples of 8: some lanes
                                     // of its 8x8 blocks have no pixel
    uint32_t tile_width, tile_height, tiles_x, tile_count;
    divider by_per_pass, by_blocks_x, by_tile_width, by_tile_height;  // block_count * 64, ...
    uint32_t pass_count;             // passes in this batch
    uint32_t max_depth;
    uint32_t preview;
};

RT_DEV uint32_t wave_lane() { return threadIdx.x & 63u; }

// Path state and queues are streamed: written by one launch, read once by the next, with
// gigabytes of other traffic in between.  Non-temporal accesses keep them from evicting the
// BVH and the triangles from the L2, which every ray re-reads.
typedef float native_f4 __attribute__((ext_vector_type(4)));
typedef float native_f2 __attribute__((ext_vector_type(2)));

RT_DEV float4 stream_load(const float4* p) {
    native_f4 const v = __builtin_nontemporal_load(reinterpret_cast<const native_f4*>(p));
    return make_float4(v.x, v.y, v.z, v.w);
}
RT_DEV float2 stream_load(const float2* p) {
    native_f2 const v = __builtin_nontemporal_load(reinterpret_cast<const native_f2*>(p));
    return make_float2(v.x, v.y);
}
RT_DEV uint32_t stream_load(const uint32_t* p) { return __builtin_nontemporal_load(p); }
// The first three floats of a float4 (a ray's origin or direction: the fourth is unused).  A
// four-float load whose last register is dead lets the compiler reuse that register at once,
// and the write then has to wait for the load: a prefetch would stall its issuer.
typedef float native_f3 __attribute__((ext_vector_type(3)));
RT_DEV f3 stream_load3(const float4* p) {
    native_f3 const v = __builtin_nontemporal_load(reinterpret_cast<const native_f3*>(p));
    return mk3(v.x, v.y, v.z);
}
RT_DEV void stream_store(float4* p, float4 v) {
    native_f4 n;
    n.x = v.x, n.y = v.y, n.z = v.z, n.w = v.w;
    __builtin_nontemporal_store(n, reinterpret_cast<native_f4*>(p));
}
RT_DEV void stream_store(float2* p, float2 v) {
    native_f2 n;
    n.x = v.x, n.y = v.y;
    __builtin_nontemporal_store(n, reinterpret_cast<native_f2*>(p));
}
RT_DEV void stream_store(uint32_t* p, uint32_t v) { __builtin_nontemporal_store(v, p); }

// Index of this wave in its launch and the number of waves launched.
RT_DEV uint32_t wave_index() { return (blockIdx.x * blockDim.x + threadIdx.x) >> 6; }
RT_DEV uint32_t wave_total() { return (gridDim.x * blockDim.x) >> 6; }

RT_DEV void add_wave_stat(wavefront_params const& p, uint32_t k, unsigned long long v) {
    // one lane per wave, no other wave touches this slot
    unsigned long long* const slot = p.wave_stats + (size_t) wave_index() * kStatWords + k;
    *slot += v;
}

RT_DEV unsigned long long wave_sum(uint32_t v) {
    unsigned long long total = 0;
    for (int offset = 32; offset > 0; offset >>= 1) {
        v += (uint32_t) __shfl_down((int) v, offset, 64);
    }
    total = (unsigned long long) (uint32_t) __builtin_amdgcn_readfirstlane((int) v);
    return total;
}

RT_DEV bool pid_pixel(wavefront_params const& p, uint32_t pid, uint32_t& x, uint32_t& y,
    uint32_t& pass) {
    uint32_t const per_pass = p.block_count * 64u;
    pass = divide(pid, p.by_per_pass);
    uint32_t const in_pass = pid - pass * per_pass;
    uint32_t const block = in_pass >> 6;
    uint32_t const lane = in_pass & 63u;
    uint32_t const block_y = divide(block, p.by_blocks_x);
    x = p.x0 + (block - block_y * p.blocks_x) * 8u + (lane & 7u);
    y = p.y0 + block_y * 8u + (lane >> 3);
    return x < p.x1 && y < p.y1;
}

// An occlusion ray that got through: its path's pending light sample becomes radiance
// (megakernel_raytracer.comp:547-549).  The trace launch does this itself, for the rays it finds
// unoccluded, instead of leaving a flag for the launches after it: those walk every live path
// and are bound by the memory they move, while the trace launch is bound by instruction issue
// and has the bandwidth to spare.  Nothing else touches rad[pid] between the shading launch
// that stored the sample and the one that would have added it, so the sum is the same.
// `rad` and `nee` are the path's rad[pid] and nee[pid], fetched together with the ray so that
// nothing waits for them here.
RT_DEV void add_light_sample(wavefront_params const& p, uint32_t pid, float4 rad, float4 nee) {
    rad.x = rad.x + nee.x;
    rad.y = rad.y + nee.y;
    rad.z = rad.z + nee.z;
    stream_store(&p.rad[pid], rad);
}

// Start of path `pid`: its pixel, its seed after the two jitter draws and its primary ray
// (megakernel_raytracer.comp:85-100).  Trip 0 has no stored rays: the trace launch derives the
// primary ray from the pid and leaves its direction and the seed in ray_d[pid] for the shading
// launch (16 bytes written and read per path against some 250 instructions of a launch that is
// bound by instruction issue: hyperion +4 %, teapot stand-in +2.6 %); the settle launch, which
// waits for memory and needs the ray of one path in seven, derives it again.
RT_DEV bool primary_of_pid(wavefront_params const& p, uint32_t pid, uint32_t& seed, f3& origin,
    f3& direction) {
    uint32_t x, y, pass;
    if (pid >= p.pass_count * p.block_count * 64u || !pid_pixel(p, pid, x, y, pass)) {
        return false;
    }
    uint32_t random_seed = p.scalar_seed;
    if (p.seeds != nullptr) {
        uint32_t const tile =
            divide(y, p.by_tile_height) * p.tiles_x + divide(x, p.by_tile_width);
        random_seed = p.seeds[(size_t) pass * p.tile_count + tile];
    }
    f3 const du = load3(p.cam.pixel_delta_u);
    f3 const dv = load3(p.cam.pixel_delta_v);
    f3 const ul = load3(p.cam.upper_left_pixel);
    f3 const pos = load3(p.cam.position);
    seed = hash_combine(random_seed, p.frame_width * y + x);
    f3 const pixel = (ul + du * (float) (int) x) + dv * (float) (int) y;
    float const ox = -0.5f + rand_01(seed);
    float const oy = -0.5f + rand_01(seed);
    f3 const pixel_sample = pixel + (du * ox + dv * oy);
    origin = pos;
    direction = normalize3(pixel_sample - pos);
    return true;
}

// Compaction of the raw slots into trip `trip`'s lists, in three small launches over tiles of
// kTileChunks chunks: (1) per-tile sums of the packed chunk counts, (2) one workgroup scans
// the tile sums and records the list lengths, (3) every tile scans its own chunk counts and
// scatters its survivors.  `in_count_ptr ? *in_count_ptr : in_count` = number of raw slots.
constexpr uint32_t kTileChunks = 1024;
constexpr uint32_t kTileThreads = 1024;

struct triple {
    uint32_t live, closest, shadow;
};

RT_DEV triple unpack_counts(uint32_t packed) {
    return triple{packed & 0xFFu, (packed >> 8) & 0xFFu, (packed >> 16) & 0xFFu};
}

// Inclusive scan of one value per thread over a workgroup of THREADS threads.
template <uint32_t THREADS>
RT_DEV triple block_inclusive_scan(triple v, uint32_t (*scratch)[THREADS]) {
    uint32_t const t = threadIdx.x;
    scratch[0][t] = v.live;
    scratch[1][t] = v.closest;
    scratch[2][t] = v.shadow;
    __syncthreads();
    for (uint32_t step = 1; step < THREADS; step <<= 1) {
        triple add{0u, 0u, 0u};
        if (t >= step) {
            add = triple{scratch[0][t - step], scratch[1][t - step], scratch[2][t - step]};
        }
        __syncthreads();
        scratch[0][t] += add.live;
        scratch[1][t] += add.closest;
        scratch[2][t] += add.shadow;
        __syncthreads();
    }
    return triple{scratch[0][t], scratch[1][t], scratch[2][t]};
}

#ifndef RT_SHADE_VARIANT_TU  // (a translation unit that only instantiates shading kernels)
// (1) also counts the raw slots, 64 at a time, and leaves the packed counts in chunk_counts:
// the producers write raw words only, in any order and from several launches.
__global__ void __launch_bounds__(kTileThreads) wf_tile_sums_kernel(wavefront_params const p,
    const uint32_t* in_count_ptr, uint32_t in_count) {
    __shared__ uint32_t scratch[3][kTileThreads];
    __shared__ uint32_t packed[kTileChunks];
    uint32_t const n = in_count_ptr ? *in_count_ptr : in_count;
    uint32_t const chunks = (n + 63u) / 64u;
    uint32_t const first = blockIdx.x * kTileChunks;
    if (first >= chunks) {
        return;
    }
    uint32_t const tile_chunks = min(kTileChunks, chunks - first);
    for (uint32_t c = threadIdx.x >> 6; c < tile_chunks; c += kTileThreads / 64u) {
        uint32_t const slot = (first + c) * 64u + wave_lane();
        uint32_t const entry = slot < n ? stream_load(&p.raw[slot]) : 0u;
        unsigned long long const live = __ballot(entry != 0u);
        unsigned long long const closest = __ballot((entry & kLiveHasClosest) != 0u);
        unsigned long long const shadow = __ballot((entry & kLiveHasShadow) != 0u);
        if (wave_lane() == 0) {
            uint32_t const counts = (uint32_t) __builtin_popcountll(live) |
                                    ((uint32_t) __builtin_popcountll(closest) << 8) |
                                    ((uint32_t) __builtin_popcountll(shadow) << 16);
            packed[c] = counts;
            p.chunk_counts[first + c] = counts;
        }
    }
    __syncthreads();
    triple sum{0u, 0u, 0u};
    for (uint32_t c = threadIdx.x; c < tile_chunks; c += kTileThreads) {
        triple const v = unpack_counts(packed[c]);
        sum.live += v.live;
        sum.closest += v.closest;
        sum.shadow += v.shadow;
    }
    triple const total = block_inclusive_scan<kTileThreads>(sum, scratch);
    if (threadIdx.x == kTileThreads - 1u) {
        p.tile_sums[3u * blockIdx.x + 0u] = total.live;
        p.tile_sums[3u * blockIdx.x + 1u] = total.closest;
        p.tile_sums[3u * blockIdx.x + 2u] = total.shadow;
    }
}

// handover_below: when fewer paths than this are alive, the trip and all later ones are handed
// to the finishing kernel: the list lengths that the trace and shade launches read become
// zero (they and every later launch of the batch return at once) and kCtlFinish holds the
// length of the live list instead.  Whether few paths are left after two bounces or after ten
// depends on the scene; the host cannot know without reading back.
__global__ void __launch_bounds__(kScanThreads) wf_scan_tiles_kernel(wavefront_params const p,
    uint32_t trip, const uint32_t* in_count_ptr, uint32_t in_count, uint32_t handover_below) {
    __shared__ uint32_t scratch[3][kScanThreads];
    uint32_t const n = in_count_ptr ? *in_count_ptr : in_count;
    uint32_t const chunks = (n + 63u) / 64u;
    uint32_t const tiles = (chunks + kTileChunks - 1u) / kTileChunks;
    // thread t takes tiles t * kScanTilesPerThread ...: their sum goes through the workgroup's
    // scan, their own offsets follow from the thread's exclusive prefix
    uint32_t const t = threadIdx.x;
    uint32_t const first_tile = t * kScanTilesPerThread;
    triple own[kScanTilesPerThread];
    triple mine{0u, 0u, 0u};
    for (uint32_t k = 0; k < kScanTilesPerThread; ++k) {
        uint32_t const tile = first_tile + k;
        own[k] = triple{0u, 0u, 0u};
        if (tile < tiles) {
            own[k] = triple{p.tile_sums[3u * tile], p.tile_sums[3u * tile + 1u],
                p.tile_sums[3u * tile + 2u]};
        }
        mine.live += own[k].live;
        mine.closest += own[k].closest;
        mine.shadow += own[k].shadow;
    }
    triple const incl = block_inclusive_scan<kScanThreads>(mine, scratch);
    triple run{incl.live - mine.live, incl.closest - mine.closest, incl.shadow - mine.shadow};
    for (uint32_t k = 0; k < kScanTilesPerThread; ++k) {
        uint32_t const tile = first_tile + k;
        if (tile < tiles) {
            p.tile_sums[3u * tile + 0u] = run.live;  // exclusive: the tile's offsets
            p.tile_sums[3u * tile + 1u] = run.closest;
            p.tile_sums[3u * tile + 2u] = run.shadow;
        }
        run.live += own[k].live;
        run.closest += own[k].closest;
        run.shadow += own[k].shadow;
    }
    if (t == kScanThreads - 1u) {
        uint32_t* const ctl = p.control + trip * kCtlWords;
        bool const handover = incl.live < handover_below;
        ctl[kCtlLive] = handover ? 0u : incl.live;
        ctl[kCtlClosest] = handover ? 0u : incl.closest;
        ctl[kCtlShadow] = handover ? 0u : incl.shadow;
        ctl[kCtlFinish] = handover ? incl.live : 0u;
        if (!handover) {
            // the finishing kernel counts the rays it traces itself; single thread of a
            // single workgroup - but two batches of a render call run side by side on two
            // streams, so the two scans may add at the same time
            atomicAdd(&p.counters[1], (unsigned long long) incl.closest);
            atomicAdd(&p.counters[2], (unsigned long long) incl.shadow);
        }
    }
}

__global__ void __launch_bounds__(kTileThreads) wf_scatter_kernel(wavefront_params const p,
    const uint32_t* in_count_ptr, uint32_t in_count) {
    __shared__ uint32_t scratch[3][kTileThreads];
    __shared__ uint32_t base[3][kTileChunks];
    uint32_t const n = in_count_ptr ? *in_count_ptr : in_count;
    uint32_t const chunks = (n + 63u) / 64u;
    uint32_t const first = blockIdx.x * kTileChunks;
    if (first >= chunks) {
        return;
    }
    // thread t owns chunks first + 4t .. first + 4t + 3
    constexpr uint32_t kPerThread = kTileChunks / kTileThreads;
    triple own[kPerThread];
    triple sum{0u, 0u, 0u};
    for (uint32_t k = 0; k < kPerThread; ++k) {
        uint32_t const c = first + threadIdx.x * kPerThread + k;
        own[k] = c < chunks ? unpack_counts(p.chunk_counts[c]) : triple{0u, 0u, 0u};
        sum.live += own[k].live;
        sum.closest += own[k].closest;
        sum.shadow += own[k].shadow;
    }
    triple const incl = block_inclusive_scan<kTileThreads>(sum, scratch);
    triple run{p.tile_sums[3u * blockIdx.x + 0u] + incl.live - sum.live,
        p.tile_sums[3u * blockIdx.x + 1u] + incl.closest - sum.closest,
        p.tile_sums[3u * blockIdx.x + 2u] + incl.shadow - sum.shadow};
    for (uint32_t k = 0; k < kPerThread; ++k) {
        uint32_t const c = threadIdx.x * kPerThread + k;
        base[0][c] = run.live;
        base[1][c] = run.closest;
        base[2][c] = run.shadow;
        run.live += own[k].live;
        run.closest += own[k].closest;
        run.shadow += own[k].shadow;
    }
    __syncthreads();
    unsigned long long const below = (1ull << wave_lane()) - 1ull;
    uint32_t const wave_in_block = threadIdx.x >> 6;
    for (uint32_t c = wave_in_block; c < kTileChunks && first + c < chunks;
         c += kTileThreads / 64u) {
        uint32_t const slot = (first + c) * 64u + wave_lane();
        uint32_t const entry = slot < n ? p.raw[slot] : 0u;
        bool const has_closest = (entry & kLiveHasClosest) != 0u;
        bool const has_shadow = (entry & kLiveHasShadow) != 0u;
        unsigned long long const live = __ballot(entry != 0u);
        unsigned long long const closest = __ballot(has_closest);
        unsigned long long const shadow = __ballot(has_shadow);
        if (entry != 0u) {
            p.live[base[0][c] + (uint32_t) __builtin_popcountll(live & below)] = entry;
        }
        if (has_closest) {
            p.q_closest[base[1][c] + (uint32_t) __builtin_popcountll(closest & below)] =
                entry & kLivePidMask;
        }
        if (has_shadow) {
            p.q_shadow[base[2][c] + (uint32_t) __builtin_popcountll(shadow & below)] =
                entry & kLivePidMask;
        }
    }
}

// Ray binning.  After a diffuse bounce the rays of neighbouring pixels point anywhere; a wave
// tracing 64 of them walks 64 different parts of the tree and most of its lanes idle at any
// one step.  Regrouping a trip's queue so that a chunk of 64 holds rays that start in the same
// cell of a coarse grid over the scene and point into the same octant makes a wave's rays
// visit the same nodes again.  One counting sort per queue and trip: every
// workgroup histograms its contiguous share of the queue in LDS (count), two small launches
// turn the [workgroup][bin] counts into offsets (columns, bases), and every workgroup walks its
// share again
// and writes each pid to the next free place of its bin (place).  Within a bin the
// order is that of the shares, i.e. still roughly pid order.  Which lane traces which ray has
// no influence on any result.
constexpr uint32_t kBinCellBits = 10;                // 1024 cells, split over the axes by extent
constexpr uint32_t kBinCount = 32u << kBinCellBits;  // x 24 directions (numbered 0..31)
constexpr uint32_t kBinThreads = 1024;

RT_DEV uint32_t ray_bin(wavefront_params const& p, float4 o, float4 d) {
    // fmaxf(NaN, 0) = 0: a ray with a NaN origin lands in cell 0
    uint32_t const cx =
        (uint32_t) fminf(fmaxf((o.x - p.sort_lo[0]) * p.sort_scale[0], 0.0f), p.sort_top[0]);
    uint32_t const cy =
        (uint32_t) fminf(fmaxf((o.y - p.sort_lo[1]) * p.sort_scale[1], 0.0f), p.sort_top[1]);
    uint32_t const cz =
        (uint32_t) fminf(fmaxf((o.z - p.sort_lo[2]) * p.sort_scale[2], 0.0f), p.sort_top[2]);
    uint32_t const cell = (cx << p.sort_shift[0]) | (cy << p.sort_shift[1]) | (cz << p.sort_shift[2]);
    // direction: octant and dominant axis (24 classes, numbered 0..31)
    uint32_t const octant = (d.x < 0.0f ? 1u : 0u) | (d.y < 0.0f ? 2u : 0u) | (d.z < 0.0f ? 4u : 0u);
    float const ax = fabsf(d.x), ay = fabsf(d.y), az = fabsf(d.z);
    uint32_t const major = ax >= ay ? (ax >= az ? 0u : 2u) : (ay >= az ? 1u : 2u);
    return (((major << 3) | octant) << kBinCellBits) | cell;  // < kBinCount
}

// The share of workgroup `g` of `groups` in a queue of n entries: [first, last), whole chunks.
RT_DEV void bin_share(uint32_t n, uint32_t g, uint32_t groups, uint32_t& first, uint32_t& last) {
    uint32_t const per_group = (((n + groups - 1u) / groups) + 63u) & ~63u;
    first = min(n, g * per_group);
    last = min(n, first + per_group);
}

// grid (groups, 2): y = 0 closest-hit queue, y = 1 occlusion queue
__global__ void __launch_bounds__(kBinThreads) wf_bin_count_kernel(wavefront_params const p,
    uint32_t trip) {
    extern __shared__ uint32_t hist[];  // kBinCount words
    uint32_t const kind = blockIdx.y;
    uint32_t const n = p.control[trip * kCtlWords + (kind ? kCtlShadow : kCtlClosest)];
    if (n == 0u) {
        return;  // the scan and place launches of this queue return as well
    }
    const uint32_t* const queue = kind ? p.q_shadow : p.q_closest;
    const float4* const src_o = kind ? p.sh_o : p.ray_o;
    const float4* const src_d = kind ? p.sh_d : p.ray_d;
    for (uint32_t b = threadIdx.x; b < kBinCount; b += kBinThreads) {
        hist[b] = 0u;
    }
    __syncthreads();
    uint32_t first, last;
    bin_share(n, blockIdx.x, gridDim.x, first, last);
    uint16_t* const keys = kind ? p.bin_keys_shadow : p.bin_keys_closest;
    for (uint32_t i = first + threadIdx.x; i < last; i += kBinThreads) {
        uint32_t const pid = queue[i];
        uint32_t const bin = ray_bin(p, src_o[pid], src_d[pid]);
        keys[i] = (uint16_t) bin;
        atomicAdd(&hist[bin], 1u);
    }
    __syncthreads();
    uint32_t* const out = p.bin_hist + ((size_t) kind * gridDim.x + blockIdx.x) * kBinCount;
    for (uint32_t b = threadIdx.x; b < kBinCount; b += kBinThreads) {
        out[b] = hist[b];
    }
}

// grid (kBinCount / kBinColumnThreads, 2): one thread per bin walks the bin's column of
// per-workgroup counts (neighbouring threads read neighbouring words), turns it into offsets
// within the bin and leaves the bin's size in bin_total.
constexpr uint32_t kBinColumnThreads = 256;
__global__ void __launch_bounds__(kBinColumnThreads) wf_bin_columns_kernel(wavefront_params const p,
    uint32_t trip, uint32_t groups) {
    uint32_t const kind = blockIdx.y;
    uint32_t const n = p.control[trip * kCtlWords + (kind ? kCtlShadow : kCtlClosest)];
    uint32_t const bin = blockIdx.x * kBinColumnThreads + threadIdx.x;
    if (n == 0u || bin >= kBinCount) {
        return;
    }
    uint32_t* const column = p.bin_hist + (size_t) kind * groups * kBinCount + bin;
    uint32_t running = 0;
    uint32_t g = 0;
    for (; g + 8u <= groups; g += 8u) {
        uint32_t c[8];
        for (uint32_t k = 0; k < 8u; ++k) {
            c[k] = column[(size_t) (g + k) * kBinCount];
        }
        for (uint32_t k = 0; k < 8u; ++k) {
            column[(size_t) (g + k) * kBinCount] = running;
            running += c[k];
        }
    }
    for (; g < groups; ++g) {
        uint32_t const c = column[(size_t) g * kBinCount];
        column[(size_t) g * kBinCount] = running;
        running += c;
    }
    p.bin_total[kind * kBinCount + bin] = running;
}

// grid (2): exclusive scan of one queue's bin sizes, in place (they become the bins' bases).
__global__ void __launch_bounds__(kBinThreads) wf_bin_bases_kernel(wavefront_params const p,
    uint32_t trip) {
    __shared__ uint32_t scratch[kBinThreads];
    uint32_t const kind = blockIdx.x;
    uint32_t const n = p.control[trip * kCtlWords + (kind ? kCtlShadow : kCtlClosest)];
    if (n == 0u) {
        return;
    }
    uint32_t* const totals = p.bin_total + kind * kBinCount;
    constexpr uint32_t kPerThread = (kBinCount + kBinThreads - 1u) / kBinThreads;
    uint32_t const first = threadIdx.x * kPerThread;
    uint32_t own[kPerThread];
    uint32_t sum = 0;
    for (uint32_t k = 0; k < kPerThread; ++k) {
        own[k] = first + k < kBinCount ? totals[first + k] : 0u;
        sum += own[k];
    }
    uint32_t const t = threadIdx.x;
    scratch[t] = sum;
    __syncthreads();
    for (uint32_t step = 1; step < kBinThreads; step <<= 1) {
        uint32_t const add = t >= step ? scratch[t - step] : 0u;
        __syncthreads();
        scratch[t] += add;
        __syncthreads();
    }
    uint32_t run = scratch[t] - sum;
    for (uint32_t k = 0; k < kPerThread; ++k) {
        if (first + k < kBinCount) {
            totals[first + k] = run;
        }
        run += own[k];
    }
}

// grid (groups, 2), the same shares as the count launch
__global__ void __launch_bounds__(kBinThreads) wf_bin_place_kernel(wavefront_params const p,
    uint32_t trip) {
    extern __shared__ uint32_t cursor[];  // kBinCount words
    uint32_t const kind = blockIdx.y;
    uint32_t const n = p.control[trip * kCtlWords + (kind ? kCtlShadow : kCtlClosest)];
    if (n == 0u) {
        return;
    }
    const uint32_t* const queue = kind ? p.q_shadow : p.q_closest;
    const uint16_t* const keys = kind ? p.bin_keys_shadow : p.bin_keys_closest;
    uint32_t* const out_queue = kind ? p.binned_shadow : p.binned_closest;
    const uint32_t* const offsets = p.bin_hist + ((size_t) kind * gridDim.x + blockIdx.x) * kBinCount;
    const uint32_t* const bases = p.bin_total + kind * kBinCount;
    for (uint32_t b = threadIdx.x; b < kBinCount; b += kBinThreads) {
        cursor[b] = bases[b] + offsets[b];
    }
    __syncthreads();
    uint32_t first, last;
    bin_share(n, blockIdx.x, gridDim.x, first, last);
    for (uint32_t i = first + threadIdx.x; i < last; i += kBinThreads) {
        uint32_t const place = atomicAdd(&cursor[keys[i]], 1u);
        if (place < n) {  // always: the counts and the places come from the same keys
            out_queue[place] = queue[i];
        }
    }
}

#endif  // RT_SHADE_VARIANT_TU

// Traces every ray of trip `trip`: first the closest-hit queue, then the occlusion queue, so
// a wave never mixes the two kinds.  Two ways of handing the rays to the lanes, chosen per
// launch by `refill` (the host measures both per uploaded scene and bounce, rt_hip.hip:
// settle_probe): fixed chunks - a wave takes 64 rays and walks them to the end of the slowest -
// or refilled lanes - every lane keeps a walk of its own, and once kRefillIdle lanes have
// finished theirs they take the wave's next rays together (refilling lane by lane, without
// that threshold, repeated the per-ray set-up at partial occupancy and was 1.8x slower on
// primary rays).  Coherent camera rays do best in fixed chunks, the rays of later bounces
// refilled.
//
// Rebuilt acceleration structure: workgroups of kExtendThreads threads; LDS holds, in this
// order, the hottest nodes, the hottest instance records (both copied in by the workgroup
// before it starts) and every wave's traversal stack.  Reference-order structure: one wave
// per workgroup, private stack, no LDS.
constexpr uint32_t kExtendThreads = 1024;
#ifndef RT_PERSISTENT_TRACE
#define RT_PERSISTENT_TRACE 1
#endif
#ifndef RT_REFILL_IDLE
#define RT_REFILL_IDLE 32
#endif
// The refilling trace launches hand their queue out in groups of this many 64-ray chunks, from
// a counter in the trip's control block: one atomic per 512 rays.  A fixed share per wave (chunks
// w, w + W, ...) left the waves with cheap rays idle at the end of every launch (hand-out by
// counter: hyperion +1.6 %, mustang +2.5 %, staircase +3.3 %, the 10 000-instance scene +4.3 %);
// groups of 2 chunks serialise on the counter (hyperion -15 %), 4 and 16 are within 1 % of 8,
// 32 bring the idle tails back (-3 ... -5 %).
#ifndef RT_HAND_OUT_CHUNKS
#define RT_HAND_OUT_CHUNKS 8
#endif
constexpr uint32_t kHandOutChunks = RT_HAND_OUT_CHUNKS;
constexpr uint32_t kRefillIdle = RT_REFILL_IDLE;  // lanes that must wait before a wave refills
constexpr uint32_t kLdsBytesPerCu = 160u * 1024u;

// First chunk of the wave's next group (all lanes get it); past the queue's end when it is
// used up.  Which wave traces which ray has no influence on any result.
RT_DEV uint32_t next_chunks(uint32_t* cursor) {
    uint32_t first = 0;
    if (wave_lane() == 0) {
        first = atomicAdd(cursor, kHandOutChunks);
    }
    return (uint32_t) __builtin_amdgcn_readfirstlane((int) first);
}

// FIRST: trip 0 - the rays are the primary rays of pids 0 .. paths-1, built on the spot.
template <int ACCEL, bool COUNT, bool FIRST>
__global__ void __launch_bounds__(ACCEL == RT_HIP_ACCEL_REBUILT ? kExtendThreads : 64)
    wf_extend_kernel(wavefront_params const p, uint32_t trip) {
    extern __shared__ float4 extend_lds[];
    const uint32_t* const ctl = p.control + trip * kCtlWords;
    uint32_t const n_closest = FIRST ? p.pass_count * p.block_count * 64u : ctl[kCtlClosest];
    uint32_t const n_shadow = FIRST ? 0u : ctl[kCtlShadow];
    uint32_t const closest_chunks = (n_closest + 63u) / 64u;
    uint32_t const total_chunks = closest_chunks + (n_shadow + 63u) / 64u;
    bool const binned = !FIRST && p.sort_rays != 0u;
    const uint32_t* const q_closest = binned ? p.binned_closest : p.q_closest;
    const uint32_t* const q_shadow = binned ? p.binned_shadow : p.q_shadow;
    fast_stack stk{};
    if (ACCEL == RT_HIP_ACCEL_REBUILT) {
        if (blockIdx.x * (kExtendThreads / 64u) >= total_chunks) {
            return;  // none of this workgroup's waves has a chunk: skip filling the caches
        }
        uint32_t const node_words = p.sc.cached_nodes * kWideNodeQuads;
        uint32_t const instance_words = p.sc.cached_instances * 4u;
        const float4* const nodes = reinterpret_cast<const float4*>(p.sc.wide_nodes);
        const float4* const records = reinterpret_cast<const float4*>(p.sc.wide_instances);
        for (uint32_t i = threadIdx.x; i < node_words; i += kExtendThreads) {
            extend_lds[i] = nodes[i];
        }
        for (uint32_t i = threadIdx.x; i < instance_words; i += kExtendThreads) {
            extend_lds[node_words + i] = records[i];
        }
        __syncthreads();
        stk.node_cache = extend_lds;
        stk.instance_cache = extend_lds + node_words;
        stk.cached_nodes = p.sc.cached_nodes;
        stk.cached_instances = p.sc.cached_instances;
        stk.lds = reinterpret_cast<uint32_t*>(extend_lds + node_words + instance_words) +
                  (threadIdx.x >> 6) * p.sc.stack_entries * 64u + wave_lane();
    }
    trace_counters cnt;
    if (FIRST && blockIdx.x == 0 && threadIdx.x == 0) {
        // for rt_hip_trip_counts; includes the unused lanes of partial 8x8 blocks
        uint32_t* const first_ctl = p.control;
        first_ctl[kCtlLive] = n_closest;
        first_ctl[kCtlClosest] = n_closest;
    }
#if RT_PERSISTENT_TRACE
    if (ACCEL == RT_HIP_ACCEL_REBUILT && p.refill != 0u) {
        // One walk per lane, refilled: the rays of a wave need very different numbers of steps
        // (the mean over the maximum of 64 is what bounds the lanes in use), so a lane whose
        // ray is done takes the wave's next ray - but only once at least kRefillIdle lanes
        // wait, so that the per-ray set-up (three exact divisions) runs on many lanes at once.
        // The rays come in groups of kHandOutChunks chunks from the queue's counter (next_chunks).
        unsigned long long const below = (1ull << wave_lane()) - 1ull;
        auto run = [&](auto walk, uint32_t n, uint32_t* cursor, auto fetch, auto deliver) {
            uint32_t const chunks = (n + 63u) / 64u;
            uint32_t group_first = 0, group_next = kHandOutChunks * 64u;  // slots used of the group
            bool exhausted = chunks == 0u;
            bool busy = false;
            uint32_t pid = 0;
            for (;;) {
                unsigned long long const idle = __ballot(!busy);
                uint32_t const idle_count = (uint32_t) __builtin_popcountll(idle);
                if (!exhausted && (idle_count >= kRefillIdle || idle_count == 64u)) {
                    if (group_next >= kHandOutChunks * 64u) {
                        group_first = next_chunks(cursor);
                        group_next = 0;
                        if (group_first >= chunks) {
                            exhausted = true;
                            continue;
                        }
                    }
                    uint32_t const take = min(idle_count, kHandOutChunks * 64u - group_next);
                    uint32_t const rank = (uint32_t) __builtin_popcountll(idle & below);
                    if (!busy && rank < take) {
                        uint32_t const i = group_first * 64u + group_next + rank;
                        f3 o, d;
                        float t_limit;
                        if (i < n && fetch(i, pid, o, d, t_limit)) {
                            walk.start(p.sc, o, d, t_limit);
                            busy = true;
                        }
                    }
                    group_next += take;
                    continue;  // (lanes whose slot held no ray try again)
                }
                if (idle_count == 64u) {
                    break;  // nothing walking and nothing left to fetch
                }
                if (busy && walk.round(p.sc, stk, cnt)) {
                    deliver(pid, walk);
                    busy = false;
                }
            }
        };
        run(fast_walk<false, COUNT, true>{}, n_closest,
            p.control + trip * kCtlWords + kCtlClosestCursor,
            [&](uint32_t i, uint32_t& pid, f3& o, f3& d, float& t_limit) {
                t_limit = kInfinity;
                if (FIRST) {
                    uint32_t seed;
                    pid = i;
                    bool const inside = primary_of_pid(p, pid, seed, o, d);
                    if (inside) {
                        stream_store(&p.ray_d[pid], make_float4(d.x, d.y, d.z, __uint_as_float(seed)));
                    }
                    return inside;
                }
                pid = stream_load(&q_closest[i]);
                float4 const ro = stream_load(&p.ray_o[pid]);
                float4 const rd = stream_load(&p.ray_d[pid]);
                o = mk3(ro.x, ro.y, ro.z);
                d = mk3(rd.x, rd.y, rd.z);
                return true;
            },
            [&](uint32_t pid, fast_walk<false, COUNT, true> const& wk) {
                stream_store(&p.hit0[pid],
                    make_float4(__uint_as_float(wk.found ? wk.best.instance : RT_HIP_MISS),
                        __uint_as_float(wk.best.triangle), wk.best.b0, wk.best.b1));
                stream_store(&p.hit1[pid], make_float2(wk.best.b2, wk.best.t));
            });
        float4 lane_rad = make_float4(0.0f, 0.0f, 0.0f, 0.0f), lane_nee = lane_rad;
        run(fast_walk<true, COUNT, true>{}, n_shadow,
            p.control + trip * kCtlWords + kCtlShadowCursor,
            [&](uint32_t i, uint32_t& pid, f3& o, f3& d, float& t_limit) {
                pid = stream_load(&q_shadow[i]);
                float4 const ro = stream_load(&p.sh_o[pid]);
                float4 const rd = stream_load(&p.sh_d[pid]);
                lane_rad = stream_load(&p.rad[pid]);
                lane_nee = stream_load(&p.nee[pid]);
                o = mk3(ro.x, ro.y, ro.z);
                d = mk3(rd.x, rd.y, rd.z);
                t_limit = ro.w;
                return true;
            },
            [&](uint32_t pid, fast_walk<true, COUNT, true> const& wk) {
                if (!wk.found) {
                    add_light_sample(p, pid, lane_rad, lane_nee);
                }
            });
    } else
#endif
    // (fixed chunks, fixed shares: wave w takes chunks w, w + W, ...  This is the loop of the
    // coherent bounces - camera rays - whose chunks cost about the same; handing them out by
    // counter as well measured 1.6 % slower on hyperion and 2.7 % on the teapot scene)
    for (uint32_t chunk = wave_index(); chunk < total_chunks; chunk += wave_total()) {
        if (chunk < closest_chunks) {
            uint32_t const i = chunk * 64u + wave_lane();
            bool active = i < n_closest;
            uint32_t const pid = FIRST ? i : (active ? stream_load(&q_closest[i]) : 0u);
            float4 o = make_float4(0.0f, 0.0f, 0.0f, 0.0f), d = make_float4(1.0f, 0.0f, 0.0f, 0.0f);
            if (FIRST) {
                uint32_t seed;
                f3 origin, direction;
                active = primary_of_pid(p, pid, seed, origin, direction);
                if (active) {
                    o = make_float4(origin.x, origin.y, origin.z, 0.0f);
                    d = make_float4(direction.x, direction.y, direction.z, 0.0f);
                    stream_store(&p.ray_d[pid], make_float4(direction.x, direction.y, direction.z,
                                                    __uint_as_float(seed)));
                }
            } else if (active) {
                o = stream_load(&p.ray_o[pid]);
                d = stream_load(&p.ray_d[pid]);
            }
            hit_result best;
            best.instance = RT_HIP_MISS;
            best.triangle = RT_HIP_MISS;
            best.b0 = best.b1 = best.b2 = 0.0f;
            best.t = kInfinity;
            bool hit;
            if (ACCEL == RT_HIP_ACCEL_REBUILT) {
                hit = trace_fast<false, COUNT, true>(p.sc, stk, active, mk3(o.x, o.y, o.z),
                    mk3(d.x, d.y, d.z), kInfinity, best, cnt);
            } else {
                hit = active && trace_ref<false, COUNT>(p.sc, mk3(o.x, o.y, o.z),
                                    mk3(d.x, d.y, d.z), kInfinity, best, cnt);
            }
            if (active) {
                stream_store(&p.hit0[pid],
                    make_float4(__uint_as_float(hit ? best.instance : RT_HIP_MISS),
                        __uint_as_float(best.triangle), best.b0, best.b1));
                stream_store(&p.hit1[pid], make_float2(best.b2, best.t));
            }
        } else {
            uint32_t const i = (chunk - closest_chunks) * 64u + wave_lane();
            bool const active = i < n_shadow;
            uint32_t const pid = active ? stream_load(&q_shadow[i]) : 0u;
            float4 o = make_float4(0.0f, 0.0f, 0.0f, 1.0f), d = make_float4(1.0f, 0.0f, 0.0f, 0.0f);
            float4 rad = d, nee = d;
            if (active) {
                o = stream_load(&p.sh_o[pid]);
                d = stream_load(&p.sh_d[pid]);
                rad = stream_load(&p.rad[pid]);
                nee = stream_load(&p.nee[pid]);
            }
            hit_result unused;
            bool blocked;
            if (ACCEL == RT_HIP_ACCEL_REBUILT) {
                blocked = trace_fast<true, COUNT, true>(p.sc, stk, active, mk3(o.x, o.y, o.z),
                    mk3(d.x, d.y, d.z), o.w, unused, cnt);
            } else {
                blocked = active && trace_ref<true, COUNT>(p.sc, mk3(o.x, o.y, o.z),
                                        mk3(d.x, d.y, d.z), o.w, unused, cnt);
            }
            if (active && !blocked) {
                add_light_sample(p, pid, rad, nee);
            }
        }
    }
    if (COUNT) {
        unsigned long long const nodes = wave_sum(cnt.nodes);
        unsigned long long const tris = wave_sum(cnt.tris);
        uint32_t deepest = cnt.max_stack;
        for (int offset = 32; offset > 0; offset >>= 1) {
            deepest = max(deepest, (uint32_t) __shfl_down((int) deepest, offset, 64));
        }
        unsigned long long const lds_nodes = wave_sum(cnt.lds_nodes);
        unsigned long long const instances = wave_sum(cnt.instances);
        unsigned long long const lds_instances = wave_sum(cnt.lds_instances);
        unsigned long long const node_wave_steps = wave_sum(cnt.node_wave_steps);
        unsigned long long const tri_wave_steps = wave_sum(cnt.tri_wave_steps);
        unsigned long long const instance_wave_steps = wave_sum(cnt.instance_wave_steps);
        if (wave_lane() == 0) {
            add_wave_stat(p, kStatNodes, nodes);
            add_wave_stat(p, kStatTris, tris);
            add_wave_stat(p, kStatLdsNodes, lds_nodes);
            add_wave_stat(p, kStatInstances, instances);
            add_wave_stat(p, kStatLdsInstances, lds_instances);
            add_wave_stat(p, kStatNodeWaveSteps, node_wave_steps);
            add_wave_stat(p, kStatTriWaveSteps, tri_wave_steps);
            add_wave_stat(p, kStatInstanceWaveSteps, instance_wave_steps);
            unsigned long long* const slot =
                p.wave_stats + (size_t) wave_index() * kStatWords + kStatMaxStack;
            *slot = max(*slot, (unsigned long long) deepest);
        }
    }
}

// Shading of trip `trip`, in two launches over its live list (trip 0: over all pids).
//
// What a path costs here differs by two orders of magnitude.  A path whose ray hit a surface
// below max_depth goes on: light proposal, BSDF sampling, two BSDF evaluations - a few thousand
// instructions in some 200 registers (two waves per SIMD).  Every other path ENDS here - it had
// no closest-hit ray, missed, hit a light or is at max_depth - after a few dozen instructions
// that mostly wait for memory; after the first bounce of an open scene that is nine paths in
// ten.  Shading both kinds in one kernel made the many cheap paths run at the occupancy of
// the few dear ones (measured inside that kernel: 50-64 % of a later bounce's time went into
// fetching list entries and path state, i.e. into memory latency at two waves per SIMD).
//
//   wf_settle_kernel  sorts the list: it settles the paths that end (their radiance becomes
//                     final, their raw slot 0) and writes the others, region by region of
//                     kRegionSlots list slots, to the `going` list.  Small, up to eight waves
//                     per SIMD: it runs at the speed of the memory system.
//   wf_shade_kernel   shades the going paths, 64 per wave whatever their slots were, and
//                     writes their raw slots.  Waves take regions from a counter in the
//                     control block (one atomic per region, not per chunk).
//
// Which lane shades which path has no influence on any result; both kernels call shade_trip,
// each with the half of it that its paths can reach.  The compaction that follows counts the
// raw slots itself (wf_tile_sums_kernel).
//
// Both stage the scene's small tables in LDS first (plan_shade_tables) and point their copy of
// the scene description at them.
constexpr uint32_t kShadeThreads = 256;
constexpr uint32_t kSettleThreads = 1024;
constexpr uint32_t kRegionSlots = 1024;
#ifdef RT_SHADE_REGION_TIMING
__device__ unsigned long long g_region_cycles[2][12];
#endif

#ifndef RT_SHADE_WAVES_PER_SIMD
#define RT_SHADE_WAVES_PER_SIMD 2
#endif
// one four-wave workgroup per wave slot of the SIMDs; the workgroups of a CU share its LDS
#ifndef RT_SHADE_GROUPS_PER_CU
#define RT_SHADE_GROUPS_PER_CU RT_SHADE_WAVES_PER_SIMD
#endif
constexpr uint32_t kShadeGroupsPerCu = RT_SHADE_GROUPS_PER_CU;
// settle: 1024-thread workgroups; two per CU = eight waves per SIMD when the registers allow
#ifndef RT_SETTLE_GROUPS_PER_CU
#define RT_SETTLE_GROUPS_PER_CU 2
#endif
constexpr uint32_t kSettleGroupsPerCu = RT_SETTLE_GROUPS_PER_CU;
constexpr uint32_t kShadeLdsBudget = (kShadeGroupsPerCu <= 2u ? 64u : 48u) * 1024u;

// Copies the tables the plan stages into `lds` and returns the scene description that points
// at the copies.  Called by every thread of the workgroup; ends with a barrier.
RT_DEV scene_dev stage_scene_tables(scene_dev const& global, float4* lds, uint32_t threads) {
    scene_dev sc = global;
    shade_plan const plan = plan_shade_tables(global, kShadeLdsBudget);
    const void* const source[kTableCount] = {global.lights, global.materials, global.meshes,
        global.textures, global.instances, global.transforms, global.normal_matrices};
    uint32_t* const words = reinterpret_cast<uint32_t*>(lds);
    for (uint32_t t = 0; t < kTableCount; ++t) {
        if (plan.offset[t] == kNotStaged) {
            continue;
        }
        const uint32_t* const src = static_cast<const uint32_t*>(source[t]);
        for (uint32_t i = threadIdx.x; i < plan.bytes[t] / 4u; i += threads) {
            words[plan.offset[t] / 4u + i] = src[i];
        }
    }
    __syncthreads();
    auto staged = [&](uint32_t t, const void* fallback) -> const void* {
        return plan.offset[t] == kNotStaged
                   ? fallback
                   : static_cast<const void*>(reinterpret_cast<const char*>(lds) + plan.offset[t]);
    };
    sc.lights = static_cast<const rt_light*>(staged(kTableLights, global.lights));
    sc.materials = static_cast<const rt_material*>(staged(kTableMaterials, global.materials));
    sc.meshes = static_cast<const rt_mesh*>(staged(kTableMeshes, global.meshes));
    sc.textures = static_cast<const texture_dev*>(staged(kTableTextures, global.textures));
    sc.instances = static_cast<const rt_instance*>(staged(kTableInstances, global.instances));
    sc.transforms = static_cast<const rt_mat4*>(staged(kTableTransforms, global.transforms));
    sc.normal_matrices =
        static_cast<const normal_matrix*>(staged(kTableNormalMatrices, global.normal_matrices));
    return sc;
}

// FIRST: trip 0 - the list is pids 0 .. paths-1, every path has a closest-hit ray and its
// initial state (nothing has been stored yet).
template <bool FIRST, uint32_t F>
__global__ void __launch_bounds__(kSettleThreads, RT_SETTLE_GROUPS_PER_CU * 4)
    wf_settle_kernel(wavefront_params const p, uint32_t trip) {
    extern __shared__ float4 shade_lds[];
    uint32_t const n_live =
        FIRST ? p.pass_count * p.block_count * 64u : p.control[trip * kCtlWords + kCtlLive];
    uint32_t const total_regions = (n_live + kRegionSlots - 1u) / kRegionSlots;
    if (blockIdx.x * (kSettleThreads / 64u) >= total_regions) {
        return;  // none of this workgroup's waves has a region
    }
    scene_dev const sc = stage_scene_tables(p.sc, shade_lds, kSettleThreads);
    // a traced path's depth is the trip's number: every trip it went through added one
    bool const last_trip = trip == p.max_depth;
    unsigned long long const below = (1ull << wave_lane()) - 1ull;
    uint32_t total_hits = 0, samples = 0, total_going = 0;
    for (uint32_t region = wave_index(); region < total_regions; region += wave_total()) {
        uint32_t const first_slot = region * kRegionSlots;
        uint32_t const region_chunks = (min(n_live - first_slot, kRegionSlots) + 63u) / 64u;
        uint32_t going = 0;
        for (uint32_t k = 0; k < region_chunks; ++k) {
            uint32_t const i = first_slot + k * 64u + wave_lane();
            bool const in_range = i < n_live;
            uint32_t const entry =
                FIRST ? (i | kLiveHasClosest) : (in_range ? stream_load(&p.live[i]) : 0u);
            uint32_t const pid = entry & kLivePidMask;
            bool traced = in_range && (entry & kLiveHasClosest) != 0u;
            bool real = in_range;
            if (FIRST && p.padded != 0u && in_range) {
                // the padding lanes of partial 8x8 blocks have no path and no hit record
                uint32_t x, y, pass;
                real = traced = pid_pixel(p, pid, x, y, pass);
            }
            float4 h0 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            bool hit = false, goes_on = false;
            if (traced) {
                h0 = stream_load(&p.hit0[pid]);
                uint32_t const instance = __float_as_uint(h0.x);
                hit = instance != RT_HIP_MISS;
                goes_on = hit && !last_trip && sc.instances[instance].light < 0;
            }
            unsigned long long const going_mask = __ballot(goes_on);
            if (goes_on) {
                p.going[first_slot + going + (uint32_t) __builtin_popcountll(going_mask & below)] =
                    make_uint2(i, entry);
            }
            going += (uint32_t) __builtin_popcountll(going_mask);
            if (in_range && !goes_on) {
                stream_store(&p.raw[i], 0u);  // the path ends here (or never was)
            }
            if (FIRST && real) {
                samples += 1u;
            }
            if (!real || goes_on) {
                continue;
            }
            // The path ends at this trip; its radiance becomes final.  Only what the case at
            // hand reads is fetched: the waves here are many and wait for memory, not for the
            // VALU, so a dependent load costs less than the bytes of an unconditional one.
            // (the outcome of its occlusion ray is in rad[pid] already: add_light_sample)
            bool const sky = traced && !hit && feat<F>(kFeatSky) && sc.sky_light >= 0;
            if (!FIRST && !sky && !hit) {
                continue;  // nothing is added: rad[pid] is final as it stands
            }
            ray_pair rp;
            rp.has_shadow = false;
            rp.has_closest = traced;
            rp.origin = rp.direction = splat3(0.0f);
            path_state ps;
            ps.radiance = splat3(0.0f);
            ps.throughput = splat3(1.0f);
            ps.nee = splat3(0.0f);
            ps.last_bsdf_pdf = 0.0f;
            ps.depth = trip;
            ps.seed = 0u;
            pair_result pr;
            pr.occluded = false;
            pr.hit = hit;
            if (FIRST) {
                if (sky || hit) {
                    primary_of_pid(p, pid, ps.seed, rp.origin, rp.direction);
                }
            } else {
                float4 const rad = stream_load(&p.rad[pid]);
                ps.radiance = mk3(rad.x, rad.y, rad.z);
                if (sky) {
                    ps.last_bsdf_pdf = stream_load(&p.nee[pid]).w;
                }
                if (sky || hit) {
                    float4 const thr = stream_load(&p.thr[pid]);
                    float4 const d = stream_load(&p.ray_d[pid]);
                    ps.throughput = mk3(thr.x, thr.y, thr.z);
                    rp.direction = mk3(d.x, d.y, d.z);
                }
                if (hit) {
                    float4 const o = stream_load(&p.ray_o[pid]);
                    rp.origin = mk3(o.x, o.y, o.z);
                }
            }
            if (hit) {
                float2 const h1 = stream_load(&p.hit1[pid]);
                pr.best.instance = __float_as_uint(h0.x);
                pr.best.triangle = __float_as_uint(h0.y);
                pr.best.b0 = h0.z;
                pr.best.b1 = h0.w;
                pr.best.b2 = h1.x;
                pr.best.t = h1.y;
            }
            uint32_t closest_hits = 0, shadow_skipped = 0;
            shade_trip<F, kTripEnds>(sc, p.max_depth, rp, pr, ps, closest_hits, shadow_skipped);
            total_hits += closest_hits;
            stream_store(&p.rad[pid], make_float4(ps.radiance.x, ps.radiance.y, ps.radiance.z,
                                          __uint_as_float(ps.depth)));
        }
        if (wave_lane() == 0) {
            p.going_count[region] = going;
        }
        total_going += going;
    }
    if (wave_lane() == 0 && total_going != 0u) {
        atomicAdd(p.control + trip * kCtlWords + kCtlGoing, total_going);
    }
    unsigned long long const hits = wave_sum(total_hits);
    unsigned long long const started = FIRST ? wave_sum(samples) : 0ull;
    if (wave_lane() == 0) {
        add_wave_stat(p, kStatClosestHits, hits);
        if (FIRST) {
            add_wave_stat(p, kStatSamples, started);  // = trip 0's closest-hit rays, too
        }
    }
}

// What wf_shade_kernel holds of a going path before it shades it.
struct going_state {
    float4 rad, thr, h0;
    f3 o, d;
    float2 h1;
};

template <bool FIRST, uint32_t F>
__global__ void __launch_bounds__(kShadeThreads, RT_SHADE_WAVES_PER_SIMD)
    wf_shade_kernel(wavefront_params const p, uint32_t trip) {
    extern __shared__ float4 shade_lds[];
    uint32_t const n_live =
        FIRST ? p.pass_count * p.block_count * 64u : p.control[trip * kCtlWords + kCtlLive];
    uint32_t const going_total = p.control[trip * kCtlWords + kCtlGoing];
    if (going_total == 0u) {
        return;
    }
    // A wave takes a BATCH of `span` consecutive regions at a time and shades their going paths
    // as one dense sequence, 64 per step, whatever region each came from: where one path in
    // fifteen goes on, a region alone would fill a wave once, partly.  `span` is chosen so that
    // a batch holds about kBatchPaths going paths on average - few enough that the batches
    // outnumber the waves many times over (waves take them from a counter, so the load
    // balances), enough that what a wave pays per batch (the counter, the counts, an empty
    // pipeline) stays small.  (256, 512, 1024, 2048, 4096 measured: 512 is 0.7-1.3 % ahead of
    // 1024 on three scenes of four and level on the fourth, 256 level with 512, 2048 and up
    // 1-6 % behind.)
    constexpr uint32_t kBatchPaths = 512;
    uint32_t const total_regions = (n_live + kRegionSlots - 1u) / kRegionSlots;
    uint32_t const span = (uint32_t) min(
        64ull, max(1ull, ((unsigned long long) kBatchPaths * total_regions + going_total - 1u) /
                             going_total));
    uint32_t const total_batches = (total_regions + span - 1u) / span;
    if (blockIdx.x * (kShadeThreads / 64u) >= total_batches) {
        return;  // the waves before this workgroup's can take every batch there is
    }
    scene_dev const sc = stage_scene_tables(p.sc, shade_lds, kShadeThreads);
    uint32_t* const cursor = p.control + trip * kCtlWords + kCtlRegionCursor;
    uint32_t total_hits = 0, total_skipped = 0;
    region_clock rc;
    rc.start();
    for (;;) {
        uint32_t batch = 0;
        if (wave_lane() == 0) {
            batch = atomicAdd(cursor, 1u);
        }
        batch = (uint32_t) __builtin_amdgcn_readfirstlane((int) batch);
        if (batch >= total_batches) {
            break;
        }
        // lane j: going paths of the batch's regions before its j-th (exclusive prefix sum)
        uint32_t const first_region = batch * span;
        uint32_t const my_region = first_region + wave_lane();
        uint32_t const mine =
            wave_lane() < span && my_region < total_regions ? p.going_count[my_region] : 0u;
        uint32_t before = mine;
        for (uint32_t step = 1; step < 64u; step <<= 1) {
            uint32_t const other = (uint32_t) __shfl_up((int) before, step, 64);
            if (wave_lane() >= step) {
                before += other;
            }
        }
        uint32_t const count = (uint32_t) __builtin_amdgcn_readlane((int) before, 63);
        before -= mine;
        // the (slot, entry) pair at position `at` of the batch's dense sequence
        auto item_at = [&](uint32_t at) {
            uint32_t j = 0;  // the last region whose prefix is <= at
            for (uint32_t step = 32u; step > 0u; step >>= 1) {
                uint32_t const probe = j + step;
                uint32_t const prefix = (uint32_t) __shfl((int) before, (int) (probe & 63u), 64);
                if (probe < span && prefix <= at) {
                    j = probe;
                }
            }
            uint32_t const prefix = (uint32_t) __shfl((int) before, (int) j, 64);
            if (at >= count) {
                return make_uint2(0u, 0u);
            }
            return p.going[(size_t) (first_region + j) * kRegionSlots + (at - prefix)];
        };
        auto fetch = [&](uint2 item, bool active, going_state& g) {
            uint32_t const pid = item.y & kLivePidMask;
            if (active) {
                if (!FIRST) {
                    g.rad = stream_load(&p.rad[pid]);
                    g.thr = stream_load(&p.thr[pid]);
                    g.o = stream_load3(&p.ray_o[pid]);
                    g.d = stream_load3(&p.ray_d[pid]);
                }
                if (FIRST) {
                    g.thr = stream_load(&p.ray_d[pid]);  // primary direction, w = seed bits
                }
                g.h0 = stream_load(&p.hit0[pid]);
                g.h1 = stream_load(&p.hit1[pid]);
            }
        };
        rc.mark(8);
        uint2 item = item_at(wave_lane());
        going_state cur{};
        fetch(item, wave_lane() < count, cur);
        for (uint32_t at = wave_lane(); at - wave_lane() < count; at += 64u) {
            bool const active = at < count;
            // the next step's list entry is on its way while this step's path is set up; its
            // state follows once this step has issued its last load (shade_trip's hook)
            uint2 const next_item = item_at(at + 64u);
            going_state next{};
            struct prefetch {
                decltype(fetch)& run;
                uint2 item;
                bool active;
                going_state& into;
                RT_DEV void after_loads() { run(item, active, into); }
            } hook{fetch, next_item, at + 64u < count, next};
            uint32_t const slot = item.x;
            uint32_t const entry = item.y;
            uint32_t const pid = entry & kLivePidMask;
            ray_pair rp;
            rp.has_shadow = false;  // (the one of the bounce before is settled: add_light_sample)
            rp.has_closest = active;
            uint32_t closest_hits = 0, shadow_skipped = 0;
            path_state ps;
            if (active) {
                pair_result pr;
                pr.hit = true;
                if (FIRST) {
                    // the trace launch left the primary ray's direction and the seed after the
                    // two jitter draws where the closest-hit ray's direction lives
                    rp.origin = load3(p.cam.position);
                    rp.direction = mk3(cur.thr.x, cur.thr.y, cur.thr.z);
                    ps.seed = __float_as_uint(cur.thr.w);
                    ps.radiance = splat3(0.0f);
                    ps.throughput = splat3(1.0f);
                    ps.nee = splat3(0.0f);
                    ps.last_bsdf_pdf = 0.0f;
                    ps.depth = 0;
                    pr.occluded = false;
                } else {
                    pr.occluded = false;
                    ps.radiance = mk3(cur.rad.x, cur.rad.y, cur.rad.z);
                    ps.depth = __float_as_uint(cur.rad.w);
                    ps.throughput = mk3(cur.thr.x, cur.thr.y, cur.thr.z);
                    ps.seed = __float_as_uint(cur.thr.w);
                    // (both are written before they are read: a path that goes on draws a new
                    // light sample or has none, and its BSDF sample sets the pdf)
                    ps.nee = splat3(0.0f);
                    ps.last_bsdf_pdf = 0.0f;
                    rp.origin = cur.o;
                    rp.direction = cur.d;
                }
                pr.best.instance = __float_as_uint(cur.h0.x);
                pr.best.triangle = __float_as_uint(cur.h0.y);
                pr.best.b0 = cur.h0.z;
                pr.best.b1 = cur.h0.w;
                pr.best.b2 = cur.h1.x;
                pr.best.t = cur.h1.y;
                rc.mark(0);
                shade_trip<F, kTripGoesOn>(sc, p.max_depth, rp, pr, ps, closest_hits,
                    shadow_skipped, rc, hook);
                rc.mark(11);
                stream_store(&p.rad[pid], make_float4(ps.radiance.x, ps.radiance.y,
                                              ps.radiance.z, __uint_as_float(ps.depth)));
                if (rp.has_closest || rp.has_shadow) {
                    stream_store(&p.thr[pid], make_float4(ps.throughput.x, ps.throughput.y,
                                                  ps.throughput.z, __uint_as_float(ps.seed)));
                    stream_store(&p.nee[pid],
                        make_float4(ps.nee.x, ps.nee.y, ps.nee.z, ps.last_bsdf_pdf));
                }
                if (rp.has_closest) {
                    stream_store(&p.ray_o[pid],
                        make_float4(rp.origin.x, rp.origin.y, rp.origin.z, 0.0f));
                    stream_store(&p.ray_d[pid],
                        make_float4(rp.direction.x, rp.direction.y, rp.direction.z, 0.0f));
                }
                if (rp.has_shadow) {
                    stream_store(&p.sh_o[pid], make_float4(rp.shadow_origin.x, rp.shadow_origin.y,
                                                   rp.shadow_origin.z, rp.shadow_t_max));
                    stream_store(&p.sh_d[pid], make_float4(rp.shadow_direction.x,
                                                   rp.shadow_direction.y, rp.shadow_direction.z, 0.0f));
                }
                bool const more = rp.has_closest || rp.has_shadow;
                stream_store(&p.raw[slot],
                    more ? (pid | (rp.has_closest ? kLiveHasClosest : 0u) |
                               (rp.has_shadow ? kLiveHasShadow : 0u))
                         : 0u);
            }
            total_hits += closest_hits;
            total_skipped += shadow_skipped;
            item = next_item;
            cur = next;
            rc.mark(9);
        }
    }
#ifdef RT_SHADE_REGION_TIMING
    if (wave_lane() == 0) {
        for (int k = 0; k < 12; ++k) {
            atomicAdd(&g_region_cycles[FIRST ? 0 : 1][k], rc.cycles[k]);
        }
    }
#endif
    unsigned long long const hits = wave_sum(total_hits);
    unsigned long long const skipped = wave_sum(total_skipped);
    if (wave_lane() == 0) {
        add_wave_stat(p, kStatClosestHits, hits);
        add_wave_stat(p, kStatShadowSkipped, skipped);
    }
}

// The two launches of one shading stage for feature set F (host side).
template <uint32_t F>
void launch_shade_stage(wavefront_params const& w, uint32_t trip, dim3 settle_grid,
    dim3 shade_grid, uint32_t lds_bytes, hipStream_t stream) {
    if (trip == 0u) {
        hipLaunchKernelGGL((wf_settle_kernel<true, F>), settle_grid, dim3(kSettleThreads),
            lds_bytes, stream, w, trip);
        hipLaunchKernelGGL((wf_shade_kernel<true, F>), shade_grid, dim3(kShadeThreads), lds_bytes,
            stream, w, trip);
    } else {
        hipLaunchKernelGGL((wf_settle_kernel<false, F>), settle_grid, dim3(kSettleThreads),
            lds_bytes, stream, w, trip);
        hipLaunchKernelGGL((wf_shade_kernel<false, F>), shade_grid, dim3(kShadeThreads), lds_bytes,
            stream, w, trip);
    }
}

// The feature sets the shading kernels are compiled for (rt_shading.h: kFeat*), each in a
// translation unit of its own (shade_variant.hip) so that they build side by side; a scene gets
// the smallest one that covers what its materials and lights can do (shade_variant_for).
//    0  diffuse / dielectric surfaces, area lights (the Cornell box)
//    7  + glass, clearcoat, metal (hyperion_rect_light)
//   13  glass, metal, textures (the staircase)
//   23  glass, clearcoat, metal under a constant sky (the generated 10 000-instance scene)
//   24  textures and an environment map (the teapot file)
//   31  everything
#define RT_SHADE_VARIANTS(X) X(0u) X(7u) X(13u) X(23u) X(24u) X(31u)

inline uint32_t shade_variant_for(uint32_t features) {
    uint32_t best = kShadeAll;
    uint32_t best_bits = 32u;
#define RT_CONSIDER(F)                                                                  \
    if ((features & ~(F)) == 0u && (uint32_t) __builtin_popcount(F) < best_bits) {      \
        best = (F);                                                                     \
        best_bits = (uint32_t) __builtin_popcount(F);                                   \
    }
    RT_SHADE_VARIANTS(RT_CONSIDER)
#undef RT_CONSIDER
    return best;
}

// Runs every path handed over at trip `trip` (kCtlFinish, see wf_scan_tiles_kernel) to its
// end inside one kernel, fused-kernel style (trace, shade, trace ... with the path in
// registers).  Once few paths are left, giving each of their remaining bounces a trace
// launch, a shade launch and a compaction costs more in launch tails than the work itself.
// Launched after every compaction; returns at once unless that compaction handed over.
template <int ACCEL, bool COUNT>
__global__ void __launch_bounds__(64, 2) wf_finish_kernel(wavefront_params const p, uint32_t trip) {
    __shared__ uint32_t lds_stack[ACCEL == RT_HIP_ACCEL_REBUILT ? kFastLdsEntries * 64u : 64u];
    fast_stack stk{};
    stk.lds = lds_stack + threadIdx.x;
    uint32_t const n_live = p.control[trip * kCtlWords + kCtlFinish];
    if (n_live == 0u) {
        return;
    }
    uint32_t const total_chunks = (n_live + 63u) / 64u;
    trace_counters cnt;
    uint32_t closest_rays = 0, shadow_rays = 0, closest_hits = 0, shadow_skipped = 0;
    for (uint32_t chunk = wave_index(); chunk < total_chunks; chunk += wave_total()) {
        uint32_t const i = chunk * 64u + wave_lane();
        bool const active = i < n_live;
        uint32_t const entry = active ? p.live[i] : 0u;
        uint32_t const pid = entry & kLivePidMask;
        ray_pair rp;
        rp.has_shadow = (entry & kLiveHasShadow) != 0u;
        rp.has_closest = (entry & kLiveHasClosest) != 0u;
        path_state ps;
        ps.radiance = ps.throughput = ps.nee = splat3(0.0f);
        ps.last_bsdf_pdf = 0.0f;
        ps.depth = 0;
        ps.seed = 0;
        rp.origin = rp.shadow_origin = splat3(0.0f);
        rp.direction = rp.shadow_direction = mk3(1.0f, 0.0f, 0.0f);
        rp.shadow_t_max = 1.0f;
        if (active) {
            float4 const rad = p.rad[pid];
            float4 const thr = p.thr[pid];
            float4 const nee = p.nee[pid];
            ps.radiance = mk3(rad.x, rad.y, rad.z);
            ps.depth = __float_as_uint(rad.w);
            ps.throughput = mk3(thr.x, thr.y, thr.z);
            ps.seed = __float_as_uint(thr.w);
            ps.nee = mk3(nee.x, nee.y, nee.z);
            ps.last_bsdf_pdf = nee.w;
            if (rp.has_closest) {
                float4 const o = p.ray_o[pid];
                float4 const d = p.ray_d[pid];
                rp.origin = mk3(o.x, o.y, o.z);
                rp.direction = mk3(d.x, d.y, d.z);
            }
            if (rp.has_shadow) {
                float4 const o = p.sh_o[pid];
                float4 const d = p.sh_d[pid];
                rp.shadow_origin = mk3(o.x, o.y, o.z);
                rp.shadow_direction = mk3(d.x, d.y, d.z);
                rp.shadow_t_max = o.w;
            }
        }
        while (rp.has_closest || rp.has_shadow) {
            pair_result pr;
            closest_rays += rp.has_closest ? 1u : 0u;
            shadow_rays += rp.has_shadow ? 1u : 0u;
            trace_pair<ACCEL, COUNT>(p.sc, stk, rp, pr, cnt);
            shade_trip(p.sc, p.max_depth, rp, pr, ps, closest_hits, shadow_skipped);
        }
        if (active) {
            p.rad[pid] = make_float4(ps.radiance.x, ps.radiance.y, ps.radiance.z,
                __uint_as_float(ps.depth));
        }
    }
    unsigned long long const sums[4] = {wave_sum(closest_rays), wave_sum(shadow_rays),
        wave_sum(closest_hits), wave_sum(shadow_skipped)};
    unsigned long long const nodes = COUNT ? wave_sum(cnt.nodes) : 0ull;
    unsigned long long const tris = COUNT ? wave_sum(cnt.tris) : 0ull;
    unsigned long long const instances = COUNT ? wave_sum(cnt.instances) : 0ull;
    uint32_t deepest = cnt.max_stack;
    for (int offset = 32; offset > 0; offset >>= 1) {
        deepest = max(deepest, (uint32_t) __shfl_down((int) deepest, offset, 64));
    }
    if (wave_lane() == 0) {
        add_wave_stat(p, kStatClosestRays, sums[0]);
        add_wave_stat(p, kStatShadowRays, sums[1]);
        add_wave_stat(p, kStatClosestHits, sums[2]);
        add_wave_stat(p, kStatShadowSkipped, sums[3]);
        if (COUNT) {
            add_wave_stat(p, kStatNodes, nodes);
            add_wave_stat(p, kStatTris, tris);
            add_wave_stat(p, kStatFinishNodes, nodes);
            add_wave_stat(p, kStatFinishTris, tris);
            add_wave_stat(p, kStatInstances, instances);
            unsigned long long* const slot =
                p.wave_stats + (size_t) wave_index() * kStatWords + kStatMaxStack;
            *slot = max(*slot, (unsigned long long) deepest);
        }
    }
}

#ifndef RT_SHADE_VARIANT_TU
// Adds the batch's finished paths to the frame, pass by pass
// (megakernel_raytracer.comp:108-113 once per pass, in pass order).
__global__ void __launch_bounds__(256) wf_resolve_kernel(wavefront_params const p) {
    uint32_t const in_pass = blockIdx.x * 256u + threadIdx.x;
    uint32_t x, y, pass;
    if (in_pass >= p.block_count * 64u || !pid_pixel(p, in_pass, x, y, pass)) {
        return;
    }
    float* const px = p.accum + 4u * ((size_t) y * p.frame_width + x);
    f3 acc = splat3(0.0f);
    if (!p.preview) {
        float4 const old = *reinterpret_cast<const float4*>(px);
        acc = mk3(old.x, old.y, old.z);
    }
    uint32_t const per_pass = p.block_count * 64u;
    for (uint32_t k = 0; k < p.pass_count; ++k) {
        float4 const r = p.rad[(size_t) k * per_pass + in_pass];
        f3 const radiance = mk3(r.x, r.y, r.z);
        acc = p.preview ? radiance : acc + radiance;
    }
    *reinterpret_cast<float4*>(px) = make_float4(acc.x, acc.y, acc.z, 1.0f);
}

#endif  // RT_SHADE_VARIANT_TU

}  // namespace rtd

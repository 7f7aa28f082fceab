// Traversal of the rebuilt acceleration structure (RT_HIP_ACCEL_REBUILT).
//
// Layout choices for gfx950 (no ray-tracing hardware, 64-wide waves, one ray
// per lane):
//   * 128-byte nodes that carry the boxes of up to FOUR children: the walk
//     runs at the latency of its dependent fetches, seven 16-byte loads issued
//     together cost one latency, and a 4-wide tree has two thirds of the steps
//     of a binary one (the reference's 40-byte binary nodes need a dependent
//     fetch per child);
//   * a positions-only 48-byte triangle stream in leaf order (the reference
//     fetches 96 bytes, 60 of them normals/uvs, for every tested triangle);
//   * TLAS and BLAS walked in ONE loop with one stack: entering an instance
//     pushes a marker and switches the ray to object space, popping the
//     marker switches back, so the wave never re-converges per instance;
//   * per-ray set-up (reciprocals, shear constants) runs for the whole wave
//     before each walk, never inside the divergent loop: with trees this
//     shallow a ray visits only a handful of nodes, so serialised set-up
//     would cost more than the walk itself;
//   * the traversal stack lives in LDS (stack[entry][lane]: conflict-free for
//     a wave); the builder bounds the tree depth so it cannot overflow;
//   * children are visited nearest first by entry distance; the others that
//     were hit are pushed farthest first and tested against the then-current
//     t_max when popped.
//
// The triangle test is the reference's arithmetic unchanged (hit_triangle),
// applied to the same object-space ray, so t and the barycentrics of a given
// (instance, triangle) pair are bit-identical to the reference-order walk.
// Box tests use one fma per plane and are widened by 2 ulp so that rounding
// can only add candidates, never lose the true closest hit.
#pragma once

#include "rt_traverse_ref.h"

namespace rtd {

// Per-lane LDS stack entries.  The builder bounds its trees so that the siblings a walk
// can have pending (TLAS + instance marker + BLAS) never exceed this, which is why
// there is no overflow path (a private overflow array would live in scratch
// memory, and scratch-heavy kernels get fewer resident waves).
constexpr uint32_t kFastLdsEntries = 40;
// 0: experiments - the trace kernel keeps no LDS copies of nodes and instance records
#ifndef RT_TRACE_COPIES
#define RT_TRACE_COPIES 1
#endif
constexpr bool kTraceCopies = RT_TRACE_COPIES != 0;
// The descent loop of the trace launches ends early once fewer lanes than this are still
// descending: the lanes that hold a leaf stop waiting for the last few.
#ifndef RT_DESCENT_MIN
#define RT_DESCENT_MIN 16
#endif
constexpr uint32_t kDescentMin = RT_DESCENT_MIN;
// Likewise the triangle loop of a leaf: once fewer lanes than this still have triangles to
// test, the rest of their leaves waits for the next round (the leaf code in `cur` is advanced
// triangle by triangle).
#ifndef RT_LEAF_MIN
#define RT_LEAF_MIN 8
#endif
constexpr uint32_t kLeafMin = RT_LEAF_MIN;
// The refilling trace loop also holds back instance entries and leaves until this many lanes
// wait for them (unless nothing else can run).
#ifndef RT_INSTANCE_MIN
#define RT_INSTANCE_MIN 16
#endif
constexpr uint32_t kInstanceMin = RT_INSTANCE_MIN;
#ifndef RT_LEAF_START
#define RT_LEAF_START 16
#endif
constexpr uint32_t kLeafStart = RT_LEAF_START;
constexpr int32_t kInstanceMarker = 0x7FFFFFFF;

struct fast_stack {
    uint32_t* lds;  // this lane's column: entry e at lds[e * 64]
    // LDS copies of the first (most visited) nodes and instance records, CACHED walks only
    const float4* node_cache;
    const float4* instance_cache;
    uint32_t cached_nodes;
    uint32_t cached_instances;
};

// A pointer into the workgroup's LDS, typed as such: through a plain pointer the compiler
// reads with flat loads (and merges an "LDS or memory" choice into one flat load of a selected
// address), which are slower than either kind.
typedef float native_f4 __attribute__((ext_vector_type(4)));
struct lds_quads {
    const __attribute__((address_space(3))) native_f4* at;
    RT_DEV float4 operator[](uint32_t i) const {
        native_f4 const v = at[i];
        return make_float4(v.x, v.y, v.z, v.w);
    }
    RT_DEV lds_quads operator+(uint32_t quads) const { return lds_quads{at + quads}; }
};
// The same with byte offsets: quad(at, plus) reads the 16 bytes at `at + plus`, `plus` a
// constant that fits the instruction's immediate offset.
struct lds_bytes {
    uint32_t base;  // LDS address
    RT_DEV float4 quad(uint32_t at, uint32_t plus = 0u) const {
        native_f4 const v =
            *(const __attribute__((address_space(3))) native_f4*) (uintptr_t) (base + at + plus);
        return make_float4(v.x, v.y, v.z, v.w);
    }
};
RT_DEV lds_bytes lds_view_bytes(const float4* p) {
    return lds_bytes{(uint32_t) (uintptr_t) p};
}
RT_DEV lds_quads lds_view(const float4* p) {
    // the low half of a generic pointer into LDS is the LDS address
    return lds_quads{(const __attribute__((address_space(3))) native_f4*) (uintptr_t) (uint32_t) (uintptr_t) p};
}

// 1 on the first of the lanes executing this line, 0 on the others.
RT_DEV uint32_t first_active_lane() {
    unsigned long long const exec = __ballot(1);
    return (uint32_t) __builtin_amdgcn_mbcnt_hi((uint32_t) (exec >> 32),
               __builtin_amdgcn_mbcnt_lo((uint32_t) exec, 0u)) == 0u ? 1u : 0u;
}

// Element `index` of a device array whose byte size stays below 4 GiB: the byte offset is
// formed in 32 bits, so the load takes the uniform base from scalar registers and the offset
// from one vector register instead of a 64-bit shift and add per lane.
template <class T>
RT_DEV const float4* quads_at(const T* base, uint32_t index) {
    return reinterpret_cast<const float4*>(
        reinterpret_cast<const char*>(base) + index * (uint32_t) sizeof(T));
}

struct slab_ray {
    f3 inv;   // 1 / direction
    f3 oinv;  // -origin * inv   (t = plane * inv + oinv)
};

// A direction component of exactly zero gets a huge FINITE reciprocal: with 1/0 = inf the
// two plane distances of that axis become inf - inf = NaN for one plane and +-inf for the other
// whenever the planes lie on different sides of zero, and the box is lost although the ray runs
// inside the slab.  With 1e18 both distances are finite and of opposite sign (inside the slab:
// no constraint) or of the same sign and far beyond any scene (outside: a miss).
#ifndef RT_SLAB_FAST_RCP
#define RT_SLAB_FAST_RCP 1
#endif
RT_DEV float slab_reciprocal(float d) {
#if RT_SLAB_FAST_RCP
    // the hardware reciprocal (1 ulp) instead of a correctly rounded division: the box test is
    // conservative and its margin (slab_box) covers the extra half ulp, which scales the near
    // and the far distance of an axis alike.  Zero and denormal components give infinity there
    // and the huge finite value here.
    float const r = __builtin_amdgcn_rcpf(d);
    return fabsf(r) <= 1.0e18f ? r : copysignf(1.0e18f, d);
#else
    return d == 0.0f ? copysignf(1.0e18f, d) : 1.0f / d;
#endif
}

RT_DEV slab_ray make_slab_ray(f3 o, f3 d) {
    slab_ray r;
    r.inv = mk3(slab_reciprocal(d.x), slab_reciprocal(d.y), slab_reciprocal(d.z));
    r.oinv = mk3(-o.x * r.inv.x, -o.y * r.inv.y, -o.z * r.inv.z);
    return r;
}

// Entry/exit distances of one box from its three NEAR and three FAR planes - the planes the ray
// meets first and last along each axis, which the sign of the direction decides once per ray
// (near_planes): the caller loads them from the node by that sign instead of loading both planes
// of an axis and taking the minimum and maximum of the two distances.  The distance to a plane
// is one rounding of plane * inv + oinv, monotone in the plane, so the near plane's distance IS
// the minimum of the two: the same values, six instructions per box less.  (The reciprocals
// are finite, see slab_reciprocal.)  An unused child slot holds an inverted box (lo = FLT_MAX,
// hi = -FLT_MAX): its near distances are huge and positive, its far ones huge and negative
// whatever the ray, so it is never entered.
RT_DEV void slab_box(slab_ray const& r, float near_x, float near_y, float near_z, float far_x,
    float far_y, float far_z, float t_max, float& t_enter, float& t_exit) {
    float const nx = __builtin_fmaf(near_x, r.inv.x, r.oinv.x);
    float const fx = __builtin_fmaf(far_x, r.inv.x, r.oinv.x);
    float const ny = __builtin_fmaf(near_y, r.inv.y, r.oinv.y);
    float const fy = __builtin_fmaf(far_y, r.inv.y, r.oinv.y);
    float const nz = __builtin_fmaf(near_z, r.inv.z, r.oinv.z);
    float const fz = __builtin_fmaf(far_z, r.inv.z, r.oinv.z);
    // widened by 6 ulp in all (rounding may add candidates, never lose the true hit); the whole
    // margin sits on the exit side, so the entry distance is also the children's sort key as is
    t_enter = fmaxf(fmaxf(nx, ny), fmaxf(nz, 0.0f));
    t_exit = fminf(fminf(fx, fy), fminf(fz, t_max)) * 1.0000007f;
}

// Where in a wide node the near and the far planes of each axis lie for this ray, as byte
// offsets (the node stores lo_x, lo_y, lo_z, hi_x, hi_y, hi_z as its quads 0..5).
struct plane_offsets {
    uint32_t near_x, near_y, near_z, far_x, far_y, far_z;
};
RT_DEV plane_offsets make_plane_offsets(slab_ray const& r) {
    bool const bx = r.inv.x < 0.0f, by = r.inv.y < 0.0f, bz = r.inv.z < 0.0f;
    return plane_offsets{bx ? 48u : 0u, by ? 64u : 16u, bz ? 80u : 32u, bx ? 0u : 48u,
        by ? 16u : 64u, bz ? 32u : 80u};
}

// One bounce's ray pair.  The occlusion ray (if any) is traced first over
// (0, shadow_t_max]; then the closest-hit ray (if any) over (0, 1e14].
struct ray_pair {
    bool has_shadow;
    bool has_closest;
    f3 shadow_origin, shadow_direction;
    float shadow_t_max;
    f3 origin, direction;
};

struct pair_result {
    bool occluded;  // the occlusion ray hit something
    bool hit;       // the closest-hit ray hit something (best is valid)
    hit_result best;
};

// One walk of the rebuilt structure: a ray, its cursor and its stack pointer (the stack itself
// lives in LDS, fast_stack).  ANY = false: closest hit (best is filled); ANY = true: occlusion
// query over (0, t_limit], over at the first hit.  COPIES: nodes / instance records below
// stk.cached_* are read from the workgroup's LDS copy, where a fetch that differs per lane costs
// a few cycles instead of one cache-line lookup per lane in the vector memory pipeline.
//
// The walk is a sequence of steps of three kinds, chosen by the cursor `cur`: an inner node
// (cur >= 0: test the four child boxes, go to the nearest hit child, push the others farthest
// first - or pop), a TLAS leaf (cur < 0 outside an instance: enter the instance, i.e. switch
// to the object-space ray and push a marker) and a BLAS leaf (cur < 0 inside an instance: test
// its triangles and pop; the cursor can also hold the REST of a leaf).  Popping the marker
// switches back to the world-space ray.  cur == kDone: nothing left.  The callers differ only
// in which lanes of a wave step together (trace_fast, fast_walk::round, wf_extend_kernel);
// a ray's own sequence of steps, and so every result, is the same in all of them.
template <bool ANY, bool COUNT, bool COPIES>
struct fast_walk {
    f3 origin, direction;
    slab_ray world_slab, slab;
    plane_offsets planes;  // of `slab`
    tri_ray tr;
    float t_max;
    hit_result best;
    uint32_t current_instance;
    uint32_t sp;
    int32_t cur;
    bool in_blas;
    bool found;

    // no leaf code equals it (the builder refuses scenes with 2^28 triangles or more)
    static constexpr int32_t kDone = (int32_t) 0x80000000u;

    RT_DEV void start(scene_dev const& sc, f3 o, f3 d, float t_limit) {
        origin = o;
        direction = d;
        world_slab = make_slab_ray(o, d);
        slab = world_slab;
        planes = make_plane_offsets(slab);
        tr = tri_ray{};  // triangles are tested inside instances only: set up on entry
        t_max = t_limit;
        found = false;
        best.instance = RT_HIP_MISS;
        best.triangle = RT_HIP_MISS;
        best.b0 = best.b1 = best.b2 = 0.0f;
        best.t = kInfinity;
        current_instance = 0;
        sp = 0;
        cur = sc.wide_tlas_root;
        in_blas = false;
    }

    // (never overflows: the builder bounds the depth and upload checks it against the stack)
    RT_DEV void push(fast_stack const& stk, trace_counters& cnt, int32_t v) {
        stk.lds[sp * 64u] = (uint32_t) v;
        ++sp;
        if (COUNT) {
            cnt.max_stack = max(cnt.max_stack, sp);
        }
    }

    // Pops the next subtree; leaving an instance restores the world-space ray.
    RT_DEV void next_subtree(fast_stack const& stk) {
        for (;;) {
            if (sp == 0) {
                cur = kDone;
                return;
            }
            --sp;
            cur = (int32_t) stk.lds[sp * 64u];
            if (cur != kInstanceMarker) {
                return;
            }
            in_blas = false;
            slab = world_slab;
            planes = make_plane_offsets(slab);
        }
    }

    // POP = false: a step that ends with nothing to go on with does not pop the next subtree
    // itself but returns true, so that the caller can pop once for lanes coming from node
    // steps and from leaves alike (the pop waits for an LDS read).
    template <bool POP = true>
    RT_DEV bool node_step(scene_dev const& sc, fast_stack const& stk, trace_counters& cnt) {
        // near / far planes of the four children along each axis
        float4 nx, ny, nz, fx, fy, fz, kids;
        uint32_t const node_at = (uint32_t) cur * (uint32_t) sizeof(wide_node);  // byte offset
        if (kTraceCopies && COPIES && (uint32_t) cur < stk.cached_nodes) {
            lds_bytes const lp = lds_view_bytes(stk.node_cache);
            nx = lp.quad(node_at + planes.near_x), ny = lp.quad(node_at + planes.near_y);
            nz = lp.quad(node_at + planes.near_z), fx = lp.quad(node_at + planes.far_x);
            fy = lp.quad(node_at + planes.far_y), fz = lp.quad(node_at + planes.far_z);
            kids = lp.quad(node_at, 96u);
            if (COUNT) {
                ++cnt.lds_nodes;
            }
        } else {
            const char* const np = reinterpret_cast<const char*>(sc.wide_nodes);
            auto quad = [&](uint32_t at) { return *reinterpret_cast<const float4*>(np + at); };
            nx = quad(node_at + planes.near_x), ny = quad(node_at + planes.near_y);
            nz = quad(node_at + planes.near_z), fx = quad(node_at + planes.far_x);
            fy = quad(node_at + planes.far_y), fz = quad(node_at + planes.far_z);
            kids = quad(node_at + 96u);
        }
        if (COUNT) {
            ++cnt.nodes;
            cnt.node_wave_steps += first_active_lane();
        }
        int32_t ref[4] = {__float_as_int(kids.x), __float_as_int(kids.y), __float_as_int(kids.z),
            __float_as_int(kids.w)};
        float key[4];
        float const miss = kInfinity * 2.0f;  // beyond any entry distance
        {
            float e, x;
            slab_box(slab, nx.x, ny.x, nz.x, fx.x, fy.x, fz.x, t_max, e, x);
            key[0] = e <= x ? e : miss;
            slab_box(slab, nx.y, ny.y, nz.y, fx.y, fy.y, fz.y, t_max, e, x);
            key[1] = e <= x ? e : miss;
            slab_box(slab, nx.z, ny.z, nz.z, fx.z, fy.z, fz.z, t_max, e, x);
            key[2] = e <= x ? e : miss;
            slab_box(slab, nx.w, ny.w, nz.w, fx.w, fy.w, fz.w, t_max, e, x);
            key[3] = e <= x ? e : miss;
        }
        // nearest first: sort the four (distance, child) pairs, misses go last
        auto order = [&](int i, int j) {
            bool const swap = key[j] < key[i];
            float const ki = swap ? key[j] : key[i];
            float const kj = swap ? key[i] : key[j];
            int32_t const ri = swap ? ref[j] : ref[i];
            int32_t const rj = swap ? ref[i] : ref[j];
            key[i] = ki, key[j] = kj, ref[i] = ri, ref[j] = rj;
        };
        order(0, 1);
        order(2, 3);
        order(0, 2);
        order(1, 3);
        order(1, 2);
        if (key[0] < miss) {
            if (key[3] < miss) {
                push(stk, cnt, ref[3]);
            }
            if (key[2] < miss) {
                push(stk, cnt, ref[2]);
            }
            if (key[1] < miss) {
                push(stk, cnt, ref[1]);
            }
            cur = ref[0];
            return false;
        }
        if (POP) {
            next_subtree(stk);
        }
        return true;
    }

    // TLAS leaf: exactly one instance.  Enters its BLAS.
    RT_DEV void enter_instance(scene_dev const& sc, fast_stack const& stk, trace_counters& cnt) {
        uint32_t const rec_index = ((uint32_t) ~cur) >> 3;
        float4 r0, r1, r2, r3;
        if (kTraceCopies && COPIES && rec_index < stk.cached_instances) {
            lds_quads const lp = lds_view(stk.instance_cache) + 4u * rec_index;
            r0 = lp[0], r1 = lp[1], r2 = lp[2], r3 = lp[3];
            if (COUNT) {
                ++cnt.lds_instances;
            }
        } else {
            const float4* ip = quads_at(sc.wide_instances, rec_index);
            r0 = ip[0], r1 = ip[1], r2 = ip[2], r3 = ip[3];
        }
        if (COUNT) {
            ++cnt.instances;
            cnt.instance_wave_steps += first_active_lane();
        }
        // m[col*3 + row]: r0 = (m00 m01 m02 m10) r1 = (m11 m12 m20 m21) r2 = (m22 m30 m31 m32)
        f3 o, d;
        o.x = ((r0.x * origin.x + r0.w * origin.y) + r1.z * origin.z) + r2.y * 1.0f;
        o.y = ((r0.y * origin.x + r1.x * origin.y) + r1.w * origin.z) + r2.z * 1.0f;
        o.z = ((r0.z * origin.x + r1.y * origin.y) + r2.x * origin.z) + r2.w * 1.0f;
        d.x = ((r0.x * direction.x + r0.w * direction.y) + r1.z * direction.z) + r2.y * 0.0f;
        d.y = ((r0.y * direction.x + r1.x * direction.y) + r1.w * direction.z) + r2.z * 0.0f;
        d.z = ((r0.z * direction.x + r1.y * direction.y) + r2.x * direction.z) + r2.w * 0.0f;
        slab = make_slab_ray(o, d);
        planes = make_plane_offsets(slab);
        tr = make_tri_ray(o, d);
        current_instance = (uint32_t) __float_as_int(r3.y);
        push(stk, cnt, kInstanceMarker);
        in_blas = true;
        cur = __float_as_int(r3.x);
    }

    // BLAS leaf (1..8 triangles, or the rest of one): tests triangles until the leaf is done
    // or - leaf_min > 0 - fewer than leaf_min of the lanes in here still have some left, in
    // which case the cursor keeps the rest.  An occlusion walk that finds a hit is over.
    template <bool POP = true>
    RT_DEV bool test_leaf(scene_dev const& sc, fast_stack const& stk, trace_counters& cnt,
        uint32_t leaf_min) {
        uint32_t const code = (uint32_t) ~cur;
        uint32_t const first = code >> 3;
        uint32_t const count = (code & 7u) + 1u;
        uint32_t k = 0;
        bool stop = false;
        do {
            const float4* tp = quads_at(sc.wide_tris, first + k);
            float4 const va = tp[0], vb = tp[1], vc = tp[2];
            if (COUNT) {
                ++cnt.tris;
                cnt.tri_wave_steps += first_active_lane();
            }
            float b0, b1, b2, th;
            if (hit_triangle(mk3(va.x, va.y, va.z), mk3(vb.x, vb.y, vb.z), mk3(vc.x, vc.y, vc.z),
                    tr, t_max, b0, b1, b2, th, sc.wide_lists_degenerate != 0u)) {
                found = true;
                if (ANY) {
                    stop = true;
                } else {
                    t_max = th;
                    best.instance = current_instance;
                    best.triangle = (uint32_t) __float_as_int(va.w);
                    best.b0 = b0;
                    best.b1 = b1;
                    best.b2 = b2;
                    best.t = th;
                }
            }
            ++k;
        } while (!stop && k < count &&
                 (leaf_min == 0u ||
                     (uint32_t) __builtin_popcountll(__ballot(k < count)) >= leaf_min));
        if (stop) {
            cur = kDone;
            return false;
        }
        if (k < count) {
            cur = (int32_t) ~(((first + k) << 3) | (count - k - 1u));  // the rest of the leaf
            return false;
        }
        if (POP) {
            next_subtree(stk);
        }
        return true;
    }

    // Node steps until this lane holds a leaf or has nothing left - or, descent_min > 0, until
    // fewer than descent_min of the lanes in here still descend.
    RT_DEV void descend(scene_dev const& sc, fast_stack const& stk, trace_counters& cnt,
        uint32_t descent_min) {
        while (cur >= 0) {
            node_step(sc, stk, cnt);
            if (descent_min != 0u &&
                (uint32_t) __builtin_popcountll(__ballot(cur >= 0)) < descent_min) {
                break;
            }
        }
    }

    // One round of a wave that keeps a walk per lane (wf_extend_kernel's refilling loop);
    // returns true when this lane's walk is over.  A kind of step runs when enough of the
    // wave's lanes wait for it (node steps: kDescentMin, instance entries: kInstanceMin,
    // triangles: kLeafStart); when no kind has enough, the one most lanes wait for runs, so
    // every round makes progress.  Must be called by every lane that has a walk, together.
    RT_DEV bool round(scene_dev const& sc, fast_stack const& stk, trace_counters& cnt) {
        auto lanes = [](bool v) { return (uint32_t) __builtin_popcountll(__ballot(v)); };
        uint32_t const n_node = lanes(cur >= 0);
        bool const ran = n_node >= max(kDescentMin, 1u);
        if (ran) {
            descend(sc, stk, cnt, kDescentMin);
        }
        bool const at_instance = cur < 0 && cur != kDone && !in_blas;
        bool const at_leaf = cur < 0 && cur != kDone && in_blas;
        uint32_t const n_instance = lanes(at_instance);
        uint32_t const n_leaf = lanes(at_leaf);
        bool run_instance = n_instance >= max(kInstanceMin, 1u);
        bool run_leaf = n_leaf >= max(kLeafStart, 1u);
        if (!ran && !run_instance && !run_leaf) {
            if (n_node >= n_instance && n_node >= n_leaf) {
                descend(sc, stk, cnt, kDescentMin);
            } else if (n_instance >= n_leaf) {
                run_instance = true;
            } else {
                run_leaf = true;
            }
        }
        if (run_instance && at_instance) {
            enter_instance(sc, stk, cnt);
        }
        if (run_leaf && at_leaf) {
            test_leaf(sc, stk, cnt, kLeafMin);
        }
        return cur == kDone;
    }
};

// One ray per lane, from start to end.  Lanes with active == false skip the walk (their ray
// does not exist); the ray set-up runs for the whole wave at once, outside the divergent loop.
// CACHED = false (the fused kernels, two waves per SIMD): one step of whatever kind per trip
// of the loop.  CACHED = true (the pipeline's trace kernel, fixed chunks of 64 rays): every
// lane descends until it holds a leaf or has nothing left - or only kDescentMin lanes still
// descend - before any lane enters an instance or tests triangles, so lanes wait for the other
// kinds of step once per round, not on every step.  (That loop measured 20 % slower in the
// fused kernels: there the lanes idling through other lanes' descents cost more than the
// shared leaf rounds save.)
template <bool ANY, bool COUNT, bool CACHED = false>
RT_DEV bool trace_fast(scene_dev const& sc, fast_stack const& stk, bool active, f3 origin,
    f3 direction, float t_limit, hit_result& best, trace_counters& cnt) {
    fast_walk<ANY, COUNT, CACHED> w;
    w.start(sc, origin, direction, t_limit);
    if (!active) {
        return false;
    }
    for (;;) {
        if constexpr (!CACHED) {
            if (w.cur >= 0) {
                if (!w.template node_step<false>(sc, stk, cnt)) {
                    continue;
                }
            } else if (!w.in_blas) {
                w.enter_instance(sc, stk, cnt);
                continue;
            } else if (!w.template test_leaf<false>(sc, stk, cnt, 0u)) {
                return true;  // (only an occlusion walk's leaf ends without a pop: a hit)
            }
            w.next_subtree(stk);
        } else {
            w.descend(sc, stk, cnt, kDescentMin);
            if (w.cur < 0 && w.cur != w.kDone) {
                if (!w.in_blas) {
                    w.enter_instance(sc, stk, cnt);
                } else {
                    w.test_leaf(sc, stk, cnt, kLeafMin);
                }
            }
        }
        if (w.cur == w.kDone) {
            if (!ANY) {
                best = w.best;
            }
            return w.found;
        }
    }
}

template <bool COUNT>
RT_DEV void trace_pair_fast(scene_dev const& sc, fast_stack const& stk, ray_pair const& rp,
    pair_result& out, trace_counters& cnt) {
    hit_result unused;
    out.occluded = trace_fast<true, COUNT>(sc, stk, rp.has_shadow, rp.shadow_origin,
        rp.shadow_direction, rp.shadow_t_max, unused, cnt);
    out.hit = trace_fast<false, COUNT>(
        sc, stk, rp.has_closest, rp.origin, rp.direction, kInfinity, out.best, cnt);
}

}  // namespace rtd

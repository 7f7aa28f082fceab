// The path-tracing loop: one lane owns one pixel and runs all of that pixel's
// passes back to back, adding each finished path to a register accumulator in
// pass order (the same order as the reference's one-dispatch-per-pass
// read-modify-write, megakernel_raytracer.comp:108-113), then writes the pixel
// once.  The bounce loop (:509-565) and the pass loop are flattened into one
// loop so a lane whose path ends starts its next pass immediately instead of
// idling until the longest path of the wave ends.
#pragma once

#include "rt_hip.h"
#include "rt_shading.h"
#include "rt_traverse_fast.h"
#include "rt_traverse_ref.h"

namespace rtd {

struct render_params {
    scene_dev sc;
    rt_camera cam;
    float* accum;            // frame_width * frame_height * 4 floats
    const uint32_t* seeds;   // [pass][tile] random_seed words, or nullptr -> scalar_seed
    uint32_t frame_width;    // imageSize().x : flat pixel index = frame_width*y + x
    uint32_t x0, y0, x1, y1; // pixel rectangle
    uint32_t tile_width, tile_height, tiles_x, tile_count;
    uint32_t pass_count;
    uint32_t max_depth;
    uint32_t preview;        // store instead of accumulate
    uint32_t scalar_seed;
    unsigned long long* counters;  // samples, closest, shadow, hits, nodes, tris, max_stack
    uint32_t* out_instance;  // primary-id outputs (primary_ids_kernel only)
    uint32_t* out_triangle;
    float* out_t;
    float* out_bary;
    const uint32_t* block_order;  // linear 8x8-block indices, most expensive first, or nullptr
    uint32_t* block_cost;         // per-block cycle counts of this launch, or nullptr
};

// Traces a bounce's occlusion ray (optional) and the following closest-hit ray
// (optional).  The reference-order mode runs its two walks back to back; the
// rebuilt mode runs both inside one loop (rt_traverse_fast.h).
template <int ACCEL, bool COUNT>
RT_DEV void trace_pair(scene_dev const& sc, fast_stack const& stk, ray_pair const& rp,
    pair_result& out, trace_counters& c) {
    if (ACCEL == RT_HIP_ACCEL_REFERENCE) {
        out.occluded = false;
        out.hit = false;
        if (rp.has_shadow) {
            hit_result unused;
            out.occluded = trace_ref<true, COUNT>(
                sc, rp.shadow_origin, rp.shadow_direction, rp.shadow_t_max, unused, c);
        }
        if (rp.has_closest) {
            out.hit = trace_ref<false, COUNT>(sc, rp.origin, rp.direction, kInfinity, out.best, c);
        }
    } else {
        trace_pair_fast<COUNT>(sc, stk, rp, out, c);
    }
}

RT_DEV f3 load3(const float* p) { return mk3(p[0], p[1], p[2]); }

// inverse(mat3(transform)) of the shader, from the table made at upload (same inverse33, same
// bits as evaluating it here).
RT_DEV m33 load_normal_matrix(scene_dev const& sc, int32_t transform) {
    normal_matrix const n = sc.normal_matrices[transform];
    m33 m;
    m.c0 = mk3(n.c0.x, n.c0.y, n.c0.z);
    m.c1 = mk3(n.c1.x, n.c1.y, n.c1.z);
    m.c2 = mk3(n.c2.x, n.c2.y, n.c2.z);
    return m;
}

// Third phase of closest_hit (megakernel_raytracer.comp:280-322): interpolate
// the hit, build the tangent frame and move it to world space.
template <uint32_t F = kShadeAll>
RT_DEV void fill_hit_state(scene_dev const& sc, hit_result const& hr, f3 origin, f3 direction,
    hit_state& st) {
    rt_instance const inst = sc.instances[hr.instance];
    const float4* tv = reinterpret_cast<const float4*>(&sc.triangles[hr.triangle]);
    float4 const a_pu = tv[0], a_nv = tv[1], b_pu = tv[2], b_nv = tv[3], c_pu = tv[4],
                 c_nv = tv[5];
    f3 const a = mk3(a_pu.x, a_pu.y, a_pu.z), b = mk3(b_pu.x, b_pu.y, b_pu.z),
             c = mk3(c_pu.x, c_pu.y, c_pu.z);
    f3 const an = mk3(a_nv.x, a_nv.y, a_nv.z), bn = mk3(b_nv.x, b_nv.y, b_nv.z),
             cn = mk3(c_nv.x, c_nv.y, c_nv.z);
    st.position = origin + direction * hr.t;
    f3 const outward = (an * hr.b0 + bn * hr.b1) + cn * hr.b2;
    // world-space ray against the object-space normal, as in the shader
    st.front_face = dot3(direction, outward) < 0.0f;
    st.normal = st.front_face ? outward : -outward;
    // The tangent frame (:299-314) is read by the normal map alone (get_surface_info, :163-169):
    // it is built for surfaces that have one - a division, two normalisations and two matrix
    // products that every other hit went through for nothing.
    bool const mapped = feat<F>(kFeatTextures) && inst.light < 0 && inst.material >= 0 &&
                        sc.materials[inst.material].normal_tex >= 0;
    st.tangent = st.bitangent = splat3(0.0f);
    if (mapped) {
        f3 const dp1 = b - a;
        f3 const dp2 = c - a;
        float const du1 = b_pu.w - a_pu.w, dv1 = b_nv.w - a_nv.w;
        float const du2 = c_pu.w - a_pu.w, dv2 = c_nv.w - a_nv.w;
        float const inv = 1.0f / (du1 * dv2 - du2 * dv1);
        st.tangent = (dp1 * dv2 - dp2 * dv1) * inv;
        st.bitangent = (dp1 * (-du2) + dp2 * du1) * inv;
    }
    if (inst.transform >= 0) {
        st.normal = normalize3(tmul33(load_normal_matrix(sc, inst.transform), st.normal));
        if (mapped) {
            rt_mat4 const& m = sc.transforms[inst.transform];
            m33 m3;
            m3.c0 = mk3(m.m[0][0], m.m[0][1], m.m[0][2]);
            m3.c1 = mk3(m.m[1][0], m.m[1][1], m.m[1][2]);
            m3.c2 = mk3(m.m[2][0], m.m[2][1], m.m[2][2]);
            st.tangent = normalize3(mul33(m3, st.tangent));
            st.bitangent = normalize3(mul33(m3, st.bitangent));
        }
    }
    st.u = (hr.b0 * a_pu.w + hr.b1 * b_pu.w) + hr.b2 * c_pu.w;
    st.v = (hr.b0 * a_nv.w + hr.b1 * b_nv.w) + hr.b2 * c_nv.w;
    st.t = hr.t;
    st.material = inst.material;
    st.light = inst.light;
}

struct light_sample {
    f3 intensity;
    float pdf;
    f3 wi;
    float shadow_t_max;  // the shadow ray covers (0, shadow_t_max]
    int32_t type;
};

// sample_light, megakernel_raytracer.comp:440-507, split around its shadow
// ray: this part draws the light sample and prepares everything; the caller
// traces the occlusion ray (origin `position`, direction ls.wi) and drops the
// sample when it is blocked.  None of the work after the shadow ray in the
// shader draws random numbers, so hoisting it keeps the random sequence.
// Returns false when the shader returns false before tracing anything.
template <uint32_t F = kShadeAll>
RT_DEV bool propose_light(scene_dev const& sc, uint32_t& seed, f3 ray_direction, f3 position,
    light_sample& ls) {
    float const light_pdf = 1.0f / (float) sc.light_count;
    rt_light const light = sc.lights[rand_uint(seed, 0u, sc.light_count - 1u)];
    ls.type = light.type;
    if (feat<F>(kFeatSky) && light.type == RT_LIGHT_SKY) {
        // hemisphere around the incoming ray direction (not the normal), as in the shader;
        // with the opt-in tables, a direction drawn in proportion to the map instead
        bool const importance =
            feat<F>(kFeatTextures) && sc.sky_density != nullptr && light.emission_tex >= 0;
        f3 const direction = importance ? sample_sky_importance(sc, seed)
                                        : uniform_sample_hemisphere(seed, ray_direction);
        f4 const ip = eval_sky_light<F>(sc, direction);
        if (importance && !(ip.w > 0.0f)) {
            return false;  // a pole of the map: zero density, nothing to weigh
        }
        ls.intensity = mk3(ip.x, ip.y, ip.z);
        ls.pdf = light_pdf * ip.w;
        ls.wi = direction;
        ls.shadow_t_max = kInfinityMinusEps;
        return true;
    }
    if (light.type == RT_LIGHT_AREA_SINGLE_SIDED || light.type == RT_LIGHT_AREA_DOUBLE_SIDED) {
        rt_mesh const mesh = sc.meshes[light.mesh];
        uint32_t const triangle_i =
            mesh.triangle_offset + rand_uint(seed, 0u, mesh.triangle_count - 1u);
        const float4* tv = reinterpret_cast<const float4*>(&sc.triangles[triangle_i]);
        float4 const a_pu = tv[0], a_nv = tv[1], b_pu = tv[2], b_nv = tv[3], c_pu = tv[4],
                     c_nv = tv[5];
        f3 const bc = uniform_sample_triangle(seed);
        f3 const local_pos =
            (mk3(a_pu.x, a_pu.y, a_pu.z) * bc.x + mk3(b_pu.x, b_pu.y, b_pu.z) * bc.y) +
            mk3(c_pu.x, c_pu.y, c_pu.z) * bc.z;
        f3 const local_nor =
            (mk3(a_nv.x, a_nv.y, a_nv.z) * bc.x + mk3(b_nv.x, b_nv.y, b_nv.z) * bc.y) +
            mk3(c_nv.x, c_nv.y, c_nv.z) * bc.z;
        // light.transform < 0 means mat4(1.0); the products are still formed.
        m33 m3;
        f3 c3 = splat3(0.0f);
        m3.c0 = mk3(1.0f, 0.0f, 0.0f);
        m3.c1 = mk3(0.0f, 1.0f, 0.0f);
        m3.c2 = mk3(0.0f, 0.0f, 1.0f);
        m33 inv3 = m3;
        if (light.transform >= 0) {
            rt_mat4 const& m = sc.transforms[light.transform];
            m3.c0 = mk3(m.m[0][0], m.m[0][1], m.m[0][2]);
            m3.c1 = mk3(m.m[1][0], m.m[1][1], m.m[1][2]);
            m3.c2 = mk3(m.m[2][0], m.m[2][1], m.m[2][2]);
            c3 = mk3(m.m[3][0], m.m[3][1], m.m[3][2]);
            inv3 = load_normal_matrix(sc, light.transform);
        }
        f3 const light_pos =
            ((m3.c0 * local_pos.x + m3.c1 * local_pos.y) + m3.c2 * local_pos.z) + c3 * 1.0f;
        f3 const light_nor = tmul33(inv3, local_nor);
        f3 const light_to_frag = light_pos - position;
        if (light.type != RT_LIGHT_AREA_DOUBLE_SIDED && dot3(light_to_frag, light_nor) > 0.0f) {
            return false;
        }
        float const dist = length3(light_to_frag);
        float const dist2 = dist * dist;
        f3 const direction = light_to_frag / dist;
        f3 intensity = mk3(light.intensity[0], light.intensity[1], light.intensity[2]);
        if (feat<F>(kFeatTextures) && light.emission_tex >= 0) {
            float const u = (bc.x * a_pu.w + bc.y * b_pu.w) + bc.z * c_pu.w;
            float const v = (bc.x * a_nv.w + bc.y * b_nv.w) + bc.z * c_nv.w;
            intensity = intensity * texture_rgb(sc, light.emission_tex, u, v);
        }
        float const pdf_on_light = 1.0f / light.direction[0];
        ls.intensity = intensity;
        ls.pdf = light_pdf * pdf_on_light * (dist2 / fabsf(dot3(light_nor, direction)));
        ls.wi = direction;
        ls.shadow_t_max = dist - kEpsilon;
        return true;
    }
    return false;  // distant lights: the shader's branch is empty
}

// Diagnostic builds (RT_SHADE_REGION_TIMING): cycles a wave spends in each part of shade_trip.
struct region_clock {
#ifdef RT_SHADE_REGION_TIMING
    unsigned long long cycles[12] = {};
    unsigned long long last = 0;
    RT_DEV void start() {
        __builtin_amdgcn_sched_barrier(0);
        last = __builtin_readcyclecounter();
        __builtin_amdgcn_sched_barrier(0);
    }
    RT_DEV void mark(int region) {
        __builtin_amdgcn_sched_barrier(0);
        unsigned long long const now = __builtin_readcyclecounter();
        cycles[region] += now - last;
        last = now;
        __builtin_amdgcn_sched_barrier(0);
    }
#else
    RT_DEV void start() {}
    RT_DEV void mark(int) {}
#endif
};

// What shade_trip calls once a going path's last load has been issued (the light's triangle):
// from there to the end of the trip - BSDF sampling and the two evaluations, more than half of
// its instructions - nothing waits for memory, so a caller can start the loads of its NEXT path
// there and find them complete when it gets to it.
struct no_prefetch {
    RT_DEV void after_loads() {}
};

// What a path carries from one bounce to the next.
struct path_state {
    f3 radiance;
    f3 throughput;
    f3 nee;               // next-event contribution waiting for its occlusion ray
    float last_bsdf_pdf;
    uint32_t depth;
    uint32_t seed;
};

// One trip of the bounce loop (megakernel_raytracer.comp:509-565) after its rays have been
// traced: `pr` holds the outcome of the previous bounce's occlusion ray and of this bounce's
// closest-hit ray (`rp` says which of them existed).  All shading of a bounce (light
// proposal, both BSDF evaluations, BSDF sampling) happens BEFORE its occlusion ray is traced:
// the next-event contribution is kept as three floats (`nee`) and added when the ray turns
// out unoccluded.  Nothing between the shader's occlusion test (:488) and the next bounce
// draws random numbers or adds to the radiance, so values and order of every addition are
// those of the shader; only the surface state no longer has to stay alive across a
// traversal.  On return `rp` describes the next trip's rays; the path is over when it has
// neither.
template <uint32_t F = kShadeAll, int PART = kTripWhole, class HOOK = no_prefetch>
RT_DEV void shade_trip(scene_dev const& sc, uint32_t max_depth, ray_pair& rp,
    pair_result const& pr, path_state& ps, uint32_t& closest_hits, uint32_t& shadow_skipped,
    region_clock& rc, HOOK& hook) {
    if (rp.has_shadow && !pr.occluded) {
        ps.radiance = ps.radiance + ps.nee;  // :547-549
    }
    rp.has_shadow = false;
    if (rp.has_closest) {
        rp.has_closest = false;  // set again below when the path goes on
        if (PART != kTripGoesOn && !pr.hit) {
            // :518-527
            f4 const ip = eval_sky_light<F>(sc, rp.direction);
            if (ip.w > 0.0f) {
                float const mis = ps.depth > 0u ? power_heuristic(ps.last_bsdf_pdf, ip.w) : 1.0f;
                ps.radiance = ps.radiance + (mk3(ip.x, ip.y, ip.z) * mis) * ps.throughput;
            }
            rc.mark(1);
        } else {
            ++closest_hits;
            hit_state st;
            fill_hit_state<F>(sc, pr.best, rp.origin, rp.direction, st);
            rc.mark(2);
            surface si;
            make_surface<F, PART>(sc, st, si);
            ps.radiance = ps.radiance + si.emission * ps.throughput;
            rc.mark(3);
            bool const goes_on = PART == kTripGoesOn
                                     ? true
                                     : (PART == kTripEnds ? false
                                                          : st.light < 0 && ps.depth != max_depth);
            if (goes_on) {
                lobe_mix const lm = make_lobe_mix(st, si, rp.direction);
                rc.mark(4);
                // state.glsl:14-16 - undoes the ray-facing flip, so the offset follows the
                // mesh's own normal direction even for back-face hits
                f3 const front_face_normal = st.front_face ? st.normal : -st.normal;
                f3 const shadow_origin = st.position + front_face_normal * kEpsilon;
                light_sample ls;
                bool const have_light =
                    propose_light<F>(sc, ps.seed, rp.direction, shadow_origin, ls);
                rc.mark(5);
                hook.after_loads();
                f3 const next_direction = sample_disney<F>(ps.seed, st, si, lm);
                rc.mark(6);
                // One eval_disney body serves both uses (:541-550 and :554-560).
#pragma nounroll
                for (int k = have_light ? 0 : 1; k < 2; ++k) {
                    f3 const l = k == 0 ? ls.wi : next_direction;
                    f4 const bsdf = eval_disney<F>(st, si, lm, l);
                    if (k == 0) {
                        if (bsdf.w > 0.0f) {
                            bool const is_delta = ls.type == RT_LIGHT_DISTANT;
                            float const mis =
                                is_delta ? 1.0f : power_heuristic(ls.pdf, bsdf.w);
                            f3 const f = mk3(bsdf.x, bsdf.y, bsdf.z);
                            ps.nee = (((f * mis) * ls.intensity) / ls.pdf) * ps.throughput;
                            rp.has_shadow = true;
                            rp.shadow_origin = shadow_origin;
                            rp.shadow_direction = ls.wi;
                            rp.shadow_t_max = ls.shadow_t_max;
                        } else {
                            // the shader traces this occlusion ray and then discards the
                            // sample (pdf <= 0): its outcome cannot reach the image
                            ++shadow_skipped;
                        }
                    } else {
                        ps.last_bsdf_pdf = bsdf.w;
                        if (bsdf.w > 0.0f) {
                            ps.throughput = ps.throughput * (mk3(bsdf.x, bsdf.y, bsdf.z) / bsdf.w);
                            rp.has_closest = true;
                        }
                    }
                }
                rc.mark(7);
                rp.direction = next_direction;
                rp.origin = st.position + next_direction * kEpsilon;
                ++ps.depth;
            }
        }
    }
}

template <uint32_t F = kShadeAll, int PART = kTripWhole>
RT_DEV void shade_trip(scene_dev const& sc, uint32_t max_depth, ray_pair& rp,
    pair_result const& pr, path_state& ps, uint32_t& closest_hits, uint32_t& shadow_skipped) {
    region_clock rc;
    no_prefetch hook;
    shade_trip<F, PART>(sc, max_depth, rp, pr, ps, closest_hits, shadow_skipped, rc, hook);
}

// Seeds a pixel for one pass and builds its jittered primary ray
// (megakernel_raytracer.comp:85-100).
RT_DEV void primary_ray(render_params const& p, uint32_t x, uint32_t y, uint32_t random_seed,
    uint32_t& seed, f3& origin, f3& direction) {
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
}

RT_DEV uint32_t pass_seed(render_params const& p, uint32_t pass, uint32_t x, uint32_t y) {
    if (p.seeds == nullptr) {
        return p.scalar_seed;
    }
    uint32_t const tile = (y / p.tile_height) * p.tiles_x + x / p.tile_width;
    return p.seeds[(size_t) pass * p.tile_count + tile];
}

}  // namespace rtd

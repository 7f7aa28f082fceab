// Known-answer probes: single calls of the device functions the render kernels are made of,
// on caller-supplied arguments, so that each can be compared with the CPU oracle's probe of the
// same reference function instead of only through whole images.  One thread per record; the
// functions called are exactly the ones the kernels call.
#pragma once

#include "rt_hip.h"
#include "rt_path.h"

namespace rtd {

__global__ void __launch_bounds__(64) probe_kernel(scene_dev const sc, uint32_t kind,
    uint32_t count, const rt_hip_probe_in* in, rt_hip_probe_out* out) {
    uint32_t const i = blockIdx.x * 64u + threadIdx.x;
    if (i >= count) {
        return;
    }
    rt_hip_probe_in const& q = in[i];
    rt_hip_probe_out r{};
    switch (kind) {
    case RT_HIP_PROBE_RAND01: {  // shaders/common/random.glsl:12-24
        uint32_t seed = q.u[0];
        for (int k = 0; k < 16; ++k) {
            r.f[k] = rand_01(seed);
            r.u[k] = seed;
        }
        break;
    }
    case RT_HIP_PROBE_HASH_COMBINE:  // random.glsl:8-10
        r.u[0] = hash_combine(q.u[0], q.u[1]);
        break;
    case RT_HIP_PROBE_RAND_UINT: {  // random.glsl:16-19
        uint32_t seed = q.u[0];
        r.u[0] = rand_uint(seed, q.u[1], q.u[2]);
        r.u[1] = seed;
        break;
    }
    case RT_HIP_PROBE_HIT_AABB: {  // shaders/raytracer/ray.glsl:11-31
        rt_bvh_node n{};
        n.x_min = q.f[0], n.x_max = q.f[1];
        n.y_min = q.f[2], n.y_max = q.f[3];
        n.z_min = q.f[4], n.z_max = q.f[5];
        f3 const o = mk3(q.f[6], q.f[7], q.f[8]);
        f3 const d = mk3(q.f[9], q.f[10], q.f[11]);
        f3 const inv = splat3(1.0f) / d;
        r.u[0] = hit_aabb_ref(n, o, inv, d.x < 0.0f, d.y < 0.0f, d.z < 0.0f, q.f[12], q.f[13])
                     ? 1u : 0u;
        // the rebuilt structure's slab test (one fma per plane, near / far planes picked by the
        // sign of the direction as the walk does, widened on the exit side), t_min = 0
        float enter, exit;
        slab_ray const sr = make_slab_ray(o, d);
        plane_offsets const po = make_plane_offsets(sr);
        slab_box(sr, po.near_x == 0u ? n.x_min : n.x_max, po.near_y == 16u ? n.y_min : n.y_max,
            po.near_z == 32u ? n.z_min : n.z_max, po.far_x == 0u ? n.x_min : n.x_max,
            po.far_y == 16u ? n.y_min : n.y_max, po.far_z == 32u ? n.z_min : n.z_max, q.f[13],
            enter, exit);
        r.u[1] = enter <= exit ? 1u : 0u;
        break;
    }
    case RT_HIP_PROBE_HIT_TRIANGLE: {  // ray.glsl:119-183 (and :47-105), t_min = 0
        f3 const o = mk3(q.f[9], q.f[10], q.f[11]);
        f3 const d = mk3(q.f[12], q.f[13], q.f[14]);
        float b0 = 0.0f, b1 = 0.0f, b2 = 0.0f, t = 0.0f;
        bool const hit = hit_triangle(mk3(q.f[0], q.f[1], q.f[2]), mk3(q.f[3], q.f[4], q.f[5]),
            mk3(q.f[6], q.f[7], q.f[8]), make_tri_ray(o, d), q.f[15], b0, b1, b2, t);
        r.u[0] = hit ? 1u : 0u;
        r.f[0] = b0, r.f[1] = b1, r.f[2] = b2, r.f[3] = t;
        break;
    }
    case RT_HIP_PROBE_TRIANGLE_DEGENERATE:  // ray.glsl:124-128, the test the builder hoists
        r.u[0] = degenerate_triangle(mk3(q.f[0], q.f[1], q.f[2]), mk3(q.f[3], q.f[4], q.f[5]),
                     mk3(q.f[6], q.f[7], q.f[8])) ? 1u : 0u;
        break;
    case RT_HIP_PROBE_EVAL_DISNEY:      // shaders/raytracer/disney.glsl:170-248
    case RT_HIP_PROBE_SAMPLE_DISNEY: {  // disney.glsl:250-315
        // the surface of megakernel_raytracer.comp:126-182 for a hit on this material alone
        scene_dev one = sc;
        rt_material material = q.material;
        derive_material(material);  // as the upload does for the scene's table
        one.materials = &material;
        hit_state st{};
        st.normal = mk3(q.f[0], q.f[1], q.f[2]);
        st.tangent = st.bitangent = st.position = splat3(0.0f);
        st.u = st.v = st.t = 0.0f;
        st.front_face = q.u[0] != 0u;
        st.material = 0;
        st.light = -1;
        surface si;
        make_surface(one, st, si);
        f3 const ray_direction = mk3(q.f[3], q.f[4], q.f[5]);
        lobe_mix const lm = make_lobe_mix(st, si, ray_direction);
        if (kind == RT_HIP_PROBE_EVAL_DISNEY) {
            f4 const e = eval_disney(st, si, lm, mk3(q.f[6], q.f[7], q.f[8]));
            r.f[0] = e.x, r.f[1] = e.y, r.f[2] = e.z, r.f[3] = e.w;
        } else {
            uint32_t seed = q.u[1];
            f3 const l = sample_disney(seed, st, si, lm);
            r.f[0] = l.x, r.f[1] = l.y, r.f[2] = l.z;
            r.u[0] = seed;
        }
        break;
    }
    case RT_HIP_PROBE_SAMPLE_TEXTURE: {  // texture(): bilinear, REPEAT, one level
        f4 const s = sample_texture(sc, (int) q.u[0], q.f[0], q.f[1]);
        r.f[0] = s.x, r.f[1] = s.y, r.f[2] = s.z, r.f[3] = s.w;
        break;
    }
    case RT_HIP_PROBE_SKY: {  // eval_sky_light, megakernel_raytracer.comp:413-438
        f4 const s = eval_sky_light(sc, mk3(q.f[0], q.f[1], q.f[2]));
        r.f[0] = s.x, r.f[1] = s.y, r.f[2] = s.z, r.f[3] = s.w;
        break;
    }
    case RT_HIP_PROBE_SKY_SAMPLE: {  // the opt-in sky distribution: one draw and its density
        uint32_t seed = q.u[0];
        f3 const d = sample_sky_importance(sc, seed);
        f4 const s = eval_sky_light(sc, d);
        r.f[0] = d.x, r.f[1] = d.y, r.f[2] = d.z, r.f[3] = s.w;
        r.u[0] = seed;
        break;
    }
    default:
        break;
    }
    out[i] = r;
}

}  // namespace rtd

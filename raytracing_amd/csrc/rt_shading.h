// Device shading: texture fetch, surface set-up, sampling helpers, the Disney
// BSDF and sky evaluation.  Each function names the reference GLSL it stands
// for (paths relative to the upstream repository root).
#pragma once

#include "rt_device_scene.h"

namespace rtd {

// What closest_hit() leaves behind (shaders/raytracer/state.glsl:1-12).
struct hit_state {
    f3 position, normal, tangent, bitangent;
    float u, v, t;
    int32_t material, light;
    bool front_face;
};

// shaders/raytracer/state.glsl:18-35
struct surface {
    f3 albedo, emission;
    float metallic, spec_trans, eta, ior, inv_ior, subsurface, roughness, specular_tint;
    float ax, ay, sheen, sheen_tint, clearcoat, clearcoat_gloss;
};

// ---------------------------------------------------------------------------
// What the uploaded scene's materials and lights can make the shading code do.  The shading
// kernel of the pipeline is compiled once per supported combination (wf_shade_kernel) and the
// upload picks the smallest one that covers the scene: a branch that no material or light of the
// scene can take is left out at compile time, which frees its registers and cannot change a
// bit of any result.  kShadeAll keeps everything (the fused kernels, the probes).
enum : uint32_t {
    kFeatGlass = 1u,      // some material has spec_trans != 0 (the glass lobe can get a weight)
    kFeatClearcoat = 2u,  // some material has clearcoat != 0
    kFeatMetal = 4u,      // some material has metallic != 0 or a metallic-roughness texture
    kFeatTextures = 8u,   // some material or light (the sky included) names a texture
    kFeatSky = 16u,       // there is a sky light
    kShadeAll = 31u,
};
template <uint32_t F>
RT_DEV constexpr bool feat(uint32_t bit) { return (F & bit) != 0u; }

// Which paths a caller of shade_trip hands it (the pipeline sorts them beforehand, so each of
// its two shading kernels carries only the code its paths can reach).
enum : int {
    kTripWhole = 0,   // any path
    kTripEnds = 1,    // only paths that end at this trip: no ray, a miss, a light, or max_depth
    kTripGoesOn = 2,  // only paths whose ray hit a surface below max_depth
};

// ---------------------------------------------------------------------------
// texture(): bilinear, REPEAT addressing, one mip level, fp32 weights, texel
// centres at +0.5 (sampler: src/vulkan/vulkan_image.cpp:309-330).
RT_DEV uint32_t wrap_repeat(int i, uint32_t size) {
    int const s = (int) size;
    if ((size & (size - 1u)) == 0u) {
        // power-of-two sizes (the usual case): the two's-complement bits are the remainder
        return (uint32_t) i & (size - 1u);
    }
    int m = i % s;
    if (m < 0) {
        m += s;
    }
    return (uint32_t) m;
}

// byte / 255.0f, correctly rounded, in three operations instead of the eleven of a general
// division: the product with the rounded reciprocal is off by at most one unit in the last
// place and one residual step repairs it - checked against the division for all 256 bytes
// (tests/test_texture_unorm.py).
RT_DEV float unorm8(uint8_t byte) {
    float const x = (float) byte;
    float const r = 1.0f / 255.0f;
    float const q = x * r;
    return __builtin_fmaf(__builtin_fmaf(-q, 255.0f, x), r, q);
}

RT_DEV f4 load_texel(texture_dev const& t, uint32_t x, uint32_t y) {
    size_t const idx = (size_t) y * t.width + x;
    if (t.format == RT_TEX_F32X4) {
        float4 const p = reinterpret_cast<const float4*>(t.data)[idx];
        return f4{p.x, p.y, p.z, p.w};
    }
    uchar4 const p = reinterpret_cast<const uchar4*>(t.data)[idx];
    return f4{unorm8(p.x), unorm8(p.y), unorm8(p.z), unorm8(p.w)};
}

RT_DEV f4 sample_texture(scene_dev const& sc, int tex, float u, float v) {
    texture_dev const t = sc.textures[tex];
    float const x = u * (float) t.width - 0.5f;
    float const y = v * (float) t.height - 0.5f;
    float const fx = floorf(x);
    float const fy = floorf(y);
    float const a = x - fx;
    float const b = y - fy;
    // the neighbour of a wrapped coordinate is its successor, wrapped once more at the edge:
    // the same texel as wrapping fx + 1 itself, without a second remainder
    uint32_t const x0 = wrap_repeat((int) fx, t.width);
    uint32_t const x1 = x0 + 1u == t.width ? 0u : x0 + 1u;
    uint32_t const y0 = wrap_repeat((int) fy, t.height);
    uint32_t const y1 = y0 + 1u == t.height ? 0u : y0 + 1u;
    f4 const t00 = load_texel(t, x0, y0);
    f4 const t10 = load_texel(t, x1, y0);
    f4 const t01 = load_texel(t, x0, y1);
    f4 const t11 = load_texel(t, x1, y1);
    f4 r;
    r.x = mix1(mix1(t00.x, t10.x, a), mix1(t01.x, t11.x, a), b);
    r.y = mix1(mix1(t00.y, t10.y, a), mix1(t01.y, t11.y, a), b);
    r.z = mix1(mix1(t00.z, t10.z, a), mix1(t01.z, t11.z, a), b);
    r.w = mix1(mix1(t00.w, t10.w, a), mix1(t01.w, t11.w, a), b);
    return r;
}

RT_DEV f3 texture_rgb(scene_dev const& sc, int tex, float u, float v) {
    f4 const s = sample_texture(sc, tex, u, v);
    return f3{s.x, s.y, s.z};
}

// ---------------------------------------------------------------------------
// What get_surface_info (:145, :178-180) and eval_disney (disney.glsl:214) compute from the
// material alone, hit after hit: 1 / ior and the anisotropic roughness pair.  The device copy
// of the material table carries them in the record's padding, filled once per upload by
// derive_material - the same operations on the same inputs, so the same bits.
RT_DEV void derive_material(rt_material& m) {
    m._p0[0] = 1.0f / m.ior;
    float const aspect = sqrtf(1.0f - m.anisotropic * 0.9f);
    m._p0[1] = fmaxf(0.001f, m.roughness / aspect);
    m._p0[2] = fmaxf(0.001f, m.roughness * aspect);
}
RT_DEV float material_inv_ior(rt_material const& m) { return m._p0[0]; }
RT_DEV float material_ax(rt_material const& m) { return m._p0[1]; }
RT_DEV float material_ay(rt_material const& m) { return m._p0[2]; }

// ---------------------------------------------------------------------------
// shaders/raytracer/megakernel_raytracer.comp:126-182 (get_surface_info).
// May replace st.normal with the normal-mapped one.
template <uint32_t F = kShadeAll, int PART = kTripWhole>
RT_DEV void make_surface(scene_dev const& sc, hit_state& st, surface& si) {
    constexpr bool kTex = feat<F>(kFeatTextures);
    si = surface{};
    if (PART != kTripGoesOn && st.light >= 0) {
        rt_light const& light = sc.lights[st.light];
        if (st.front_face || light.type == RT_LIGHT_AREA_DOUBLE_SIDED) {
            si.emission = mk3(light.intensity[0], light.intensity[1], light.intensity[2]);
            if (kTex && light.emission_tex >= 0) {
                si.emission = si.emission * texture_rgb(sc, light.emission_tex, st.u, st.v);
            }
        }
    } else if (st.material >= 0) {
        rt_material const m = sc.materials[st.material];
        si.albedo = mk3(m.albedo[0], m.albedo[1], m.albedo[2]);
        si.emission = mk3(m.emission[0], m.emission[1], m.emission[2]);
        si.metallic = m.metallic;
        si.spec_trans = m.spec_trans;
        si.eta = st.front_face ? material_inv_ior(m) : m.ior;
        si.ior = m.ior;
        si.inv_ior = material_inv_ior(m);
        si.subsurface = m.subsurface;
        si.roughness = m.roughness;
        si.specular_tint = m.specular_tint;
        si.sheen = m.sheen;
        si.sheen_tint = m.sheen_tint;
        si.clearcoat = m.clearcoat;
        si.clearcoat_gloss = m.clearcoat_gloss;
        if (kTex && m.albedo_tex >= 0) {
            si.albedo = si.albedo * texture_rgb(sc, m.albedo_tex, st.u, st.v);
        }
        if (kTex && m.emission_tex >= 0) {
            si.emission = si.emission * texture_rgb(sc, m.emission_tex, st.u, st.v);
        }
        if (kTex && m.normal_tex >= 0) {
            f3 const texel = texture_rgb(sc, m.normal_tex, st.u, st.v);
            f3 const n = normalize3(texel * 2.0f - splat3(1.0f));
            st.normal = normalize3((st.tangent * n.x + st.bitangent * n.y) + st.normal * n.z);
        }
        if (kTex && m.metallic_roughness_tex >= 0) {
            f4 const texel = sample_texture(sc, m.metallic_roughness_tex, st.u, st.v);
            si.metallic = texel.z;  // .bg swizzle: blue = metallic, green = roughness
            si.roughness = fmaxf(texel.y * texel.y, 0.001f);
        }
        // The anisotropy split uses the material's roughness, not the textured one: it is a
        // constant of the material (derive_material).
        si.ax = material_ax(m);
        si.ay = material_ay(m);
    }
}

// ---------------------------------------------------------------------------
// shaders/raytracer/sampling.glsl
RT_DEV float power_heuristic(float a, float b) {
    float const a2 = a * a;
    return a2 / (a2 + b * b);
}

// :6-18.  phi = 2*r1 (no pi), as written in the shader.
RT_DEV f3 uniform_sample_hemisphere(uint32_t& seed, f3 around) {
    float const r0 = rand_01(seed);
    float const r1 = rand_01(seed);
    float const cos_theta = 1.0f - 2.0f * r0;
    float const sin_theta = sqrtf(fmaxf(0.0f, 1.0f - cos_theta * cos_theta));
    float const phi = 2.0f * r1;
    f3 const r = mk3(sin_theta * cosf(phi), sin_theta * sinf(phi), cos_theta);
    return dot3(r, around) < 0.0f ? -r : r;
}

// :20-33
RT_DEV f3 uniform_sample_triangle(uint32_t& seed) {
    float const r0 = rand_01(seed);
    float const r1 = rand_01(seed);
    float b0, b1;
    if (r0 < r1) {
        b0 = 0.5f * r0;
        b1 = r1 - b0;
    } else {
        b1 = 0.5f * r1;
        b0 = r0 - b1;
    }
    return mk3(b0, b1, (1.0f - b0) - b1);
}

// The point of the unit disk that cosine_sample_hemisphere (:35-44) and sample_ggx_vndf
// (:46-63) both start from: r = sqrt(r1), phi = 2 pi r2, (r cos phi, r sin phi).  Whichever
// of the two a lane runs, these are the same operations on the same two random numbers, so
// sample_disney computes them once for the wave instead of once per divergent branch.
struct disk_point {
    float x, y;
};
RT_DEV disk_point sample_disk(float r1, float r2) {
    float const r = sqrtf(r1);
    float const phi = kTwoPi * r2;
    return disk_point{r * cosf(phi), r * sinf(phi)};
}

// :35-44
RT_DEV f3 cosine_sample_hemisphere(disk_point d) {
    float const x = d.x;
    float const y = d.y;
    float const z = sqrtf(fmaxf(0.0f, (1.0f - x * x) - y * y));
    return mk3(x, y, z);
}

// :46-63 (Heitz 2018 visible-normal sampling)
RT_DEV f3 sample_ggx_vndf(disk_point d, f3 v, float ax, float ay) {
    f3 const vh = normalize3(mk3(ax * v.x, ay * v.y, v.z));
    float const lensq = vh.x * vh.x + vh.y * vh.y;
    f3 const t1v = lensq > 0.0f ? mk3(-vh.y, vh.x, 0.0f) * (1.0f / sqrtf(lensq)) :
                                  mk3(1.0f, 0.0f, 0.0f);
    f3 const t2v = cross3(vh, t1v);
    float const t1 = d.x;
    float t2 = d.y;
    float const s = 0.5f * (1.0f + vh.z);
    t2 = (1.0f - s) * sqrtf(1.0f - t1 * t1) + s * t2;
    f3 const nh =
        (t1v * t1 + t2v * t2) + vh * sqrtf(fmaxf(0.0f, (1.0f - t1 * t1) - t2 * t2));
    return normalize3(mk3(ax * nh.x, ay * nh.y, fmaxf(0.0f, nh.z)));
}

// :65-77
RT_DEV f3 sample_gtr1(float r1, float r2, float roughness) {
    float const a = fmaxf(0.001f, roughness);
    float const a2 = a * a;
    float const phi = r1 * kTwoPi;
    float const cos_theta = sqrtf((1.0f - powf(a2, 1.0f - r2)) / (1.0f - a2));
    float const sin_theta = clamp1(sqrtf(1.0f - (cos_theta * cos_theta)), 0.0f, 1.0f);
    float const sin_phi = sinf(phi);
    float const cos_phi = cosf(phi);
    return mk3(sin_theta * cos_phi, sin_theta * sin_phi, cos_theta);
}

// ---------------------------------------------------------------------------
// shaders/raytracer/disney.glsl

// :27-31
RT_DEV float schlick_weight(float u) {
    float const m = clamp1(1.0f - u, 0.0f, 1.0f);
    float const m2 = m * m;
    return m2 * m2 * m;
}

// :61-71
RT_DEV float dielectric_fresnel(float cos_theta_i, float eta) {
    float const sin_theta_t2 = eta * eta * (1.0f - cos_theta_i * cos_theta_i);
    if (sin_theta_t2 > 1.0f) {
        return 1.0f;
    }
    float const cos_theta_t = sqrtf(fmaxf(1.0f - sin_theta_t2, 0.0f));
    float const rs = (eta * cos_theta_t - cos_theta_i) / (eta * cos_theta_t + cos_theta_i);
    float const rp = (eta * cos_theta_i - cos_theta_t) / (eta * cos_theta_i + cos_theta_t);
    return 0.5f * (rs * rs + rp * rp);
}

// :73-82
RT_DEV float gtr2_aniso(float n_h, float h_x, float h_y, float ax, float ay) {
    float const a = h_x / ax;
    float const b = h_y / ay;
    float const c = (a * a + b * b) + n_h * n_h;
    return 1.0f / (kPi * ax * ay * c * c);
}

// :84-93
RT_DEV float smith_g_aniso(float n_v, float v_x, float v_y, float ax, float ay) {
    float const a = v_x * ax;
    float const b = v_y * ay;
    float const c = n_v;
    return (2.0f * n_v) / (n_v + sqrtf((a * a + b * b) + c * c));
}

// :136-143
RT_DEV float gtr1(float n_h, float a) {
    if (a >= 1.0f) {
        return kOneOverPi;
    }
    float const a2 = a * a;
    float const t = 1.0f + (a2 - 1.0f) * n_h * n_h;
    return (a2 - 1.0f) / (kPi * logf(a2) * t);
}

// :145-150
RT_DEV float smith_g(float n_v, float alpha) {
    float const a = alpha * alpha;
    float const b = n_v * n_v;
    return (2.0f * n_v) / (n_v + sqrtf((a + b) - a * b));
}

// Per-hit quantities that eval_disney (:183-201) and sample_disney (:259-277)
// both derive from the surface and the local view vector: computed once.
struct lobe_mix {
    f3 t, b;             // tangent frame around the shading normal
    f3 v;                // view vector (towards the eye) in that frame
    f3 csheen, cspec0;
    float f0;
    float dielectric_wt, metal_wt, glass_wt;
    float diffuse_pr, dielectric_pr, metal_pr, glass_pr, clearcoat_pr;
};

RT_DEV lobe_mix make_lobe_mix(hit_state const& st, surface const& si, f3 ray_direction) {
    lobe_mix lm;
    onb(st.normal, lm.t, lm.b);
    f3 const minus_v = -ray_direction;
    lm.v = mk3(dot3(minus_v, lm.t), dot3(minus_v, lm.b), dot3(minus_v, st.normal));
    // tint_colors, :15-25
    float const lum = luminance(si.albedo);
    f3 const ctint = lum > 0.0f ? si.albedo / lum : splat3(1.0f);
    float f0 = (1.0f - si.eta) / (1.0f + si.eta);
    f0 *= f0;
    lm.f0 = f0;
    lm.cspec0 = mix3(splat3(1.0f), ctint, si.specular_tint) * f0;
    lm.csheen = mix3(splat3(1.0f), ctint, si.sheen_tint);
    lm.dielectric_wt = (1.0f - si.metallic) * (1.0f - si.spec_trans);
    lm.metal_wt = si.metallic;
    lm.glass_wt = (1.0f - si.metallic) * si.spec_trans;
    float const schlick_wt = schlick_weight(lm.v.z);
    lm.diffuse_pr = lm.dielectric_wt * luminance(si.albedo);
    lm.dielectric_pr = lm.dielectric_wt * luminance(mix3(lm.cspec0, splat3(1.0f), schlick_wt));
    lm.metal_pr = lm.metal_wt * luminance(mix3(si.albedo, splat3(1.0f), schlick_wt));
    lm.glass_pr = lm.glass_wt;
    lm.clearcoat_pr = 0.25f * si.clearcoat;
    float const one_pr_sum = 1.0f / ((((lm.diffuse_pr + lm.dielectric_pr) + lm.metal_pr) +
                                         lm.glass_pr) +
                                        lm.clearcoat_pr);
    lm.diffuse_pr *= one_pr_sum;
    lm.dielectric_pr *= one_pr_sum;
    lm.metal_pr *= one_pr_sum;
    lm.glass_pr *= one_pr_sum;
    lm.clearcoat_pr *= one_pr_sum;
    return lm;
}

// GGX reflection lobe, :95-110.  Returns f (xyz, before the Fresnel colour) and pdf.
struct lobe_term {
    float f;    // scalar part common to the three channels
    float pdf;
};

// eval_disney, :170-248.  l_world: direction leaving the surface.  Returns (f*|cos|, pdf).
template <uint32_t F = kShadeAll>
RT_DEV f4 eval_disney(hit_state const& st, surface const& si, lobe_mix const& lm, f3 l_world) {
    constexpr bool kGlass = feat<F>(kFeatGlass);
    constexpr bool kMetal = feat<F>(kFeatMetal);
    constexpr bool kClearcoat = feat<F>(kFeatClearcoat);
    f3 const v = lm.v;
    f3 const l = mk3(dot3(l_world, lm.t), dot3(l_world, lm.b), dot3(l_world, st.normal));
    f3 h = l.z > 0.0f ? normalize3(l + v) : normalize3(l + v * si.eta);
    h = h.z < 0.0f ? -h : h;
    bool const reflect = l.z * v.z > 0.0f;
    float const abs_v_h = fabsf(dot3(v, h));
    f3 f = splat3(0.0f);
    float pdf = 0.0f;

    // diffuse + retro-reflection + fake subsurface + sheen, :33-59
    if (lm.diffuse_pr > 0.0f && reflect) {
        if (l.z > 0.0f) {
            float const l_h = dot3(l, h);
            float const rr = 2.0f * si.roughness * l_h * l_h;
            float const fl = schlick_weight(l.z);
            float const fv = schlick_weight(v.z);
            float const fretro = rr * ((fl + fv) + fl * fv * (rr - 1.0f));
            float const fd = (1.0f - 0.5f * fl) * (1.0f - 0.5f * fv);
            float const fss90 = 0.5f * rr;
            float const fss = mix1(1.0f, fss90, fl) * mix1(1.0f, fss90, fv);
            float const ss = 1.25f * (fss * (1.0f / (l.z + v.z) - 0.5f) + 0.5f);
            float const fh = schlick_weight(l_h);
            f3 const fsheen = lm.csheen * (fh * si.sheen);
            f3 const contrib =
                (si.albedo * kOneOverPi) * mix1(fd + fretro, ss, si.subsurface) + fsheen;
            f = f + contrib * lm.dielectric_wt;
            pdf += lm.diffuse_pr * (l.z * kOneOverPi);
        }
    }

    // The three reflection lobes share D, G1, G2 (:95-110); they differ in the Fresnel colour.
    bool const want_reflection =
        reflect && l.z > 0.0f &&
        (lm.dielectric_pr > 0.0f || (kMetal && lm.metal_pr > 0.0f) ||
            (kGlass && lm.glass_pr > 0.0f));
    float d_ggx = 0.0f, g1 = 0.0f, g2 = 0.0f;
    if (want_reflection || (kGlass && lm.glass_pr > 0.0f && !reflect && l.z < 0.0f)) {
        d_ggx = gtr2_aniso(h.z, h.x, h.y, si.ax, si.ay);
        g1 = smith_g_aniso(fabsf(v.z), v.x, v.y, si.ax, si.ay);
        g2 = g1 * smith_g_aniso(fabsf(l.z), l.x, l.y, si.ax, si.ay);
    }
    float const refl_denom = 4.0f * l.z * v.z;
    float const refl_pdf = g1 * d_ggx / (4.0f * v.z);

    if (lm.dielectric_pr > 0.0f && reflect && l.z > 0.0f) {
        float const fr = (dielectric_fresnel(abs_v_h, si.inv_ior) - lm.f0) / (1.0f - lm.f0);
        f3 const fc = mix3(lm.cspec0, splat3(1.0f), fr);
        f3 const contrib = mk3(fc.x * d_ggx * g2 / refl_denom, fc.y * d_ggx * g2 / refl_denom,
            fc.z * d_ggx * g2 / refl_denom);
        f = f + contrib * lm.dielectric_wt;
        pdf += lm.dielectric_pr * refl_pdf;
    }
    if (kMetal && lm.metal_pr > 0.0f && reflect && l.z > 0.0f) {
        f3 const fc = mix3(si.albedo, splat3(1.0f), schlick_weight(abs_v_h));
        f3 const contrib = mk3(fc.x * d_ggx * g2 / refl_denom, fc.y * d_ggx * g2 / refl_denom,
            fc.z * d_ggx * g2 / refl_denom);
        f = f + contrib * lm.metal_wt;
        pdf += lm.metal_pr * refl_pdf;
    }
    if (kGlass && lm.glass_pr > 0.0f) {
        float const fr = dielectric_fresnel(abs_v_h, si.eta);
        if (reflect) {
            if (l.z > 0.0f) {
                float const c = fr * d_ggx * g2 / refl_denom;
                f = f + splat3(c) * lm.glass_wt;
                pdf += (lm.glass_pr * fr) * refl_pdf;
            }
        } else if (l.z < 0.0f) {
            // eval_microfacet_refraction, :112-134
            float const l_h = dot3(l, h);
            float const v_h = dot3(v, h);
            float const denom = l_h + v_h * si.eta;
            float const denom2 = denom * denom;
            float const eta2 = si.eta * si.eta;
            float const jacobian = fabsf(l_h) / pad_above_zero(denom2);
            float const tail = pad_above_zero(fabsf(l.z * v.z));
            float const abs_vh = fabsf(v_h);
            float const one_minus_f = 1.0f - fr;
            f3 const tint =
                mk3(powf(si.albedo.x, 0.5f), powf(si.albedo.y, 0.5f), powf(si.albedo.z, 0.5f));
            f3 const contrib =
                mk3(tint.x * one_minus_f * d_ggx * g2 * abs_vh * jacobian * eta2 / tail,
                    tint.y * one_minus_f * d_ggx * g2 * abs_vh * jacobian * eta2 / tail,
                    tint.z * one_minus_f * d_ggx * g2 * abs_vh * jacobian * eta2 / tail);
            f = f + contrib * lm.glass_wt;
            pdf += (lm.glass_pr * (1.0f - fr)) *
                   (g1 * fmaxf(0.0f, v_h) * d_ggx * jacobian / pad_above_zero(v.z));
        }
    }
    // clearcoat, :152-168
    if (kClearcoat && lm.clearcoat_pr > 0.0f && reflect && l.z > 0.0f) {
        float const v_h = dot3(v, h);
        float const fc = mix1(0.04f, 1.0f, schlick_weight(v_h));
        float const d = gtr1(h.z, si.clearcoat_gloss);
        float const g = smith_g(l.z, 0.25f) * smith_g(v.z, 0.25f);
        float const jacobian = 1.0f / (4.0f * v_h);
        f = f + splat3(fc * d * g) * (0.25f * si.clearcoat);
        pdf += lm.clearcoat_pr * (d * h.z * jacobian);
    }
    float const cos_l = fabsf(l.z);
    return f4{f.x * cos_l, f.y * cos_l, f.z * cos_l, pdf};
}

RT_DEV f3 reflect3(f3 i, f3 n) { return i - n * (2.0f * dot3(n, i)); }
RT_DEV f3 refract3(f3 i, f3 n, float eta) {
    float const d = dot3(n, i);
    float const k = 1.0f - eta * eta * (1.0f - d * d);
    if (k < 0.0f) {
        return splat3(0.0f);
    }
    return i * eta - n * (eta * d + sqrtf(k));
}

// sample_disney, :250-315.  Returns the sampled world-space direction.
template <uint32_t F = kShadeAll>
RT_DEV f3 sample_disney(uint32_t& seed, hit_state const& st, surface const& si,
    lobe_mix const& lm) {
    // Without a glass material glass_pr is zero, cdf3 == cdf2 and the Fresnel draw never
    // happens.  The clearcoat branch stays whatever the scene holds: the four probabilities
    // can add up to a float just below one, and a draw of pr at or above it lands there.
    constexpr bool kGlass = feat<F>(kFeatGlass);
    f3 const v = lm.v;
    float const cdf0 = lm.diffuse_pr;
    float const cdf1 = cdf0 + lm.dielectric_pr;
    float const cdf2 = cdf1 + lm.metal_pr;
    float const cdf3 = cdf2 + lm.glass_pr;
    float const pr = rand_01(seed);
    // every lobe's sampler draws exactly two more numbers before anything else
    float const r1 = rand_01(seed);
    float const r2 = rand_01(seed);
    disk_point disk{0.0f, 0.0f};
    if (pr < cdf3) {
        disk = sample_disk(r1, r2);
    }
    f3 l;
    if (pr < cdf0) {
        l = cosine_sample_hemisphere(disk);
    } else if (pr < cdf3) {
        f3 h = sample_ggx_vndf(disk, v, si.ax, si.ay);
        if (h.z < 0.0f) {
            h = -h;
        }
        bool do_reflect = true;
        if (kGlass && !(pr < cdf2)) {
            float const fr = dielectric_fresnel(fabsf(dot3(v, h)), si.eta);
            float const pr_f = rand_01(seed);
            do_reflect = pr_f < fr;
        }
        l = do_reflect ? normalize3(reflect3(-v, h)) : normalize3(refract3(-v, h, si.eta));
    } else {
        f3 h = sample_gtr1(r1, r2, si.clearcoat_gloss);
        if (h.z < 0.0f) {
            h = -h;
        }
        l = normalize3(reflect3(-v, h));
    }
    return (lm.t * l.x + lm.b * l.y) + st.normal * l.z;
}

// ---------------------------------------------------------------------------
// eval_sky_light, megakernel_raytracer.comp:413-438.  Returns (radiance, pdf).
template <uint32_t F = kShadeAll>
RT_DEV f4 eval_sky_light(scene_dev const& sc, f3 direction) {
    if (!feat<F>(kFeatSky) || sc.sky_light < 0) {
        return f4{0.0f, 0.0f, 0.0f, 0.0f};
    }
    rt_light const& sky = sc.lights[sc.sky_light];
    f3 const intensity = mk3(sky.intensity[0], sky.intensity[1], sky.intensity[2]);
    if (!feat<F>(kFeatTextures) || sky.emission_tex < 0) {
        return f4{intensity.x, intensity.y, intensity.z, kOneOverFourPi};
    }
    float const theta = acosf(clamp1(direction.y, -1.0f, 1.0f));
    float const u = (kPi + atan2f(direction.z, direction.x)) * kOneOverTwoPi;
    float const v = theta * kOneOverPi;
    f3 const texel = texture_rgb(sc, sky.emission_tex, u, v);
    f3 const mapped = texel / (texel + splat3(1.0f));  // to_ldr.glsl:1-3
    float pdf;
    if (sc.sky_density != nullptr) {
        // opt-in: the density the direction is actually drawn with (sample_sky_importance)
        uint32_t const x = min(sc.sky_w - 1u, (uint32_t) (u * (float) sc.sky_w));
        uint32_t const y = min(sc.sky_h - 1u, (uint32_t) (v * (float) sc.sky_h));
        float const sin_theta = sinf(theta);
        pdf = sin_theta > 0.0f ? sc.sky_density[(size_t) y * sc.sky_w + x] / (kTwoPiPi * sin_theta)
                               : 0.0f;
    } else {
        float const padded_sin_theta = pad_above_zero(sinf(theta));
        pdf = ((luminance(intensity) / sky.direction[0]) * sky.direction[1] * sky.direction[2]) /
              (kTwoPiPi * padded_sin_theta);
    }
    return f4{intensity.x * mapped.x, intensity.y * mapped.y, intensity.z * mapped.z, pdf};
}

// Opt-in replacement of the reference's uniform sky sampling (no counterpart in the shader,
// whose sample_light draws a uniform direction, megakernel_raytracer.comp:447-457): inverts
// the piecewise-constant distribution over the map's texels (luminance of the tone-mapped,
// intensity-scaled texel x sin(theta)), the marginal over rows first, then the row's
// conditional, with the remainder of each random number placing the sample inside the texel.
// Two rand_01 draws, like the sampling it replaces.
RT_DEV float sample_cdf(const float* cdf, uint32_t n, float r) {
    // largest i in [0, n - 1] with cdf[i] <= r   (cdf has n + 1 entries, cdf[0] = 0, cdf[n] = 1)
    uint32_t lo = 0, hi = n;
    while (hi - lo > 1u) {
        uint32_t const mid = (lo + hi) >> 1;
        if (cdf[mid] <= r) {
            lo = mid;
        } else {
            hi = mid;
        }
    }
    float const width = cdf[lo + 1u] - cdf[lo];
    float within = r - cdf[lo];
    if (width > 0.0f) {
        within = within / width;
    }
    return ((float) lo + fminf(within, 1.0f)) / (float) n;
}

RT_DEV f3 sample_sky_importance(scene_dev const& sc, uint32_t& seed) {
    float const r0 = rand_01(seed);
    float const r1 = rand_01(seed);
    float const v = sample_cdf(sc.sky_marginal, sc.sky_h, r0);
    uint32_t const row = min(sc.sky_h - 1u, (uint32_t) (v * (float) sc.sky_h));
    float const u = sample_cdf(sc.sky_conditional + (size_t) row * (sc.sky_w + 1u), sc.sky_w, r1);
    float const theta = v * kPi;
    float const phi = u * kTwoPi - kPi;
    float const sin_theta = sinf(theta);
    return mk3(sin_theta * cosf(phi), cosf(theta), sin_theta * sinf(phi));
}

}  // namespace rtd

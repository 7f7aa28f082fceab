// Reference-order traversal: walks the uploaded 40-byte TLAS/BLAS nodes in the
// order of the reference shader so that the winning hit - including which of
// two equal-t hits wins - is the one the reference algorithm reports.
//
// shaders/raytracer/megakernel_raytracer.comp:184-276 (closest_hit) and
// :325-411 (any_hit) first collect every instance whose TLAS leaf box is hit
// into a 15-entry list and then walk the instances' BLASes in list order.
// The TLAS walk tests boxes against the ray's *initial* interval (it never
// sees the shrinking t_max), so visiting each instance at the moment its leaf
// is reached, while still testing TLAS boxes against the initial interval,
// visits the same instances in the same order without the list and its
// 15-entry limit.  One stack serves both levels: the BLAS part grows on top
// of the suspended TLAS part.
#pragma once

#include "rt_device_scene.h"

namespace rtd {

struct hit_result {
    uint32_t instance;  // index into the uploaded instances[]
    uint32_t triangle;  // index into the uploaded triangles[]
    float b0, b1, b2, t;
};

struct trace_counters {
    uint32_t nodes = 0;
    uint32_t tris = 0;
    uint32_t max_stack = 0;
    // rebuilt structure only: how the fetches were served
    uint32_t lds_nodes = 0;      // of `nodes`: read from the workgroup's LDS copy
    uint32_t instances = 0;      // instance records fetched (TLAS leaves entered)
    uint32_t lds_instances = 0;  // of `instances`: read from LDS
    // trace launches only: how often a WAVE executed each kind of step (counted by the first
    // lane taking part), so that steps / (64 x wave steps) is the share of lanes in use
    uint32_t node_wave_steps = 0;
    uint32_t tri_wave_steps = 0;
    uint32_t instance_wave_steps = 0;
};

constexpr uint32_t kRefStackCapacity = 160;

// shaders/raytracer/ray.glsl:11-31.  The per-axis early exit of the shader is
// equivalent to one final comparison: t_min never decreases and t_max never
// increases, so a failed test on one axis stays failed.
RT_DEV bool hit_aabb_ref(rt_bvh_node const& n, f3 origin, f3 inv_d, bool neg_x, bool neg_y,
    bool neg_z, float t_min, float t_max) {
    float const x0 = ((neg_x ? n.x_max : n.x_min) - origin.x) * inv_d.x;
    float const x1 = ((neg_x ? n.x_min : n.x_max) - origin.x) * inv_d.x;
    t_min = fmaxf(x0, t_min);
    t_max = fminf(x1, t_max);
    float const y0 = ((neg_y ? n.y_max : n.y_min) - origin.y) * inv_d.y;
    float const y1 = ((neg_y ? n.y_min : n.y_max) - origin.y) * inv_d.y;
    t_min = fmaxf(y0, t_min);
    t_max = fminf(y1, t_max);
    float const z0 = ((neg_z ? n.z_max : n.z_min) - origin.z) * inv_d.z;
    float const z1 = ((neg_z ? n.z_min : n.z_max) - origin.z) * inv_d.z;
    t_min = fmaxf(z0, t_min);
    t_max = fminf(z1, t_max);
    return !(t_min > t_max);
}

// shaders/raytracer/ray.glsl:40-45
RT_DEV int max_component_index(f3 v) {
    float const ax = fabsf(v.x), ay = fabsf(v.y), az = fabsf(v.z);
    return ax > ay ? (ax > az ? 0 : 2) : (ay > az ? 1 : 2);
}

// Ray constants of the watertight test that do not depend on the triangle
// (ray.glsl:137-147): the axis permutation and the shear.
struct tri_ray {
    f3 origin;
    int kx, ky, kz;
    float sx, sy, sz;
};

RT_DEV tri_ray make_tri_ray(f3 origin, f3 direction) {
    tri_ray r;
    r.origin = origin;
    r.kz = max_component_index(direction);
    r.kx = r.kz + 1 == 3 ? 0 : r.kz + 1;
    r.ky = r.kx + 1 == 3 ? 0 : r.kx + 1;
    float const dx = comp3(direction, r.kx);
    float const dy = comp3(direction, r.ky);
    float const dz = comp3(direction, r.kz);
    r.sx = -dx / dz;
    r.sy = -dy / dz;
    r.sz = 1.0f / dz;
    return r;
}

// shaders/raytracer/ray.glsl:119-183 (and :47-105, which is the same test
// without the outputs).  Accepts 0 < t <= t_max, so a later hit at the same t
// replaces an earlier one.
// check_degenerate = false: the caller knows the triangle passes the shader's first test
// (|cross(edge1, edge2)| >= 1e-8, which does not depend on the ray): the rebuilt structure
// leaves the triangles that fail it out of its leaves when it is built.
// The first test of hit_triangle (ray.glsl:124-128): a triangle whose edge cross product is
// shorter than 1e-8 is never hit, whatever the ray.  The rebuilt structure evaluates the same
// test once per triangle on the host (accel_build.cpp: triangle_never_hit) and leaves such
// triangles out of its leaves.
RT_DEV bool degenerate_triangle(f3 a, f3 b, f3 c) {
    f3 const edge1 = b - a;
    f3 const edge2 = c - a;
    return fabsf(length3(cross3(edge1, edge2))) < 1e-8f;
}

RT_DEV bool hit_triangle(f3 a, f3 b, f3 c, tri_ray const& ray, float t_max, float& b0,
    float& b1, float& b2, float& t, bool check_degenerate = true) {
    if (check_degenerate && degenerate_triangle(a, b, c)) {
        return false;
    }
    f3 const at0 = a - ray.origin;
    f3 const bt0 = b - ray.origin;
    f3 const ct0 = c - ray.origin;
    float atx = comp3(at0, ray.kx), aty = comp3(at0, ray.ky), atz = comp3(at0, ray.kz);
    float btx = comp3(bt0, ray.kx), bty = comp3(bt0, ray.ky), btz = comp3(bt0, ray.kz);
    float ctx = comp3(ct0, ray.kx), cty = comp3(ct0, ray.ky), ctz = comp3(ct0, ray.kz);
    atx += ray.sx * atz;
    aty += ray.sy * atz;
    btx += ray.sx * btz;
    bty += ray.sy * btz;
    ctx += ray.sx * ctz;
    cty += ray.sy * ctz;
    float const e0 = btx * cty - bty * ctx;
    float const e1 = ctx * aty - cty * atx;
    float const e2 = atx * bty - aty * btx;
    if ((e0 < 0.0f || e1 < 0.0f || e2 < 0.0f) && (e0 > 0.0f || e1 > 0.0f || e2 > 0.0f)) {
        return false;
    }
    float const det = e0 + e1 + e2;
    if (det == 0.0f) {
        return false;
    }
    atz *= ray.sz;
    btz *= ray.sz;
    ctz *= ray.sz;
    float const t_scaled = (e0 * atz + e1 * btz) + e2 * ctz;
    float const t_min = 0.0f;
    if (det < 0.0f && (t_scaled >= t_min * det || t_scaled < t_max * det)) {
        return false;
    } else if (det > 0.0f && (t_scaled <= t_min * det || t_scaled > t_max * det)) {
        return false;
    }
    float const inv_det = 1.0f / det;
    b0 = e0 * inv_det;
    b1 = e1 * inv_det;
    b2 = e2 * inv_det;
    t = t_scaled * inv_det;
    return true;
}

// World ray -> object space of one instance (megakernel_raytracer.comp:229-238).
// The w = 0 / w = 1 products are kept: (-0) + (+0) = +0 matters for 1/direction.
RT_DEV void to_object_space(scene_dev const& sc, rt_instance const& inst, f3 origin,
    f3 direction, f3& o_out, f3& d_out) {
    if (inst.transform >= 0) {
        rt_mat4 const& m = sc.inverse_transforms[inst.transform];
        o_out.x = ((m.m[0][0] * origin.x + m.m[1][0] * origin.y) + m.m[2][0] * origin.z) +
                  m.m[3][0] * 1.0f;
        o_out.y = ((m.m[0][1] * origin.x + m.m[1][1] * origin.y) + m.m[2][1] * origin.z) +
                  m.m[3][1] * 1.0f;
        o_out.z = ((m.m[0][2] * origin.x + m.m[1][2] * origin.y) + m.m[2][2] * origin.z) +
                  m.m[3][2] * 1.0f;
        d_out.x = ((m.m[0][0] * direction.x + m.m[1][0] * direction.y) +
                      m.m[2][0] * direction.z) +
                  m.m[3][0] * 0.0f;
        d_out.y = ((m.m[0][1] * direction.x + m.m[1][1] * direction.y) +
                      m.m[2][1] * direction.z) +
                  m.m[3][1] * 0.0f;
        d_out.z = ((m.m[0][2] * direction.x + m.m[1][2] * direction.y) +
                      m.m[2][2] * direction.z) +
                  m.m[3][2] * 0.0f;
    } else {
        // mat4(1.0) * vec4: every product is still formed.
        o_out.x = ((1.0f * origin.x + 0.0f * origin.y) + 0.0f * origin.z) + 0.0f * 1.0f;
        o_out.y = ((0.0f * origin.x + 1.0f * origin.y) + 0.0f * origin.z) + 0.0f * 1.0f;
        o_out.z = ((0.0f * origin.x + 0.0f * origin.y) + 1.0f * origin.z) + 0.0f * 1.0f;
        d_out.x = ((1.0f * direction.x + 0.0f * direction.y) + 0.0f * direction.z) + 0.0f * 0.0f;
        d_out.y = ((0.0f * direction.x + 1.0f * direction.y) + 0.0f * direction.z) + 0.0f * 0.0f;
        d_out.z = ((0.0f * direction.x + 0.0f * direction.y) + 1.0f * direction.z) + 0.0f * 0.0f;
    }
}

// ANY = false: closest hit (best is filled, returns whether anything was hit).
// ANY = true : occlusion query over (0, t_limit]; returns on the first hit.
template <bool ANY, bool COUNT>
RT_DEV bool trace_ref(scene_dev const& sc, f3 origin, f3 direction, float t_limit,
    hit_result& best, trace_counters& cnt) {
    uint32_t stack[kRefStackCapacity];
    uint32_t top = 0;
    float t_max = t_limit;
    bool found = false;
    f3 const inv_w = splat3(1.0f) / direction;
    bool const nwx = direction.x < 0.0f, nwy = direction.y < 0.0f, nwz = direction.z < 0.0f;
    uint32_t tlas_node = 0;
    for (;;) {
        rt_bvh_node const node = sc.tlas[tlas_node];
        if (COUNT) {
            ++cnt.nodes;
        }
        bool pop = true;
        if (hit_aabb_ref(node, origin, inv_w, nwx, nwy, nwz, 0.0f, t_limit)) {
            if (node.obj_count > 0) {
                for (uint32_t i = node.first_obj; i < node.first_obj + node.obj_count; ++i) {
                    rt_instance const inst = sc.instances[i];
                    rt_mesh const mesh = sc.meshes[inst.mesh];
                    f3 o, d;
                    to_object_space(sc, inst, origin, direction, o, d);
                    f3 const inv_o = splat3(1.0f) / d;
                    bool const nox = d.x < 0.0f, noy = d.y < 0.0f, noz = d.z < 0.0f;
                    tri_ray const tr = make_tri_ray(o, d);
                    uint32_t const base = top;
                    uint32_t blas_node = mesh.bvh_start;
                    for (;;) {
                        rt_bvh_node const bn = sc.blas[blas_node];
                        if (COUNT) {
                            ++cnt.nodes;
                        }
                        bool bpop = true;
                        if (hit_aabb_ref(bn, o, inv_o, nox, noy, noz, 0.0f, t_max)) {
                            if (bn.obj_count > 0) {
                                for (uint32_t t = bn.first_obj; t < bn.first_obj + bn.obj_count;
                                     ++t) {
                                    rt_triangle const& tri = sc.triangles[t];
                                    f3 const a = mk3(tri.a.position_texu[0],
                                        tri.a.position_texu[1], tri.a.position_texu[2]);
                                    f3 const b = mk3(tri.b.position_texu[0],
                                        tri.b.position_texu[1], tri.b.position_texu[2]);
                                    f3 const c = mk3(tri.c.position_texu[0],
                                        tri.c.position_texu[1], tri.c.position_texu[2]);
                                    if (COUNT) {
                                        ++cnt.tris;
                                    }
                                    float b0, b1, b2, th;
                                    if (hit_triangle(a, b, c, tr, t_max, b0, b1, b2, th)) {
                                        if (ANY) {
                                            return true;
                                        }
                                        t_max = th;
                                        found = true;
                                        best.instance = i;
                                        best.triangle = t;
                                        best.b0 = b0;
                                        best.b1 = b1;
                                        best.b2 = b2;
                                        best.t = th;
                                    }
                                }
                            } else {
                                bool const negative = bn.split_axis == 0 ? nox :
                                                      (bn.split_axis == 1 ? noy : noz);
                                uint32_t const near_child = negative ? bn.right : blas_node + 1;
                                uint32_t const far_child = negative ? blas_node + 1 : bn.right;
                                blas_node = near_child;
                                if (top < kRefStackCapacity) {
                                    stack[top] = far_child;
                                }
                                ++top;
                                if (COUNT) {
                                    cnt.max_stack = max(cnt.max_stack, top);
                                }
                                bpop = false;
                            }
                        }
                        if (bpop) {
                            if (top == base) {
                                break;
                            }
                            --top;
                            blas_node = stack[top < kRefStackCapacity ? top : 0];
                        }
                    }
                }
            } else {
                bool const negative =
                    node.split_axis == 0 ? nwx : (node.split_axis == 1 ? nwy : nwz);
                uint32_t const near_child = negative ? node.right : tlas_node + 1;
                uint32_t const far_child = negative ? tlas_node + 1 : node.right;
                tlas_node = near_child;
                if (top < kRefStackCapacity) {
                    stack[top] = far_child;
                }
                ++top;
                if (COUNT) {
                    cnt.max_stack = max(cnt.max_stack, top);
                }
                pop = false;
            }
        }
        if (pop) {
            if (top == 0) {
                break;
            }
            --top;
            tlas_node = stack[top < kRefStackCapacity ? top : 0];
        }
    }
    return found;
}

}  // namespace rtd

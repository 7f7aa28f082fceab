// Small float vector / matrix helpers for the host layer.
//
// The reference host leans on glm (an un-vendored submodule) for
// translate / quaternion->matrix / scale / inverse and mat4*vec4
// (src/asset/scene.cpp:221-246, src/renderer/megakernel_raytracer.cpp:277-282,
// src/renderer/aabb.cpp:20-38).  These are restatements of the published glm
// formulas with the operation order written out, so host-produced matrices are
// reproducible without glm.
#pragma once

#include <cmath>
#include <cstdint>

namespace rt {

struct vec3 {
    float x = 0.0f, y = 0.0f, z = 0.0f;
    float operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
};

struct vec4 {
    float x = 0.0f, y = 0.0f, z = 0.0f, w = 0.0f;
};

inline vec3 operator+(vec3 a, vec3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline vec3 operator-(vec3 a, vec3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline vec3 operator-(vec3 a) { return {-a.x, -a.y, -a.z}; }
inline vec3 operator*(float s, vec3 a) { return {s * a.x, s * a.y, s * a.z}; }
inline vec3 operator*(vec3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline vec3 operator/(vec3 a, float s) { return {a.x / s, a.y / s, a.z / s}; }
inline vec3 operator/(vec3 a, vec3 b) { return {a.x / b.x, a.y / b.y, a.z / b.z}; }
inline float dot(vec3 a, vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline vec3 cross(vec3 a, vec3 b) {
    return {a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y};
}
inline float length(vec3 a) { return std::sqrt(dot(a, a)); }
inline vec3 normalize(vec3 a) { return a * (1.0f / std::sqrt(dot(a, a))); }

inline vec4 operator+(vec4 a, vec4 b) { return {a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w}; }
inline vec4 operator*(vec4 a, float s) { return {a.x * s, a.y * s, a.z * s, a.w * s}; }

// Column-major 4x4, c[col] is a column; same memory layout as rt_mat4.
struct mat4 {
    vec4 c[4];
    static mat4 identity() {
        mat4 m;
        m.c[0] = {1, 0, 0, 0};
        m.c[1] = {0, 1, 0, 0};
        m.c[2] = {0, 0, 1, 0};
        m.c[3] = {0, 0, 0, 1};
        return m;
    }
};
static_assert(sizeof(mat4) == 64, "mat4 must be 16 tightly packed floats");

// (m0*x + m1*y) + (m2*z + m3*w): glm's pairing for mat4 * vec4.
inline vec4 mul(mat4 const& m, vec4 v) {
    vec4 const a = m.c[0] * v.x + m.c[1] * v.y;
    vec4 const b = m.c[2] * v.z + m.c[3] * v.w;
    return a + b;
}

inline mat4 mul(mat4 const& a, mat4 const& b) {
    mat4 r;
    for (int j = 0; j < 4; ++j) {
        vec4 const col = b.c[j];
        r.c[j] = ((a.c[0] * col.x + a.c[1] * col.y) + a.c[2] * col.z) + a.c[3] * col.w;
    }
    return r;
}

inline mat4 translate(mat4 const& m, vec3 v) {
    mat4 r = m;
    r.c[3] = ((m.c[0] * v.x + m.c[1] * v.y) + m.c[2] * v.z) + m.c[3];
    return r;
}

inline mat4 scale(mat4 const& m, vec3 v) {
    mat4 r;
    r.c[0] = m.c[0] * v.x;
    r.c[1] = m.c[1] * v.y;
    r.c[2] = m.c[2] * v.z;
    r.c[3] = m.c[3];
    return r;
}

// Rotation matrix of a quaternion given as (w, x, y, z).
inline mat4 quat_to_mat4(float w, float x, float y, float z) {
    float const xx = x * x, yy = y * y, zz = z * z;
    float const xz = x * z, xy = x * y, yz = y * z;
    float const wx = w * x, wy = w * y, wz = w * z;
    mat4 r = mat4::identity();
    r.c[0] = {1.0f - 2.0f * (yy + zz), 2.0f * (xy + wz), 2.0f * (xz - wy), 0.0f};
    r.c[1] = {2.0f * (xy - wz), 1.0f - 2.0f * (xx + zz), 2.0f * (yz + wx), 0.0f};
    r.c[2] = {2.0f * (xz + wy), 2.0f * (yz - wx), 1.0f - 2.0f * (xx + yy), 0.0f};
    return r;
}

// General 4x4 inverse by 2x2 sub-determinants (cofactor expansion).
inline mat4 inverse(mat4 const& in) {
    float m[4][4];
    for (int c = 0; c < 4; ++c) {
        m[c][0] = in.c[c].x;
        m[c][1] = in.c[c].y;
        m[c][2] = in.c[c].z;
        m[c][3] = in.c[c].w;
    }
    float const s00 = m[2][2] * m[3][3] - m[3][2] * m[2][3];
    float const s02 = m[1][2] * m[3][3] - m[3][2] * m[1][3];
    float const s03 = m[1][2] * m[2][3] - m[2][2] * m[1][3];
    float const s04 = m[2][1] * m[3][3] - m[3][1] * m[2][3];
    float const s06 = m[1][1] * m[3][3] - m[3][1] * m[1][3];
    float const s07 = m[1][1] * m[2][3] - m[2][1] * m[1][3];
    float const s08 = m[2][1] * m[3][2] - m[3][1] * m[2][2];
    float const s10 = m[1][1] * m[3][2] - m[3][1] * m[1][2];
    float const s11 = m[1][1] * m[2][2] - m[2][1] * m[1][2];
    float const s12 = m[2][0] * m[3][3] - m[3][0] * m[2][3];
    float const s14 = m[1][0] * m[3][3] - m[3][0] * m[1][3];
    float const s15 = m[1][0] * m[2][3] - m[2][0] * m[1][3];
    float const s16 = m[2][0] * m[3][2] - m[3][0] * m[2][2];
    float const s18 = m[1][0] * m[3][2] - m[3][0] * m[1][2];
    float const s19 = m[1][0] * m[2][2] - m[2][0] * m[1][2];
    float const s20 = m[2][0] * m[3][1] - m[3][0] * m[2][1];
    float const s22 = m[1][0] * m[3][1] - m[3][0] * m[1][1];
    float const s23 = m[1][0] * m[2][1] - m[2][0] * m[1][1];

    float inv[4][4];
    inv[0][0] = +(m[1][1] * s00 - m[1][2] * s04 + m[1][3] * s08);
    inv[0][1] = -(m[0][1] * s00 - m[0][2] * s04 + m[0][3] * s08);
    inv[0][2] = +(m[0][1] * s02 - m[0][2] * s06 + m[0][3] * s10);
    inv[0][3] = -(m[0][1] * s03 - m[0][2] * s07 + m[0][3] * s11);
    inv[1][0] = -(m[1][0] * s00 - m[1][2] * s12 + m[1][3] * s16);
    inv[1][1] = +(m[0][0] * s00 - m[0][2] * s12 + m[0][3] * s16);
    inv[1][2] = -(m[0][0] * s02 - m[0][2] * s14 + m[0][3] * s18);
    inv[1][3] = +(m[0][0] * s03 - m[0][2] * s15 + m[0][3] * s19);
    inv[2][0] = +(m[1][0] * s04 - m[1][1] * s12 + m[1][3] * s20);
    inv[2][1] = -(m[0][0] * s04 - m[0][1] * s12 + m[0][3] * s20);
    inv[2][2] = +(m[0][0] * s06 - m[0][1] * s14 + m[0][3] * s22);
    inv[2][3] = -(m[0][0] * s07 - m[0][1] * s15 + m[0][3] * s23);
    inv[3][0] = -(m[1][0] * s08 - m[1][1] * s16 + m[1][2] * s20);
    inv[3][1] = +(m[0][0] * s08 - m[0][1] * s16 + m[0][2] * s20);
    inv[3][2] = -(m[0][0] * s10 - m[0][1] * s18 + m[0][2] * s22);
    inv[3][3] = +(m[0][0] * s11 - m[0][1] * s19 + m[0][2] * s23);

    float const det = (m[0][0] * inv[0][0] + m[0][1] * inv[1][0]) +
                      (m[0][2] * inv[2][0] + m[0][3] * inv[3][0]);
    float const one_over_det = 1.0f / det;
    mat4 r;
    for (int c = 0; c < 4; ++c) {
        r.c[c] = {inv[c][0] * one_over_det, inv[c][1] * one_over_det,
            inv[c][2] * one_over_det, inv[c][3] * one_over_det};
    }
    return r;
}

}  // namespace rt

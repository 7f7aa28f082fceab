// Device-side float3 helpers and shader constants for the gfx950 kernels.
//
// Operation order is fixed and spelled out (this translation unit is built
// with -ffp-contract=off): dot = (x*x' + y*y') + z*z', mat*vec sums its
// columns left to right, normalize multiplies by 1/sqrt(dot).  The CPU oracle
// pins the same choices, which is what makes hit ids and barycentrics
// comparable bit for bit.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#define RT_DEV __device__ __forceinline__

namespace rtd {

// shaders/common/utils.glsl:1-9; constant expressions folded in double, then rounded.
constexpr double kPiD = 3.1415926535;
constexpr float kPi = (float) kPiD;
constexpr float kTwoPi = (float) (2.0 * kPiD);
constexpr float kOneOverPi = (float) (1.0 / kPiD);
constexpr float kOneOverTwoPi = (float) (1.0 / (2.0 * kPiD));
constexpr float kOneOverFourPi = (float) (1.0 / (4.0 * kPiD));
constexpr float kTwoPiPi = (float) (2.0 * kPiD * kPiD);
constexpr float kInfinity = (float) (1.0 / 0.00000000000001);
constexpr float kInfinityMinusEps = (float) ((1.0 / 0.00000000000001) - 0.0001);
constexpr float kEpsilon = 0.0001f;

struct f3 {
    float x, y, z;
};
struct f4 {
    float x, y, z, w;
};

RT_DEV f3 mk3(float x, float y, float z) { return f3{x, y, z}; }
RT_DEV f3 splat3(float s) { return f3{s, s, s}; }
RT_DEV f3 operator+(f3 a, f3 b) { return f3{a.x + b.x, a.y + b.y, a.z + b.z}; }
RT_DEV f3 operator-(f3 a, f3 b) { return f3{a.x - b.x, a.y - b.y, a.z - b.z}; }
RT_DEV f3 operator-(f3 a) { return f3{-a.x, -a.y, -a.z}; }
RT_DEV f3 operator*(f3 a, f3 b) { return f3{a.x * b.x, a.y * b.y, a.z * b.z}; }
RT_DEV f3 operator*(f3 a, float s) { return f3{a.x * s, a.y * s, a.z * s}; }
RT_DEV f3 operator*(float s, f3 a) { return f3{a.x * s, a.y * s, a.z * s}; }
RT_DEV f3 operator/(f3 a, f3 b) { return f3{a.x / b.x, a.y / b.y, a.z / b.z}; }
RT_DEV f3 operator/(f3 a, float s) { return f3{a.x / s, a.y / s, a.z / s}; }
RT_DEV float dot3(f3 a, f3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
RT_DEV f3 cross3(f3 a, f3 b) {
    return f3{a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y};
}
RT_DEV float length3(f3 a) { return sqrtf(dot3(a, a)); }
RT_DEV f3 normalize3(f3 a) { return a * (1.0f / sqrtf(dot3(a, a))); }
RT_DEV float mix1(float x, float y, float a) { return x * (1.0f - a) + y * a; }
RT_DEV f3 mix3(f3 x, f3 y, float a) { return x * (1.0f - a) + y * a; }
RT_DEV float clamp1(float x, float lo, float hi) { return fminf(fmaxf(x, lo), hi); }
RT_DEV float sign1(float x) { return x > 0.0f ? 1.0f : (x < 0.0f ? -1.0f : 0.0f); }
RT_DEV float comp3(f3 a, int i) { return i == 0 ? a.x : (i == 1 ? a.y : a.z); }

// Column-major 3x3 (columns c0..c2).
struct m33 {
    f3 c0, c1, c2;
};
RT_DEV f3 mul33(m33 const& m, f3 v) { return (m.c0 * v.x + m.c1 * v.y) + m.c2 * v.z; }
RT_DEV f3 tmul33(m33 const& m, f3 v) { return f3{dot3(m.c0, v), dot3(m.c1, v), dot3(m.c2, v)}; }
RT_DEV m33 inverse33(m33 const& m) {
    float const a00 = m.c0.x, a01 = m.c0.y, a02 = m.c0.z;
    float const a10 = m.c1.x, a11 = m.c1.y, a12 = m.c1.z;
    float const a20 = m.c2.x, a21 = m.c2.y, a22 = m.c2.z;
    float const det = (a00 * (a11 * a22 - a21 * a12) - a10 * (a01 * a22 - a21 * a02)) +
                      a20 * (a01 * a12 - a11 * a02);
    float const inv_det = 1.0f / det;
    m33 r;
    r.c0.x = (a11 * a22 - a21 * a12) * inv_det;
    r.c1.x = -(a10 * a22 - a20 * a12) * inv_det;
    r.c2.x = (a10 * a21 - a20 * a11) * inv_det;
    r.c0.y = -(a01 * a22 - a21 * a02) * inv_det;
    r.c1.y = (a00 * a22 - a20 * a02) * inv_det;
    r.c2.y = -(a00 * a21 - a20 * a01) * inv_det;
    r.c0.z = (a01 * a12 - a11 * a02) * inv_det;
    r.c1.z = -(a00 * a12 - a10 * a02) * inv_det;
    r.c2.z = (a00 * a11 - a10 * a01) * inv_det;
    return r;
}

// shaders/common/utils.glsl:11-17
RT_DEV float pad_above_zero(float v) { return fabsf(v) < kEpsilon ? sign1(v) * kEpsilon : v; }
RT_DEV float luminance(f3 c) { return (0.212671f * c.x + 0.715160f * c.y) + 0.072169f * c.z; }

// shaders/common/utils.glsl:62-66
RT_DEV void onb(f3 n, f3& t, f3& b) {
    f3 const up = fabsf(n.z) < 0.9999999f ? mk3(0.0f, 0.0f, 1.0f) : mk3(1.0f, 0.0f, 0.0f);
    t = normalize3(cross3(up, n));
    b = cross3(n, t);
}

// shaders/common/random.glsl:4-24
RT_DEV uint32_t rng_step(uint32_t r) { return r * 747796405u + 1u; }
RT_DEV uint32_t hash_combine(uint32_t seed, uint32_t v) {
    return v + 0x9e3779b9u + (seed << 6) + (seed >> 2);
}
RT_DEV uint32_t rand_uint(uint32_t& seed, uint32_t lo, uint32_t hi) {
    seed = rng_step(seed);
    uint32_t const range = hi - lo + 1u;
    return lo + seed % range;
}
RT_DEV float rand_01(uint32_t& seed) {
    seed = rng_step(seed);
    uint32_t word = ((seed >> ((seed >> 28) + 4u)) ^ seed) * 277803737u;
    word = (word >> 22) ^ word;
    return (float) word / 4294967295.0f;
}

}  // namespace rtd

// device_math.h -- f32 device math of the gfx950 ray-transport kernels.
//
// Numeric contract: IEEE f32, no fast-math, compiled with -ffp-contract=off; every
// fused multiply-add is an explicit __builtin_fmaf in a fixed order; '/' and sqrtf are the
// correctly rounded forms (hipcc default -fhip-fp32-correctly-rounded-divide-sqrt); the only
// transcendental of radiance mode is the fixed polynomial sincos_pi4().  Under this contract the
// radiance kernels reproduce the CPU oracle bit for bit; ultrasound mode additionally calls
// sinf/cosf/expf/acosf (ocml), which agree with libm to a few ulp only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define DEV __device__ __forceinline__

// Exact unsigned division by a launch-uniform divisor (Granlund-Montgomery round-up multiplier, the
// branch-free 33-bit form):  n / d == (((n - hi) >> 1) + hi) >> shift  with hi = mulhi(n, magic), for every
// 32-bit n and d >= 2; d == 1 is flagged.  5 VALU instead of the ~20 of a division by a runtime value.
struct FastDiv {
    uint32_t magic, shift, is_one, pad;
};
inline FastDiv make_fastdiv(uint32_t d) {  // host
    FastDiv f = {0, 0, 0, 0};
    if (d <= 1) {
        f.is_one = 1;
        return f;
    }
    uint32_t L = 31;
    while (!(d >> L)) --L;  // floor(log2 d)
    if ((d & (d - 1)) == 0) {
        f.shift = L - 1;  // magic 0: ((n - 0) >> 1) >> (L - 1)
        return f;
    }
    const uint64_t num = 1ull << (32 + L);
    uint64_t m = num / d;
    const uint64_t rem = num % d;
    m += m;
    if (2 * rem >= d) m += 1;
    f.magic = (uint32_t)(m + 1);
    f.shift = L;
    return f;
}
inline uint32_t udiv_fast_host(uint32_t n, FastDiv f) {
    if (f.is_one) return n;
    const uint32_t hi = (uint32_t)(((uint64_t)n * f.magic) >> 32);
    return (((n - hi) >> 1) + hi) >> f.shift;
}
DEV uint32_t udiv_fast(uint32_t n, FastDiv f) {
    const uint32_t hi = __umulhi(n, f.magic);
    const uint32_t q = (((n - hi) >> 1) + hi) >> f.shift;
    return f.is_one ? n : q;
}

struct V3 {
    float x, y, z;
};

DEV V3 v3(float x, float y, float z) { return V3{x, y, z}; }
DEV V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
DEV V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
DEV V3 operator-(V3 a) { return {-a.x, -a.y, -a.z}; }
DEV V3 operator*(V3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
DEV V3 operator*(V3 a, V3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
DEV float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
DEV float dot(V3 a, V3 b) { return fma_(a.x, b.x, fma_(a.y, b.y, a.z * b.z)); }
DEV V3 cross(V3 a, V3 b) {
    return {fma_(a.y, b.z, -(a.z * b.y)), fma_(a.z, b.x, -(a.x * b.z)), fma_(a.x, b.y, -(a.y * b.x))};
}
DEV V3 madd(V3 a, float s, V3 b) { return {fma_(a.x, s, b.x), fma_(a.y, s, b.y), fma_(a.z, s, b.z)}; }
DEV V3 normalize(V3 v) {
    float inv = 1.0f / sqrtf(dot(v, v));
    return v * inv;
}
DEV float max3(V3 v) { return fmaxf(v.x, fmaxf(v.y, v.z)); }

#define K_PI 3.14159265358979323846f
#define K_INV_PI 0.31830988618379067154f
#define K_PI_OVER_4 0.78539816339744830962f
#define K_RAY_EPS (1500.0f / 16777216.0f)
#define K_SHADOW_EPS (15000.0f / 16777216.0f)
#define K_INF __builtin_huge_valf()

// ---- counter-based RNG: pcg4d keyed (a, b, c, seed) ---------------------------------------------
struct F4 {
    float x, y, z, w;
};
DEV float u01(uint32_t x) { return (float)(x >> 8) * (1.0f / 16777216.0f); }
DEV F4 rng4(uint32_t a, uint32_t b, uint32_t c, uint32_t seed) {
    uint32_t x = a * 1664525u + 1013904223u;
    uint32_t y = b * 1664525u + 1013904223u;
    uint32_t z = c * 1664525u + 1013904223u;
    uint32_t w = seed * 1664525u + 1013904223u;
    x += y * w;
    y += z * x;
    z += x * y;
    w += y * z;
    x ^= x >> 16;
    y ^= y >> 16;
    z ^= z >> 16;
    w ^= w >> 16;
    x += y * w;
    y += z * x;
    z += x * y;
    w += y * z;
    return {u01(x), u01(y), u01(z), u01(w)};
}

// ---- sin/cos on [-pi/4, pi/4] -------------------------------------------------------------------
DEV void sincos_pi4(float x, float *s, float *c) {
    float x2 = x * x;
    float ps = fma_(x2, 2.7557319223985893e-06f, -1.9841269841269841e-04f);
    ps = fma_(x2, ps, 8.3333333333333332e-03f);
    ps = fma_(x2, ps, -1.6666666666666666e-01f);
    *s = fma_(x * x2, ps, x);
    float pc = fma_(x2, 2.4801587301587302e-05f, -1.3888888888888889e-03f);
    pc = fma_(x2, pc, 4.1666666666666664e-02f);
    pc = fma_(x2, pc, -0.5f);
    *c = fma_(x2, pc, 1.0f);
}

// Mitsuba warp::square_to_uniform_disk_concentric (CustomBSDF.py:48)
DEV void square_to_disk(float sx, float sy, float *dx, float *dy) {
    float x = fma_(2.0f, sx, -1.0f), y = fma_(2.0f, sy, -1.0f);
    bool is_zero = (x == 0.0f) && (y == 0.0f);
    bool q13 = fabsf(x) < fabsf(y);
    float r = q13 ? y : x, rp = q13 ? x : y;
    float a = is_zero ? 0.0f : K_PI_OVER_4 * (rp / r);
    float s, c;
    sincos_pi4(a, &s, &c);
    *dx = r * (q13 ? s : c);
    *dy = r * (q13 ? c : s);
}
DEV V3 square_to_cosine_hemisphere(float sx, float sy) {
    float dx, dy;
    square_to_disk(sx, sy, &dx, &dy);
    float z2 = 1.0f - fma_(dx, dx, dy * dy);
    return {dx, dy, sqrtf(fmaxf(z2, 0.0f))};
}
DEV V3 square_to_uniform_hemisphere(float sx, float sy) {
    float dx, dy;
    square_to_disk(sx, sy, &dx, &dy);
    float z = 1.0f - fma_(dx, dx, dy * dy);
    float k = sqrtf(z + 1.0f);
    return {dx * k, dy * k, z};
}

// Mitsuba coordinate_system(n) == Frame3f(n) (CustomBSDF.py:32)
struct Frame {
    V3 s, t, n;
};
DEV Frame make_frame(V3 n) {
    float sign = copysignf(1.0f, n.z);
    float a = -1.0f / (sign + n.z);
    float b = n.x * n.y * a;
    Frame f;
    f.s = {fma_(sign * n.x, n.x * a, 1.0f), sign * b, -sign * n.x};
    f.t = {b, fma_(n.y, n.y * a, sign), -n.y};
    f.n = n;
    return f;
}
// Mitsuba SurfaceInteraction::initialize_sh_frame(): s = normalize(dp_du - n (n . dp_du)), t = n x s -- the frame
// si.wi, si.to_local and si.to_world use (CustomBSDF.py:90,165; CustomIntegrator.py:358).  Zero dp_du: coordinate_system(n).
DEV Frame make_sh_frame(V3 n, V3 dp_du) {
    V3 s = madd(n, -dot(n, dp_du), dp_du);
    float l2 = dot(s, s);
    if (!(l2 > 0.0f)) return make_frame(n);
    Frame f;
    f.s = s * (1.0f / sqrtf(l2));
    f.t = cross(n, f.s);
    f.n = n;
    return f;
}
DEV V3 to_local(const Frame &f, V3 v) { return {dot(v, f.s), dot(v, f.t), dot(v, f.n)}; }
DEV V3 to_world(const Frame &f, V3 v) {
    return {fma_(f.s.x, v.x, fma_(f.t.x, v.y, f.n.x * v.z)), fma_(f.s.y, v.x, fma_(f.t.y, v.y, f.n.y * v.z)),
            fma_(f.s.z, v.x, fma_(f.t.z, v.y, f.n.z * v.z))};
}

// Mitsuba SurfaceInteraction::offset_p / spawn_ray (CustomIntegrator.py:324,359)
DEV V3 offset_origin(V3 p, V3 ng, V3 d) {
    float mag = (1.0f + fmaxf(fabsf(p.x), fmaxf(fabsf(p.y), fabsf(p.z)))) * K_RAY_EPS;
    mag = copysignf(mag, dot(ng, d));
    return madd(ng, mag, p);
}

// row-major 3x4 affine transform
DEV V3 xf_point(const float *m, V3 p) {
    return {fma_(m[0], p.x, fma_(m[1], p.y, fma_(m[2], p.z, m[3]))), fma_(m[4], p.x, fma_(m[5], p.y, fma_(m[6], p.z, m[7]))),
            fma_(m[8], p.x, fma_(m[9], p.y, fma_(m[10], p.z, m[11])))};
}
DEV V3 xf_vec(const float *m, V3 v) {
    return {fma_(m[0], v.x, fma_(m[1], v.y, m[2] * v.z)), fma_(m[4], v.x, fma_(m[5], v.y, m[6] * v.z)),
            fma_(m[8], v.x, fma_(m[9], v.y, m[10] * v.z))};
}

DEV float mis_weight(float a, float b) {
    a *= a;
    b *= b;
    float w = a / (a + b);
    return __builtin_isfinite(w) ? w : 0.0f;
}

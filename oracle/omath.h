// omath.h -- scalar f32 math of the CPU oracle (TEST INFRASTRUCTURE, not product code).
//
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use anything under
// oracle/.  The product path (libpbrt_hip.so) never links, includes or calls this.
//
// Numeric contract shared with the HIP kernels (DESIGN.md "Numeric contract"):
//   * IEEE f32, round-to-nearest, no fast-math, no implicit contraction (-ffp-contract=off);
//     every fused multiply-add is spelled fmaf() explicitly, in the order written here;
//   * / and sqrtf are correctly rounded on both sides;
//   * the only transcendental in radiance mode is the fixed polynomial sincos_pi4() below;
//     ultrasound mode additionally uses libm sinf/cosf/expf/acosf (tolerance, not bit-exact).
#pragma once
#include <cmath>
#include <cstdint>

namespace orc {

struct V3 {
    float x, y, z;
};

static inline V3 v3(float x, float y, float z) { return V3{x, y, z}; }
static inline V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
static inline V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
static inline V3 operator-(V3 a) { return {-a.x, -a.y, -a.z}; }
static inline V3 operator*(V3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
static inline V3 operator*(V3 a, V3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
static inline float dot(V3 a, V3 b) { return fmaf(a.x, b.x, fmaf(a.y, b.y, a.z * b.z)); }
static inline V3 cross(V3 a, V3 b) {
    return {fmaf(a.y, b.z, -(a.z * b.y)), fmaf(a.z, b.x, -(a.x * b.z)), fmaf(a.x, b.y, -(a.y * b.x))};
}
// a*s + b
static inline V3 madd(V3 a, float s, V3 b) { return {fmaf(a.x, s, b.x), fmaf(a.y, s, b.y), fmaf(a.z, s, b.z)}; }
static inline V3 normalize(V3 v) {
    float inv = 1.0f / sqrtf(dot(v, v));
    return v * inv;
}
static inline float max3(V3 v) { return fmaxf(v.x, fmaxf(v.y, v.z)); }

static const float kPi = 3.14159265358979323846f;
static const float kInvPi = 0.31830988618379067154f;
static const float kPiOver4 = 0.78539816339744830962f;
static const float kRayEps = 1500.0f / 16777216.0f;      // Mitsuba RayEpsilon<float> = 1500 * 2^-24
static const float kShadowEps = 15000.0f / 16777216.0f;  // Mitsuba ShadowEpsilon = 10 * RayEpsilon
static const float kInf = INFINITY;

// ---- counter-based RNG: pcg4d (Jarzynski & Olano 2020), keyed (a, b, c, seed) ---------------
struct U4 {
    uint32_t x, y, z, w;
};
static inline U4 pcg4d(uint32_t a, uint32_t b, uint32_t c, uint32_t d) {
    U4 v{a, b, c, d};
    v.x = v.x * 1664525u + 1013904223u;
    v.y = v.y * 1664525u + 1013904223u;
    v.z = v.z * 1664525u + 1013904223u;
    v.w = v.w * 1664525u + 1013904223u;
    v.x += v.y * v.w;
    v.y += v.z * v.x;
    v.z += v.x * v.y;
    v.w += v.y * v.z;
    v.x ^= v.x >> 16;
    v.y ^= v.y >> 16;
    v.z ^= v.z >> 16;
    v.w ^= v.w >> 16;
    v.x += v.y * v.w;
    v.y += v.z * v.x;
    v.z += v.x * v.y;
    v.w += v.y * v.z;
    return v;
}
static inline float u01(uint32_t x) { return (float)(x >> 8) * (1.0f / 16777216.0f); }
struct F4 {
    float x, y, z, w;
};
static inline F4 rng4(uint32_t a, uint32_t b, uint32_t c, uint32_t seed) {
    U4 v = pcg4d(a, b, c, seed);
    return {u01(v.x), u01(v.y), u01(v.z), u01(v.w)};
}

// ---- sin/cos on [-pi/4, pi/4]: fixed Taylor polynomials, Horner with fmaf --------------------
static inline void sincos_pi4(float x, float *s, float *c) {
    float x2 = x * x;
    float ps = fmaf(x2, 2.7557319223985893e-06f, -1.9841269841269841e-04f);
    ps = fmaf(x2, ps, 8.3333333333333332e-03f);
    ps = fmaf(x2, ps, -1.6666666666666666e-01f);
    *s = fmaf(x * x2, ps, x);
    float pc = fmaf(x2, 2.4801587301587302e-05f, -1.3888888888888889e-03f);
    pc = fmaf(x2, pc, 4.1666666666666664e-02f);
    pc = fmaf(x2, pc, -0.5f);
    *c = fmaf(x2, pc, 1.0f);
}

// Mitsuba warp::square_to_uniform_disk_concentric (called at CustomBSDF.py:48)
static inline void square_to_disk(float sx, float sy, float *dx, float *dy) {
    float x = fmaf(2.0f, sx, -1.0f), y = fmaf(2.0f, sy, -1.0f);
    bool is_zero = (x == 0.0f) && (y == 0.0f);
    bool q13 = fabsf(x) < fabsf(y);
    float r = q13 ? y : x, rp = q13 ? x : y;
    float a = is_zero ? 0.0f : kPiOver4 * (rp / r);
    float s, c;
    sincos_pi4(a, &s, &c);
    // phi = q13 ? pi/2 - a : a  ->  (cos phi, sin phi) = q13 ? (s, c) : (c, s)
    *dx = r * (q13 ? s : c);
    *dy = r * (q13 ? c : s);
}
// Mitsuba warp::square_to_cosine_hemisphere
static inline V3 square_to_cosine_hemisphere(float sx, float sy) {
    float dx, dy;
    square_to_disk(sx, sy, &dx, &dy);
    float z2 = 1.0f - fmaf(dx, dx, dy * dy);
    return {dx, dy, sqrtf(fmaxf(z2, 0.0f))};
}
// Mitsuba warp::square_to_uniform_hemisphere (UltraSensor.sample_ray)
static inline V3 square_to_uniform_hemisphere(float sx, float sy) {
    float dx, dy;
    square_to_disk(sx, sy, &dx, &dy);
    float z = 1.0f - fmaf(dx, dx, dy * dy);
    float k = sqrtf(z + 1.0f);
    return {dx * k, dy * k, z};
}

// Mitsuba coordinate_system(n) (Duff et al. branchless ONB) == Frame3f(n) (CustomBSDF.py:32)
struct Frame {
    V3 s, t, n;
};
static inline Frame make_frame(V3 n) {
    float sign = copysignf(1.0f, n.z);
    float a = -1.0f / (sign + n.z);
    float b = n.x * n.y * a;
    Frame f;
    f.s = {fmaf(sign * n.x, n.x * a, 1.0f), sign * b, -sign * n.x};
    f.t = {b, fmaf(n.y, n.y * a, sign), -n.y};
    f.n = n;
    return f;
}
// Mitsuba SurfaceInteraction::initialize_sh_frame(): the shading frame of an intersection is NOT coordinate_system(n)
// but s = normalize(dp_du - n (n . dp_du)), t = n x s, with the shape's dp_du (si.wi = sh_frame.to_local(-ray.d),
// read at CustomBSDF.py:90; si.to_local / si.to_world at CustomBSDF.py:165, CustomIntegrator.py:358).
// A zero dp_du (no parameterisation) falls back to coordinate_system(n).
static inline Frame make_sh_frame(V3 n, V3 dp_du) {
    V3 s = madd(n, -dot(n, dp_du), dp_du);
    float l2 = dot(s, s);
    if (!(l2 > 0.0f)) return make_frame(n);
    Frame f;
    f.s = s * (1.0f / sqrtf(l2));
    f.t = cross(n, f.s);
    f.n = n;
    return f;
}
static inline V3 to_local(const Frame &f, V3 v) { return {dot(v, f.s), dot(v, f.t), dot(v, f.n)}; }
static inline V3 to_world(const Frame &f, V3 v) {
    return {fmaf(f.s.x, v.x, fmaf(f.t.x, v.y, f.n.x * v.z)), fmaf(f.s.y, v.x, fmaf(f.t.y, v.y, f.n.y * v.z)),
            fmaf(f.s.z, v.x, fmaf(f.t.z, v.y, f.n.z * v.z))};
}

// Mitsuba SurfaceInteraction::offset_p / spawn_ray (called at CustomIntegrator.py:324,359)
static inline V3 offset_origin(V3 p, V3 ng, V3 d) {
    float mag = (1.0f + fmaxf(fabsf(p.x), fmaxf(fabsf(p.y), fabsf(p.z)))) * kRayEps;
    mag = copysignf(mag, dot(ng, d));
    return madd(ng, mag, p);
}

// row-major 3x4 affine transform
static inline V3 xf_point(const float *m, V3 p) {
    return {fmaf(m[0], p.x, fmaf(m[1], p.y, fmaf(m[2], p.z, m[3]))), fmaf(m[4], p.x, fmaf(m[5], p.y, fmaf(m[6], p.z, m[7]))),
            fmaf(m[8], p.x, fmaf(m[9], p.y, fmaf(m[10], p.z, m[11])))};
}
static inline V3 xf_vec(const float *m, V3 v) {
    return {fmaf(m[0], v.x, fmaf(m[1], v.y, m[2] * v.z)), fmaf(m[4], v.x, fmaf(m[5], v.y, m[6] * v.z)),
            fmaf(m[8], v.x, fmaf(m[9], v.y, m[10] * v.z))};
}

}  // namespace orc

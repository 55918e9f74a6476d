// oracle.cpp -- CPU restatement of the hot path (TEST INFRASTRUCTURE, not product code).
//
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load liboracle.so.
// The product (libpbrt_hip.so + physics-based-ray-tracing_amd/) never links or calls it.
//
// PARITY STATUS: "parity unpinned" at the third-party boundary.  The reference's arithmetic for
// radiance mode lives entirely in Mitsuba 3 / Dr.Jit (un-vendored, un-pinned, absent from the
// container), so radiance mode restates the published Mitsuba-3 algorithms
// (path / direct integrators, diffuse / conductor / dielectric BSDFs, area / point emitters,
// perspective sensor, hdrfilm + tent/box filters).  Ultrasound mode restates the
// reference's OWN arithmetic line by line (CustomIntegrator.py:235-376, CustomBSDF.py:30-175,
// CustomEmmitter.py:30-107, CustomSensor.py:29-59) and is pinned by the known-answer fixtures
// K1-K7 in tests/golden/ (values derived from the reference's formulas and from its only
// importable file, sampling_test.py).
//
// Scalar, one path at a time, brute-force or median-split BVH intersection, f32 arithmetic under
// the numeric contract of omath.h.  Same counter-based RNG keys as the HIP kernels.
#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>

#include "../include/pbrt_hip.h"
#include "omath.h"

using namespace orc;

namespace {

// ------------------------------------------------------------------------------------------------
// Scene
// ------------------------------------------------------------------------------------------------
struct BvhNode {
    float lo[3], hi[3];
    int32_t left;   // internal: index of left child (right = left + 1 ... stored explicitly below)
    int32_t right;  // internal: index of right child; leaf: -1
    int32_t first, count;  // leaf: range in prim_order
};

struct Scene {
    std::vector<pbrt_prim> prims;
    std::vector<pbrt_material> mats;
    std::vector<pbrt_emitter> emitters;
    std::vector<uint32_t> light_prims;
    std::vector<float> light_cdf;
    std::vector<float> vnormals;  // optional [n_prims][9] vertex normals (pbrt_scene_desc.vertex_normals), caller order
    bool use_bvh = false;
    std::vector<pbrt_prim> occ;  // brute-force scenes: primitives that can occlude a segment (find_occluders)
    std::vector<BvhNode> nodes;
    std::vector<uint32_t> order;
};

inline V3 g3(const pbrt_prim &P, int i) { return {P.g[i], P.g[i + 1], P.g[i + 2]}; }

// Analytic cone ([DEFINE] D8: 'cone' of MitsubaScenes/Cone_Box.xml:36-47 = closed unit cone under to_world): the ray in
// object space (world -> object matrix in g, t preserved) against x^2 + y^2 = (1 - z)^2, 0 <= z <= 1, and the base disc.
// flag 0: lateral surface, 1: base disc.
inline bool cone_hit(const pbrt_prim &P, V3 o, V3 d, float tmax, float *t, float *flag) {
    const V3 r0 = g3(P, 0), r1 = g3(P, 4), r2 = g3(P, 8);
    const V3 oo = {dot(r0, o) + P.g[3], dot(r1, o) + P.g[7], dot(r2, o) + P.g[11]};
    const V3 dd = {dot(r0, d), dot(r1, d), dot(r2, d)};
    const float ow = 1.0f - oo.z;
    const float A = fmaf(dd.x, dd.x, fmaf(dd.y, dd.y, -(dd.z * dd.z)));
    const float b = fmaf(oo.x, dd.x, fmaf(oo.y, dd.y, ow * dd.z));
    const float C = fmaf(oo.x, oo.x, fmaf(oo.y, oo.y, -(ow * ow)));
    const float disc = fmaf(b, b, -(A * C));
    float best = tmax;
    float fl = 0.0f;
    bool found = false;
    if (disc >= 0.0f) {
        const float q = -(b + copysignf(sqrtf(disc), b));
        const float ta = q / A, tb = C / q;
        const float za = fmaf(ta, dd.z, oo.z), zb = fmaf(tb, dd.z, oo.z);
        if (ta >= 0.0f && ta <= best && za >= 0.0f && za <= 1.0f) {
            best = ta;
            found = true;
        }
        if (tb >= 0.0f && tb <= best && zb >= 0.0f && zb <= 1.0f && (!found || tb < best)) {
            best = tb;
            found = true;
        }
    }
    const float tc = -oo.z / dd.z;
    const float x = fmaf(tc, dd.x, oo.x), y = fmaf(tc, dd.y, oo.y);
    if (tc >= 0.0f && tc <= best && fmaf(x, x, y * y) <= 1.0f && (!found || tc < best)) {
        best = tc;
        fl = 1.0f;
        found = true;
    }
    *t = best;
    *flag = fl;
    return found;
}

// Closest-hit candidate test for one primitive.  Mitsuba: Mesh::ray_intersect_triangle,
// Sphere::ray_intersect_preliminary, Rectangle::ray_intersect_preliminary (all reached through
// scene.ray_intersect, CustomIntegrator.py:309).  Triangles and parallelograms share the
// Moeller-Trumbore set-up; the barycentric acceptance test is done on det-scaled values so the
// division only happens for accepted candidates (DESIGN.md "Intersection").
inline bool prim_hit(const pbrt_prim &P, V3 o, V3 d, float tmax, float *t, float *u, float *v) {
    if (P.type == PBRT_PRIM_SPHERE) {
        V3 c = g3(P, 0);
        float r = P.g[3];
        V3 f = o - c;
        float bp = -dot(f, d);
        V3 perp = madd(d, bp, f);
        float disc = fmaf(r, r, -dot(perp, perp));
        if (!(disc >= 0.0f)) return false;
        float sq = sqrtf(disc);
        float q = bp + copysignf(sq, bp);
        float cc = fmaf(-r, r, dot(f, f));
        float t0 = cc / q, t1 = q;
        float tn = fminf(t0, t1), tf = fmaxf(t0, t1);
        if (!(tn <= tmax && tf >= 0.0f)) return false;
        if (tn < 0.0f && tf > tmax) return false;
        *t = tn < 0.0f ? tf : tn;
        *u = 0.0f;
        *v = 0.0f;
        return true;
    }
    if (P.type == PBRT_PRIM_TRIANGLE || P.type == PBRT_PRIM_PARALLELOGRAM) {
        V3 v0 = g3(P, 0), e1 = g3(P, 3), e2 = g3(P, 6);
        V3 pvec = cross(d, e2);
        float det = dot(e1, pvec);
        V3 tvec = o - v0;
        V3 qvec = cross(tvec, e1);
        float us = dot(tvec, pvec), vs = dot(d, qvec), ts = dot(e2, qvec);
        if (det < 0.0f) {
            det = -det;
            us = -us;
            vs = -vs;
            ts = -ts;
        }
        bool ok = det > 0.0f && us >= 0.0f && vs >= 0.0f && ts >= 0.0f;
        if (P.type == PBRT_PRIM_TRIANGLE)
            ok = ok && (us + vs <= det);
        else
            ok = ok && (us <= det) && (vs <= det);
        if (!ok) return false;
        float inv = 1.0f / det;
        float tt = ts * inv;
        if (!(tt <= tmax)) return false;
        *t = tt;
        *u = us * inv;
        *v = vs * inv;
        return true;
    }
    if (P.type == PBRT_PRIM_CONE) {
        *v = 0.0f;
        return cone_hit(P, o, d, tmax, t, u);
    }
    return false;  // unknown primitive type
}

// Brute-force scenes (<= 32 primitives, walked in index order by every ray): candidate hit as the
// ratio t = num / den (den > 0), barycentrics scaled by den.  No division per candidate: the range
// test is ts <= tmax * det, candidates are ranked by cross-multiplication (closest_hit below) and
// the one division happens after the loop.  Spheres report (t, 1).  DESIGN.md "Intersection".
inline bool prim_candidate(const pbrt_prim &P, V3 o, V3 d, float tmax, float *num, float *den, float *us_, float *vs_) {
    if (P.type == PBRT_PRIM_SPHERE || P.type == PBRT_PRIM_CONE) {
        float t, u, v;
        if (!prim_hit(P, o, d, tmax, &t, &u, &v)) return false;
        *num = t;
        *den = 1.0f;
        *us_ = u;  // 0 for spheres; cone: lateral / base flag
        *vs_ = 0.0f;
        return true;
    }
    if (P.type == PBRT_PRIM_TRIANGLE || P.type == PBRT_PRIM_PARALLELOGRAM) {
        V3 v0 = g3(P, 0), e1 = g3(P, 3), e2 = g3(P, 6);
        V3 pvec = cross(d, e2);
        float det = dot(e1, pvec);
        V3 tvec = o - v0;
        V3 qvec = cross(tvec, e1);
        float us = dot(tvec, pvec), vs = dot(d, qvec), ts = dot(e2, qvec);
        if (det < 0.0f) {
            det = -det;
            us = -us;
            vs = -vs;
            ts = -ts;
        }
        bool ok = det > 0.0f && us >= 0.0f && vs >= 0.0f && ts >= 0.0f && ts <= tmax * det;
        if (P.type == PBRT_PRIM_TRIANGLE)
            ok = ok && (us + vs <= det);
        else
            ok = ok && (us <= det) && (vs <= det);
        if (!ok) return false;
        *num = ts;
        *den = det;
        *us_ = us;
        *vs_ = vs;
        return true;
    }
    return false;
}

inline bool box_hit(const BvhNode &n, V3 o, V3 inv_d, float tbest) {
    // the slabs are widened a little: a ray with a zero direction component whose origin lies exactly in a box
    // face (0 * inf = NaN) must not be culled -- e.g. the probe axis through the centre vertex of a cone's base fan
    const float px = 1e-6f * (1.0f + fmaxf(fabsf(n.lo[0]), fabsf(n.hi[0])));
    const float py = 1e-6f * (1.0f + fmaxf(fabsf(n.lo[1]), fabsf(n.hi[1])));
    const float pz = 1e-6f * (1.0f + fmaxf(fabsf(n.lo[2]), fabsf(n.hi[2])));
    float tx0 = (n.lo[0] - px - o.x) * inv_d.x, tx1 = (n.hi[0] + px - o.x) * inv_d.x;
    float ty0 = (n.lo[1] - py - o.y) * inv_d.y, ty1 = (n.hi[1] + py - o.y) * inv_d.y;
    float tz0 = (n.lo[2] - pz - o.z) * inv_d.z, tz1 = (n.hi[2] + pz - o.z) * inv_d.z;
    float tn = fmaxf(fmaxf(fminf(tx0, tx1), fminf(ty0, ty1)), fmaxf(fminf(tz0, tz1), 0.0f));
    float tf = fminf(fminf(fmaxf(tx0, tx1), fmaxf(ty0, ty1)), fminf(fmaxf(tz0, tz1), tbest));
    // conservative slack so that the box test can never cull a primitive hit the brute-force loop
    // would report (the primitive test, not the box, decides)
    return tn <= tf * 1.0000004f + 1e-30f;
}

struct Hit {
    float t, u, v;
    uint32_t prim;
};

bool closest_hit(const Scene &sc, V3 o, V3 d, float tmax, Hit *h) {
    h->t = tmax;
    h->prim = 0xffffffffu;
    h->u = h->v = 0.0f;
    bool found = false;
    if (!sc.use_bvh) {
        float bn = 0.0f, bd = 1.0f, bu = 0.0f, bv = 0.0f;
        for (uint32_t i = 0; i < sc.prims.size(); ++i) {
            float num, den, us, vs;
            if (prim_candidate(sc.prims[i], o, d, tmax, &num, &den, &us, &vs) && (!found || num * bd < bn * den)) {
                bn = num;
                bd = den;
                bu = us;
                bv = vs;
                h->prim = i;
                found = true;
            }
        }
        if (found) {
            float inv = 1.0f / bd;
            h->t = bn * inv;
            h->u = bu * inv;
            h->v = bv * inv;
        }
        return found;
    }
    V3 inv_d = {1.0f / d.x, 1.0f / d.y, 1.0f / d.z};
    int stack[64], sp = 0;
    stack[sp++] = 0;
    while (sp) {
        const BvhNode &n = sc.nodes[stack[--sp]];
        if (!box_hit(n, o, inv_d, h->t)) continue;
        if (n.right < 0) {
            for (int k = 0; k < n.count; ++k) {
                uint32_t i = sc.order[n.first + k];
                float t, u, v;
                if (prim_hit(sc.prims[i], o, d, h->t, &t, &u, &v) &&
                    (!found || t < h->t || (t == h->t && i < h->prim))) {
                    *h = {t, u, v, i};
                    found = true;
                }
            }
        } else {
            stack[sp++] = n.left;
            stack[sp++] = n.right;
        }
    }
    return found;
}

bool any_hit(const Scene &sc, V3 o, V3 d, float tmax) {
    if (!sc.use_bvh) {
        for (uint32_t i = 0; i < sc.prims.size(); ++i) {
            float num, den, us, vs;
            if (prim_candidate(sc.prims[i], o, d, tmax, &num, &den, &us, &vs)) return true;
        }
        return false;
    }
    V3 inv_d = {1.0f / d.x, 1.0f / d.y, 1.0f / d.z};
    int stack[64], sp = 0;
    stack[sp++] = 0;
    while (sp) {
        const BvhNode &n = sc.nodes[stack[--sp]];
        if (!box_hit(n, o, inv_d, tmax)) continue;
        if (n.right < 0) {
            for (int k = 0; k < n.count; ++k) {
                float t, u, v;
                if (prim_hit(sc.prims[sc.order[n.first + k]], o, d, tmax, &t, &u, &v)) return true;
            }
        } else {
            stack[sp++] = n.left;
            stack[sp++] = n.right;
        }
    }
    return false;
}

// Next-event shadow rays are SEGMENTS between two points of the scene.  In brute-force scenes they are
// tested against Scene::occ only: every primitive except the planar ones that lie on a supporting plane of
// the scene's convex hull (all other geometry and every point emitter in one closed half-space of their
// plane) -- a segment whose end points are in the hull cannot cross those (DESIGN.md "Intersection").
// Unbounded occlusion rays (ultrasound mode, CustomIntegrator.py:324) always test every primitive.
bool any_hit_segment(const Scene &sc, V3 o, V3 d, float tmax, bool full_list = false) {
    if (sc.use_bvh || full_list) return any_hit(sc, o, d, tmax);  // full_list: PBRT_FILM_NO_OCCLUDER_PRUNING
    for (const pbrt_prim &P : sc.occ) {
        float num, den, us, vs;
        if (prim_candidate(P, o, d, tmax, &num, &den, &us, &vs)) return true;
    }
    return false;
}

// CONE primitives keep the world -> object matrix only; bounds and hull tests need the world-space base ellipse
// (centre c, conjugate radii a, b) and the apex: solve M x = e - t by Cramer's rule in f64.
static bool cone_frame(const pbrt_prim &P, double c[3], double a[3], double b[3], double apex[3]) {
    const double M[3][3] = {{P.g[0], P.g[1], P.g[2]}, {P.g[4], P.g[5], P.g[6]}, {P.g[8], P.g[9], P.g[10]}};
    auto det3 = [](const double m[3][3]) {
        return m[0][0] * (m[1][1] * m[2][2] - m[1][2] * m[2][1]) - m[0][1] * (m[1][0] * m[2][2] - m[1][2] * m[2][0]) +
               m[0][2] * (m[1][0] * m[2][1] - m[1][1] * m[2][0]);
    };
    const double D = det3(M);
    if (!std::isfinite(D) || D == 0.0) return false;
    auto solve = [&](const double rhs[3], double x[3]) {
        for (int k = 0; k < 3; ++k) {
            double m[3][3];
            for (int i = 0; i < 3; ++i)
                for (int j = 0; j < 3; ++j) m[i][j] = (j == k) ? rhs[i] : M[i][j];
            x[k] = det3(m) / D;
        }
    };
    const double t[3] = {P.g[3], P.g[7], P.g[11]};
    const double r0[3] = {-t[0], -t[1], -t[2]}, rx[3] = {1 - t[0], -t[1], -t[2]}, ry[3] = {-t[0], 1 - t[1], -t[2]},
                 rz[3] = {-t[0], -t[1], 1 - t[2]};
    double px[3], py[3];
    solve(r0, c);
    solve(rx, px);
    solve(ry, py);
    solve(rz, apex);
    for (int k = 0; k < 3; ++k) {
        a[k] = px[k] - c[k];
        b[k] = py[k] - c[k];
        if (!std::isfinite(c[k]) || !std::isfinite(a[k]) || !std::isfinite(b[k]) || !std::isfinite(apex[k])) return false;
    }
    return true;
}

void find_occluders(Scene &sc) {
    sc.occ.clear();
    for (size_t i = 0; i < sc.prims.size(); ++i) {
        const pbrt_prim &P = sc.prims[i];
        bool keep = true;
        if (P.type == PBRT_PRIM_TRIANGLE || P.type == PBRT_PRIM_PARALLELOGRAM) {
            const double n[3] = {P.g[9], P.g[10], P.g[11]}, p0[3] = {P.g[0], P.g[1], P.g[2]};
            double lo = INFINITY, hi = -INFINITY;
            auto side = [&](double x, double y, double z) { return n[0] * (x - p0[0]) + n[1] * (y - p0[1]) + n[2] * (z - p0[2]); };
            auto acc = [&](double s) {
                lo = std::min(lo, s);
                hi = std::max(hi, s);
            };
            for (size_t j = 0; j < sc.prims.size(); ++j) {
                if (j == i) continue;
                const pbrt_prim &Q = sc.prims[j];
                if (Q.type == PBRT_PRIM_SPHERE) {
                    double s = side(Q.g[0], Q.g[1], Q.g[2]);
                    acc(s - (double)Q.g[3]);
                    acc(s + (double)Q.g[3]);
                } else if (Q.type == PBRT_PRIM_CONE) {
                    double cc[3], ca[3], cb[3], cx[3];
                    cone_frame(Q, cc, ca, cb, cx);
                    const double s = side(cc[0], cc[1], cc[2]);
                    const double an = n[0] * ca[0] + n[1] * ca[1] + n[2] * ca[2], bn = n[0] * cb[0] + n[1] * cb[1] + n[2] * cb[2];
                    const double r = std::sqrt(an * an + bn * bn) * (1.0 + 1e-12);
                    acc(s - r);
                    acc(s + r);
                    acc(side(cx[0], cx[1], cx[2]));
                } else {
                    const double v0[3] = {Q.g[0], Q.g[1], Q.g[2]}, e1[3] = {Q.g[3], Q.g[4], Q.g[5]}, e2[3] = {Q.g[6], Q.g[7], Q.g[8]};
                    acc(side(v0[0], v0[1], v0[2]));
                    acc(side(v0[0] + e1[0], v0[1] + e1[1], v0[2] + e1[2]));
                    acc(side(v0[0] + e2[0], v0[1] + e2[1], v0[2] + e2[2]));
                    if (Q.type == PBRT_PRIM_PARALLELOGRAM)
                        acc(side(v0[0] + e1[0] + e2[0], v0[1] + e1[1] + e2[1], v0[2] + e1[2] + e2[2]));
                }
            }
            for (const pbrt_emitter &E : sc.emitters)
                if (E.type == PBRT_EMIT_POINT) acc(side(E.pos[0], E.pos[1], E.pos[2]));
            if (lo >= 0.0 || hi <= 0.0) keep = false;
            // a scene lit by ONE single-primitive area light: every shadow segment ends (1 - ShadowEpsilon)
            // short of that primitive's plane, so the light itself never occludes
            if (sc.emitters.size() == 1 && sc.emitters[0].type == PBRT_EMIT_AREA && sc.emitters[0].count == 1 &&
                sc.light_prims[sc.emitters[0].first] == i)
                keep = false;
        }
        if (keep) sc.occ.push_back(P);
    }
}

void prim_bounds(const pbrt_prim &P, float lo[3], float hi[3]) {
    if (P.type == PBRT_PRIM_SPHERE) {
        for (int k = 0; k < 3; ++k) {
            lo[k] = P.g[k] - P.g[3];
            hi[k] = P.g[k] + P.g[3];
        }
        return;
    }
    if (P.type == PBRT_PRIM_CONE) {
        double cc[3], ca[3], cb[3], cx[3];
        cone_frame(P, cc, ca, cb, cx);
        for (int k = 0; k < 3; ++k) {
            const double r = std::sqrt(ca[k] * ca[k] + cb[k] * cb[k]);
            lo[k] = std::nextafter((float)std::min(cc[k] - r, cx[k]), -kInf);
            hi[k] = std::nextafter((float)std::max(cc[k] + r, cx[k]), kInf);
        }
        return;
    }
    for (int k = 0; k < 3; ++k) {
        float a = P.g[k], b = P.g[k] + P.g[3 + k], c = P.g[k] + P.g[6 + k];
        float d = (P.type == PBRT_PRIM_PARALLELOGRAM) ? P.g[k] + P.g[3 + k] + P.g[6 + k] : a;
        lo[k] = std::min(std::min(a, b), std::min(c, d));
        hi[k] = std::max(std::max(a, b), std::max(c, d));
    }
}

int build_node(Scene &sc, std::vector<float> &cent, int first, int count) {
    int idx = (int)sc.nodes.size();
    sc.nodes.push_back(BvhNode{});
    float lo[3] = {kInf, kInf, kInf}, hi[3] = {-kInf, -kInf, -kInf};
    float clo[3] = {kInf, kInf, kInf}, chi[3] = {-kInf, -kInf, -kInf};
    for (int k = 0; k < count; ++k) {
        uint32_t i = sc.order[first + k];
        float a[3], b[3];
        prim_bounds(sc.prims[i], a, b);
        for (int c = 0; c < 3; ++c) {
            lo[c] = std::min(lo[c], a[c]);
            hi[c] = std::max(hi[c], b[c]);
            clo[c] = std::min(clo[c], cent[3 * i + c]);
            chi[c] = std::max(chi[c], cent[3 * i + c]);
        }
    }
    for (int c = 0; c < 3; ++c) {
        sc.nodes[idx].lo[c] = lo[c];
        sc.nodes[idx].hi[c] = hi[c];
    }
    int axis = 0;
    for (int c = 1; c < 3; ++c)
        if (chi[c] - clo[c] > chi[axis] - clo[axis]) axis = c;
    if (count <= 4 || !(chi[axis] > clo[axis])) {
        sc.nodes[idx].left = sc.nodes[idx].right = -1;
        sc.nodes[idx].first = first;
        sc.nodes[idx].count = count;
        return idx;
    }
    int mid = count / 2;
    std::nth_element(sc.order.begin() + first, sc.order.begin() + first + mid, sc.order.begin() + first + count,
                     [&](uint32_t a, uint32_t b) { return cent[3 * a + axis] < cent[3 * b + axis]; });
    int l = build_node(sc, cent, first, mid);
    int r = build_node(sc, cent, first + mid, count - mid);
    sc.nodes[idx].left = l;
    sc.nodes[idx].right = r;
    sc.nodes[idx].first = sc.nodes[idx].count = 0;
    return idx;
}

void build_bvh(Scene &sc) {
    size_t n = sc.prims.size();
    sc.order.resize(n);
    std::vector<float> cent(3 * n);
    for (size_t i = 0; i < n; ++i) {
        sc.order[i] = (uint32_t)i;
        float a[3], b[3];
        prim_bounds(sc.prims[i], a, b);
        for (int c = 0; c < 3; ++c) cent[3 * i + c] = 0.5f * (a[c] + b[c]);
    }
    sc.nodes.clear();
    build_node(sc, cent, 0, (int)n);
}

// ------------------------------------------------------------------------------------------------
// Surface interaction
// ------------------------------------------------------------------------------------------------
struct SI {
    V3 p, n;  // hit point, geometric normal (rays are offset along it)
    V3 ns;    // shading normal si.sh_frame.n: interpolated vertex normals where the mesh has them, else n
    const pbrt_prim *prim;
};

inline SI make_si(const Scene &sc, V3 o, V3 d, const Hit &h) {
    const pbrt_prim &P = sc.prims[h.prim];
    SI si;
    si.prim = &P;
    if (P.type == PBRT_PRIM_SPHERE) {
        V3 c = g3(P, 0);
        V3 p = madd(d, h.t, o);
        si.n = normalize(p - c);
        si.p = madd(si.n, P.g[3], c);
    } else if (P.type == PBRT_PRIM_CONE) {
        const V3 r0 = g3(P, 0), r1 = g3(P, 4), r2 = g3(P, 8);
        si.p = madd(d, h.t, o);
        V3 no = {0.0f, 0.0f, -1.0f};
        if (h.u == 0.0f) {
            no = {dot(r0, si.p) + P.g[3], dot(r1, si.p) + P.g[7], 1.0f - (dot(r2, si.p) + P.g[11])};
            if (!(dot(no, no) > 0.0f)) no = {0.0f, 0.0f, 1.0f};
        }
        si.n = normalize(V3{fmaf(r0.x, no.x, fmaf(r1.x, no.y, r2.x * no.z)), fmaf(r0.y, no.x, fmaf(r1.y, no.y, r2.y * no.z)),
                            fmaf(r0.z, no.x, fmaf(r1.z, no.y, r2.z * no.z))});
    } else {
        si.p = madd(g3(P, 6), h.v, madd(g3(P, 3), h.u, g3(P, 0)));
        si.n = g3(P, 9);
    }
    // Mitsuba Mesh::compute_surface_interaction: sh_frame.n = normalize(b0 n0 + b1 n1 + b2 n2)
    si.ns = si.n;
    if (!sc.vnormals.empty() && (P.type == PBRT_PRIM_TRIANGLE || P.type == PBRT_PRIM_PARALLELOGRAM)) {
        const float *r = &sc.vnormals[(size_t)h.prim * 9];
        const V3 n0 = {r[0], r[1], r[2]}, n1 = {r[3], r[4], r[5]}, n2 = {r[6], r[7], r[8]};
        if (dot(n0, n0) + dot(n1, n1) + dot(n2, n2) > 0.0f) {
            if (P.type == PBRT_PRIM_PARALLELOGRAM) {
                si.ns = normalize(n0);
            } else {
                const float b0 = 1.0f - h.u - h.v;
                si.ns = normalize(madd(n0, b0, madd(n1, h.u, n2 * h.v)));
            }
        }
    }
    return si;
}

// dp_du of the interaction, input of Mitsuba's shading frame (omath.h make_sh_frame): first edge of a triangle /
// parallelogram (Mesh without texture coordinates: p1 - p0; `rectangle`: to_world * (2,0,0)), 2 pi (-y, x, 0) about the
// centre of a sphere (object axes taken parallel to the world's), none for the [DEFINE] cone (-> coordinate_system(n))
inline V3 si_dp_du(const SI &si) {
    const pbrt_prim &P = *si.prim;
    if (P.type == PBRT_PRIM_SPHERE) return {-(si.p.y - P.g[1]), si.p.x - P.g[0], 0.0f};
    if (P.type == PBRT_PRIM_CONE) return {0.0f, 0.0f, 0.0f};
    return g3(P, 3);
}

// ------------------------------------------------------------------------------------------------
// BSDFs, radiance mode (Mitsuba diffuse / conductor / dielectric)
// ------------------------------------------------------------------------------------------------
struct BSample {
    V3 wo;
    float pdf;
    V3 weight;
    float eta;
    bool delta;
    bool valid;
    uint32_t lobe;  // 0 reflection, 1 transmission
};

void ultra_sample(const pbrt_material &m, uint32_t quirks, V3 wi, V3 n_geo, V3 n_sh, const Frame &shf, float s1, float s2x,
                  float s2y, V3 *wo, float *pdf, float *amp, uint32_t *lobe);

inline bool mat_smooth(const pbrt_material &m) { return m.type == PBRT_MAT_DIFFUSE; }

// value * cos(theta_o) and pdf; zero for delta / ultrasound BSDFs (CustomBSDF.py:177-184)
inline void bsdf_eval_pdf(const pbrt_material &m, V3 wi, V3 wo, V3 *f, float *pdf) {
    *f = {0, 0, 0};
    *pdf = 0.0f;
    if (m.type == PBRT_MAT_DIFFUSE && wi.z > 0.0f && wo.z > 0.0f) {
        float c = kInvPi * wo.z;
        *f = v3(m.p[0], m.p[1], m.p[2]) * c;
        *pdf = c;
    }
}

// shf: the interaction's shading frame (si.to_local at CustomBSDF.py:165; read by ULTRA only)
inline BSample bsdf_sample(const pbrt_material &m, uint32_t quirks, V3 wi, V3 n_geo, V3 n_sh, const Frame &shf, float s1,
                           float s2x, float s2y) {
    BSample b;
    b.valid = false;
    b.delta = false;
    b.eta = 1.0f;
    b.pdf = 0.0f;
    b.weight = {0, 0, 0};
    b.wo = {0, 0, 1};
    b.lobe = 0;
    switch (m.type) {
        case PBRT_MAT_DIFFUSE: {
            if (!(wi.z > 0.0f)) return b;
            b.wo = square_to_cosine_hemisphere(s2x, s2y);
            b.pdf = kInvPi * b.wo.z;
            if (!(b.pdf > 0.0f)) return b;
            b.weight = v3(m.p[0], m.p[1], m.p[2]);
            b.valid = true;
            return b;
        }
        case PBRT_MAT_CONDUCTOR: {
            if (!(wi.z > 0.0f)) return b;
            b.wo = {-wi.x, -wi.y, wi.z};
            b.pdf = 1.0f;
            b.weight = v3(m.p[0], m.p[1], m.p[2]);
            b.delta = true;
            b.valid = true;
            return b;
        }
        case PBRT_MAT_DIELECTRIC: {
            // Mitsuba fresnel(cos_theta_i, eta) + SmoothDielectric::sample
            float eta = m.p[0];
            float ci = wi.z;
            bool outside = ci >= 0.0f;
            float rcp_eta = 1.0f / eta;
            float eta_it = outside ? eta : rcp_eta, eta_ti = outside ? rcp_eta : eta;
            float ct2 = fmaf(-fmaf(-ci, ci, 1.0f), eta_ti * eta_ti, 1.0f);
            float cia = fabsf(ci), cta = sqrtf(fmaxf(ct2, 0.0f));
            float a_s = fmaf(-eta_it, cta, cia) / fmaf(eta_it, cta, cia);
            float a_p = fmaf(-eta_it, cia, cta) / fmaf(eta_it, cia, cta);
            float r = 0.5f * fmaf(a_s, a_s, a_p * a_p);
            if (eta == 1.0f) r = 0.0f;
            else if (cia == 0.0f) r = 1.0f;
            float ct = copysignf(cta, -ci);
            b.delta = true;
            b.valid = true;
            if (s1 <= r) {
                b.wo = {-wi.x, -wi.y, wi.z};
                b.pdf = r;
                b.weight = {1, 1, 1};
                b.lobe = 0;
            } else {
                b.wo = {-eta_ti * wi.x, -eta_ti * wi.y, ct};
                b.pdf = 1.0f - r;
                float f2 = eta_ti * eta_ti;  // radiance transport
                b.weight = {f2, f2, f2};
                b.eta = eta_it;
                b.lobe = 1;
            }
            return b;
        }
        case PBRT_MAT_ULTRA: {
            float amp;
            ultra_sample(m, quirks, wi, n_geo, n_sh, shf, s1, s2x, s2y, &b.wo, &b.pdf, &amp, &b.lobe);
            b.weight = {amp, amp, amp};
            b.delta = true;  // components are declared Delta* (CustomBSDF.py:22-26)
            b.valid = true;
            return b;
        }
        default:
            return b;
    }
}

// ------------------------------------------------------------------------------------------------
// UltraBSDF.sample -- CustomBSDF.py:87-175 with _ggx_sample :30-61 and ggx_pdf :64-83.
// `quirks` selects, bit by bit, the literal behaviour of the reference or the
// documented intent; PBRT_USQ_REFERENCE is the literal restatement.
//   wi     si.wi, incident direction in the local shading frame       (CustomBSDF.py:90)
//   n_geo  si.n, world geometric normal                                (:95)
//   n_sh   si.sh_frame.n, world shading normal                         (:91)
//   s1     'sample1' (micro-normal), s2x 'sample2' (lobe choice); s2y second micro-normal variate
//          used only when the diagonal-broadcast quirk A2 is off.
// Returns wo in the local shading frame when called through the BSDF API; the integrator undoes
// that with si.to_world (CustomIntegrator.py:358), so ultra_dir() below is what transport uses.
// ------------------------------------------------------------------------------------------------
struct UltraOut {
    V3 chosen;  // direction as computed at CustomBSDF.py:147 (before si.to_local)
    float pdf, amp;
    bool reflect;
};

UltraOut ultra_core(const pbrt_material &m, uint32_t quirks, V3 wi_in, V3 n_geo, V3 n_sh, float s1, float s2, float s1b) {
    float impedance = m.p[0], alpha = m.p[1], medium_z = m.p[2];
    // --- _ggx_sample(si.wi, si.n, sample1)  (:30-61)
    V3 wi = wi_in;
    if (quirks & PBRT_USQ_DOUBLE_LOCAL) wi = to_local(make_frame(n_geo), wi_in);  // :32-33 (A1)
    V3 ws = normalize(v3(alpha * wi.x, alpha * wi.y, wi.z));                     // :37-38
    float inv_len = 1.0f / sqrtf(fmaxf(fmaf(-ws.z, ws.z, 1.0f), 1e-7f));          // :41
    V3 T1 = {ws.y * inv_len, -ws.x * inv_len, 0.0f};                               // :42-44
    V3 T2 = cross(ws, T1);                                                         // :45
    float dx, dy;
    if (quirks & PBRT_USQ_DIAG_SAMPLE)
        square_to_disk(s1, s1, &dx, &dy);  // :48 scalar broadcast to Point2f(s,s) (A2)
    else
        square_to_disk(s1, s1b, &dx, &dy);
    float S = 0.5f * (1.0f + ws.z);                                               // :51
    dy = fmaf(1.0f - S, sqrtf(fmaxf(fmaf(-dx, dx, 1.0f), 0.0f)), S * dy);          // :52
    float mz = sqrtf(fmaxf(1.0f - fmaf(dx, dx, dy * dy), 0.0f));                   // :55
    V3 ms = madd(ws, mz, madd(T2, dy, T1 * dx));                                   // :55
    V3 mm = normalize(v3(alpha * ms.x, alpha * ms.y, ms.z));                       // :56-59
    // --- sample()  (:87-175)
    V3 inc = wi_in;                                                                // :90
    if (!(dot(mm, inc) < 0.0f)) mm = -mm;                                          // :100
    float cos_wi_m = dot(inc, mm);                                                 // :101
    bool entering;
    if (quirks & PBRT_USQ_NEVER_ENTER)
        entering = dot(mm, inc) > 0.0f;  // :104, false after :100 (A4)
    else
        entering = wi_in.z > 0.0f;  // intent: the wave arrives from the coupling medium
    float Z1 = entering ? medium_z : impedance;                                    // :106
    float Z2 = entering ? impedance : medium_z;                                    // :107
    float ratio = Z1 / Z2;                                                         // :111
    float cosTr = fabsf(dot(mm, inc));                                             // :119
    float sqrt_arg = fmaf(-(ratio * ratio), fmaf(-cosTr, cosTr, 1.0f), 1.0f);      // :120
    float cosTt = sqrtf(fmaxf(sqrt_arg, 0.0f));                                    // :121
    float denom = fmaf(Z1, cosTr, Z2 * cosTt);                                     // :122
    float Ar = fmaf(Z1, cosTr, -(Z2 * cosTt)) / denom;                             // :123
    float At = 1.0f - Ar;                                                          // :124
    V3 refl, trans;
    if (quirks & PBRT_USQ_REF_REFLECT) {
        refl = madd(mm, 2.0f * cos_wi_m, inc);                                     // :130 (A5)
        trans = madd(mm, fmaf(ratio, cosTr, -cosTt), refl * ratio);                // :131
    } else {
        // intent: mirror direction 2(wi.m)m - wi and Snell refraction about the facet facing wi (-m)
        refl = madd(mm, 2.0f * cos_wi_m, -inc);
        trans = madd(mm, -fmaf(ratio, cosTr, -cosTt), (-inc) * ratio);
    }
    bool tir = sqrt_arg < 0.0f;                                                    // :137
    float prob_reflect = Ar * Ar;                                                  // :142
    bool select_reflect = tir ? true : (s2 < prob_reflect);                        // :144-145
    V3 chosen = select_reflect ? refl : trans;                                     // :147
    float pdf_m;
    if (quirks & PBRT_USQ_UNIT_GGX_PDF) {
        pdf_m = 1.0f;  // :81-82 pdf_max / pdf_max (A7)
    } else {
        // intent: GGX normal distribution D(m) * |m.z| with alpha = roughness
        float c = fabsf(mm.z), a2 = alpha * alpha;
        float dd = fmaf(fmaf(a2, 1.0f, -1.0f) * c, c, 1.0f);
        pdf_m = a2 / (kPi * dd * dd) * c;
    }
    float pdf_reflect = pdf_m / (4.0f * fabsf(cos_wi_m));                          // :154
    float cos_wo_m = dot(trans, mm);                                               // :155
    V3 nref = (quirks & PBRT_USQ_MIXED_FRAMES) ? n_sh : v3(0, 0, 1);               // :156-157 (A8)
    float abs_n_wi = fabsf(dot(nref, inc));                                        // :156
    float abs_n_wo = fmaxf(fabsf(dot(nref, trans)), 1e-7f);                        // :157
    float pdf_trans = pdf_m * (ratio * ratio) * fabsf(cos_wo_m) / (abs_n_wi * abs_n_wo);  // :158
    UltraOut o;
    o.chosen = chosen;
    o.pdf = select_reflect ? pdf_reflect : pdf_trans;                              // :166
    o.amp = select_reflect ? Ar : At;                                              // :170
    o.reflect = select_reflect;
    return o;
}

void ultra_sample(const pbrt_material &m, uint32_t quirks, V3 wi, V3 n_geo, V3 n_sh, const Frame &shf, float s1, float s2x,
                  float s2y, V3 *wo, float *pdf, float *amp, uint32_t *lobe) {
    UltraOut o = ultra_core(m, quirks, wi, n_geo, n_sh, s1, s2x, s2y);
    *wo = to_local(shf, o.chosen);  // :165 bs.wo = si.to_local(chosen_dir)
    *pdf = o.pdf;
    *amp = o.amp;
    *lobe = o.reflect ? 0u : 1u;
}

// ------------------------------------------------------------------------------------------------
// Emitter sampling (Scene::sample_emitter_direction, area / point emitters)
// ------------------------------------------------------------------------------------------------
struct ESample {
    V3 q, d;       // sampled point, unit direction from the reference point
    float dist, pdf;
    V3 weight;     // radiance / pdf (no visibility)
    bool delta, valid;
    uint32_t emitter;
};

inline ESample sample_emitter(const Scene &sc, V3 p, F4 u) {
    ESample e;
    e.valid = false;
    e.delta = false;
    e.pdf = 0.0f;
    e.dist = 0.0f;
    e.weight = {0, 0, 0};
    e.q = e.d = {0, 0, 0};
    e.emitter = 0;
    uint32_t nE = (uint32_t)sc.emitters.size();
    if (nE == 0) return e;
    uint32_t ei = std::min((uint32_t)(u.x * (float)nE), nE - 1);
    const pbrt_emitter &E = sc.emitters[ei];
    e.emitter = ei;
    float sel = (float)nE;  // 1 / emitter pmf
    if (E.type == PBRT_EMIT_POINT) {
        e.q = v3(E.pos[0], E.pos[1], E.pos[2]);
        V3 dv = e.q - p;
        float d2 = dot(dv, dv);
        e.dist = sqrtf(d2);
        float inv = 1.0f / e.dist;
        e.d = dv * inv;
        e.pdf = 1.0f;
        e.delta = true;
        float k = (inv * inv) * sel;
        e.weight = v3(E.radiance[0], E.radiance[1], E.radiance[2]) * k;
        e.valid = true;
        return e;
    }
    // area emitter: primitive by area CDF, uniform point on it
    uint32_t k = 0;
    while (k + 1 < E.count && !(u.y < sc.light_cdf[E.first + k])) ++k;
    const pbrt_prim &P = sc.prims[sc.light_prims[E.first + k]];
    float b1, b2;
    if (P.type == PBRT_PRIM_TRIANGLE) {
        float t = sqrtf(fmaxf(1.0f - u.z, 0.0f));  // warp::square_to_uniform_triangle
        b1 = 1.0f - t;
        b2 = t * u.w;
    } else {
        b1 = u.z;
        b2 = u.w;
    }
    e.q = madd(g3(P, 6), b2, madd(g3(P, 3), b1, g3(P, 0)));
    V3 nl = g3(P, 9);
    V3 dv = e.q - p;
    float d2 = dot(dv, dv);
    e.dist = sqrtf(d2);
    float inv = 1.0f / e.dist;
    e.d = dv * inv;
    float cosl = -dot(nl, e.d);
    if (!(cosl > 0.0f)) return e;  // back side of a one-sided area light
    e.pdf = d2 / (cosl * E.area * sel);
    float w = 1.0f / e.pdf;
    e.weight = v3(E.radiance[0], E.radiance[1], E.radiance[2]) * w;
    e.valid = true;
    return e;
}

inline float mis_weight(float a, float b) {
    a *= a;
    b *= b;
    float w = a / (a + b);
    return std::isfinite(w) ? w : 0.0f;
}

// ------------------------------------------------------------------------------------------------
// Sensor.sample_ray: Mitsuba 'perspective' 
// ------------------------------------------------------------------------------------------------
inline void camera_ray(const pbrt_camera &cam, float sx, float sy, V3 *o, V3 *d, float *tmax) {
    float tx = cam.tan_half_fov_x;
    float ty = tx * (float)cam.film_h / (float)cam.film_w;
    V3 dc = normalize(v3(fmaf(-2.0f, sx, 1.0f) * tx, fmaf(-2.0f, sy, 1.0f) * ty, 1.0f));
    V3 dw = normalize(xf_vec(cam.to_world, dc));
    float inv_z = 1.0f / dc.z;
    V3 org = {cam.to_world[3], cam.to_world[7], cam.to_world[11]};
    *o = madd(dw, cam.near_clip * inv_z, org);
    *d = dw;
    *tmax = (cam.far_clip - cam.near_clip) * inv_z;
}

// ------------------------------------------------------------------------------------------------
// Integrator.sample: Mitsuba 'path' (and 'direct' == path with max_depth 2); SURVEY.md App. D.
// RNG blocks per path: c = 0 camera jitter; c = 1 + 2k emitter sample of bounce k;
// c = 2 + 2k BSDF sample (x: lobe, y,z: direction) and Russian roulette (w) of bounce k.
// ------------------------------------------------------------------------------------------------
struct PathStats {
    uint64_t segments = 0, shadow = 0;
};

V3 path_radiance(const Scene &sc, V3 o, V3 d, float tmax, uint32_t ka, uint32_t kb, uint32_t seed, uint32_t max_depth,
                 uint32_t rr_depth, PathStats *st, bool no_pruning = false) {
    V3 thr = {1, 1, 1}, L = {0, 0, 0};
    float eta = 1.0f, prev_pdf = 1.0f;
    bool prev_delta = true;
    uint32_t nE = (uint32_t)sc.emitters.size();
    for (uint32_t depth = 0;; ) {
        Hit h;
        if (!closest_hit(sc, o, d, tmax, &h)) break;
        if (st) st->segments++;
        SI si = make_si(sc, o, d, h);
        const pbrt_prim &P = *si.prim;
        // ---- direct emission (area emitters are one-sided)
        if (P.emitter >= 0) {
            const pbrt_emitter &E = sc.emitters[P.emitter];
            float cosl = -dot(si.ns, d);  // Frame::cos_theta(si.wi): the shading frame (== n on emitters)
            if (cosl > 0.0f) {
                float w = 1.0f;
                if (!prev_delta) {
                    float pdf_em = (h.t * h.t) / (cosl * E.area * (float)nE);
                    w = mis_weight(prev_pdf, pdf_em);
                }
                V3 Le = v3(E.radiance[0], E.radiance[1], E.radiance[2]);
                L = {fmaf(thr.x * Le.x, w, L.x), fmaf(thr.y * Le.y, w, L.y), fmaf(thr.z * Le.z, w, L.z)};
            }
        }
        if (depth + 1 >= max_depth) break;
        const pbrt_material &M = sc.mats[P.material];
        Frame fr = make_frame(si.ns);
        V3 wi = to_local(fr, -d);
        // ---- emitter sampling
        if (mat_smooth(M) && nE > 0) {
            F4 u = rng4(ka, kb, 1 + 2 * depth, seed);
            ESample es = sample_emitter(sc, si.p, u);
            if (es.valid) {
                V3 wo = to_local(fr, es.d);
                V3 f;
                float bpdf;
                bsdf_eval_pdf(M, wi, wo, &f, &bpdf);
                if (bpdf > 0.0f) {
                    // Interaction::spawn_ray_to
                    V3 so = offset_origin(si.p, si.n, es.d);
                    V3 sv = es.q - so;
                    float sd = sqrtf(dot(sv, sv));
                    V3 sdir = sv * (1.0f / sd);
                    if (st) st->shadow++;
                    if (!any_hit_segment(sc, so, sdir, sd * (1.0f - kShadowEps), no_pruning)) {
                        float mis = es.delta ? 1.0f : mis_weight(es.pdf, bpdf);
                        L = {fmaf(thr.x * f.x, es.weight.x * mis, L.x), fmaf(thr.y * f.y, es.weight.y * mis, L.y),
                             fmaf(thr.z * f.z, es.weight.z * mis, L.z)};
                    }
                }
            }
        }
        // ---- BSDF sampling
        F4 ub = rng4(ka, kb, 2 + 2 * depth, seed);
        BSample bs = bsdf_sample(M, PBRT_USQ_REFERENCE, wi, si.n, si.ns, fr, ub.x, ub.y, ub.z);
        if (!bs.valid) break;
        thr = thr * bs.weight;
        eta *= bs.eta;
        V3 nd = to_world(fr, bs.wo);
        if (M.type == PBRT_MAT_ULTRA) nd = normalize(nd);
        o = offset_origin(si.p, si.n, nd);
        d = nd;
        tmax = kInf;
        prev_pdf = bs.pdf;
        prev_delta = bs.delta;
        depth += 1;
        float tm = max3(thr);
        if (depth >= rr_depth) {
            float q = fminf(tm * eta * eta, 0.95f);
            bool cont = ub.w < q;
            float rq = 1.0f / q;
            thr = thr * rq;
            if (!cont) break;
        }
        if (tm == 0.0f) break;
    }
    return L;
}

// ------------------------------------------------------------------------------------------------
// Film: hdrfilm + reconstruction filter.  Deterministic gather:
//   pixel(ix,iy) = sum_{s asc} sum_{ny,nx raster over the footprint} w * L / sum w
// ------------------------------------------------------------------------------------------------
inline int filter_halo(uint32_t f) { return f == PBRT_FILTER_BOX ? 0 : (f == PBRT_FILTER_TENT ? 1 : 2); }
inline float filter_1d(uint32_t f, float x) {
    if (f == PBRT_FILTER_TENT) return fmaxf(0.0f, 1.0f - fabsf(x));
    // gaussian, stddev 0.5, radius 2
    const float alpha = -2.0f;
    return fmaxf(0.0f, expf(alpha * x * x) - expf(alpha * 4.0f));
}

struct RenderJob {
    const Scene *sc;
    const pbrt_camera *cam;
    const pbrt_film_desc *film;
};

}  // namespace

// ================================================================================================
// C entry points (mirror of include/pbrt_hip.h without the device context)
// ================================================================================================
extern "C" {

struct oracle_scene {
    Scene sc;
};

int oracle_scene_create(const pbrt_scene_desc *desc, oracle_scene **out) {
    if (!desc || !out) return PBRT_E_INVALID;
    oracle_scene *s = new oracle_scene();
    s->sc.prims.assign(desc->prims, desc->prims + desc->n_prims);
    s->sc.mats.assign(desc->materials, desc->materials + desc->n_materials);
    s->sc.emitters.assign(desc->emitters, desc->emitters + desc->n_emitters);
    s->sc.light_prims.assign(desc->light_prims, desc->light_prims + desc->n_light_prims);
    s->sc.light_cdf.assign(desc->light_cdf, desc->light_cdf + desc->n_light_prims);
    if (desc->vertex_normals) s->sc.vnormals.assign(desc->vertex_normals, desc->vertex_normals + (size_t)desc->n_prims * 9);
    for (auto &p : s->sc.prims) {
        double cc[3], ca[3], cb[3], cx[3];
        if (p.type > PBRT_PRIM_CONE || p.material >= desc->n_materials ||
            (p.emitter >= 0 && (uint32_t)p.emitter >= desc->n_emitters) ||
            (p.type == PBRT_PRIM_CONE && !cone_frame(p, cc, ca, cb, cx))) {
            const int rc = p.type > PBRT_PRIM_CONE ? PBRT_E_UNSUPPORTED : PBRT_E_INVALID;  // (p lives in *s)
            delete s;
            return rc;
        }
    }
    s->sc.use_bvh = desc->accel == PBRT_ACCEL_BVH || desc->accel == PBRT_ACCEL_BVH_GLOBAL ||
                    (desc->accel == PBRT_ACCEL_AUTO && desc->n_prims > 32);
    if (s->sc.use_bvh)
        build_bvh(s->sc);
    else
        find_occluders(s->sc);
    *out = s;
    return PBRT_OK;
}
int oracle_scene_update_material(oracle_scene *s, uint32_t index, const pbrt_material *m) {
    if (!s || !m || index >= s->sc.mats.size()) return PBRT_E_INVALID;
    s->sc.mats[index] = *m;
    return PBRT_OK;
}
int oracle_scene_destroy(oracle_scene *s) {
    delete s;
    return PBRT_OK;
}

// stats out: [0] segments, [1] shadow rays (may be NULL)
int oracle_render_radiance(oracle_scene *s, const pbrt_camera *cam, const pbrt_film_desc *film, float *out,
                           int n_threads, uint64_t *stats) {
    if (!s || !cam || !film || !out) return PBRT_E_INVALID;
    const Scene &sc = s->sc;
    const int W = (int)cam->film_w, H = (int)cam->film_h;
    const int R = filter_halo(film->filter);
    const int cx0 = (int)film->crop_x, cy0 = (int)film->crop_y, cw = (int)film->crop_w, ch = (int)film->crop_h;
    if (cx0 + cw > W || cy0 + ch > H || cw <= 0 || ch <= 0) return PBRT_E_INVALID;
    const int rx0 = std::max(cx0 - R, 0), ry0 = std::max(cy0 - R, 0);
    const int rx1 = std::min(cx0 + cw + R, W), ry1 = std::min(cy0 + ch + R, H);
    const int rw = rx1 - rx0, rh = ry1 - ry0;
    std::vector<float> Ls((size_t)rw * rh * 3), jit((size_t)rw * rh * 2);
    std::vector<float> acc((size_t)cw * ch * 4, 0.0f);
    if (n_threads < 1) n_threads = 1;
    std::atomic<uint64_t> seg{0}, shd{0};
    for (uint32_t si = 0; si < film->spp; ++si) {
        const uint32_t s_idx = film->sample_offset + si;
        std::atomic<int> next_row{0};
        auto trace_rows = [&]() {
            PathStats st;
            for (;;) {
                int ry = next_row.fetch_add(1);
                if (ry >= rh) break;
                for (int rx = 0; rx < rw; ++rx) {
                    int x = rx0 + rx, y = ry0 + ry;
                    uint32_t pix = (uint32_t)(y * W + x);
                    F4 uj = rng4(pix, s_idx, 0, film->seed);
                    float jx = uj.x, jy = uj.y;
                    float px = (float)x + jx, py = (float)y + jy;
                    V3 o, d;
                    float tmax;
                    camera_ray(*cam, px / (float)W, py / (float)H, &o, &d, &tmax);
                    V3 L = path_radiance(sc, o, d, tmax, pix, s_idx, film->seed, film->max_depth, film->rr_depth, &st,
                                         (film->flags & PBRT_FILM_NO_OCCLUDER_PRUNING) != 0);
                    size_t k = (size_t)ry * rw + rx;
                    Ls[3 * k] = L.x;
                    Ls[3 * k + 1] = L.y;
                    Ls[3 * k + 2] = L.z;
                    jit[2 * k] = px;
                    jit[2 * k + 1] = py;
                }
            }
            seg += st.segments;
            shd += st.shadow;
        };
        std::vector<std::thread> th;
        for (int t = 1; t < n_threads; ++t) th.emplace_back(trace_rows);
        trace_rows();
        for (auto &t : th) t.join();
        // gather
        for (int iy = 0; iy < ch; ++iy)
            for (int ix = 0; ix < cw; ++ix) {
                int x = cx0 + ix, y = cy0 + iy;
                float *a = &acc[4 * ((size_t)iy * cw + ix)];
                float ccx = (float)x + 0.5f, ccy = (float)y + 0.5f;
                for (int ny = y - R; ny <= y + R; ++ny)
                    for (int nx = x - R; nx <= x + R; ++nx) {
                        if (nx < 0 || ny < 0 || nx >= W || ny >= H) continue;
                        size_t k = (size_t)(ny - ry0) * rw + (nx - rx0);
                        float w;
                        if (film->filter == PBRT_FILTER_BOX)
                            w = 1.0f;
                        else
                            w = filter_1d(film->filter, ccx - jit[2 * k]) * filter_1d(film->filter, ccy - jit[2 * k + 1]);
                        if (w > 0.0f) {
                            a[0] = fmaf(w, Ls[3 * k], a[0]);
                            a[1] = fmaf(w, Ls[3 * k + 1], a[1]);
                            a[2] = fmaf(w, Ls[3 * k + 2], a[2]);
                            a[3] += w;
                        }
                    }
            }
    }
    if (film->flags & PBRT_FILM_RAW_ACCUM) {
        std::memcpy(out, acc.data(), acc.size() * sizeof(float));
    } else {
        for (size_t k = 0; k < (size_t)cw * ch; ++k) {
            float w = acc[4 * k + 3];
            float inv = w > 0.0f ? 1.0f / w : 0.0f;
            out[3 * k] = acc[4 * k] * inv;
            out[3 * k + 1] = acc[4 * k + 1] * inv;
            out[3 * k + 2] = acc[4 * k + 2] * inv;
        }
    }
    if (stats) {
        stats[0] = seg.load();
        stats[1] = shd.load();
    }
    return PBRT_OK;
}

// Integrator.sample(scene, sampler, ray, medium, active) on caller-supplied rays (signature: CustomIntegrator.py:52; the
// reference's body is the stub `return Color1f(0)`, the radiance estimator is Mitsuba's `path`, App. D): twin of
// pbrt_integrator_sample.  Ray i draws from the key (index_offset + i, sample_index): with index_offset = 0, rays listed
// in pixel order and generated with the jitter of rng4(pixel, sample_index, 0, seed) these are exactly the paths of a
// render, so rgb[i] is the value of sample `sample_index` of pixel i.   rgb [3][n].
int oracle_integrator_sample(oracle_scene *s, uint32_t n, const float *o, const float *d, const float *tmax,
                             uint32_t index_offset, uint32_t sample_index, uint32_t seed, uint32_t max_depth,
                             uint32_t rr_depth, float *rgb) {
    if (!s || !o || !d || !tmax || !rgb || max_depth == 0) return PBRT_E_INVALID;
    for (uint32_t i = 0; i < n; ++i) {
        V3 L = path_radiance(s->sc, v3(o[i], o[n + i], o[2 * n + i]), v3(d[i], d[n + i], d[2 * n + i]), tmax[i],
                             index_offset + i, sample_index, seed, max_depth, rr_depth, nullptr);
        rgb[i] = L.x;
        rgb[n + i] = L.y;
        rgb[2 * n + i] = L.z;
    }
    return PBRT_OK;
}

// ------------------------------------------------------------------------------------------------
// Ultrasound acquisition: UltraIntegrator.simulate_acquisition_parallel (CustomIntegrator.py:235-405)
// RNG block per (ray = a*N+e, path k, bounce b): rng4(ray, k, b, seed) = (recv pick :319, s1 :337,
// s2 :337, Russian roulette :365).
// ------------------------------------------------------------------------------------------------
int oracle_us_tx_delays(const pbrt_us_params *p, float *tx) {
    if (!p || !tx || p->n_angles > PBRT_US_MAX_ANGLES) return PBRT_E_INVALID;
    for (uint32_t a = 0; a < p->n_angles; ++a) {
        double ar = (double)p->angles_deg[a] * (M_PI / 180.0);  // np.deg2rad (:247)
        for (uint32_t e = 0; e < p->n_elements; ++e) {
            // :248 elem_x is float32: pitch * (arange_f32 - (N-1)/2)
            float ex = (float)((double)p->pitch * ((double)(float)e - ((double)p->n_elements - 1.0) / 2.0));
            tx[a * p->n_elements + e] = (float)(((double)ex * sin(ar)) / (double)p->sound_speed);  // :254,257
        }
    }
    return PBRT_OK;
}

static inline float us_elem_x(const pbrt_us_params *p, uint32_t e) {
    return (float)((double)p->pitch * ((double)(float)e - ((double)p->n_elements - 1.0) / 2.0));
}

// One ray of CustomEmitter.sample_ray (CustomEmmitter.py:51-107): element pick (:56-57), element centre and normal of the linear
// (:33-38) or convex (:41-47) array, jitter inside the element (:64-68), steering angle (:85-87), direction (:90), steering delay
// (:93-94), cosine weight (:97-98).  Shared by the leaf operator and by the acquisition's emitter-primary mode.
struct EmitRay {
    V3 o, d;
    float time, weight, pdf_pos;
};
// sin / cos of a steering or element angle: the fixed polynomial shared with the device up to 45 degrees (omath.h sincos_pi4), the
// library routines beyond -- so that the first bounce of an emitter ray sees the same bits on both sides (kernels_us.h emit_sincos)
static inline void emit_sincos(float x, float *s, float *c) {
    if (fabsf(x) <= kPiOver4) {
        sincos_pi4(x, s, c);
    } else {
        *s = sinf(x);
        *c = cosf(x);
    }
}
static inline EmitRay emitter_ray(const pbrt_us_emitter &e, float time, float s1, float s2x, float s2y, float s3) {
    const float N = (float)e.number_of_elements;
    const float total_rays = (float)(e.number_of_elements * e.number_of_rays_per_element);  // :17
    float idx = fminf(floorf(s1 * N), N - 1.0f);                                            // :56-57
    V3 c, nrm;
    if (e.radius == 0.0f) {                                                                 // :33-38 linspace
        float lo = -(N - 1.0f) / 2.0f * e.pitch, hi = (N - 1.0f) / 2.0f * e.pitch;
        float x = N > 1.0f ? fmaf(idx, (hi - lo) / (N - 1.0f), lo) : lo;
        c = {x, 0.0f, 0.0f};
        nrm = {0.0f, 0.0f, 1.0f};
    } else {                                                                                // :41-47
        float span = e.opening_angle * (kPi / 180.0f);
        float lo = -span / 2.0f, hi = span / 2.0f;
        float th = N > 1.0f ? fmaf(idx, (hi - lo) / (N - 1.0f), lo) : lo;
        float sth, cth;
        emit_sincos(th, &sth, &cth);
        c = {e.radius * sth, 0.0f, e.radius * cth};
        nrm = normalize(v3(sth, 0.0f, cth));                                                // :49
    }
    float dx = (s2x - 0.5f) * e.element_width, dy = (s2y - 0.5f) * e.element_height;        // :64-65
    EmitRay r;
    r.o = c + v3(dx, dy, 0.0f);                                                             // :68
    r.pdf_pos = 1.0f / (N * e.element_width * e.element_height);                            // :77
    float pmin = e.steering_angle_min * (kPi / 180.0f), pmax = e.steering_angle_max * (kPi / 180.0f);
    float psi = fmaf(s3, pmax - pmin, pmin);                                                // :85-87
    float spsi, cpsi;
    emit_sincos(psi, &spsi, &cpsi);
    r.d = {spsi, 0.0f, cpsi};                                                               // :90
    float delay = -(r.o.x * spsi) / e.speed_of_sound;                                       // :93
    r.time = time + delay;                                                                  // :94
    float fd = fmaxf(0.0f, dot(r.d, nrm));                                                  // :97
    r.weight = fd / total_rays;                                                             // :98
    return r;
}

static inline float directivity_weight_i(V3 sec_dir, V3 tn, float am, float ac) {  // :289-304
    V3 w = -sec_dir;
    float dt = dot(tn, w);
    float alpha = fabsf(acosf(dt));
    float mid = (ac - alpha) / (ac - am);
    return alpha <= am ? 1.0f : (alpha <= ac ? mid : 0.0f);
}

int oracle_us_acquire(oracle_scene *s, const pbrt_us_params *p, uint32_t seed, uint32_t paths_per_ray,
                      uint32_t path_offset, uint32_t norm_paths, float *channel_buf, float *tx_delays, uint64_t *stats) {
    if (!s || !p || !channel_buf || p->n_angles > PBRT_US_MAX_ANGLES) return PBRT_E_INVALID;
    const Scene &sc = s->sc;
    const uint32_t NA = p->n_angles, NE = p->n_elements, T = p->time_samples;
    std::vector<float> tx(NA * NE);
    oracle_us_tx_delays(p, tx.data());
    if (tx_delays) std::memcpy(tx_delays, tx.data(), tx.size() * sizeof(float));
    std::vector<double> acc((size_t)NA * NE * T, 0.0);  // deterministic order, double accumulators
    const float num_rays = (float)(NA * NE);                                                   // :243
    const V3 tn = normalize(xf_vec(p->sensor_to_world, v3(0, 0, 1)));                          // :292,369
    const float am = p->main_beam_angle * (kPi / 180.0f), ac = p->cutoff_angle * (kPi / 180.0f);
    const float cos_min = cosf(ac);                                                            // :370
    const float katt = (float)(-(double)p->attenuation * (double)p->frequency * 1e-6);         // :328
    const float two_pi_f = (float)(2.0 * M_PI * (double)p->frequency);                         // :330
    const float inv_c = 1.0f / p->sound_speed;
    uint64_t segs = 0, shadows = 0;
    const bool emit = p->primary == PBRT_US_PRIMARY_EMITTER;
    if (emit && p->emitter.number_of_elements != NE) return PBRT_E_INVALID;
    for (uint32_t a = 0; a < NA; ++a)
        for (uint32_t e = 0; e < NE; ++e) {
            const uint32_t ray_id = a * NE + e;
            const float a_rad = (float)((double)p->angles_deg[a] * (M_PI / 180.0));
            const float x_elem = us_elem_x(p, e);
            const float t0_elem = tx[ray_id];                                                  // :267
            const V3 o0 = xf_point(p->sensor_to_world, v3(x_elem, 0, 0));                      // :270,273
            const V3 d0 = normalize(xf_vec(p->sensor_to_world, v3(sinf(a_rad), 0.0f, cosf(a_rad))));  // :271,273
            for (uint32_t kk = 0; kk < paths_per_ray; ++kk) {
                const uint32_t k = path_offset + kk;
                V3 o = o0, d = d0;
                float amp = 1.0f, atten = 1.0f, tof = 0.0f, geo_len = 0.0f;                    // :276-279
                float t0 = t0_elem, w_ray = 1.0f;
                if (emit) {
                    // PBRT_US_PRIMARY_EMITTER (include/pbrt_hip.h, DESIGN D15): the path's own ray from CustomEmitter.sample_ray,
                    // the (angle, element) grid stratifying the emitter's element pick and steering angle; RNG block 0x80000000
                    const F4 ue = rng4(ray_id, k, 0x80000000u, seed);
                    const float s1 = ((float)e + 0.5f) / (float)NE, s3 = ((float)a + ue.z) / (float)NA;
                    const EmitRay r = emitter_ray(p->emitter, 0.0f, s1, ue.x, ue.y, s3);       // CustomEmmitter.py:81-107
                    o = xf_point(p->sensor_to_world, r.o);                                     // (:272-273 for the integrator's own ray)
                    d = normalize(xf_vec(p->sensor_to_world, r.d));
                    w_ray = r.weight;   // CustomEmmitter.py:97-98; multiplies every echo of the path, as Mitsuba's render loop multiplies
                                        // what Integrator.sample returns by the ray weight (amp itself starts at 1, :276)
                    tof = r.time;                                                              // :93-94: the steering delay starts the clock
                    t0 = 0.0f;
                }
                uint32_t depth = 0;
                bool active = true;
                while (active && depth < p->max_depth && geo_len < p->max_path_len) {          // :307
                    Hit h;
                    if (!closest_hit(sc, o, d, kInf, &h)) break;                               // :309-312
                    ++segs;
                    SI si = make_si(sc, o, d, h);
                    const float distance = h.t;                                                // :314
                    geo_len += distance;                                                       // :315
                    if (!(p->quirks & PBRT_USQ_NO_TOF_ACCUM)) tof += distance * inv_c;         // :316
                    // B1 (Dr.Jit variant): the draws are constants of the traced loop body -- every bounce reuses block 0
                    const uint32_t block = (p->quirks & PBRT_USQ_FROZEN_DRAWS) ? 0u : depth;
                    F4 u = rng4(ray_id, k, block, seed);
                    uint32_t recv = std::min((uint32_t)(u.x * (float)NE), NE - 1);             // :319
                    V3 target = xf_point(p->sensor_to_world, v3(us_elem_x(p, recv), 0, 0));    // :320-321
                    V3 tv = target - si.p;
                    float dist_recv = sqrtf(dot(tv, tv));
                    V3 sec_dir = tv * (1.0f / dist_recv);                                      // :322
                    ++shadows;
                    bool visible = !any_hit(sc, offset_origin(si.p, si.n, sec_dir), sec_dir, kInf);  // :324-325 (B7)
                    atten *= expf(katt * distance / 8.686f);                                   // :328
                    float tof_hit = (p->quirks & PBRT_USQ_NO_TOF_ACCUM) ? tof + distance * inv_c : tof;
                    float total_time = t0 + tof_hit + dist_recv * inv_c;                       // :329
                    float phase = two_pi_f * total_time;                                       // :330
                    const pbrt_material &M = sc.mats[si.prim->material];
                    // si.sh_frame as Mitsuba builds it: from the shape's dp_du (initialize_sh_frame), not coordinate_system(n)
                    const Frame fr = make_sh_frame(si.ns, si_dp_du(si));
                    V3 wi = to_local(fr, -d);                                                  // si.wi (CustomBSDF.py:90)
                    float a_resp, bpdf;
                    V3 new_dir;
                    if (M.type == PBRT_MAT_ULTRA) {
                        // intent arithmetic (A2 off): the micro-normal's second variate comes from a second block of the
                        // path's stream; u.w decides the roulette (:365) and must not steer the facet as well
                        const float s1b = (p->quirks & PBRT_USQ_DIAG_SAMPLE) ? u.w : rng4(ray_id, k, block | 0x40000000u, seed).x;
                        UltraOut uo = ultra_core(M, p->quirks, wi, si.n, si.ns, u.y, u.z, s1b); // :338
                        a_resp = uo.amp;
                        bpdf = uo.pdf;
                        // :165 + :358: to_world(to_local(chosen))
                        new_dir = to_world(fr, to_local(fr, uo.chosen));
                    } else {
                        BSample bs = bsdf_sample(M, p->quirks, wi, si.n, si.ns, fr, u.y, u.z, u.w);
                        if (!bs.valid) break;
                        a_resp = bs.weight.x;
                        bpdf = bs.pdf;
                        new_dir = to_world(fr, bs.wo);
                    }
                    float cos_theta = dot(si.ns, -d);                                          // :340 (si.sh_frame.n)
                    amp *= a_resp * cos_theta * fmaxf(bpdf, 1e-6f);                            // :341
                    float w_o = dot(d, si.ns) / num_rays;                                      // :286-287,345 (si.sh_frame.n)
                    float fd = directivity_weight_i(sec_dir, tn, am, ac) * w_o;                // :345
                    float pressure = atten * amp * fd * ((p->quirks & PBRT_USQ_NO_CARRIER) ? 1.0f : sinf(phase));  // :348 / f-3
                    pressure *= w_ray;                                                         // (x 1, or the emitter ray's weight)
                    float tf = rintf(total_time * p->fs);                                      // :351-352 (half-to-even)
                    if (p->quirks & PBRT_USQ_CLAMP_TIME) tf = fminf(fmaxf(tf, 0.0f), (float)(T - 1));
                    if (tf >= 0.0f && tf < (float)T && visible)                                // :353
                        acc[((size_t)a * NE + recv) * T + (size_t)tf] += (double)pressure;     // :354
                    d = normalize(new_dir);                                                    // :358-359
                    o = offset_origin(si.p, si.n, d);
                    depth += 1;                                                                // :361
                    bool survive = true;                                                       // (B5 repaired)
                    if (p->quirks & PBRT_USQ_SIGNED_RR) {                                      // Dr.Jit variant :219-224
                        const float rr_prob = fminf(atten * amp, 1.0f);
                        survive = u.w < rr_prob;
                        atten = survive ? atten / rr_prob : 0.0f;
                    } else {
                        const float rr_prob = fminf(fabsf(atten * amp), 1.0f);                 // :364
                        if (u.w > rr_prob) survive = false;                                    // :365-366
                        atten /= rr_prob;                                                      // :367
                    }
                    bool within = dot(d, tn) >= cos_min;                                       // :371
                    active = active && within && (geo_len < p->max_path_len) && (depth < p->max_depth) && survive;  // :372-376
                }
            }
        }
    const double inv_norm = 1.0 / (double)(norm_paths ? norm_paths : 1);
    for (size_t i = 0; i < acc.size(); ++i) channel_buf[i] = (float)(acc[i] * inv_norm);
    if (stats) {
        stats[0] = segs;
        stats[1] = shadows;
    }
    return PBRT_OK;
}

// ------------------------------------------------------------------------------------------------
// Leaf operators
// ------------------------------------------------------------------------------------------------
int oracle_ray_intersect(oracle_scene *s, uint32_t n, const float *o, const float *d, const float *tmax, float *t,
                         uint32_t *prim, float *u, float *v) {
    if (!s) return PBRT_E_INVALID;
    for (uint32_t i = 0; i < n; ++i) {
        Hit h;
        bool f = closest_hit(s->sc, v3(o[i], o[n + i], o[2 * n + i]), v3(d[i], d[n + i], d[2 * n + i]), tmax[i], &h);
        t[i] = f ? h.t : kInf;
        prim[i] = f ? h.prim : 0xffffffffu;
        u[i] = f ? h.u : 0.0f;
        v[i] = f ? h.v : 0.0f;
    }
    return PBRT_OK;
}
int oracle_ray_test(oracle_scene *s, uint32_t n, const float *o, const float *d, const float *tmax, uint8_t *hit) {
    if (!s) return PBRT_E_INVALID;
    for (uint32_t i = 0; i < n; ++i)
        hit[i] = any_hit(s->sc, v3(o[i], o[n + i], o[2 * n + i]), v3(d[i], d[n + i], d[2 * n + i]), tmax[i]) ? 1 : 0;
    return PBRT_OK;
}

int oracle_bsdf_sample(const pbrt_material *m, uint32_t quirks, uint32_t n, const float *wi, const float *n_geo,
                       const float *n_sh, const float *sh_s, const float *s1, const float *s2, float *wo, float *pdf,
                       float *weight, uint32_t *sampled) {
    for (uint32_t i = 0; i < n; ++i) {
        V3 ng = n_geo ? v3(n_geo[i], n_geo[n + i], n_geo[2 * n + i]) : v3(0, 0, 1);
        V3 ns = n_sh ? v3(n_sh[i], n_sh[n + i], n_sh[2 * n + i]) : v3(0, 0, 1);
        const Frame shf = sh_s ? make_sh_frame(ns, v3(sh_s[i], sh_s[n + i], sh_s[2 * n + i])) : make_frame(ns);
        BSample b = bsdf_sample(*m, quirks, v3(wi[i], wi[n + i], wi[2 * n + i]), ng, ns, shf, s1[i], s2[i], s2[n + i]);
        wo[i] = b.wo.x;
        wo[n + i] = b.wo.y;
        wo[2 * n + i] = b.wo.z;
        pdf[i] = b.pdf;
        weight[i] = b.weight.x;
        weight[n + i] = b.weight.y;
        weight[2 * n + i] = b.weight.z;
        sampled[i] = b.valid ? b.lobe : 0xffffffffu;
    }
    return PBRT_OK;
}
int oracle_bsdf_eval_pdf(const pbrt_material *m, uint32_t n, const float *wi, const float *wo, float *f, float *pdf) {
    for (uint32_t i = 0; i < n; ++i) {
        V3 fv;
        bsdf_eval_pdf(*m, v3(wi[i], wi[n + i], wi[2 * n + i]), v3(wo[i], wo[n + i], wo[2 * n + i]), &fv, &pdf[i]);
        f[i] = fv.x;
        f[n + i] = fv.y;
        f[2 * n + i] = fv.z;
    }
    return PBRT_OK;
}
int oracle_emitter_sample_direction(oracle_scene *s, uint32_t n, const float *p, const float *u, float *d, float *dist,
                                    float *pdf, float *weight, float *q, uint32_t *emitter) {
    if (!s) return PBRT_E_INVALID;
    for (uint32_t i = 0; i < n; ++i) {
        ESample e = sample_emitter(s->sc, v3(p[i], p[n + i], p[2 * n + i]), F4{u[i], u[n + i], u[2 * n + i], u[3 * n + i]});
        d[i] = e.d.x; d[n + i] = e.d.y; d[2 * n + i] = e.d.z;
        q[i] = e.q.x; q[n + i] = e.q.y; q[2 * n + i] = e.q.z;
        dist[i] = e.dist;
        pdf[i] = e.valid ? e.pdf : 0.0f;
        weight[i] = e.valid ? e.weight.x : 0.0f;
        weight[n + i] = e.valid ? e.weight.y : 0.0f;
        weight[2 * n + i] = e.valid ? e.weight.z : 0.0f;
        emitter[i] = e.emitter;
    }
    return PBRT_OK;
}
int oracle_sensor_sample_ray(const pbrt_camera *cam, uint32_t n, const float *pos, float *o, float *d, float *tmax) {
    for (uint32_t i = 0; i < n; ++i) {
        V3 oo, dd;
        camera_ray(*cam, pos[i], pos[n + i], &oo, &dd, &tmax[i]);
        o[i] = oo.x; o[n + i] = oo.y; o[2 * n + i] = oo.z;
        d[i] = dd.x; d[n + i] = dd.y; d[2 * n + i] = dd.z;
    }
    return PBRT_OK;
}

// UltraSensor.sample_ray (class recovered from bytecode)
int oracle_us_sensor_sample_ray(const pbrt_us_sensor *s, int use_hemisphere_warp, uint32_t n, const float *time,
                                const float *wavelength_sample, const float *position_sample,
                                const float *aperture_sample, float *o, float *d, float *weight) {
    const float N = (float)s->num_elements;
    for (uint32_t i = 0; i < n; ++i) {
        float px = position_sample[i], py = position_sample[n + i];
        float ax = aperture_sample[i], ay = aperture_sample[n + i];
        float ei = fminf(floorf(px * N), N - 1.0f);
        float ex, ez;
        if (std::isinf(s->radius)) {
            ex = fmaf(ei, s->pitch, -((N - 1.0f) * s->pitch) / 2.0f);
            ez = 0.0f;
        } else {
            float th = (ei - N / 2.0f) * (s->pitch / s->radius);
            ex = s->radius * sinf(th);
            ez = s->radius * (1.0f - cosf(th));
        }
        float offx = (ax - 0.5f) * s->element_width, offy = (ay - 0.5f) * s->element_height;
        V3 ol = {ex + offx, offy, ez};
        V3 dl;
        if (use_hemisphere_warp) {
            dl = square_to_uniform_hemisphere(ax, ay);
        } else {
            float phi = 2.0f * kPi * py, ct = wavelength_sample[i];
            float st = sqrtf(fmaxf(0.0f, 1.0f - ct * ct));
            dl = {st * cosf(phi), st * sinf(phi), ct};
        }
        V3 ow = xf_point(s->to_world, ol);
        V3 dw = normalize(xf_vec(s->to_world, dl));
        float dweight = fabsf(dl.z) * s->directivity;
        weight[i] = cosf(2.0f * kPi * s->center_frequency * time[i]) * dweight;
        o[i] = ow.x; o[n + i] = ow.y; o[2 * n + i] = ow.z;
        d[i] = dw.x; d[n + i] = dw.y; d[2 * n + i] = dw.z;
    }
    return PBRT_OK;
}

// CustomEmitter.sample_position + sample_ray (CustomEmmitter.py:30-107)
int oracle_us_emitter_sample_ray(const pbrt_us_emitter *e, uint32_t n, const float *time, const float *s1,
                                 const float *s2, const float *s3, float *o, float *d, float *ray_time, float *weight,
                                 float *pdf_pos) {
    for (uint32_t i = 0; i < n; ++i) {
        const EmitRay r = emitter_ray(*e, time[i], s1[i], s2[i], s2[n + i], s3[i]);
        pdf_pos[i] = r.pdf_pos;
        ray_time[i] = r.time;
        weight[i] = r.weight;
        o[i] = r.o.x; o[n + i] = r.o.y; o[2 * n + i] = r.o.z;
        d[i] = r.d.x; d[n + i] = r.d.y; d[2 * n + i] = r.d.z;
    }
    return PBRT_OK;
}

// CustomSensor.put_data (CustomSensor.py:29-59), sequential in ray order
int oracle_us_put_data(const pbrt_us_receiver *r, uint32_t n, const float *ox, const float *time, const float *d,
                       const float *amplitude, float *channel_buffer) {
    for (uint32_t i = 0; i < n; ++i) {
        // :36 int(np.round(x / pitch + N / 2))  (float64 arithmetic on a float32 x; half-to-even)
        double idx = nearbyint((double)ox[i] / (double)r->pitch + (double)r->number_of_elements / 2.0);
        double ti = nearbyint((double)time[i] * (double)r->sample_rate);                      // :43
        V3 dir = normalize(-v3(d[i], d[n + i], d[2 * n + i]));                                // :46
        float gain = fmaxf(0.0f, dot(dir, v3(0, 0, 1)));                                      // :51
        float amp = amplitude[i] * gain;                                                      // :53
        if (idx >= 0 && idx < (double)r->number_of_elements && ti >= 0 && ti < (double)r->time_samples)  // :58
            channel_buffer[(size_t)idx * r->time_samples + (size_t)ti] += amp;                // :59
    }
    return PBRT_OK;
}

// sampling_test.py:3-23 (inverse-CDF GGX angle, degrees) and :25-43 (D*sin, un-normalised)
void oracle_ggx_angle_deg(double alpha, uint32_t n, const double *xi, double *theta_deg) {
    for (uint32_t i = 0; i < n; ++i) {
        double c = sqrt((1.0 - xi[i]) / (1.0 + (alpha * alpha - 1.0) * xi[i]));
        theta_deg[i] = acos(c) * 180.0 / M_PI;
    }
}
void oracle_ggx_pdf_raw(double alpha, uint32_t n, const double *theta_deg, double *pdf) {
    for (uint32_t i = 0; i < n; ++i) {
        double t = theta_deg[i] * M_PI / 180.0, c = cos(t), s = sin(t);
        double den = (alpha * alpha - 1.0) * c * c + 1.0;
        pdf[i] = alpha * alpha / (M_PI * den * den) * s;
    }
}
// scalar helpers for the K3/K4/K5 fixtures
float oracle_us_attenuation(float attenuation, float frequency, float distance) {
    float katt = (float)(-(double)attenuation * (double)frequency * 1e-6);
    return expf(katt * distance / 8.686f);
}
float oracle_us_directivity_i(float angle_deg, float main_deg, float cutoff_deg) {
    float a = angle_deg * (kPi / 180.0f);
    V3 sec = -v3(sinf(a), 0.0f, cosf(a));  // direction whose reverse makes `angle` with +z
    return directivity_weight_i(sec, v3(0, 0, 1), main_deg * (kPi / 180.0f), cutoff_deg * (kPi / 180.0f));
}
// Ar, At, Ar^2, pdf_reflect, tir for a micro-normal aligned so that |m.wi| = cosTr (K5)
void oracle_us_impedance(float Z1, float Z2, float cosTr, float *out5) {
    float ratio = Z1 / Z2;
    float sqrt_arg = fmaf(-(ratio * ratio), fmaf(-cosTr, cosTr, 1.0f), 1.0f);
    float cosTt = sqrtf(fmaxf(sqrt_arg, 0.0f));
    float Ar = fmaf(Z1, cosTr, -(Z2 * cosTt)) / fmaf(Z1, cosTr, Z2 * cosTt);
    out5[0] = Ar;
    out5[1] = 1.0f - Ar;
    out5[2] = Ar * Ar;
    out5[3] = 1.0f / (4.0f * cosTr);
    out5[4] = sqrt_arg < 0.0f ? 1.0f : 0.0f;
}

}  // extern "C"

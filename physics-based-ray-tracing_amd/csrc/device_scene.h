// device_scene.h -- scene access, intersection, BSDFs, emitter sampling and the perspective
// sensor on the device.  Semantics: Mitsuba 3 built-ins for radiance mode,
// the reference's own UltraBSDF arithmetic (CustomBSDF.py:30-175) for ultrasound mode.
#pragma once
#include "../../include/pbrt_hip.h"
#include "device_math.h"

// BVH4 inner node, 64 bytes (bvh_build.h HostNode4): lower corner of the node's box, a power-of-two grid step per axis and
// the boxes of FOUR children on that 8-bit grid, rounded outward -- one node read decides four descents.
// Child reference c: bit 31 set = leaf, bits 27..30 = primitive count (0..15; 0 = empty slot), bits 0..26 = first leaf record;
// bit 31 clear = index of an inner node.  Traversal keeps a short per-lane stack of node entries in LDS (BvhStack).
// This is the record in GLOBAL memory (one 64-byte read per node through the vector caches).  The LDS image of a small tree
// holds the same four quads as PLANES (TreeLds below).
struct DevNode4 {
    float org[3];
    uint32_t exps;      // biased exponents of the grid step: x | y << 8 | z << 16; bits 24..25: split axis of the collapsed BVH2 node
    uint32_t child[4];
    uint32_t qlo[3];    // qlo[axis]: byte k = lower plane of child k
    uint32_t qhi[3];
    uint32_t pad[2];
};
// Leaf record, 40 bytes: what a primitive TEST reads (v0 / e1 / e2; sphere: centre + radius), type << 28 | caller's index.
// Normal, material and emitter of the primitive that was hit come from the full 64-byte table (prims_by_id) afterwards.
struct DevLeafPrim {
    float g[9];
    uint32_t meta;
};
// Pointers into the LDS carry their address space in the TYPE: an access through them can only be a ds_* instruction, and it
// cannot be merged with an access to private (scratch) or global memory into one FLAT instruction through a selected pointer --
// FLAT accesses to LDS are not ordered against the DS instructions around them (round 3: a traversal stack whose LDS rows and
// scratch overflow had been merged that way cycled; tests/test_asm_address_spaces.py checks the compiled kernels).
#define LDS_AS __attribute__((address_space(3)))
#define PRIV_AS __attribute__((address_space(5)))
typedef uint32_t __attribute__((ext_vector_type(4))) u32x4;
typedef uint32_t __attribute__((ext_vector_type(2))) u32x2;
struct NodeQ {   // the four quads of a node: (org, exps) | child[4] | (qlo[3], qhi[0]) | (qhi[1], qhi[2])
    u32x4 w0, w1, w2;
    u32x2 w3;
};
struct LeafQ {   // the 40 bytes of a leaf record
    u32x2 a, b, c, d, e;
};
// The tree as a kernel reads it.  TreeGlobal: nodes and leaf records through the vector caches.
struct TreeGlobal {
    const DevNode4 *nodes;
    const DevLeafPrim *lprims;
    DEV NodeQ node(uint32_t n) const {
        const DevNode4 *np = nodes + n;
        NodeQ q;
        q.w0 = *reinterpret_cast<const u32x4 *>(&np->org[0]);
        q.w1 = *reinterpret_cast<const u32x4 *>(&np->child[0]);
        q.w2 = *reinterpret_cast<const u32x4 *>(&np->qlo[0]);
        q.w3 = *reinterpret_cast<const u32x2 *>(&np->qhi[1]);
        return q;
    }
    DEV NodeQ node_boxes(uint32_t n) const {  // without the child references (w1 undefined)
        const DevNode4 *np = nodes + n;
        NodeQ q;
        q.w0 = *reinterpret_cast<const u32x4 *>(&np->org[0]);
        q.w1 = q.w0;
        q.w2 = *reinterpret_cast<const u32x4 *>(&np->qlo[0]);
        q.w3 = *reinterpret_cast<const u32x2 *>(&np->qhi[1]);
        return q;
    }
    DEV uint32_t child(uint32_t n, uint32_t slot) const { return nodes[n].child[slot]; }
    DEV LeafQ leaf(uint32_t s) const {
        const u32x2 *q = reinterpret_cast<const u32x2 *>(&lprims[s]);
        return {q[0], q[1], q[2], q[3], q[4]};
    }
};
// TreeLds: the image a workgroup staged into its LDS (kernels_radiance.h stage_tree_lds), n = nodes of the tree:
//   [plane 0: (org, exps)  16 B x n][plane 1: child[4]  16 B x n][plane 2: (qlo, qhi.x)  16 B x n][plane 3: (qhi.y, qhi.z)  8 B x n]
//   [leaf records  40 B x primitives]
// Planes instead of the 64-byte records because of the banks: a ds_read_b128 is served in groups of 16 lanes over 64 banks, and
// quad k of a 64-byte record n starts at bank 16 (n mod 4) + 4 k -- only FOUR bank sets for the 16 lanes of a group, whatever
// nodes they read (35 % of the LDS cycles of round 3's k_trace were conflicts).  Quad k of node n in a plane starts at bank
// 4 n mod 64: sixteen sets, the most a 16-byte read can have; the 4-byte read of one child reference (bvh_pop) goes from 8 to 32
// banks.  56 instead of 64 bytes per node on top (TestRing: 65.4 -> 62.8 KB, which makes room for a fourth stack row).
struct TreeLds {
    const LDS_AS uint32_t *img;
    uint32_t plane;     // bytes of a 16-byte plane: 16 n
    uint32_t leaf_off;  // byte offset of the leaf records: 56 n
    DEV NodeQ node(uint32_t n) const {
        const LDS_AS char *b = reinterpret_cast<const LDS_AS char *>(img);
        const uint32_t a = n * 16u;
        NodeQ q;
        q.w0 = *reinterpret_cast<const LDS_AS u32x4 *>(b + a);
        q.w1 = *reinterpret_cast<const LDS_AS u32x4 *>(b + plane + a);
        q.w2 = *reinterpret_cast<const LDS_AS u32x4 *>(b + 2u * plane + a);
        q.w3 = *reinterpret_cast<const LDS_AS u32x2 *>(b + 3u * plane + n * 8u);
        return q;
    }
    DEV NodeQ node_boxes(uint32_t n) const {
        const LDS_AS char *b = reinterpret_cast<const LDS_AS char *>(img);
        const uint32_t a = n * 16u;
        NodeQ q;
        q.w0 = *reinterpret_cast<const LDS_AS u32x4 *>(b + a);
        q.w1 = q.w0;
        q.w2 = *reinterpret_cast<const LDS_AS u32x4 *>(b + 2u * plane + a);
        q.w3 = *reinterpret_cast<const LDS_AS u32x2 *>(b + 3u * plane + n * 8u);
        return q;
    }
    DEV uint32_t child(uint32_t n, uint32_t slot) const {
        const LDS_AS char *b = reinterpret_cast<const LDS_AS char *>(img);
        return *reinterpret_cast<const LDS_AS uint32_t *>(b + plane + n * 16u + slot * 4u);
    }
    DEV LeafQ leaf(uint32_t s) const {
        const LDS_AS u32x2 *q = reinterpret_cast<const LDS_AS u32x2 *>(reinterpret_cast<const LDS_AS char *>(img) + leaf_off + s * 40u);
        return {q[0], q[1], q[2], q[3], q[4]};
    }
};
#define LDS_IMAGE_NODE_BYTES 56u
#define BVH_LEAF 0x80000000u
#define BVH_SENT 0xffffffffu  // bottom of the traversal stack

struct DevScene {
    const pbrt_prim *prims;  // caller order (== prims_by_id); Hit::slot indexes it
    const DevLeafPrim *lprims;  // BVH: the leaf records in leaf order
    const DevNode4 *nodes;
    const pbrt_material *mats;
    const pbrt_emitter *emitters;
    const uint32_t *light_prims;  // caller's primitive indices
    const float *light_cdf;
    const pbrt_prim *prims_by_id;  // caller order (== prims for BRUTE); used by emitter sampling
    // BRUTE only: the primitives that can occlude a SEGMENT between two points of the scene (copies, any
    // order).  Planar primitives on a supporting plane of the scene's convex hull (all other geometry and
    // every point emitter on one side) can never be crossed by such a segment and are left out, e.g. the
    // five walls of the Cornell box (pbrt_api.hip: find_occluders).  Unbounded occlusion rays (ultrasound
    // mode, pbrt_ray_test) always walk the full list.
    const pbrt_prim *occ_prims;
    uint32_t n_occ;
#ifdef PBRT_BRUTE_PAIRS
    const struct PairItem *pair_items;  // ACCEL_K_BRUTE: sc.prims as pairs of planar primitives (brute_closest_pairs)
    uint32_t n_pair_items;
#endif
    // optional: the vertex normals of mesh primitives, [n_prims][9], caller order (indexed by Hit::slot); nullptr: face normals
    const float *vnormals;
    uint32_t n_prims, n_nodes, n_emitters, n_mats, n_light_prims;
};

struct Hit {
    float t, u, v;
    uint32_t prim;  // caller's primitive index
    uint32_t slot;  // index into sc.prims
};

DEV V3 g3(const pbrt_prim &P, int i) { return {P.g[i], P.g[i + 1], P.g[i + 2]}; }

// Analytic cone ('cone' shapes of MitsubaScenes/Cone_Box.xml:36-47; [DEFINE] D8): the ray goes to object space
// through the primitive's world -> object matrix (t is preserved, the direction is not re-normalised), there the
// closed unit cone is the quadric x^2 + y^2 = (1 - z)^2 cut to 0 <= z <= 1 plus the base disc z = 0, r <= 1.
// Roots by the cancellation-free form q = -(b + sign(b) sqrt(disc)), t = q / A, C / q (a ray parallel to the
// lateral surface has A = 0: q / A is +-inf or NaN and fails the range tests, C / q is its one crossing).
// *flag = 0: lateral surface, 1: base disc.  Identical arithmetic in the oracle's cone_hit.
DEV bool cone_hit(const pbrt_prim &P, V3 o, V3 d, float tmax, float *t, float *flag) {
    const V3 r0 = g3(P, 0), r1 = g3(P, 4), r2 = g3(P, 8);
    const V3 oo = {dot(r0, o) + P.g[3], dot(r1, o) + P.g[7], dot(r2, o) + P.g[11]};
    const V3 dd = {dot(r0, d), dot(r1, d), dot(r2, d)};
    const float ow = 1.0f - oo.z;  // w = 1 - z, dw = -dz
    const float A = fma_(dd.x, dd.x, fma_(dd.y, dd.y, -(dd.z * dd.z)));
    const float b = fma_(oo.x, dd.x, fma_(oo.y, dd.y, ow * dd.z));
    const float C = fma_(oo.x, oo.x, fma_(oo.y, oo.y, -(ow * ow)));
    const float disc = fma_(b, b, -(A * C));
    float best = tmax;
    float fl = 0.0f;
    bool found = false;
    if (disc >= 0.0f) {
        const float q = -(b + copysignf(sqrtf(disc), b));
        const float ta = q / A, tb = C / q;
        const float za = fma_(ta, dd.z, oo.z), zb = fma_(tb, dd.z, oo.z);
        if (ta >= 0.0f && ta <= best && za >= 0.0f && za <= 1.0f) {
            best = ta;
            found = true;
        }
        if (tb >= 0.0f && tb <= best && zb >= 0.0f && zb <= 1.0f && (!found || tb < best)) {
            best = tb;
            found = true;
        }
    }
    const float tc = -oo.z / dd.z;  // base plane; dz = 0: +-inf or NaN, rejected below
    const float x = fma_(tc, dd.x, oo.x), y = fma_(tc, dd.y, oo.y);
    if (tc >= 0.0f && tc <= best && fma_(x, x, y * y) <= 1.0f && (!found || tc < best)) {
        best = tc;
        fl = 1.0f;
        found = true;
    }
    *t = best;
    *flag = fl;
    return found;
}

// One primitive against one ray; identical arithmetic to the oracle's prim_hit (Mitsuba
// Mesh::ray_intersect_triangle / Sphere / Rectangle reached via scene.ray_intersect,
// CustomIntegrator.py:309).  The barycentric test runs on det-scaled values so that the division
// is only executed for accepted candidates.  The per-type forms take the geometry as plain vectors: the 64-byte
// records of the brute-force loop and the 40-byte leaf records of the BVH go through the same arithmetic.
DEV bool sphere_hit(V3 c, float r, V3 o, V3 d, float tmax, float *t) {
    V3 f = o - c;
    float bp = -dot(f, d);
    V3 perp = madd(d, bp, f);
    float disc = fma_(r, r, -dot(perp, perp));
    if (!(disc >= 0.0f)) return false;
    float sq = sqrtf(disc);
    float q = bp + copysignf(sq, bp);
    float cc = fma_(-r, r, dot(f, f));
    float t0 = cc / q, t1 = q;
    float tn = fminf(t0, t1), tf = fmaxf(t0, t1);
    if (!(tn <= tmax && tf >= 0.0f)) return false;
    if (tn < 0.0f && tf > tmax) return false;
    *t = tn < 0.0f ? tf : tn;
    return true;
}
DEV bool planar_hit(bool triangle, V3 v0, V3 e1, V3 e2, V3 o, V3 d, float tmax, float *t, float *u, float *v) {
    V3 pvec = cross(d, e2);
    float det = dot(e1, pvec);
    V3 tvec = o - v0;
    V3 qvec = cross(tvec, e1);
    float us = dot(tvec, pvec), vs = dot(d, qvec), ts = dot(e2, qvec);
    // det < 0: all four change sign.  As a sign transfer (x ^ sign bit of det) instead of a branch; -0 becomes +0 and fails
    // det > 0 like before, a NaN stays one.
    const uint32_t sgn = __float_as_uint(det) & 0x80000000u;
    det = __uint_as_float(__float_as_uint(det) ^ sgn);
    us = __uint_as_float(__float_as_uint(us) ^ sgn);
    vs = __uint_as_float(__float_as_uint(vs) ^ sgn);
    ts = __uint_as_float(__float_as_uint(ts) ^ sgn);
    // the acceptance tests without short circuits: five compares and their conjunction, no control flow (the BVH stream kernels
    // are bound by instruction issue, and every divergent branch costs three to five scalar instructions)
    const float lim = triangle ? us + vs : fmaxf(us, vs);  // (both <= det  <=>  the larger one is; NaNs have failed x >= 0)
    const bool ok = (det > 0.0f) & (us >= 0.0f) & (vs >= 0.0f) & (ts >= 0.0f) & (lim <= det);
    if (!ok) return false;
    float inv = 1.0f / det;
    float tt = ts * inv;
    if (!(tt <= tmax)) return false;
    *t = tt;
    *u = us * inv;
    *v = vs * inv;
    return true;
}
DEV bool prim_hit(const pbrt_prim &P, V3 o, V3 d, float tmax, float *t, float *u, float *v) {
    const uint32_t type = P.type;
    if (type == PBRT_PRIM_SPHERE) {
        *u = 0.0f;
        *v = 0.0f;
        return sphere_hit(g3(P, 0), P.g[3], o, d, tmax, t);
    }
    if (type == PBRT_PRIM_TRIANGLE || type == PBRT_PRIM_PARALLELOGRAM)
        return planar_hit(type == PBRT_PRIM_TRIANGLE, g3(P, 0), g3(P, 3), g3(P, 6), o, d, tmax, t, u, v);
    if (type == PBRT_PRIM_CONE) {
        *v = 0.0f;
        return cone_hit(P, o, d, tmax, t, u);
    }
    return false;
}

// ---- brute force: every lane walks the same primitive sequence, so the records come through the
// scalar cache (s_load) and cost no vector memory traffic -----------------------------------------
//
// Candidates are ratios t = num / den (den > 0): the range test is ts <= tmax * det, candidates are
// ranked by cross-multiplication and the one division happens after the loop, so the loop body of a
// triangle / parallelogram is straight-line code (selects, no divergent branch, no division).
// The records are read through the constant address space: a uniform load from it is always selected as
// s_load, independent of the compiler's "is this memory clobbered earlier in the kernel" analysis (any barrier,
// fence or inline asm ahead of the loop would otherwise turn the 16 dwords of every record into per-lane
// global loads held in 16 VGPRs).  The scene is immutable while a kernel runs, so the promise holds.
DEV pbrt_prim load_prim_uniform(const pbrt_prim *p) {
    typedef uint32_t __attribute__((ext_vector_type(4))) u32x4;
    typedef const u32x4 __attribute__((address_space(4))) *cptr;
    struct Raw {
        u32x4 q[4];
    };
    static_assert(sizeof(Raw) == sizeof(pbrt_prim), "pbrt_prim is 16 dwords");
    cptr q = (cptr)(uintptr_t)p;
    Raw r = {{q[0], q[1], q[2], q[3]}};
    return __builtin_bit_cast(pbrt_prim, r);
}

// CONES: the kernel variant understands PBRT_PRIM_CONE records.  Compile-time, because the cone branch costs the
// 64-VGPR brute-force radiance kernel 9 VGPRs and puts it into scratch (measured: 54 / 60 -> 63 VGPRs + 4 spilled);
// small scenes with a cone run the ACCEL_K_BRUTE_BIG variant instead (pbrt_api.hip).
template <bool ANY, bool SEGMENT = false, bool CONES = true>
DEV bool brute_intersect(const DevScene &sc, V3 o, V3 d, float tmax, Hit *h) {
    bool found = false;
    float bn = 0.0f, bd = 1.0f, bu = 0.0f, bv = 0.0f;
    uint32_t bp = 0xffffffffu;
    const pbrt_prim *list = (ANY && SEGMENT) ? sc.occ_prims : sc.prims;
    // readfirstlane: keeps the loop counter and the record address in SGPRs (s_min / s_lshl / s_add) -- without it
    // the compiler carries n_list - 1 in a VGPR and spends 5 VALU + 2 readfirstlane per record on the address
    const uint32_t n_list = (uint32_t)__builtin_amdgcn_readfirstlane((int)((ANY && SEGMENT) ? sc.n_occ : sc.n_prims));
    if (n_list == 0) return false;
    // software pipeline over the (wave-uniform) primitive records: the 64-byte scalar load of primitive
    // i + 1 is in flight while primitive i is tested
#ifdef PBRT_BRUTE_NO_PREFETCH  // A/B (round 5): one record in scalar registers instead of two (16 SGPRs for kernels that spill them)
    for (uint32_t i = 0; i < n_list; ++i) {
        const pbrt_prim P = load_prim_uniform(list + i);
#else
    pbrt_prim nxt = load_prim_uniform(list);
    for (uint32_t i = 0; i < n_list; ++i) {
        const pbrt_prim P = nxt;
        nxt = load_prim_uniform(list + min(i + 1, n_list - 1));
#endif
        const uint32_t type = P.type;  // wave-uniform
        bool ok;
        float num, den, us, vs;
        if (type == PBRT_PRIM_SPHERE || (CONES && type == PBRT_PRIM_CONE)) {  // curved primitives report (t, 1)
            float t, u, v;
            ok = prim_hit(P, o, d, tmax, &t, &u, &v);
            num = t;
            den = 1.0f;
            us = u;  // 0 for spheres; cone: 0 lateral surface / 1 base disc (den = 1: passes through unscaled)
            vs = 0.0f;
        } else {
            V3 v0 = g3(P, 0), e1 = g3(P, 3), e2 = g3(P, 6);
            V3 pvec = cross(d, e2);
            float det = dot(e1, pvec);
            V3 tvec = o - v0;
            V3 qvec = cross(tvec, e1);
            us = dot(tvec, pvec);
            vs = dot(d, qvec);
            float ts = dot(e2, qvec);
            const bool neg = det < 0.0f;
            det = neg ? -det : det;
            us = neg ? -us : us;
            vs = neg ? -vs : vs;
            ts = neg ? -ts : ts;
            // (us >= 0 & vs >= 0 & ts >= 0) and (us <= det & vs <= det) folded into min3 / max: same truth
            // value for finite operands (all operands are finite here), three compares less per primitive
            ok = (det > 0.0f) & (fminf(fminf(us, vs), ts) >= 0.0f) & (ts <= tmax * det);
            if (type == PBRT_PRIM_TRIANGLE)
                ok = ok & (us + vs <= det);
            else
                ok = ok & (fmaxf(us, vs) <= det);
            num = ts;
            den = det;
        }
        if (ANY) {
            found = found | ok;
            if (__builtin_amdgcn_ballot_w64(!found) == 0) break;  // every active lane is occluded
        } else {
            const bool better = ok & (!found | (num * bd < bn * den));
            bn = better ? num : bn;
            bd = better ? den : bd;
            bu = better ? us : bu;
            bv = better ? vs : bv;
            bp = better ? i : bp;
            found = found | better;
        }
    }
    if (!ANY && found) {
        float inv = 1.0f / bd;
        h->t = bn * inv;
        h->u = bu * inv;
        h->v = bv * inv;
        h->prim = bp;
        h->slot = bp;
    }
    return found;
}

// ---- closest hit, two planar primitives per iteration (ACCEL_K_BRUTE): diagnostic build -DPBRT_BRUTE_PAIRS ------
// Measured: bit-identical film, but 8.77 against 8.33 ms on cbox -- the 27 packed instructions replace 48, yet the
// compare / select / ranking part of each primitive grows (the `found` flag moves into a VGPR, 20-26 SGPRs of the
// 2 x 32-dword records spill), 85 against 92 VALU per pair in the end, and the 128-byte records double the scalar
// loads.  Kept as a build switch; the shipped library uses brute_intersect.
#ifdef PBRT_BRUTE_PAIRS
// The kernel is instruction-issue bound and the Moeller-Trumbore set-up is 27 multiply-adds per primitive.  The host
// interleaves the v0 / e1 / e2 of two consecutive planar primitives into one 128-byte record (pbrt_api.hip
// build_pair_items), so that every one of those operations is ONE packed instruction for both (v_pk_mul_f32 /
// v_pk_fma_f32 / v_pk_add_f32, the record in SGPR pairs): 27 instead of 48 VALU per pair.  Packed f32 is IEEE per
// component and the operation order is that of cross() / dot() above, candidates are ranked A before B, so the
// result is the same bit for bit as brute_intersect<false>.  Spheres (kind 1) take the scalar path.
typedef float __attribute__((ext_vector_type(2))) f2;
struct PairItem {         // 32 dwords, wave-uniform
    uint32_t dw[32];      // kind 0: dw[2k], dw[2k+1] = g[k] of A, B for k = 0..8 | kind 1: dw[0..15] = the pbrt_prim
};                        // dw[18] typeA, dw[19] typeB (0xffffffff: none), dw[20] kind, dw[21] index of A in sc.prims
DEV PairItem load_pair_uniform(const PairItem *p) {
    typedef uint32_t __attribute__((ext_vector_type(4))) u32x4;
    typedef const u32x4 __attribute__((address_space(4))) *cptr;
    struct Raw {
        u32x4 q[8];
    };
    static_assert(sizeof(Raw) == sizeof(PairItem), "PairItem is 32 dwords");
    cptr q = (cptr)(uintptr_t)p;
    Raw r = {{q[0], q[1], q[2], q[3], q[4], q[5], q[6], q[7]}};
    return __builtin_bit_cast(PairItem, r);
}
DEV f2 pk_fma(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }
DEV f2 pk_splat(float x) { return {x, x}; }

DEV bool brute_closest_pairs(const DevScene &sc, V3 o, V3 d, float tmax, Hit *h) {
    bool found = false;
    float bn = 0.0f, bd = 1.0f, bu = 0.0f, bv = 0.0f;
    uint32_t bp = 0xffffffffu;
    const uint32_t n_items = (uint32_t)__builtin_amdgcn_readfirstlane((int)sc.n_pair_items);
    if (n_items == 0) return false;
    auto rank = [&](bool ok, float num, float den, float us, float vs, uint32_t idx) {
        const bool better = ok & (!found | (num * bd < bn * den));
        bn = better ? num : bn;
        bd = better ? den : bd;
        bu = better ? us : bu;
        bv = better ? vs : bv;
        bp = better ? idx : bp;
        found = found | better;
    };
    auto planar = [&](float det, float us, float vs, float ts, uint32_t type, uint32_t idx) {
        const bool neg = det < 0.0f;
        det = neg ? -det : det;
        us = neg ? -us : us;
        vs = neg ? -vs : vs;
        ts = neg ? -ts : ts;
        bool ok = (det > 0.0f) & (fminf(fminf(us, vs), ts) >= 0.0f) & (ts <= tmax * det);
        if (type == PBRT_PRIM_TRIANGLE)
            ok = ok & (us + vs <= det);
        else
            ok = ok & (fmaxf(us, vs) <= det);
        rank(ok, ts, det, us, vs, idx);
    };
    const f2 dx = pk_splat(d.x), dy = pk_splat(d.y), dz = pk_splat(d.z);
    const f2 ox = pk_splat(o.x), oy = pk_splat(o.y), oz = pk_splat(o.z);
    PairItem nxt = load_pair_uniform(sc.pair_items);
    for (uint32_t i = 0; i < n_items; ++i) {
        const PairItem R = nxt;
        nxt = load_pair_uniform(sc.pair_items + min(i + 1, n_items - 1));
        const uint32_t idA = R.dw[21];
        if (R.dw[20] != 0u) {  // one primitive of any type: the scalar test
            struct Half {
                uint32_t w[16];
            };
            Half hw;
#pragma unroll
            for (int k = 0; k < 16; ++k) hw.w[k] = R.dw[k];
            const pbrt_prim P = __builtin_bit_cast(pbrt_prim, hw);
            __builtin_assume(P.type == PBRT_PRIM_SPHERE);  // planar primitives travel in pairs, cones never reach this variant
            float t, u, v;
            const bool ok = prim_hit(P, o, d, tmax, &t, &u, &v);
            rank(ok, t, 1.0f, u, 0.0f, idA);
            continue;
        }
        auto g2 = [&](int k) -> f2 { return {__uint_as_float(R.dw[2 * k]), __uint_as_float(R.dw[2 * k + 1])}; };
        const f2 v0x = g2(0), v0y = g2(1), v0z = g2(2), e1x = g2(3), e1y = g2(4), e1z = g2(5), e2x = g2(6), e2y = g2(7), e2z = g2(8);
        // pvec = cross(d, e2)
        const f2 px = pk_fma(dy, e2z, -(dz * e2y)), py = pk_fma(dz, e2x, -(dx * e2z)), pz = pk_fma(dx, e2y, -(dy * e2x));
        const f2 det = pk_fma(e1x, px, pk_fma(e1y, py, e1z * pz));                    // dot(e1, pvec)
        const f2 tx = ox - v0x, ty = oy - v0y, tz = oz - v0z;                           // tvec = o - v0
        // qvec = cross(tvec, e1)
        const f2 qx = pk_fma(ty, e1z, -(tz * e1y)), qy = pk_fma(tz, e1x, -(tx * e1z)), qz = pk_fma(tx, e1y, -(ty * e1x));
        const f2 us = pk_fma(tx, px, pk_fma(ty, py, tz * pz));                        // dot(tvec, pvec)
        const f2 vs = pk_fma(dx, qx, pk_fma(dy, qy, dz * qz));                        // dot(d, qvec)
        const f2 ts = pk_fma(e2x, qx, pk_fma(e2y, qy, e2z * qz));                     // dot(e2, qvec)
        planar(det.x, us.x, vs.x, ts.x, R.dw[18], idA);
        if (R.dw[19] != 0xffffffffu) planar(det.y, us.y, vs.y, ts.y, R.dw[19], idA + 1u);
    }
    if (found) {
        float inv = 1.0f / bd;
        h->t = bn * inv;
        h->u = bu * inv;
        h->v = bv * inv;
        h->prim = bp;
        h->slot = bp;
    }
    return found;
}
#endif  // PBRT_BRUTE_PAIRS

// ---- BVH4 traversal; NodeP / PrimP are global or LDS pointers ------------------------------------
// Box tests are conservative (boxes are padded at build time and rounded outward onto the node's grid, reciprocal
// directions are approximate); only the primitive tests decide, and ties in t go to the lowest primitive id, so the result
// does not depend on the tree or on the visiting order.
//
// Slab form: a child plane sits at org + q * step (q = 0..255), so  t = (org + q step - o) / d = q * A + B  with
// A = step * (1/d) and B = (org - o) * (1/d) per node and axis: one v_cvt_f32_ubyte + one fma per plane.  The
// rounding error is that of moving the plane by a few 2^-23 * (|plane| + |o|), two orders of magnitude inside the
// builder's padding (2e-5 * scene size, bvh_build.h), so a box the exact test accepts is never culled.  A zero direction
// component would make the products infinite (inf - inf = NaN hides the slab, but -inf - inf = -inf culls a box the ray is
// inside of), so components below 1e-18 are replaced by +-1e-18 for the box tests only: all products stay finite and the
// sign of (plane - o) survives for every plane further than the padding from the ray.
struct BoxRay {
    V3 o, inv;  // origin (shared with the primitive tests), 1 / d
};
DEV BoxRay make_box_ray(V3 o, V3 d) {
    const float tiny = 1e-18f;
    const V3 ds = {fabsf(d.x) < tiny ? copysignf(tiny, d.x) : d.x, fabsf(d.y) < tiny ? copysignf(tiny, d.y) : d.y,
                   fabsf(d.z) < tiny ? copysignf(tiny, d.z) : d.z};
    return {o, {__builtin_amdgcn_rcpf(ds.x), __builtin_amdgcn_rcpf(ds.y), __builtin_amdgcn_rcpf(ds.z)}};
}

// Per-lane traversal stack.  An entry stands for ONE node whose other hit children are still to be visited:
//   node index << 8 | count (1..3) << 6 | slots of those children in the order of their entry distances, 2 bits each, the
//   next one in bits 1..0
// so the stack is never deeper than the tree (one entry per level), and taking the next child of the newest entry is register
// arithmetic plus one 4-byte read of the node's child reference.  The newest entry lives in a register (`tos`), rows
// 0 .. n_rows - 1 of an LDS array [row][thread] (threads a power of two: the row offset is a shift) hold the next ones,
// anything deeper goes to a private array in scratch memory.  The two live in different address spaces BY TYPE (LDS_AS /
// PRIV_AS): the rows can only be reached by ds_* instructions, the overflow only by scratch_* ones, and no pointer can stand
// for both.  The bottom of the stack is BVH_SENT: the first push stores it in row 0, the last pop brings it back.  The host
// checks depth <= BVH_STK_MAX and nodes < 2^24.
#define BVH_STK_OVF 30
#define BVH_STK_MAX (BVH_STK_OVF + 2)  // guaranteed capacity whatever n_rows is (>= 2 rows are always there)
struct BvhStack {
    LDS_AS uint32_t *col;  // this thread's column: entry of row r at col[r << shift]; nullptr: brute-force kernels
    uint32_t shift;        // log2(threads of the workgroup)
    uint32_t n_rows;       // LDS rows (>= 2)
};
struct BvhCursor {
    uint32_t cur, tos, sp;
};
struct BvhOvf {
    uint32_t m[BVH_STK_OVF];
    DEV PRIV_AS uint32_t *at(uint32_t i) { return (PRIV_AS uint32_t *)(&m[0]) + min(i, (uint32_t)BVH_STK_OVF - 1u); }
};
DEV void bvh_push(const BvhStack &st, BvhCursor &c, BvhOvf &ovf, bool on, uint32_t entry) {
    if (on) {
        if (c.sp < st.n_rows)
            st.col[c.sp << st.shift] = c.tos;
        else
            *ovf.at(c.sp - st.n_rows) = c.tos;
        c.sp += 1u;
        c.tos = entry;
    }
}
// the next child reference off the stack (BVH_SENT: the traversal has finished)
template <typename Tree>
DEV uint32_t bvh_pop(const Tree &tr, const BvhStack &st, BvhCursor &c, BvhOvf &ovf) {
    const uint32_t e = c.tos;
    if (e == BVH_SENT) return BVH_SENT;
    const uint32_t ref = tr.child(e >> 8, e & 3u);
    const uint32_t n = (e >> 6) & 3u;
    if (n > 1u) {
        c.tos = (e & 0xffffff00u) | ((n - 1u) << 6) | ((e & 0x3fu) >> 2);
    } else {  // the entry is used up: the next one comes off the rows (nothing reads it before the next pop or push)
        const uint32_t sp1 = c.sp - 1u;  // (sp >= 1: BVH_SENT lies below every entry)
        if (sp1 < st.n_rows)
            c.tos = st.col[sp1 << st.shift];
        else
            c.tos = *ovf.at(sp1 - st.n_rows);
        c.sp = sp1;
    }
    return ref;
}

// The four child boxes of a node against a ray: key[k] = entry distance of child k with the slot k in its two lowest bits
// (tn >= 0, so the keys are ordered like the floats); a miss is 0xffffffff, or 0xfffffffc | k with MISS_SLOT (the packet walk
// orders the children by ONE lane's keys and needs the slots of the children that lane misses).
template <bool MISS_SLOT = false>
DEV void bvh_child_keys(const NodeQ &q, const BoxRay &r, float best, uint32_t key[4]) {
    const uint32_t exps = q.w0.w;
    const float Ax = __uint_as_float((exps & 0xffu) << 23) * r.inv.x, Ay = __uint_as_float(((exps >> 8) & 0xffu) << 23) * r.inv.y,
                Az = __uint_as_float(((exps >> 16) & 0xffu) << 23) * r.inv.z;
    const float Bx = (__uint_as_float(q.w0.x) - r.o.x) * r.inv.x, By = (__uint_as_float(q.w0.y) - r.o.y) * r.inv.y,
                Bz = (__uint_as_float(q.w0.z) - r.o.z) * r.inv.z;
    const bool nx = r.inv.x < 0.0f, ny = r.inv.y < 0.0f, nz = r.inv.z < 0.0f;
    const uint32_t qnx = nx ? q.w2.w : q.w2.x, qfx = nx ? q.w2.x : q.w2.w;   // planes the ray enters / leaves through
    const uint32_t qny = ny ? q.w3.x : q.w2.y, qfy = ny ? q.w2.y : q.w3.x;
    const uint32_t qnz = nz ? q.w3.y : q.w2.z, qfz = nz ? q.w2.z : q.w3.y;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float tnx = fma_((float)((qnx >> (8 * k)) & 0xffu), Ax, Bx), tfx = fma_((float)((qfx >> (8 * k)) & 0xffu), Ax, Bx);
        const float tny = fma_((float)((qny >> (8 * k)) & 0xffu), Ay, By), tfy = fma_((float)((qfy >> (8 * k)) & 0xffu), Ay, By);
        const float tnz = fma_((float)((qnz >> (8 * k)) & 0xffu), Az, Bz), tfz = fma_((float)((qfz >> (8 * k)) & 0xffu), Az, Bz);
        const float tn = fmaxf(fmaxf(tnx, tny), fmaxf(tnz, 0.0f));
        const float tf = fminf(fminf(tfx, tfy), fminf(tfz, best));
        key[k] = (tn <= tf) ? ((__float_as_uint(tn) & ~3u) | (uint32_t)k) : (MISS_SLOT ? (0xfffffffcu | (uint32_t)k) : 0xffffffffu);
    }
}

// One inner node: the four child boxes against the ray; the slots of the hit children sorted by entry distance (a
// 5-exchange network on distance | slot keys); the nearest becomes the cursor, the others one stack entry; no hit pops.
DEV void bvh_cex(uint32_t &ka, uint32_t &kb) {
    const uint32_t lo = min(ka, kb), hi = max(ka, kb);
    ka = lo;
    kb = hi;
}
template <typename Tree>
DEV void bvh_visit(const Tree &tr, const BvhStack &st, BvhCursor &c, BvhOvf &ovf, const BoxRay &r, float best) {
    const uint32_t node = c.cur;
    NodeQ q = tr.node(node);
#ifndef PBRT_BVH_LATE_REFS
    // (keeps the load of the child references with the other three: left alone, the compiler sinks it into the branch that uses
    // them, a second LDS round trip per node)
    asm volatile("" : "+v"(q.w1));
#endif
    uint32_t key[4];
    bvh_child_keys(q, r, best, key);
    bvh_cex(key[0], key[1]);
    bvh_cex(key[2], key[3]);
    bvh_cex(key[0], key[2]);
    bvh_cex(key[1], key[3]);
    bvh_cex(key[1], key[2]);
    // (the reference of the nearest child is picked before the branch, by selects: the four references arrive with the rest of
    // the node instead of in a second, dependent LDS read)
    const bool s_odd = (key[0] & 1u) != 0u, s_high = (key[0] & 2u) != 0u;
    const uint32_t r_lo = s_odd ? q.w1.y : q.w1.x, r_hi = s_odd ? q.w1.w : q.w1.z;
    const uint32_t near_ref = s_high ? r_hi : r_lo;
    if (key[0] == 0xffffffffu) {
        c.cur = bvh_pop(tr, st, c, ovf);
        return;
    }
    c.cur = near_ref;
    const uint32_t n_more = (key[1] != 0xffffffffu ? 1u : 0u) + (key[2] != 0xffffffffu ? 1u : 0u) + (key[3] != 0xffffffffu ? 1u : 0u);
    // (slots of misses are 3, 3, ...: never looked at, the count says how many are real)
    const uint32_t entry = (node << 8) | (n_more << 6) | (key[1] & 3u) | ((key[2] & 3u) << 2) | ((key[3] & 3u) << 4);
    bvh_push(st, c, ovf, n_more != 0u, entry);
}

// One leaf record against the ray (`full`: the 64-byte table, read for cones only).  CURVED = false: the scene holds triangles
// and parallelograms only (the host knows), the sphere and cone tests are compiled out.
template <bool CURVED = true>
DEV bool lprim_hit(const LeafQ &L, const pbrt_prim *full, V3 o, V3 d, float tmax, float *t, float *u, float *v, uint32_t *id) {
    const uint32_t meta = L.e.y, type = meta >> 28;
    *id = meta & 0x0fffffffu;
    const V3 v0 = {__uint_as_float(L.a.x), __uint_as_float(L.a.y), __uint_as_float(L.b.x)};
    if (CURVED && type == PBRT_PRIM_SPHERE) {
        *u = 0.0f;
        *v = 0.0f;
        return sphere_hit(v0, __uint_as_float(L.b.y), o, d, tmax, t);
    }
    if (CURVED && type == PBRT_PRIM_CONE) {
        *v = 0.0f;
        return cone_hit(full[*id], o, d, tmax, t, u);
    }
    const V3 e1 = {__uint_as_float(L.b.y), __uint_as_float(L.c.x), __uint_as_float(L.c.y)};
    const V3 e2 = {__uint_as_float(L.d.x), __uint_as_float(L.d.y), __uint_as_float(L.e.x)};
    return planar_hit(type == PBRT_PRIM_TRIANGLE, v0, e1, e2, o, d, tmax, t, u, v);
}

// Two-phase ("while-while") traversal: a lane walks inner nodes until it holds a leaf or has finished, and the primitives of
// the held leaves are tested when EVERY lane of the wave has got that far (the primitive test is the long part of a step).
// Holding a leaf delays the update of `best`, so a lane may visit a node more than it would have: still conservative.
#ifdef PBRT_BVH_PROBE  // diagnostic builds: how many of a wave's traversal trips does a lane use?  (tools/bvh_probe.py)
__device__ unsigned long long g_bvh_probe[8];  // closest hit: wave trips x 64, lane trips (node walk), same for primitive tests; any hit: +4
#define BVH_PROBE_FIRST_LANE() \
    (__builtin_amdgcn_mbcnt_hi((uint32_t)(__ballot(true) >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)__ballot(true), 0u)) == 0u)
#endif
template <bool ANY, typename Tree>
DEV bool bvh_intersect(const Tree &tr, const pbrt_prim *full, const BvhStack &st, V3 o, V3 d, float tmax, Hit *h) {
    const BoxRay br = make_box_ray(o, d);
    BvhCursor c;
    BvhOvf ovf;
    c.cur = 0;
    c.tos = BVH_SENT;
    c.sp = 0;
    bool found = false;
    float best = tmax;
    for (;;) {
        while ((int32_t)c.cur >= 0) {
#ifdef PBRT_BVH_PROBE
            atomicAdd(&g_bvh_probe[(ANY ? 4 : 0) + 1], 1ull);
            if (BVH_PROBE_FIRST_LANE()) atomicAdd(&g_bvh_probe[(ANY ? 4 : 0) + 0], 64ull);
#endif
            bvh_visit(tr, st, c, ovf, br, best);
        }
        if (c.cur == BVH_SENT) break;  // this lane has finished (the wave leaves the loop when every lane has)
        const uint32_t first = c.cur & 0x07ffffffu, count = (c.cur >> 27) & 15u;
        for (uint32_t k = 0; k < count; ++k) {
            float t, u, v;
            uint32_t id;
#ifdef PBRT_BVH_PROBE
            atomicAdd(&g_bvh_probe[(ANY ? 4 : 0) + 3], 1ull);
            if (BVH_PROBE_FIRST_LANE()) atomicAdd(&g_bvh_probe[(ANY ? 4 : 0) + 2], 64ull);
#endif
            if (lprim_hit(tr.leaf(first + k), full, o, d, best, &t, &u, &v, &id)) {
                if (ANY) return true;
                if (!found || t < best || (t == best && id < h->prim)) {
                    best = t;
                    h->t = t;
                    h->u = u;
                    h->v = v;
                    h->prim = id;
                    h->slot = id;
                    found = true;
                }
            }
        }
        c.cur = bvh_pop(tr, st, c, ovf);
    }
    return found;
}

// Wave-synchronous closest hit for COHERENT rays (the camera rays of an 8 x 8 pixel tile: one origin, a narrow cone of
// directions).  The wave walks ONE path through the tree: the node index is wave-uniform (a node read is one broadcast, no bank
// conflicts, no per-lane stack), every lane tests the node's four child boxes with its own ray and its own closest hit so far, a
// child goes on the wave's stack if ANY lane hits it, and at a leaf every lane tests every primitive.  The stack is the 64 lanes
// of one VGPR (written by compare + select, read by v_readlane with a scalar lane index; the host checks 3 * depth <= 64); the children are ordered by
// the entry distances of one representative lane (`rep`, wave-uniform).
// Same result per lane as bvh_intersect<false>: a lane sees a superset of the primitives its own traversal would have tested --
// a primitive in a box its ray misses cannot be hit (the boxes are conservative), one behind its closest hit loses the
// comparison -- and ties in t go to the lowest index either way.
// `best`: in = tmax of the lane's ray (< 0: the lane has no ray), out = t of the hit.
template <bool CURVED, typename Tree>
DEV bool bvh_packet_closest(const Tree &tr, const pbrt_prim *full, V3 o, V3 d, uint32_t rep, float &best, float &hu, float &hv,
                            uint32_t &hid) {
    const BoxRay r = make_box_ray(o, d);
    const uint32_t lane = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    uint32_t stk = 0;  // lane i: stack entry i
    uint32_t sp = 0;   // wave-uniform
    uint32_t cur = 0;  // wave-uniform child reference
    bool found = false;
    for (;;) {
        if ((int32_t)cur >= 0) {
            const NodeQ q = tr.node_boxes(cur);
            const uint32_t reftab = tr.child(cur, lane & 3u);  // lane k (k < 4): reference of child k
            uint32_t key[4], mask = 0;
            bvh_child_keys<true>(q, r, best, key);
#pragma unroll
            for (int k = 0; k < 4; ++k) mask |= __builtin_amdgcn_ballot_w64(key[k] < 0xfffffffcu) != 0ull ? (1u << k) : 0u;
            uint32_t pend = BVH_SENT;
            if (mask != 0u && (mask & (mask - 1u)) == 0u) {  // one child: no order to find
                pend = (uint32_t)__builtin_amdgcn_readlane((int)reftab, (int)__builtin_ctz(mask));
            } else if (mask != 0u) {
                // the representative lane's order, far to near: every hit child but the nearest goes on the stack
                uint32_t s0 = (uint32_t)__builtin_amdgcn_readlane((int)key[0], (int)rep), s1 = (uint32_t)__builtin_amdgcn_readlane((int)key[1], (int)rep),
                         s2 = (uint32_t)__builtin_amdgcn_readlane((int)key[2], (int)rep), s3 = (uint32_t)__builtin_amdgcn_readlane((int)key[3], (int)rep);
                uint32_t t;
                t = min(s0, s1), s1 = max(s0, s1), s0 = t;
                t = min(s2, s3), s3 = max(s2, s3), s2 = t;
                t = min(s0, s2), s2 = max(s0, s2), s0 = t;
                t = min(s1, s3), s3 = max(s1, s3), s1 = t;
                t = min(s1, s2), s2 = max(s1, s2), s1 = t;
                const uint32_t order[4] = {s3, s2, s1, s0};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const uint32_t slot = order[i] & 3u;
                    if ((mask >> slot) & 1u) {
                        if (pend != BVH_SENT) {
                            stk = lane == sp ? pend : stk;  // (v_writelane by hand: compare + select)
                            ++sp;
                        }
                        pend = (uint32_t)__builtin_amdgcn_readlane((int)reftab, (int)slot);
                    }
                }
            }
            cur = pend;
        } else {
            const uint32_t first = cur & 0x07ffffffu, count = (cur >> 27) & 15u;
            for (uint32_t k = 0; k < count; ++k) {
                float t, u, v;
                uint32_t id;
                if (lprim_hit<CURVED>(tr.leaf(first + k), full, o, d, best, &t, &u, &v, &id)) {
                    if (!found || t < best || (t == best && id < hid)) {
                        best = t;
                        hu = u;
                        hv = v;
                        hid = id;
                        found = true;
                    }
                }
            }
            cur = BVH_SENT;
        }
        if (cur == BVH_SENT) {  // nothing to descend into: the next entry of the stack
            if (sp == 0u) break;
            --sp;
            cur = (uint32_t)__builtin_amdgcn_readlane((int)stk, (int)sp);
        }
    }
    return found;
}

// ---- surface interaction ------------------------------------------------------------------------
struct SI {
    V3 p, n;  // hit point, geometric normal (rays are offset along it)
    V3 ns;    // shading normal si.sh_frame.n: interpolated vertex normals where the mesh has them, else n
};
// Mitsuba Mesh::compute_surface_interaction: sh_frame.n = normalize(b0 n0 + b1 n1 + b2 n2) with the barycentrics of the hit.
// SHN: compiled into the BVH and the _BIG brute-force kernels only (a small scene with vertex normals takes the _BIG
// variant, like one with a cone: the 64-VGPR Cornell-box kernel has no registers to spare).
template <bool SHN>
DEV V3 shading_normal(const pbrt_prim &P, V3 n, float u, float v, const float *vn, uint32_t slot) {
    if (!SHN || vn == nullptr || (P.type != PBRT_PRIM_TRIANGLE && P.type != PBRT_PRIM_PARALLELOGRAM)) return n;
    const float *r = vn + 9u * slot;
    const V3 n0 = {r[0], r[1], r[2]}, n1 = {r[3], r[4], r[5]}, n2 = {r[6], r[7], r[8]};
    if (!(dot(n0, n0) + dot(n1, n1) + dot(n2, n2) > 0.0f)) return n;  // no vertex normals on this primitive
    if (P.type == PBRT_PRIM_PARALLELOGRAM) return normalize(n0);      // a merged quad: all its vertex normals agree
    const float b0 = 1.0f - u - v;
    return normalize(madd(n0, b0, madd(n1, u, n2 * v)));
}
template <bool CONES = true>
DEV SI make_si(const pbrt_prim &P, V3 o, V3 d, float t, float u, float v, const float *vn = nullptr, uint32_t slot = 0) {
    SI si;
    if (P.type == PBRT_PRIM_SPHERE) {
        V3 c = g3(P, 0);
        V3 p = madd(d, t, o);
        si.n = normalize(p - c);
        si.p = madd(si.n, P.g[3], c);
    } else if (CONES && P.type == PBRT_PRIM_CONE) {
        // outward normal: object-space gradient of x^2 + y^2 - (1 - z)^2 (lateral) or -z (base disc), to world
        // space by the transpose of the world -> object matrix
        const V3 r0 = g3(P, 0), r1 = g3(P, 4), r2 = g3(P, 8);
        si.p = madd(d, t, o);
        V3 no = {0.0f, 0.0f, -1.0f};
        if (u == 0.0f) {
            no = {dot(r0, si.p) + P.g[3], dot(r1, si.p) + P.g[7], 1.0f - (dot(r2, si.p) + P.g[11])};
            if (!(dot(no, no) > 0.0f)) no = {0.0f, 0.0f, 1.0f};  // the apex itself
        }
        si.n = normalize(v3(fma_(r0.x, no.x, fma_(r1.x, no.y, r2.x * no.z)), fma_(r0.y, no.x, fma_(r1.y, no.y, r2.y * no.z)),
                            fma_(r0.z, no.x, fma_(r1.z, no.y, r2.z * no.z))));
    } else {
        si.p = madd(g3(P, 6), v, madd(g3(P, 3), u, g3(P, 0)));
        si.n = g3(P, 9);
    }
    si.ns = shading_normal<CONES>(P, si.n, u, v, vn, slot);
    return si;
}

// dp_du of the interaction, the input of Mitsuba's shading frame (make_sh_frame):
//   triangle / parallelogram: the first edge e1 (Mesh without texture coordinates: dp_du = p1 - p0; `rectangle`:
//     to_world * (2, 0, 0), which is e1 of its parallelogram record)
//   sphere: 2 pi (-y, x, 0) of the hit point about the centre (Sphere::compute_surface_interaction; the record holds
//     centre + radius only, so the sphere's object axes are taken parallel to the world's)
//   cone ([DEFINE] shape, no Mitsuba definition): none -> coordinate_system(n)
template <bool CONES = true>
DEV V3 si_dp_du(const pbrt_prim &P, const SI &si) {
    if (P.type == PBRT_PRIM_SPHERE) return {-(si.p.y - P.g[1]), si.p.x - P.g[0], 0.0f};
    if (CONES && P.type == PBRT_PRIM_CONE) return {0.0f, 0.0f, 0.0f};
    return g3(P, 3);
}

// ---- UltraBSDF.sample: CustomBSDF.py:87-175 (+ _ggx_sample :30-61, ggx_pdf :64-83) --------------
struct UltraOut {
    V3 chosen;
    float pdf, amp;
    bool reflect;
};
DEV UltraOut ultra_core(const pbrt_material &m, uint32_t quirks, V3 wi_in, V3 n_geo, V3 n_sh, float s1, float s2,
                        float s1b) {
    const float impedance = m.p[0], alpha = m.p[1], medium_z = m.p[2];
    V3 wi = wi_in;
    if (quirks & PBRT_USQ_DOUBLE_LOCAL) wi = to_local(make_frame(n_geo), wi_in);  // :32-33
    V3 ws = normalize(v3(alpha * wi.x, alpha * wi.y, wi.z));                     // :37-38
    float inv_len = 1.0f / sqrtf(fmaxf(fma_(-ws.z, ws.z, 1.0f), 1e-7f));          // :41
    V3 T1 = {ws.y * inv_len, -ws.x * inv_len, 0.0f};                              // :42-44
    V3 T2 = cross(ws, T1);                                                        // :45
    float dx, dy;
    if (quirks & PBRT_USQ_DIAG_SAMPLE)
        square_to_disk(s1, s1, &dx, &dy);  // :48
    else
        square_to_disk(s1, s1b, &dx, &dy);
    float S = 0.5f * (1.0f + ws.z);                                               // :51
    dy = fma_(1.0f - S, sqrtf(fmaxf(fma_(-dx, dx, 1.0f), 0.0f)), S * dy);          // :52
    float mz = sqrtf(fmaxf(1.0f - fma_(dx, dx, dy * dy), 0.0f));                   // :55
    V3 ms = madd(ws, mz, madd(T2, dy, T1 * dx));                                  // :55
    V3 mm = normalize(v3(alpha * ms.x, alpha * ms.y, ms.z));                      // :56-59
    V3 inc = wi_in;                                                               // :90
    if (!(dot(mm, inc) < 0.0f)) mm = -mm;                                         // :100
    float cos_wi_m = dot(inc, mm);                                                // :101
    bool entering;
    if (quirks & PBRT_USQ_NEVER_ENTER)
        entering = dot(mm, inc) > 0.0f;  // :104
    else
        entering = wi_in.z > 0.0f;
    float Z1 = entering ? medium_z : impedance;                                   // :106
    float Z2 = entering ? impedance : medium_z;                                   // :107
    float ratio = Z1 / Z2;                                                        // :111
    float cosTr = fabsf(dot(mm, inc));                                            // :119
    float sqrt_arg = fma_(-(ratio * ratio), fma_(-cosTr, cosTr, 1.0f), 1.0f);      // :120
    float cosTt = sqrtf(fmaxf(sqrt_arg, 0.0f));                                   // :121
    float denom = fma_(Z1, cosTr, Z2 * cosTt);                                    // :122
    float Ar = fma_(Z1, cosTr, -(Z2 * cosTt)) / denom;                            // :123
    float At = 1.0f - Ar;                                                         // :124
    V3 refl, trans;
    if (quirks & PBRT_USQ_REF_REFLECT) {
        refl = madd(mm, 2.0f * cos_wi_m, inc);                                    // :130
        trans = madd(mm, fma_(ratio, cosTr, -cosTt), refl * ratio);               // :131
    } else {
        refl = madd(mm, 2.0f * cos_wi_m, -inc);
        trans = madd(mm, -fma_(ratio, cosTr, -cosTt), (-inc) * ratio);
    }
    bool tir = sqrt_arg < 0.0f;                                                   // :137
    float prob_reflect = Ar * Ar;                                                 // :142
    bool select_reflect = tir ? true : (s2 < prob_reflect);                       // :144-145
    V3 chosen = select_reflect ? refl : trans;                                    // :147
    float pdf_m;
    if (quirks & PBRT_USQ_UNIT_GGX_PDF) {
        pdf_m = 1.0f;  // :81-82
    } else {
        float c = fabsf(mm.z), a2 = alpha * alpha;
        float dd = fma_(fma_(a2, 1.0f, -1.0f) * c, c, 1.0f);
        pdf_m = a2 / (K_PI * dd * dd) * c;
    }
    float pdf_reflect = pdf_m / (4.0f * fabsf(cos_wi_m));                         // :154
    float cos_wo_m = dot(trans, mm);                                              // :155
    V3 nref = (quirks & PBRT_USQ_MIXED_FRAMES) ? n_sh : v3(0, 0, 1);              // :156-157
    float abs_n_wi = fabsf(dot(nref, inc));                                       // :156
    float abs_n_wo = fmaxf(fabsf(dot(nref, trans)), 1e-7f);                       // :157
    float pdf_trans = pdf_m * (ratio * ratio) * fabsf(cos_wo_m) / (abs_n_wi * abs_n_wo);  // :158
    UltraOut o;
    o.chosen = chosen;
    o.pdf = select_reflect ? pdf_reflect : pdf_trans;                             // :166
    o.amp = select_reflect ? Ar : At;                                             // :170
    o.reflect = select_reflect;
    return o;
}

// ---- BSDF.sample / eval_pdf, radiance mode (Mitsuba diffuse / conductor / dielectric) -----------
struct BSample {
    V3 wo;
    float pdf;
    V3 weight;
    float eta;
    bool delta, valid;
    uint32_t lobe;
};

DEV void bsdf_eval_pdf(const pbrt_material &m, V3 wi, V3 wo, V3 *f, float *pdf) {
    *f = {0, 0, 0};
    *pdf = 0.0f;
    if (m.type == PBRT_MAT_DIFFUSE && wi.z > 0.0f && wo.z > 0.0f) {
        float c = K_INV_PI * wo.z;
        *f = v3(m.p[0], m.p[1], m.p[2]) * c;
        *pdf = c;
    }
}

// shf: the interaction's shading frame (si.to_local at CustomBSDF.py:165; read by ULTRA only)
DEV BSample bsdf_sample(const pbrt_material &m, uint32_t quirks, V3 wi, V3 n_geo, V3 n_sh, const Frame &shf, float s1,
                        float s2x, float s2y) {
    BSample b;
    b.valid = false;
    b.delta = false;
    b.eta = 1.0f;
    b.pdf = 0.0f;
    b.weight = {0, 0, 0};
    b.wo = {0, 0, 1};
    b.lobe = 0;
    const uint32_t type = m.type;
    if (type == PBRT_MAT_DIFFUSE) {
        if (!(wi.z > 0.0f)) return b;
        b.wo = square_to_cosine_hemisphere(s2x, s2y);
        b.pdf = K_INV_PI * b.wo.z;
        if (!(b.pdf > 0.0f)) return b;
        b.weight = v3(m.p[0], m.p[1], m.p[2]);
        b.valid = true;
    } else if (type == PBRT_MAT_CONDUCTOR) {
        if (!(wi.z > 0.0f)) return b;
        b.wo = {-wi.x, -wi.y, wi.z};
        b.pdf = 1.0f;
        b.weight = v3(m.p[0], m.p[1], m.p[2]);
        b.delta = true;
        b.valid = true;
    } else if (type == PBRT_MAT_DIELECTRIC) {
        float eta = m.p[0];
        float ci = wi.z;
        bool outside = ci >= 0.0f;
        float rcp_eta = 1.0f / eta;
        float eta_it = outside ? eta : rcp_eta, eta_ti = outside ? rcp_eta : eta;
        float ct2 = fma_(-fma_(-ci, ci, 1.0f), eta_ti * eta_ti, 1.0f);
        float cia = fabsf(ci), cta = sqrtf(fmaxf(ct2, 0.0f));
        float a_s = fma_(-eta_it, cta, cia) / fma_(eta_it, cta, cia);
        float a_p = fma_(-eta_it, cia, cta) / fma_(eta_it, cia, cta);
        float r = 0.5f * fma_(a_s, a_s, a_p * a_p);
        if (eta == 1.0f)
            r = 0.0f;
        else if (cia == 0.0f)
            r = 1.0f;
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
            float f2 = eta_ti * eta_ti;
            b.weight = {f2, f2, f2};
            b.eta = eta_it;
            b.lobe = 1;
        }
    } else if (type == PBRT_MAT_ULTRA) {
        UltraOut o = ultra_core(m, quirks, wi, n_geo, n_sh, s1, s2x, s2y);
        b.wo = to_local(shf, o.chosen);  // CustomBSDF.py:165
        b.pdf = o.pdf;
        b.weight = {o.amp, o.amp, o.amp};
        b.lobe = o.reflect ? 0u : 1u;
        b.delta = true;
        b.valid = true;
    }
    return b;
}

// ---- Emitter.sample_direction (Mitsuba area / point; Scene::sample_emitter_direction) -----------
struct ESample {
    V3 q, d;
    float dist, pdf;
    V3 weight;
    bool delta, valid;
    uint32_t emitter;
};
// The per-lane (gathered) scene tables of the shading code: global memory, or -- for brute-force
// scenes, where every table has at most 32 entries -- copies staged into LDS by the workgroup.
struct Tables {
    const pbrt_prim *prims_by_slot;  // indexed by Hit::slot
    const pbrt_prim *prims_by_id;    // indexed by the caller's primitive index (light_prims entries)
    const pbrt_material *mats;
    const pbrt_emitter *emitters;
    const uint32_t *light_prims;
    const float *light_cdf;
    uint32_t n_emitters;
};
DEV Tables global_tables(const DevScene &sc) {
    return {sc.prims, sc.prims_by_id, sc.mats, sc.emitters, sc.light_prims, sc.light_cdf, sc.n_emitters};
}

DEV ESample sample_emitter(const Tables &sc, V3 p, F4 u) {
    ESample e;
    e.valid = false;
    e.delta = false;
    e.pdf = 0.0f;
    e.dist = 0.0f;
    e.weight = {0, 0, 0};
    e.q = {0, 0, 0};
    e.d = {0, 0, 0};
    e.emitter = 0;
    const uint32_t nE = sc.n_emitters;
    if (nE == 0) return e;
    uint32_t ei = min((uint32_t)(u.x * (float)nE), nE - 1);
    const pbrt_emitter &E = sc.emitters[ei];
    e.emitter = ei;
    const float sel = (float)nE;
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
    uint32_t k = 0;
    while (k + 1 < E.count && !(u.y < sc.light_cdf[E.first + k])) ++k;
    const pbrt_prim &P = sc.prims_by_id[sc.light_prims[E.first + k]];
    float b1, b2;
    if (P.type == PBRT_PRIM_TRIANGLE) {
        float t = sqrtf(fmaxf(1.0f - u.z, 0.0f));
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
    if (!(cosl > 0.0f)) return e;
    e.pdf = d2 / (cosl * E.area * sel);
    float w = 1.0f / e.pdf;
    e.weight = v3(E.radiance[0], E.radiance[1], E.radiance[2]) * w;
    e.valid = true;
    return e;
}

// ---- Sensor.sample_ray: Mitsuba 'perspective' ---------------------------------------------------
DEV void camera_ray(const pbrt_camera &cam, float sx, float sy, V3 *o, V3 *d, float *tmax) {
    float tx = cam.tan_half_fov_x;
    float ty = tx * (float)cam.film_h / (float)cam.film_w;
    V3 dc = normalize(v3(fma_(-2.0f, sx, 1.0f) * tx, fma_(-2.0f, sy, 1.0f) * ty, 1.0f));
    V3 dw = normalize(xf_vec(cam.to_world, dc));
    float inv_z = 1.0f / dc.z;
    V3 org = {cam.to_world[3], cam.to_world[7], cam.to_world[11]};
    *o = madd(dw, cam.near_clip * inv_z, org);
    *d = dw;
    *tmax = (cam.far_clip - cam.near_clip) * inv_z;
}

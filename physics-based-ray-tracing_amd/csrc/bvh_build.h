// bvh_build.h -- host-side BVH2 builder (binned SAH, <= 4 primitives per leaf) for scenes that do
// not fit the brute-force loop (scenes/meshes/teapot.ply: 2256 triangles, TestRing/TestRing.obj:
// 1152 triangles).  Output is the flat DevNode array the kernels stage into LDS, the primitives in
// leaf order, and the map from leaf-order slot to the caller's primitive index.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <vector>

#include "../../include/pbrt_hip.h"

struct HostNode {
    float lo[3];
    uint32_t a;
    float hi[3];
    uint32_t b;
};

struct HostBvh {
    std::vector<HostNode> nodes;
    std::vector<uint32_t> order;  // leaf-order slot -> caller index
    uint32_t max_depth = 0;
};

// World-space frame of a CONE primitive (g = world -> object 3x4): base centre c, the images a, b of the object x / y
// unit vectors (the base disc is c + a cos + b sin), and the apex.  False if the matrix is singular or not finite.
inline bool cone_world_frame(const pbrt_prim &P, double c[3], double a[3], double b[3], double apex[3]) {
    const double m[3][3] = {{P.g[0], P.g[1], P.g[2]}, {P.g[4], P.g[5], P.g[6]}, {P.g[8], P.g[9], P.g[10]}};
    const double t[3] = {P.g[3], P.g[7], P.g[11]};
    const double det = m[0][0] * (m[1][1] * m[2][2] - m[1][2] * m[2][1]) - m[0][1] * (m[1][0] * m[2][2] - m[1][2] * m[2][0]) +
                       m[0][2] * (m[1][0] * m[2][1] - m[1][1] * m[2][0]);
    if (!std::isfinite(det) || det == 0.0) return false;
    double inv[3][3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            const int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
            inv[j][i] = (m[i1][j1] * m[i2][j2] - m[i1][j2] * m[i2][j1]) / det;  // cofactor transpose
        }
    for (int k = 0; k < 3; ++k) {
        c[k] = -(inv[k][0] * t[0] + inv[k][1] * t[1] + inv[k][2] * t[2]);  // W (0, 0, 0)
        a[k] = inv[k][0];
        b[k] = inv[k][1];
        apex[k] = c[k] + inv[k][2];
        if (!std::isfinite(c[k]) || !std::isfinite(a[k]) || !std::isfinite(b[k]) || !std::isfinite(apex[k])) return false;
    }
    return true;
}

// SAH parameters (A/B-able): cost of one node step relative to one primitive test, and the largest leaf.
// Measured on MI355X (ring 1024^2 x 64 spp / 896-triangle cone phantom): 0.25 / 0.5 / 1 / 2 at <= 4 per leaf:
// 21.95 / 22.05 / 22.10 / 24.13 ms; leaves of <= 2 / 6 / 8: 22.07 / 22.13 / 24.04 ms -- flat around the defaults.
// Trees that stay in global memory (BVH_CTRAV_GLOBAL, round 4): a node step is a 64-byte gather per lane there, a primitive test a
// 40-byte one -- 0.25 / 0.5 / 1 / 2: bunny.ply 23.5 / 22.4 / 21.3 / 21.9 ms, suzanne.ply 19.1 / 19.1 / 18.4 / 18.5 ms (1024^2 x 64),
// while the ring phantom in LDS loses 3 % in ultrasound mode at 1 (profiles/r04_bunny_global_tree.txt).
#ifndef BVH_CTRAV
#define BVH_CTRAV 0.5f
#endif
#ifndef BVH_CTRAV_GLOBAL
#define BVH_CTRAV_GLOBAL 1.0f
#endif
#ifndef BVH_MAX_LEAF
#define BVH_MAX_LEAF 4
#endif

namespace bvh_detail {

struct Box {
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    void grow(const Box &o) {
        for (int k = 0; k < 3; ++k) {
            lo[k] = std::min(lo[k], o.lo[k]);
            hi[k] = std::max(hi[k], o.hi[k]);
        }
    }
    float area() const {
        float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
        return (dx < 0) ? 0.0f : 2.0f * (dx * dy + dy * dz + dz * dx);
    }
};

inline Box prim_box(const pbrt_prim &P) {
    Box b;
    if (P.type == PBRT_PRIM_SPHERE) {
        for (int k = 0; k < 3; ++k) {
            b.lo[k] = P.g[k] - P.g[3];
            b.hi[k] = P.g[k] + P.g[3];
        }
        return b;
    }
    if (P.type == PBRT_PRIM_CONE) {  // hull of the base ellipse and the apex (padded like every box by the builder)
        double c[3], a[3], bb[3], apex[3];
        cone_world_frame(P, c, a, bb, apex);  // validated at scene creation
        for (int k = 0; k < 3; ++k) {
            const double r = std::sqrt(a[k] * a[k] + bb[k] * bb[k]);
            b.lo[k] = std::nextafter((float)std::min(c[k] - r, apex[k]), -INFINITY);
            b.hi[k] = std::nextafter((float)std::max(c[k] + r, apex[k]), INFINITY);
        }
        return b;
    }
    for (int k = 0; k < 3; ++k) {
        float c0 = P.g[k], c1 = P.g[k] + P.g[3 + k], c2 = P.g[k] + P.g[6 + k];
        float c3 = P.type == PBRT_PRIM_PARALLELOGRAM ? P.g[k] + P.g[3 + k] + P.g[6 + k] : c0;
        b.lo[k] = std::min(std::min(c0, c1), std::min(c2, c3));
        b.hi[k] = std::max(std::max(c0, c1), std::max(c2, c3));
    }
    return b;
}

struct Builder {
    const pbrt_prim *prims;
    std::vector<Box> boxes;
    std::vector<float> cent;
    HostBvh *out;
    float pad;
    float ctrav = BVH_CTRAV;

    // nodes[idx] must already exist
    void build(uint32_t idx, uint32_t first, uint32_t count, uint32_t depth) {
        out->max_depth = std::max(out->max_depth, depth);
        Box bb, cb;
        for (uint32_t k = 0; k < count; ++k) {
            uint32_t i = out->order[first + k];
            bb.grow(boxes[i]);
            for (int c = 0; c < 3; ++c) {
                cb.lo[c] = std::min(cb.lo[c], cent[3 * i + c]);
                cb.hi[c] = std::max(cb.hi[c], cent[3 * i + c]);
            }
        }
        HostNode n;
        for (int c = 0; c < 3; ++c) {
            n.lo[c] = bb.lo[c] - pad;
            n.hi[c] = bb.hi[c] + pad;
        }
        const int NB = 16;
        int best_axis = -1, best_split = -1;
        float best_cost = (float)count * bb.area();  // leaf cost
        if (count > BVH_MAX_LEAF) best_cost = INFINITY;  // force a split above the leaf limit
        for (int axis = 0; axis < 3 && count > 1; ++axis) {
            float lo = cb.lo[axis], ext = cb.hi[axis] - lo;
            if (!(ext > 0)) continue;
            Box bins[NB];
            uint32_t cnt[NB] = {0};
            for (uint32_t k = 0; k < count; ++k) {
                uint32_t i = out->order[first + k];
                int b = std::min(NB - 1, (int)((cent[3 * i + axis] - lo) / ext * NB));
                bins[b].grow(boxes[i]);
                cnt[b]++;
            }
            float la[NB], ra[NB];
            uint32_t lc[NB], rc[NB];
            Box acc;
            uint32_t c = 0;
            for (int b = 0; b < NB; ++b) {
                acc.grow(bins[b]);
                c += cnt[b];
                la[b] = acc.area();
                lc[b] = c;
            }
            acc = Box();
            c = 0;
            for (int b = NB - 1; b >= 0; --b) {
                acc.grow(bins[b]);
                c += cnt[b];
                ra[b] = acc.area();
                rc[b] = c;
            }
            for (int b = 0; b + 1 < NB; ++b) {
                if (lc[b] == 0 || rc[b + 1] == 0) continue;
                float cost = ctrav * bb.area() + la[b] * lc[b] + ra[b + 1] * rc[b + 1];
                if (cost < best_cost) {
                    best_cost = cost;
                    best_axis = axis;
                    best_split = b;
                }
            }
        }
        uint32_t mid = 0;
        if (best_axis >= 0) {
            float lo = cb.lo[best_axis], ext = cb.hi[best_axis] - lo;
            auto it = std::partition(out->order.begin() + first, out->order.begin() + first + count, [&](uint32_t i) {
                int b = std::min(NB - 1, (int)((cent[3 * i + best_axis] - lo) / ext * NB));
                return b <= best_split;
            });
            mid = (uint32_t)(it - (out->order.begin() + first));
        } else if (count > 8) {  // degenerate centroids: split in the middle
            best_axis = 0;
            mid = count / 2;
        }
        if (best_axis < 0 || mid == 0 || mid == count) {
            n.a = first;
            n.b = count;  // leaf (count <= 8 guaranteed by the branch above)
            out->nodes[idx] = n;
            return;
        }
        uint32_t left = (uint32_t)out->nodes.size();
        out->nodes.push_back(HostNode{});
        out->nodes.push_back(HostNode{});
        n.a = left;
        n.b = 0x80000000u | (uint32_t)best_axis;
        out->nodes[idx] = n;
        build(left, first, mid, depth + 1);
        build(left + 1, first + mid, count - mid, depth + 1);
    }
};

}  // namespace bvh_detail

inline void build_bvh(const pbrt_prim *prims, uint32_t n, HostBvh *out, float ctrav = BVH_CTRAV) {
    bvh_detail::Builder b;
    b.ctrav = ctrav;
    b.prims = prims;
    b.out = out;
    b.boxes.resize(n);
    b.cent.resize(3 * (size_t)n);
    out->order.resize(n);
    bvh_detail::Box all;
    for (uint32_t i = 0; i < n; ++i) {
        b.boxes[i] = bvh_detail::prim_box(prims[i]);
        all.grow(b.boxes[i]);
        for (int c = 0; c < 3; ++c) b.cent[3 * i + c] = 0.5f * (b.boxes[i].lo[c] + b.boxes[i].hi[c]);
        out->order[i] = i;
    }
    float dx = all.hi[0] - all.lo[0], dy = all.hi[1] - all.lo[1], dz = all.hi[2] - all.lo[2];
    float diag = std::sqrt(dx * dx + dy * dy + dz * dz);
    float mag = 0;
    for (int c = 0; c < 3; ++c) mag = std::max(mag, std::max(std::fabs(all.lo[c]), std::fabs(all.hi[c])));
    // conservative padding: the slab test may never cull a primitive the brute-force loop would hit
    b.pad = 2e-5f * (diag + mag) + 1e-30f;
    out->nodes.clear();
    out->nodes.push_back(HostNode{});
    out->max_depth = 0;
    b.build(0, 0, n, 0);
}

// ---- BVH4 for the device: the SAH BVH2 above collapsed to four children per node, child boxes on an 8-bit grid --------------
// A node holds the lower corner of its (padded) box, one power-of-two grid step per axis, and for every child six grid
// coordinates rounded OUTWARD (a child's grid box contains its padded float box), so one 64-byte read decides four descents and
// the tree has about a third of the BVH2's nodes.  Children keep the left-to-right order of the BVH2 (the subtree below the
// split plane first), `axis` = the split axis of the collapsed BVH2 node: a ray with a negative direction component on it
// visits the children in descending order.  Leaves reference ranges of the 40-byte leaf records (HostLeafPrim).
struct HostNode4 {
    float org[3];
    uint32_t exps;      // biased exponents of the grid step: x | y << 8 | z << 16; bits 24..25: split axis
    uint32_t child[4];  // inner: node index; leaf: 0x80000000 | count << 27 | first; empty slot: 0x80000000
    uint32_t qlo[3];    // qlo[axis]: byte k = lower plane of child k
    uint32_t qhi[3];    // qhi[axis]: byte k = upper plane of child k
    uint32_t pad[2];
};
struct HostLeafPrim {  // what a primitive test reads: v0 / e1 / e2 (sphere: centre, radius) and type << 28 | caller's index
    float g[9];
    uint32_t meta;
};

struct HostBvh4 {
    std::vector<HostNode4> nodes;
    uint32_t depth = 0;  // levels of inner nodes (root = 1)
};

namespace bvh_detail {
inline float box_area(const HostNode &n) {
    const float dx = n.hi[0] - n.lo[0], dy = n.hi[1] - n.lo[1], dz = n.hi[2] - n.lo[2];
    return 2.0f * (dx * dy + dy * dz + dz * dx);
}
inline bool is_leaf2(const HostNode &n) { return !(n.b & 0x80000000u); }

inline uint32_t emit4(const HostBvh &b, uint32_t idx2, uint32_t level, HostBvh4 *out) {
    out->depth = std::max(out->depth, level);
    const HostNode &n = b.nodes[idx2];
    std::vector<uint32_t> kids = {n.a, n.a + 1};
    while (kids.size() < 4) {  // open the inner child with the largest box, in place (keeps the left-to-right order)
        int best = -1;
        float ba = -1.0f;
        for (size_t k = 0; k < kids.size(); ++k) {
            const HostNode &c = b.nodes[kids[k]];
            if (is_leaf2(c)) continue;
            const float a = box_area(c);
            if (a > ba) {
                ba = a;
                best = (int)k;
            }
        }
        if (best < 0) break;
        const uint32_t l = b.nodes[kids[best]].a;
        kids[best] = l;
        kids.insert(kids.begin() + best + 1, l + 1);
    }
    const uint32_t my = (uint32_t)out->nodes.size();
    out->nodes.push_back(HostNode4{});
    HostNode4 N{};
    double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (uint32_t k : kids)
        for (int c = 0; c < 3; ++c) {
            lo[c] = std::min(lo[c], (double)b.nodes[k].lo[c]);
            hi[c] = std::max(hi[c], (double)b.nodes[k].hi[c]);
        }
    int e[3];
    for (int c = 0; c < 3; ++c) {
        N.org[c] = (float)lo[c];  // min of floats: exact
        const double ext = hi[c] - lo[c];
        int ex = -126;
        if (ext > 0.0) {
            int fe;
            std::frexp(ext / 255.0, &fe);  // ext / 255 = m * 2^fe, m in [0.5, 1)  =>  2^fe > ext / 255
            ex = std::max(fe, -126);
        }
        while (std::ldexp(255.0, ex) < ext) ++ex;
        e[c] = ex;
    }
    for (int c = 0; c < 3; ++c) {
        uint32_t wl = 0, wh = 0;
        for (size_t k = 0; k < 4; ++k) {
            uint32_t ql = 255, qh = 0;  // empty slot: inverted, never hit
            if (k < kids.size()) {
                const double step = std::ldexp(1.0, e[c]);
                const double fl = std::floor(((double)b.nodes[kids[k]].lo[c] - lo[c]) / step);
                const double ch = std::ceil(((double)b.nodes[kids[k]].hi[c] - lo[c]) / step);
                ql = (uint32_t)std::min(255.0, std::max(0.0, fl));
                qh = (uint32_t)std::min(255.0, std::max(0.0, ch));
                // the grid box must contain the float box (exact in f64: lo + q * 2^e)
                while (ql > 0 && lo[c] + ql * step > (double)b.nodes[kids[k]].lo[c]) --ql;
                while (qh < 255 && lo[c] + qh * step < (double)b.nodes[kids[k]].hi[c]) ++qh;
            }
            wl |= ql << (8 * k);
            wh |= qh << (8 * k);
        }
        N.qlo[c] = wl;
        N.qhi[c] = wh;
    }
    N.exps = (uint32_t)(e[0] + 127) | ((uint32_t)(e[1] + 127) << 8) | ((uint32_t)(e[2] + 127) << 16) | ((n.b & 3u) << 24);
    for (size_t k = 0; k < 4; ++k) {
        if (k >= kids.size()) {
            N.child[k] = 0x80000000u;
            continue;
        }
        const HostNode &c = b.nodes[kids[k]];
        if (is_leaf2(c))
            N.child[k] = 0x80000000u | ((c.b & 15u) << 27) | (c.a & 0x07ffffffu);
        else
            N.child[k] = emit4(b, kids[k], level + 1, out);
    }
    out->nodes[my] = N;
    return my;
}
}  // namespace bvh_detail

inline void to_bvh4(const HostBvh &b, HostBvh4 *out) {
    out->nodes.clear();
    out->depth = 0;
    const HostNode &root = b.nodes[0];
    if (bvh_detail::is_leaf2(root)) {  // the whole scene is one leaf: a node with one child
        HostNode4 N{};
        for (int c = 0; c < 3; ++c) {
            N.org[c] = root.lo[c];
            const double ext = (double)root.hi[c] - (double)root.lo[c];
            int ex = -126;
            while (std::ldexp(255.0, ex) < ext) ++ex;
            const double step = std::ldexp(1.0, ex);
            uint32_t qh = (uint32_t)std::min(255.0, std::ceil(ext / step));
            while (qh < 255 && (double)root.lo[c] + qh * step < (double)root.hi[c]) ++qh;
            N.qlo[c] = 0xffffff00u;  // children 1..3 empty: lo 255
            N.qhi[c] = qh;           // ... hi 0
            N.exps |= (uint32_t)(ex + 127) << (8 * c);
        }
        N.child[0] = 0x80000000u | ((root.b & 15u) << 27) | (root.a & 0x07ffffffu);
        N.child[1] = N.child[2] = N.child[3] = 0x80000000u;
        out->nodes.push_back(N);
        out->depth = 1;
        return;
    }
    bvh_detail::emit4(b, 0, 1, out);
    // Breadth-first order: nodes [0, n) are the top of the tree for every n, so a kernel whose tree lives in global memory can keep
    // its first nodes -- the ones every ray visits -- in LDS (kernels_wavefront.h: k_trace, ACCEL_K_BVH_GLOBAL).
    const size_t n = out->nodes.size();
    std::vector<uint32_t> bfs;  // bfs[new] = old
    bfs.reserve(n);
    bfs.push_back(0);
    for (size_t i = 0; i < bfs.size(); ++i)
        for (uint32_t c : out->nodes[bfs[i]].child)
            if (!(c & 0x80000000u)) bfs.push_back(c);
    std::vector<uint32_t> renum(n);
    for (size_t i = 0; i < n; ++i) renum[bfs[i]] = (uint32_t)i;
    std::vector<HostNode4> sorted(n);
    for (size_t i = 0; i < n; ++i) {
        HostNode4 N = out->nodes[bfs[i]];
        for (uint32_t &c : N.child)
            if (!(c & 0x80000000u)) c = renum[c];
        sorted[i] = N;
    }
    out->nodes.swap(sorted);
}

// the leaf records in leaf order (order[slot] = caller's index)
inline void make_leaf_prims(const pbrt_prim *prims, const std::vector<uint32_t> &order, std::vector<HostLeafPrim> *out) {
    out->resize(order.size());
    for (size_t s = 0; s < order.size(); ++s) {
        const pbrt_prim &P = prims[order[s]];
        HostLeafPrim L{};
        for (int k = 0; k < 9; ++k) L.g[k] = P.g[k];  // cone: the record is read from the full table (meta holds the index)
        L.meta = (P.type << 28) | (order[s] & 0x0fffffffu);
        (*out)[s] = L;
    }
}

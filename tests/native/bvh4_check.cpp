// Host check of the BVH4 builder (csrc/bvh_build.h) and of the traversal scheme of device_scene.h (sorted children, the
// short stack with its overflow), restated in plain C++: every ray must find the same closest primitive as a loop over
// all triangles, in a bounded number of node visits.  Reads triangles "x y z x y z x y z" per line from stdin.
#include <cstdio>
#include <cstring>
#include <random>
#include "../../physics-based-ray-tracing_amd/csrc/bvh_build.h"

struct V3 { float x, y, z; };
static V3 sub(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
static float dot(V3 a, V3 b) { return fmaf(a.x, b.x, fmaf(a.y, b.y, a.z * b.z)); }
static V3 cross(V3 a, V3 b) { return {fmaf(a.y, b.z, -(a.z * b.y)), fmaf(a.z, b.x, -(a.x * b.z)), fmaf(a.x, b.y, -(a.y * b.x))}; }
static bool tri_hit(const float *g, V3 o, V3 d, float tmax, float *t) {
    V3 v0{g[0], g[1], g[2]}, e1{g[3], g[4], g[5]}, e2{g[6], g[7], g[8]};
    V3 pvec = cross(d, e2);
    float det = dot(e1, pvec);
    V3 tvec = sub(o, v0), qvec = cross(tvec, e1);
    float us = dot(tvec, pvec), vs = dot(d, qvec), ts = dot(e2, qvec);
    if (det < 0) { det = -det; us = -us; vs = -vs; ts = -ts; }
    if (!(det > 0 && us >= 0 && vs >= 0 && ts >= 0 && us + vs <= det)) return false;
    float tt = ts * (1.0f / det);
    if (!(tt <= tmax)) return false;
    *t = tt;
    return true;
}
static const uint32_t SENT = 0xffffffffu;
// device_scene.h BvhStack / bvh_push / bvh_pop: one entry per node (index << 8 | count << 6 | sorted slots), the newest in `tos`
struct Stack {
    uint32_t rows[64]; uint32_t n_rows; uint32_t ovf[30];
    uint32_t cur, tos, sp;
};
static void push(Stack &c, bool on, uint32_t entry) {
    if (!on) return;
    if (c.sp < c.n_rows) c.rows[c.sp] = c.tos; else c.ovf[std::min(c.sp - c.n_rows, 29u)] = c.tos;
    c.sp += 1; c.tos = entry;
}
static uint32_t pop(const HostBvh4 &b4, Stack &c) {
    const uint32_t e = c.tos;
    if (e == SENT) return SENT;
    const uint32_t ref = b4.nodes[e >> 8].child[e & 3u], n = (e >> 6) & 3u;
    if (n > 1u) {
        c.tos = (e & 0xffffff00u) | ((n - 1u) << 6) | ((e & 0x3fu) >> 2);
    } else {
        const uint32_t sp1 = c.sp - 1u;
        c.tos = sp1 < c.n_rows ? c.rows[sp1] : c.ovf[std::min(sp1 - c.n_rows, 29u)];
        c.sp = sp1;
    }
    return ref;
}
static void cex(uint32_t &ka, uint32_t &kb) { uint32_t lo = std::min(ka, kb), hi = std::max(ka, kb); ka = lo; kb = hi; }
int main(int argc, char **argv) {
    std::vector<pbrt_prim> prims;
    float v[9];
    while (scanf("%f %f %f %f %f %f %f %f %f", v, v + 1, v + 2, v + 3, v + 4, v + 5, v + 6, v + 7, v + 8) == 9) {
        pbrt_prim P{};
        for (int k = 0; k < 3; ++k) { P.g[k] = v[k]; P.g[3 + k] = v[3 + k] - v[k]; P.g[6 + k] = v[6 + k] - v[k]; }
        P.type = PBRT_PRIM_TRIANGLE;
        prims.push_back(P);
    }
    const uint32_t n = (uint32_t)prims.size();
    // argv[3]: the SAH traversal cost (default: the constant of trees that fit the LDS; the library builds the others with BVH_CTRAV_GLOBAL)
    HostBvh b; build_bvh(prims.data(), n, &b, argc > 3 ? (float)atof(argv[3]) : BVH_CTRAV);
    HostBvh4 b4; to_bvh4(b, &b4);
    std::vector<HostLeafPrim> lp; make_leaf_prims(prims.data(), b.order, &lp);
    // structural check: every leaf record's box lies inside the grid box of its slot in the parent, all the way up
    uint64_t bad = 0;
    struct Item { uint32_t node; double lo[3], hi[3]; };
    std::vector<Item> todo{{0, {-1e30, -1e30, -1e30}, {1e30, 1e30, 1e30}}};
    while (!todo.empty()) {
        Item it = todo.back(); todo.pop_back();
        const HostNode4 &N = b4.nodes[it.node];
        for (int k = 0; k < 4; ++k) {
            double lo[3], hi[3];
            for (int c = 0; c < 3; ++c) {
                int e = (int)((N.exps >> (8 * c)) & 0xff) - 127;
                lo[c] = (double)N.org[c] + std::ldexp((double)((N.qlo[c] >> (8 * k)) & 0xff), e);
                hi[c] = (double)N.org[c] + std::ldexp((double)((N.qhi[c] >> (8 * k)) & 0xff), e);
            }
            uint32_t cr = N.child[k];
            if (cr & 0x80000000u) {
                uint32_t first = cr & 0x07ffffffu, cnt = (cr >> 27) & 15u;
                for (uint32_t s = first; s < first + cnt; ++s)
                    for (int c = 0; c < 3; ++c) {
                        float a0 = lp[s].g[c], a1 = a0 + lp[s].g[3 + c], a2 = a0 + lp[s].g[6 + c];
                        if (std::min({a0, a1, a2}) < lo[c] || std::max({a0, a1, a2}) > hi[c]) ++bad;
                        if (lo[c] < it.lo[c] - 1e-12 && false) ++bad;
                    }
            } else {
                Item ch{cr, {lo[0], lo[1], lo[2]}, {hi[0], hi[1], hi[2]}};
                todo.push_back(ch);
            }
        }
    }
    // breadth-first order (bvh_build.h to_bvh4): the inner children of node i, taken over i = 0, 1, 2, ..., are 1, 2, 3, ... -- every
    // prefix of the array is the top of the tree
    {
        uint32_t next = 1;
        for (size_t i = 0; i < b4.nodes.size(); ++i)
            for (uint32_t cr : b4.nodes[i].child)
                if (!(cr & 0x80000000u) && cr != next++) ++bad;
        if (next != b4.nodes.size()) ++bad;
    }
    // traversal check
    std::mt19937 rng(7);
    std::uniform_real_distribution<float> U(-1.f, 1.f);
    float lo[3] = {1e30f, 1e30f, 1e30f}, hi[3] = {-1e30f, -1e30f, -1e30f};
    for (auto &P : prims) for (int c = 0; c < 3; ++c) { lo[c] = std::min(lo[c], P.g[c]); hi[c] = std::max(hi[c], P.g[c]); }
    const uint32_t n_rows = argc > 1 ? (uint32_t)atoi(argv[1]) : 4;
    uint64_t mism = 0, visits = 0, maxsp = 0, runaway = 0;
    const int NR = argc > 2 ? atoi(argv[2]) : 200000;
    for (int r = 0; r < NR; ++r) {
        V3 o{lo[0] + (hi[0] - lo[0]) * (0.5f + U(rng)), lo[1] + (hi[1] - lo[1]) * (0.5f + U(rng)), lo[2] + (hi[2] - lo[2]) * (0.5f + U(rng))};
        V3 d{U(rng), U(rng), U(rng)};
        float l = std::sqrt(dot(d, d)); if (!(l > 1e-3f)) continue;
        d = {d.x / l, d.y / l, d.z / l};
        if (r % 7 == 0) d.x = 0;
        if (r % 11 == 0) d.y = 0;
        float bt = INFINITY; uint32_t bi = SENT;
        for (uint32_t i = 0; i < n; ++i) { float t; if (tri_hit(prims[i].g, o, d, bt, &t) && (bi == SENT || t < bt || (t == bt && i < bi))) { bt = t; bi = i; } }
        const float tiny = 1e-18f;
        V3 ds{std::fabs(d.x) < tiny ? std::copysign(tiny, d.x) : d.x, std::fabs(d.y) < tiny ? std::copysign(tiny, d.y) : d.y, std::fabs(d.z) < tiny ? std::copysign(tiny, d.z) : d.z};
        V3 inv{1.0f / ds.x, 1.0f / ds.y, 1.0f / ds.z};
        Stack c{}; c.n_rows = n_rows; c.cur = 0; c.tos = SENT; c.sp = 0;
        float best = INFINITY; uint32_t hid = SENT; bool found = false; uint64_t vis = 0;
        for (;;) {
            while ((int32_t)c.cur >= 0) {
                if (++vis > 100000) break;
                const HostNode4 &N = b4.nodes[c.cur];
                float A[3], B[3]; const float iv[3] = {inv.x, inv.y, inv.z}, oc[3] = {o.x, o.y, o.z};
                uint32_t qn[3], qf[3];
                for (int a = 0; a < 3; ++a) {
                    uint32_t eb = ((N.exps >> (8 * a)) & 0xff) << 23; float s; memcpy(&s, &eb, 4);
                    A[a] = s * iv[a]; B[a] = (N.org[a] - oc[a]) * iv[a];
                    bool neg = iv[a] < 0; qn[a] = neg ? N.qhi[a] : N.qlo[a]; qf[a] = neg ? N.qlo[a] : N.qhi[a];
                }
                uint32_t key[4];
                for (int k = 0; k < 4; ++k) {
                    float tn = 0, tf = best;
                    for (int a = 0; a < 3; ++a) {
                        tn = std::fmax(tn, fmaf((float)((qn[a] >> (8 * k)) & 0xff), A[a], B[a]));
                        tf = std::fmin(tf, fmaf((float)((qf[a] >> (8 * k)) & 0xff), A[a], B[a]));
                    }
                    uint32_t tb; memcpy(&tb, &tn, 4);
                    key[k] = (tn <= tf) ? ((tb & ~3u) | (uint32_t)k) : SENT;
                }
                cex(key[0], key[1]); cex(key[2], key[3]); cex(key[0], key[2]); cex(key[1], key[3]); cex(key[1], key[2]);
                if (key[0] == SENT) { c.cur = pop(b4, c); continue; }
                const uint32_t node = c.cur;
                c.cur = N.child[key[0] & 3u];
                const uint32_t nm = (key[1] != SENT) + (key[2] != SENT) + (key[3] != SENT);
                push(c, nm != 0, (node << 8) | (nm << 6) | (key[1] & 3u) | ((key[2] & 3u) << 2) | ((key[3] & 3u) << 4));
                maxsp = std::max<uint64_t>(maxsp, c.sp);
            }
            if (vis > 100000) { ++runaway; break; }
            if (c.cur == SENT) break;
            uint32_t first = c.cur & 0x07ffffffu, cnt = (c.cur >> 27) & 15u;
            for (uint32_t k = 0; k < cnt; ++k) {
                float t; uint32_t id = lp[first + k].meta & 0x0fffffffu;
                if (tri_hit(lp[first + k].g, o, d, best, &t) && (!found || t < best || (t == best && id < hid))) { best = t; hid = id; found = true; }
            }
            c.cur = pop(b4, c);
        }
        visits += vis;
        if ((found ? hid : SENT) != bi || (found && best != bt)) ++mism;
    }
    printf("{\"prims\": %u, \"nodes4\": %zu, \"depth4\": %u, \"image_bytes\": %zu, \"outside_grid\": %llu, \"mismatches\": %llu, \"runaway\": %llu, "
           "\"visits_per_ray\": %.2f, \"max_sp\": %llu}\n",
           n, b4.nodes.size(), b4.depth, b4.nodes.size() * 64 + (size_t)n * 40, (unsigned long long)bad, (unsigned long long)mism,
           (unsigned long long)runaway, (double)visits / NR, (unsigned long long)maxsp);
    return (bad || mism || runaway) ? 1 : 0;
}

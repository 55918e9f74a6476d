// kernels_wavefront.h -- radiance mode on BVH scenes (gfx950): intersection and shading as separate streams.
//
// The fused bounce kernel (k_bounce<.., BVH>) walked the tree until the LAST lane of a wave had finished, shaded in whatever
// lanes had hit something and walked the tree again for their shadow rays, with the registers of all three parts live at
// once (96 - 128 VGPRs: 4 waves per SIMD beside the LDS image).  Here a bounce is two launches:
//   k_trace  a STREAM of ray queries against the LDS-resident BVH4: the shadow rays the previous bounce emitted (any hit) and
//            the continuation rays of the live paths (closest hit).  A lane whose ray has finished takes the next ray of the
//            workgroup's queue -- the record of that ray was requested one refill earlier, so no lane waits on HBM -- and
//            joins the two-phase walk of the others.  It reads 32 bytes per ray and writes 16 (hit) or 4 (visibility);
//            nothing else is live, so two 1024-thread workgroups with their own image fit a CU (8 waves per SIMD).
//   k_shade  per region: (1) adds the contributions of the previous bounce's unoccluded shadow rays to the radiance of their
//            paths, in the fma order of the fused kernel; (2) ends the paths whose ray left the scene and lists the others;
//            (3) shades the listed hits in full waves: emission + MIS, emitter sample -> shadow record, BSDF sample, Russian
//            roulette; survivors and shadow records are packed to the front of the region (ballot + one LDS atomic per wave).
// Same arithmetic per path, same RNG keys, same order of the radiance sums as bounce_step: the film does not change by a bit.
//
// Layouts (all float4 records, 64-byte path and shadow records so that a lane's gather is one or two 32-byte sectors):
//   path state  q0 = (o, eta)  q1 = (d, prev_pdf)  q2 = (throughput, home)  q3 = (L, -)
//   hit         (t, u, v, primitive index | 0xffffffff: none)
//   shadow ray  q0 = (origin, tmax; k_trace overwrites tmax by the visibility 1 / 0)  q1 = (direction, dest)  q2 = (A, -)  q3 = (B, -)
//               the contribution is L = fma(A, B, L) per channel; dest = state slot of the survivor, or 0x80000000 | home
#pragma once
#include "kernels_radiance.h"

#define WF_REGION 4096u      // slots per region (compaction domain of k_shade)
#define WF_KMAX 32u          // regions one k_trace workgroup walks at most
#ifndef WF_REFILL_MIN
#define WF_REFILL_MIN 16u    // idle lanes that make a wave fetch new rays
#endif
#ifndef WF_WALK_MIN
#define WF_WALK_MIN 16u      // the node walk of a turn stops when fewer lanes are still in it (while new rays can be had)
#endif
#ifndef WF_TRACE_WAVES_PER_EU
#define WF_TRACE_WAVES_PER_EU 8
#endif
#define WF_SHADE_THREADS 256u
#define WF_DEAD 0x80000000u

struct WfArgs {
    DevScene sc;
    pbrt_camera cam;
    float4 *st_in, *st_out;      // [cap][4] path state
    float4 *hits;                // [cap]
    float4 *shd_in, *shd_out;    // [cap][4] shadow rays emitted by the previous / this bounce
    float *Lhome;                // [cap] float4 records (r, g, b, 0) indexed by home
    const uint32_t *seg_in, *nsh_in;   // live paths / shadow rays of the previous bounce per region (nsh_in: nullptr at depth 0)
    uint32_t *seg_out, *nsh_out;
    unsigned long long *stats;   // per-region rows as in RadArgs
    uint32_t stat_stride, cap, n_paths, n_regions;
    uint32_t depth, max_depth, rr_depth, seed;
    uint32_t key_mode, rx0, ry0, rw, npix_r, s_first, film_w, film_h, tile_rows;
    FastDiv div_npix, div_rw;
    uint32_t index_offset, sample_index;
    uint32_t lds_bytes;          // ACCEL_K_BVH_LDS: bytes of the staged image
    uint32_t stk_rows;           // rows of the traversal stacks behind the image (BvhStack::n_rows; one more row follows)
};

DEV RadArgs wf_key_args(const WfArgs &a) {  // path_key reads these members only
    RadArgs r;
    r.key_mode = a.key_mode;
    r.rx0 = a.rx0;
    r.ry0 = a.ry0;
    r.rw = a.rw;
    r.npix_r = a.npix_r;
    r.s_first = a.s_first;
    r.film_w = a.film_w;
    r.film_h = a.film_h;
    r.tile_rows = a.tile_rows;
    r.div_npix = a.div_npix;
    r.div_rw = a.div_rw;
    r.index_offset = a.index_offset;
    r.sample_index = a.sample_index;
    return r;
}
DEV void wf_camera_ray(const WfArgs &a, uint32_t home, V3 *o, V3 *d, float *tmax, uint32_t *ka, uint32_t *kb) {
    uint32_t px, py;
    const RadArgs ra = wf_key_args(a);
    path_key<true>(ra, home, ka, kb, &px, &py);
    F4 uj = rng4(*ka, *kb, 0, a.seed);
    float fx = (float)px + uj.x, fy = (float)py + uj.y;
    camera_ray(a.cam, fx / (float)a.film_w, fy / (float)a.film_h, o, d, tmax);
}

// Every loop of the stream has an exit that each wave reaches; the guard below is the net under it: a wave that exceeds
// WF_GUARD_TURNS turns leaves, records its state in g_wf_guard and the host reports PBRT_E_DEVICE instead of hanging the box.
#define WF_GUARD_TURNS (1u << 22)
__device__ uint32_t g_wf_guard[16];  // [0] trips, [1..] state of the last wave that tripped

// ---- k_trace ---------------------------------------------------------------------------------------------------------------
// grid: G workgroups; workgroup w walks the regions w, w + G, w + 2 G, ... (at most WF_KMAX of them) as ONE queue: for each of
// its regions first the shadow rays, then the continuation rays.  dynamic LDS: [image | stack rows].
template <bool FIRST, int ACCEL>
__global__ __launch_bounds__(1024, WF_TRACE_WAVES_PER_EU) void k_trace(const WfArgs a) {
    static_assert(ACCEL == ACCEL_K_BVH_GLOBAL || ACCEL == ACCEL_K_BVH_LDS, "k_trace: BVH scenes");
    extern __shared__ __attribute__((aligned(16))) uint32_t dyn_lds[];
    __shared__ uint32_t q_in;
    __shared__ uint32_t cum[2 * WF_KMAX + 1];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, T = blockDim.x, G = gridDim.x;
    const uint32_t K = (a.n_regions - blockIdx.x + G - 1u) / G;  // regions of this workgroup (host: <= WF_KMAX)
    if (tid == 0) {
        uint32_t run = 0;
        for (uint32_t j = 0; j < K; ++j) {
            const uint32_t r = blockIdx.x + j * G;
            cum[2 * j] = run;
            run += (!FIRST && a.nsh_in) ? a.nsh_in[r] : 0u;
            cum[2 * j + 1] = run;
            const uint32_t b = r * WF_REGION;
            run += FIRST ? (a.n_paths > b ? min(a.n_paths - b, WF_REGION) : 0u) : a.seg_in[r];
        }
        cum[2 * K] = run;
        q_in = 0;
    }
    LdsScene ls;
    uint32_t *stk_lds = dyn_lds;
    if (ACCEL == ACCEL_K_BVH_LDS) {
        stage_scene_lds(a.sc, dyn_lds, &ls);  // ends with a barrier
        stk_lds = dyn_lds + (a.lds_bytes >> 2);
    } else {
        ls.nodes = a.sc.nodes;
        ls.lprims = a.sc.lprims;
        __syncthreads();
    }
    const BvhStack st = {stk_lds + tid, T, a.stk_rows};
    const uint32_t total = cum[2 * K];
    if (total == 0) return;  // uniform
    const auto nodes = ls.nodes;
    const auto lprims = ls.lprims;

    bool busy = false, found = false;
    uint32_t rslot = 0;  // slot of the ray's record | shadow ray << 31
    V3 o = {0, 0, 0}, d = {0, 0, 1};
    BoxRay br = make_box_ray(o, d);
    float best = 0.0f, hu = 0.0f, hv = 0.0f;
    uint32_t hid = 0xffffffffu;
    BvhCursor c;
    c.cur = BVH_SENT;
    c.tos = BVH_SENT;
    c.sp = 0;
    // the record of the lane's NEXT ray, requested one refill ahead
    bool has_next = false;
    uint32_t n_slot = 0;
    float4 n_q0 = {0, 0, 0, 0}, n_q1 = {0, 0, 1, 0};
    bool q_empty = false;   // wave-uniform
    uint32_t s_hint = 0;    // wave-uniform: segment of the wave's last fetch (queue indices only grow)
    uint32_t turns = 0, visits = 0;
    for (;;) {
        if (++turns > WF_GUARD_TURNS) {
            const unsigned long long bb = __ballot(busy), bi = __ballot(busy && (int32_t)c.cur >= 0), bn = __ballot(has_next);
            if (lane == 0) {
                atomicAdd(&g_wf_guard[0], 1u);
                g_wf_guard[1] = blockIdx.x;
                g_wf_guard[2] = tid >> 6;
                g_wf_guard[3] = (uint32_t)__popcll(bb);
                g_wf_guard[4] = (uint32_t)__popcll(bi);
                g_wf_guard[5] = (uint32_t)__popcll(bn);
                g_wf_guard[6] = q_empty ? 1u : 0u;
                g_wf_guard[7] = total;
                g_wf_guard[8] = q_in;
                g_wf_guard[9] = visits;
                g_wf_guard[10] = a.depth | (FIRST ? 0x100u : 0u);
                g_wf_guard[11] = K;
            }
            break;
        }
        // ---- retire
        if (busy && c.cur == BVH_SENT) {
            if (rslot & 0x80000000u) {
                reinterpret_cast<float *>(a.shd_in)[(size_t)(rslot & 0x7fffffffu) * 16u + 3u] = found ? 0.0f : 1.0f;
            } else {
                const float4 rec = {best, hu, hv, __uint_as_float(found ? hid : 0xffffffffu)};
                a.hits[rslot] = rec;
            }
            busy = false;
        }
        const uint32_t n_idle = (uint32_t)__popcll(__ballot(!busy));
        // a refill helps only through rays that IDLE lanes can take: new queue entries, or their own prefetched records
        const bool idle_next = !FIRST && __ballot(!busy && has_next) != 0;
        if ((n_idle >= WF_REFILL_MIN || n_idle == 64u) && (!q_empty || idle_next)) {
            if (!FIRST && !busy && has_next) {  // the prefetched record becomes the lane's ray
                rslot = n_slot;
                o = {n_q0.x, n_q0.y, n_q0.z};
                d = {n_q1.x, n_q1.y, n_q1.z};
                best = (rslot & 0x80000000u) ? n_q0.w : ((a.key_mode == 1 && a.depth == 0) ? n_q0.w : K_INF);
                has_next = false;
                busy = true;
                found = false;
                br = make_box_ray(o, d);
                c.cur = 0;
                c.tos = BVH_SENT;
                c.sp = 0;
            }
            if (!q_empty) {
                const bool want = FIRST ? !busy : !has_next;
                const unsigned long long bw = __ballot(want);
                const uint32_t nw = (uint32_t)__popcll(bw);
                if (nw) {
                    uint32_t got = 0;
                    if (lane == 0) got = atomicAdd(&q_in, nw);
                    const uint32_t i0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)got);
                    if (i0 + nw >= total) q_empty = true;
                    const uint32_t i = i0 + __builtin_amdgcn_mbcnt_hi((uint32_t)(bw >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bw, 0u));
                    if (i0 < total) {
                        while (s_hint + 1u < 2u * K && i0 >= cum[s_hint + 1u]) ++s_hint;  // uniform
                    }
                    if (want && i < total) {
                        uint32_t s = s_hint;
                        while (i >= cum[s + 1u]) ++s;
                        const uint32_t r = blockIdx.x + (s >> 1) * G;
                        const uint32_t slot = r * WF_REGION + (i - cum[s]);
                        const bool shadow = !(s & 1u);
                        if (FIRST) {
                            uint32_t ka, kb;
                            float tm;
                            wf_camera_ray(a, slot, &o, &d, &tm, &ka, &kb);
                            rslot = slot;
                            best = tm;
                            busy = true;
                            found = false;
                            br = make_box_ray(o, d);
                            c.cur = 0;
                            c.tos = BVH_SENT;
                            c.sp = 0;
                        } else {
                            const float4 *rec = (shadow ? a.shd_in : a.st_in) + (size_t)slot * 4u;
                            n_q0 = rec[0];
                            n_q1 = rec[1];
                            n_slot = slot | (shadow ? 0x80000000u : 0u);
                            has_next = true;
                        }
                    }
                }
            }
        }
        if (__ballot(busy) == 0) {
            if (FIRST || __ballot(has_next) == 0) {
                if (q_empty) break;
            }
            continue;
        }
        // ---- one turn of the walk: inner nodes until every busy lane holds a leaf or has finished (or, while new rays can be
        // had, until fewer than WF_WALK_MIN lanes are still walking), then the held leaves
        const bool can_refill = !q_empty;  // wave-uniform (a prefetched record only serves its own lane, once that lane is idle)
        while (busy && (int32_t)c.cur >= 0) {
            if (can_refill && (uint32_t)__popcll(__ballot(true)) < WF_WALK_MIN) break;
            if (++visits > WF_GUARD_TURNS) {  // (the main loop's guard reports)
                turns = WF_GUARD_TURNS;
                break;
            }
            bvh_visit(nodes, st, c, br, best);
        }
        if (busy && (int32_t)c.cur < 0 && c.cur != BVH_SENT) {
            const uint32_t first = c.cur & 0x07ffffffu, count = (c.cur >> 27) & 15u;
            const bool any = (rslot & 0x80000000u) != 0u;
            bool stop = false;
            for (uint32_t k = 0; k < count && !stop; ++k) {
                float t, u, v;
                uint32_t id;
                if (lprim_hit(lprims, first + k, a.sc.prims, o, d, best, &t, &u, &v, &id)) {
                    if (any) {
                        found = true;
                        stop = true;
                    } else if (!found || t < best || (t == best && id < hid)) {
                        best = t;
                        hu = u;
                        hv = v;
                        hid = id;
                        found = true;
                    }
                }
            }
            c.cur = bvh_pop(st, c);
            if (stop) c.cur = BVH_SENT;
        }
    }
}

// ---- k_shade ---------------------------------------------------------------------------------------------------------------
struct WfShadow {
    V3 so, sdir, A, B;
    float tmax;
    bool on;
};
// bounce_step from the shading on, with the shadow ray handed out instead of traced (same statements, same order)
template <int ACCEL>
DEV bool wf_shade_step(const WfArgs &a, const Tables &tb, uint32_t depth, uint32_t ka, uint32_t kb, const Hit &h, V3 &o, V3 &d,
                       V3 &thr, V3 &L, float &eta, float &prev_pdf, WfShadow &sh) {
    bool survive = false;
    sh.on = false;
    const uint32_t nE = a.sc.n_emitters;
    const pbrt_prim &P = tb.prims_by_slot[h.slot];
    SI si = make_si<true>(P, o, d, h.t, h.u, h.v, a.sc.vnormals, h.slot);
    const int32_t emitter = P.emitter;
    const uint32_t mat_id = P.material;
    if (emitter >= 0) {
        const pbrt_emitter &E = tb.emitters[emitter];
        float cosl = -dot(si.ns, d);
        if (cosl > 0.0f) {
            float w = 1.0f;
            if (prev_pdf >= 0.0f) {
                float pdf_em = (h.t * h.t) / (cosl * E.area * (float)nE);
                w = mis_weight(prev_pdf, pdf_em);
            }
            L = {fma_(thr.x * E.radiance[0], w, L.x), fma_(thr.y * E.radiance[1], w, L.y), fma_(thr.z * E.radiance[2], w, L.z)};
        }
    }
    if (depth + 1 < a.max_depth) {
        const pbrt_material M = tb.mats[mat_id];
        Frame fr = make_frame(si.ns);
        V3 wi = to_local(fr, -d);
        if (M.type == PBRT_MAT_DIFFUSE && nE > 0) {
            F4 u = rng4(ka, kb, 1 + 2 * depth, a.seed);
            ESample es = sample_emitter(tb, si.p, u);
            if (es.valid) {
                V3 wo = to_local(fr, es.d);
                V3 f;
                float bpdf;
                bsdf_eval_pdf(M, wi, wo, &f, &bpdf);
                if (bpdf > 0.0f) {
                    V3 so = offset_origin(si.p, si.n, es.d);
                    V3 sv = es.q - so;
                    float sd = sqrtf(dot(sv, sv));
                    V3 sdir = sv * (1.0f / sd);
                    float mis = es.delta ? 1.0f : mis_weight(es.pdf, bpdf);
                    sh.on = true;
                    sh.so = so;
                    sh.sdir = sdir;
                    sh.tmax = sd * (1.0f - K_SHADOW_EPS);
                    sh.A = {thr.x * f.x, thr.y * f.y, thr.z * f.z};
                    sh.B = {es.weight.x * mis, es.weight.y * mis, es.weight.z * mis};
                }
            }
        }
        F4 ub = rng4(ka, kb, 2 + 2 * depth, a.seed);
        BSample bs = bsdf_sample(M, PBRT_USQ_REFERENCE, wi, si.n, si.ns, fr, ub.x, ub.y, ub.z);
        if (bs.valid) {
            thr = thr * bs.weight;
            eta *= bs.eta;
            V3 nd = to_world(fr, bs.wo);
            if (M.type == PBRT_MAT_ULTRA) nd = normalize(nd);
            o = offset_origin(si.p, si.n, nd);
            d = nd;
            prev_pdf = bs.delta ? -1.0f : bs.pdf;
            float tm = max3(thr);
            survive = true;
            if (depth + 1 >= a.rr_depth) {
                float q = fminf(tm * eta * eta, 0.95f);
                float rq = 1.0f / q;
                thr = thr * rq;
                if (!(ub.w < q)) survive = false;
            }
            if (tm == 0.0f) survive = false;
        }
    }
    return survive;
}

template <bool FIRST, int ACCEL>
__global__ __launch_bounds__(WF_SHADE_THREADS) void k_shade(const WfArgs a) {
    constexpr uint32_t T = WF_SHADE_THREADS, W = T / 64;
    __shared__ uint32_t hit_list[WF_REGION];
    __shared__ uint32_t n_hit, q_out, q_shd;
    const uint32_t r = xcd_swizzle(blockIdx.x, gridDim.x), base = r * WF_REGION;
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint32_t cnt_in = FIRST ? (a.n_paths > base ? min(a.n_paths - base, WF_REGION) : 0u) : a.seg_in[r];
    const uint32_t n_shd = (!FIRST && a.nsh_in) ? a.nsh_in[r] : 0u;
    if (cnt_in == 0 && n_shd == 0) {  // uniform
        if (tid == 0) {
            a.seg_out[r] = 0;
            a.nsh_out[r] = 0;
        }
        return;
    }
    if (tid == 0) {
        n_hit = 0;
        q_out = 0;
        q_shd = 0;
    }
    float4 *Lh = reinterpret_cast<float4 *>(a.Lhome);
    // ---- (1) the unoccluded shadow rays of the previous bounce: L = fma(A, B, L) on the path's radiance
    for (uint32_t k = tid; k < n_shd; k += T) {
        const float4 *rec = a.shd_in + (size_t)(base + k) * 4u;
        const float4 q0 = rec[0];
        if (q0.w != 0.0f) {
            const float4 q1 = rec[1], A = rec[2], B = rec[3];
            const uint32_t dest = __float_as_uint(q1.w);
            float4 *Lp = (dest & WF_DEAD) ? Lh + (dest & 0x7fffffffu) : a.st_in + (size_t)dest * 4u + 3u;
            float4 Lv = *Lp;
            Lv.x = fma_(A.x, B.x, Lv.x);
            Lv.y = fma_(A.y, B.y, Lv.y);
            Lv.z = fma_(A.z, B.z, Lv.z);
            *Lp = Lv;
        }
    }
    __threadfence_block();
    __syncthreads();
    // ---- (2) paths whose ray left the scene end here; the others are listed
    for (uint32_t s0 = 0; s0 < cnt_in; s0 += T) {
        const uint32_t s = s0 + tid;
        const bool valid = s < cnt_in;
        bool is_hit = false;
        if (valid) {
            const uint32_t id = __float_as_uint(a.hits[base + s].w);
            is_hit = id != 0xffffffffu;
            if (!is_hit) {
                if (FIRST) {
                    const float4 z = {0.0f, 0.0f, 0.0f, 0.0f};
                    Lh[base + s] = z;
                } else {
                    const float4 *stp = a.st_in + (size_t)(base + s) * 4u;
                    const uint32_t home = __float_as_uint(stp[2].w);
                    float4 Lv = stp[3];
                    Lv.w = 0.0f;
                    Lh[home] = Lv;
                }
            }
        }
        const unsigned long long bh = __ballot(is_hit);
        if (bh) {
            uint32_t got = 0;
            if (lane == 0) got = atomicAdd(&n_hit, (uint32_t)__popcll(bh));
            const uint32_t off = (uint32_t)__builtin_amdgcn_readfirstlane((int)got);
            if (is_hit) hit_list[off + __builtin_amdgcn_mbcnt_hi((uint32_t)(bh >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bh, 0u))] = s;
        }
    }
    __syncthreads();
    const uint32_t nh = n_hit;
    const Tables tb = global_tables(a.sc);
    uint32_t n_shadow_w = 0;
    // ---- (3) the listed hits, in full waves
    for (uint32_t e0 = 0; e0 < nh; e0 += T) {
        const uint32_t e = e0 + tid;
        const bool act = e < nh;
        bool survive = false;
        WfShadow sh;
        sh.on = false;
        V3 o = {0, 0, 0}, d = {0, 0, 1}, thr = {1, 1, 1}, L = {0, 0, 0};
        float eta = 1.0f, prev_pdf = -1.0f;
        uint32_t home = 0;
        if (act) {
            const uint32_t s = hit_list[e];
            const float4 hr = a.hits[base + s];
            Hit h;
            h.t = hr.x;
            h.u = hr.y;
            h.v = hr.z;
            h.prim = __float_as_uint(hr.w);
            h.slot = h.prim;
            uint32_t ka, kb;
            if (FIRST) {
                float tm;
                home = base + s;
                wf_camera_ray(a, home, &o, &d, &tm, &ka, &kb);
            } else {
                const float4 *stp = a.st_in + (size_t)(base + s) * 4u;
                const float4 q0 = stp[0], q1 = stp[1], q2 = stp[2], q3 = stp[3];
                o = {q0.x, q0.y, q0.z};
                d = {q1.x, q1.y, q1.z};
                thr = {q2.x, q2.y, q2.z};
                L = {q3.x, q3.y, q3.z};
                eta = (a.key_mode == 1 && a.depth == 0) ? 1.0f : q0.w;  // caller rays carry tmax in the eta slot
                prev_pdf = q1.w;
                home = __float_as_uint(q2.w);
                uint32_t px, py;
                const RadArgs ra = wf_key_args(a);
                path_key<true>(ra, home, &ka, &kb, &px, &py);
            }
            survive = wf_shade_step<ACCEL>(a, tb, a.depth, ka, kb, h, o, d, thr, L, eta, prev_pdf, sh);
        }
        // survivors -> front of the region of the `out` state
        const unsigned long long bs = __ballot(survive);
        uint32_t out_slot = 0;
        {
            uint32_t got = 0;
            if (lane == 0 && bs) got = atomicAdd(&q_out, (uint32_t)__popcll(bs));
            const uint32_t off = (uint32_t)__builtin_amdgcn_readfirstlane((int)got);
            out_slot = base + off + __builtin_amdgcn_mbcnt_hi((uint32_t)(bs >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bs, 0u));
        }
        if (survive) {
            float4 *stp = a.st_out + (size_t)out_slot * 4u;
            const float4 q0 = {o.x, o.y, o.z, eta}, q1 = {d.x, d.y, d.z, prev_pdf}, q2 = {thr.x, thr.y, thr.z, __uint_as_float(home)},
                         q3 = {L.x, L.y, L.z, 0.0f};
            stp[0] = q0;
            stp[1] = q1;
            stp[2] = q2;
            stp[3] = q3;
        } else if (act) {  // the path ends (its pending shadow ray, if any, is added to this record by the next k_shade)
            const float4 rec = {L.x, L.y, L.z, 0.0f};
            Lh[home] = rec;
        }
        const unsigned long long bsh = __ballot(sh.on);
        if (bsh) {
            uint32_t got = 0;
            if (lane == 0) got = atomicAdd(&q_shd, (uint32_t)__popcll(bsh));
            const uint32_t off = (uint32_t)__builtin_amdgcn_readfirstlane((int)got);
            n_shadow_w += (uint32_t)__popcll(bsh);
            if (sh.on) {
                const uint32_t k = base + off + __builtin_amdgcn_mbcnt_hi((uint32_t)(bsh >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bsh, 0u));
                float4 *rec = a.shd_out + (size_t)k * 4u;
                const float4 q0 = {sh.so.x, sh.so.y, sh.so.z, sh.tmax},
                             q1 = {sh.sdir.x, sh.sdir.y, sh.sdir.z, __uint_as_float(survive ? out_slot : (WF_DEAD | home))},
                             q2 = {sh.A.x, sh.A.y, sh.A.z, 0.0f}, q3 = {sh.B.x, sh.B.y, sh.B.z, 0.0f};
                rec[0] = q0;
                rec[1] = q1;
                rec[2] = q2;
                rec[3] = q3;
            }
        }
    }
    __syncthreads();
    if (tid == 0) {
        a.seg_out[r] = q_out;
        a.nsh_out[r] = q_shd;
        unsigned long long *row = a.stats + r;
        const size_t stride = a.stat_stride;
        row[0] += nh;
        row[stride] += q_shd;
        row[(2 + min(a.depth, (uint32_t)MAX_DEPTH_STATS - 1)) * stride] += cnt_in;
        row[(HIT_ROW0 + min(a.depth, (uint32_t)MAX_DEPTH_STATS - 1)) * stride] += nh;  // hits of this depth (byte model)
    }
    (void)n_shadow_w;
    (void)W;
}

// upload of caller rays for Integrator.sample(): o, d [3][n] SoA + tmax -> path-state records (tmax rides in the eta slot)
__global__ __launch_bounds__(256) void k_init_rays_wf(float4 *st, uint32_t *seg_cnt, uint32_t n_regions, uint32_t n, const float *o,
                                                      const float *d, const float *tmax) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_regions) seg_cnt[i] = n > i * WF_REGION ? min(n - i * WF_REGION, WF_REGION) : 0u;
    if (i >= n) return;
    float4 *s = st + (size_t)i * 4u;
    const float4 q0 = {o[i], o[n + i], o[2 * n + i], tmax[i]}, q1 = {d[i], d[n + i], d[2 * n + i], -1.0f},
                 q2 = {1.0f, 1.0f, 1.0f, __uint_as_float(i)}, q3 = {0.0f, 0.0f, 0.0f, 0.0f};
    s[0] = q0;
    s[1] = q1;
    s[2] = q2;
    s[3] = q3;
}

// kernels_wavefront.h -- radiance mode on BVH scenes (gfx950): intersection and shading as separate streams.
//
// The fused bounce kernel (k_bounce<.., BVH>) walked the tree until the LAST lane of a wave had finished, shaded in whatever
// lanes had hit something and walked the tree again for their shadow rays, with the registers of all three parts live at
// once (96 - 128 VGPRs: 4 waves per SIMD beside the LDS image).  Here a bounce is two launches:
//   k_trace  a STREAM of ray queries against the LDS-resident BVH4: the shadow rays the previous bounce emitted (any hit) and
//            the continuation rays of the live paths (closest hit).  A lane whose ray has finished takes the next ray of the
//            workgroup's queue when enough lanes of its wave are idle (the other seven waves of the SIMD cover the load) and
//            joins the two-phase walk of the others.  It reads 32 bytes per ray and writes 4: the index of the primitive that
//            was hit, or the visibility; nothing else is live, so two 1024-thread workgroups with their own image fit a CU (8
//            waves per SIMD).
//   k_shade  every wave on its own, no barrier after the set-up: it reads the hit indices of its 64-slot chunks, ends the
//            paths whose ray left the scene, keeps the slots of the others on a list of its own in LDS and shades 64 of them
//            at a time with every lane busy: (t, u, v) of the hit from the primitive's record (the test k_trace made, repeated),
//            pending shadow contribution of the previous bounce, emission + MIS, emitter sample -> shadow record, BSDF sample,
//            Russian roulette; survivors and shadow records are packed to the front of the region (ballot + one LDS atomic
//            per wave).
// The contribution of a shadow ray is added where the fused kernel added it -- L = fma(A, B, L) before the next bounce touches
// L -- so the film does not change by a bit: same arithmetic per path, same RNG keys, same order of the radiance sums.
//
// Layouts: float4 PLANES -- component q of record i at base[q * cap + i] -- so that the streaming accesses (a wave writes the
// survivors it packed, reads the rays of consecutive queue entries, ends the missed paths of a chunk) are 1 KiB contiguous per
// wave-instruction; only the state of the paths that hit something is gathered:
//   path state, 96 B  q0 = (o, eta)  q1 = (d, prev_pdf)  q2 = (throughput, -)  q3 = (L, home)
//                     q4 = (A, visibility of the path's shadow ray: written 0 by k_shade, set by k_trace)  q5 = (B, -)
//   hit               hit_id[i] = primitive index | 0xffffffff: none (4 B)
//   shadow ray, 64 B  q0 = (origin, tmax)  q1 = (direction, dest)  and, for paths that ended at the bounce that emitted it,
//                     q2 = (A, visibility)  q3 = (B, home); dest = state slot of the survivor | WF_DEAD | own record index.
//                     Rays of survivors fill a region's records from the front, rays of ended paths from the back.
#pragma once
#include "kernels_radiance.h"

#define WF_REGION 4096u      // slots per region (compaction domain of k_shade)
#define WF_KMAX 21u          // regions one k_trace workgroup walks at most (three queue segments each)
#ifndef WF_REFILL_MIN
#define WF_REFILL_MIN 32u    // idle lanes that make a wave fetch new rays (ring 1024^2 x 256: 16 / 32 -> 48.9 / 48.1 ms)
#endif
#ifndef WF_WALK_MIN
#define WF_WALK_MIN 16u      // the node walk of a turn stops when fewer lanes are still in it (while new rays can be had)
#endif
#ifndef WF_TRACE_WAVES_PER_EU
#define WF_TRACE_WAVES_PER_EU 8
#endif
#ifndef WF_SHADE_THREADS
#define WF_SHADE_THREADS 256u
#endif
#ifndef WF_SHADE_CHUNKS
#define WF_SHADE_CHUNKS 2      // 64-slot chunks of hit indices (+ 48 B of state each at bounces >= 1) a wave of k_shade requests per step: twice the
#endif                         // bytes in flight where the kernel had fewest (round 4, ring 1024^2 x 256: 1 / 2 / 3 -> 49.0 / 46.9 / 47.8 ms)
#ifndef WF_SHADE_WAVES_PER_EU
#define WF_SHADE_WAVES_PER_EU 4   // 128 VGPRs, nothing spilled (round 4, ring 1024^2 x 256: 5 / 4 / 3 waves -> 49.3 / 47.1 / 47.4 ms; at 5 the
#endif                            // 12 spilled registers of k_shade<false> were scratch traffic of a kernel that waits on HBM)
// a wave whose walkers fall below WF_WALK_MIN stops walking to fetch rays: its idle lanes must then reach WF_REFILL_MIN (64 -
// walkers, when no lane holds a leaf), or it would neither walk nor fetch (the turn guard caught exactly that with 48 / 32)
static_assert(WF_REFILL_MIN + WF_WALK_MIN <= 65u, "k_trace: a wave below WF_WALK_MIN walkers must be able to refill");
#define WF_DEAD 0x40000000u   // shadow ray of a path that has ended: dest = WF_DEAD | index of the ray's own record
#define WF_SHADOW 0x80000000u
#define WF_STATE_Q 6u         // float4s per path-state record

struct WfArgs {
    DevScene sc;
    pbrt_camera cam;
    float4 *st_in, *st_out;      // [cap][6] path state
    uint32_t *hit_id;            // [cap] primitive index | 0xffffffff
    float4 *shd_in, *shd_out;    // [cap][4] shadow rays emitted by the previous / this bounce
    float *Lhome;                // [cap] float4 records (r, g, b, 0) indexed by home
    const uint32_t *seg_in, *nsh_in;   // per region: live paths; shadow rays of the previous bounce (survivors' | ended paths' << 16)
    uint32_t *seg_out, *nsh_out;       // (nsh_in: nullptr at depth 0)
    unsigned long long *stats;   // per-region rows as in RadArgs
    uint32_t stat_stride, cap, n_paths, n_regions;  // n_regions: regions of THIS launch, the first one is region0
    uint32_t region0;
    uint32_t depth, max_depth, rr_depth, seed;
    uint32_t key_mode, rx0, ry0, rw, npix_r, s_first, film_w, film_h, tile_rows;
    FastDiv div_npix, div_rw;
    uint32_t index_offset, sample_index;
    uint32_t lds_bytes;          // ACCEL_K_BVH_LDS: bytes of the staged image
    uint32_t stk_rows, stk_shift;  // traversal stacks behind the image: rows, log2(threads of the workgroup)
    uint32_t vis_q;              // plane of the path state whose .w takes a shadow ray's visibility (radiance: 4, ultrasound: 3)
    uint32_t *guard;             // WF_GUARD_WORDS words of the context: [0] waves that ran into the turn guard, [1..30] state of one of them,
                                 // [WF_GUARD_REHIT] hits whose (t, u, v) the shading kernels could not reproduce
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
// WF_GUARD_TURNS turns leaves AS A WHOLE, records its state in the context's guard words (WfArgs::guard) and the host reports
// PBRT_E_DEVICE instead of hanging the box.  `turns` counts trips of the main loop and of the node walk; both loops are
// wave-uniform, so it lives in a scalar register and every lane of the wave sees the trip.
#define WF_GUARD_TURNS (1u << 22)
#define WF_GUARD_REHIT 31u  // guard word: hits of k_trace that k_shade / k_us_shade could not reproduce (must stay 0)
#ifdef PBRT_WF_PROBE  // diagnostic builds: where do the lanes of k_trace go?  words 32.. of the guard buffer, printed by the host
#define WF_GUARD_WORDS 64u
#define WF_PROBE(i, v) probe[i] += (v)
#else
#define WF_GUARD_WORDS 32u
#define WF_PROBE(i, v)
#endif
#define WF_IDLE 0xfffffffeu     // cursor of a lane without a ray (BVH_SENT = 0xffffffff: its ray has finished)
#ifndef WF_WALK_UNROLL
#define WF_WALK_UNROLL 1     // node visits between two looks at the wave's walkers (the refill test and the guard)
#endif

template <int ACCEL>
struct WfTree {
    typedef TreeGlobal type;
};
template <>
struct WfTree<ACCEL_K_BVH_LDS> {
    typedef TreeLds type;
};

// ---- k_trace ---------------------------------------------------------------------------------------------------------------
// grid: G workgroups; workgroup w walks the regions w, w + G, w + 2 G, ... (at most WF_KMAX of them) as ONE queue: for each of
// its regions the shadow rays of the survivors, those of the ended paths, then the continuation rays.
// dynamic LDS: [image | stack rows].  CURVED = false: the scene holds triangles and parallelograms only.
template <bool FIRST, int ACCEL, bool CURVED>
__global__ __launch_bounds__(1024, WF_TRACE_WAVES_PER_EU) void k_trace(const WfArgs a) {
    static_assert(ACCEL == ACCEL_K_BVH_GLOBAL || ACCEL == ACCEL_K_BVH_LDS, "k_trace: BVH scenes");
    extern __shared__ __attribute__((aligned(16))) uint32_t dyn_lds[];
    __shared__ uint32_t q_in;
    __shared__ uint32_t cum[3 * WF_KMAX + 1];   // queue index at which a segment starts
    __shared__ uint32_t seg0[3 * WF_KMAX];      // record index of a segment's first ray | WF_SHADOW | WF_DEAD
    const uint32_t tid = threadIdx.x, lane = tid & 63u, G = gridDim.x;
    const uint32_t K = (a.n_regions - blockIdx.x + G - 1u) / G;  // regions of this workgroup (host: <= WF_KMAX)
    if (tid == 0) {
        uint32_t run = 0;
        for (uint32_t j = 0; j < K; ++j) {
            const uint32_t r = a.region0 + blockIdx.x + j * G, b = r * WF_REGION;
            const uint32_t ns = (!FIRST && a.nsh_in) ? a.nsh_in[r] : 0u;
            cum[3 * j] = run;
            seg0[3 * j] = b | WF_SHADOW;
            run += ns & 0xffffu;
            cum[3 * j + 1] = run;
            seg0[3 * j + 1] = (b + WF_REGION - (ns >> 16)) | WF_SHADOW | WF_DEAD;
            run += ns >> 16;
            cum[3 * j + 2] = run;
            seg0[3 * j + 2] = b;
            run += FIRST ? (a.n_paths > b ? min(a.n_paths - b, WF_REGION) : 0u) : a.seg_in[r];
        }
        cum[3 * K] = run;
        q_in = 0;
    }
    __syncthreads();
    const uint32_t total = cum[3 * K];
    if (total == 0) return;  // uniform: nothing to trace (late bounces of an ultrasound pass), before the image is staged
    typename WfTree<ACCEL>::type tr;
    LDS_AS uint32_t *stk_lds = (LDS_AS uint32_t *)dyn_lds;
    if constexpr (ACCEL == ACCEL_K_BVH_LDS) {
        tr = stage_tree_lds(a.sc, dyn_lds);  // ends with a barrier
        stk_lds += a.lds_bytes >> 2;
    } else {
        tr = TreeGlobal{a.sc.nodes, a.sc.lprims};
    }
    const BvhStack st = {stk_lds + tid, a.stk_shift, a.stk_rows};

    // A lane's state is its cursor: an inner node (>= 0) or a held leaf (bit 31 set) while its ray is being traced, BVH_SENT when
    // the ray has finished and waits to be retired, WF_IDLE when the lane has no ray.  The closest hit so far is (best, hu, hv,
    // hid); hid == 0xffffffff: none yet.
    uint32_t rslot = 0;  // closest hit: slot of the path | shadow ray: WF_SHADOW | dest (state slot, or WF_DEAD | record)
    V3 o = {0, 0, 0}, d = {0, 0, 1};
    BoxRay br = make_box_ray(o, d);
    float best = 0.0f, hu = 0.0f, hv = 0.0f;
    uint32_t hid = 0xffffffffu;
    BvhCursor c;
    BvhOvf ovf;
    c.cur = WF_IDLE;
    c.tos = BVH_SENT;
    c.sp = 0;
    bool q_empty = false;   // wave-uniform
    uint32_t s_hint = 0;    // wave-uniform: segment of the wave's last fetch (queue indices only grow)
    uint32_t turns = 0;     // wave-uniform (kept in a scalar register by the readfirstlane at its updates)
#ifdef PBRT_WF_PROBE
    unsigned long long probe[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // [0] walk trips [1] lanes in them [2] leaf trips [3] lanes [4] main trips [5] rays fetched [6] lanes with a ray per main trip [7] lanes holding a leaf, summed over the walk trips
#endif
    for (;;) {
        if (turns > WF_GUARD_TURNS) {  // (uniform: the whole wave reports and leaves)
            const unsigned long long bb = __ballot(c.cur != WF_IDLE), bwk = __ballot((int32_t)c.cur >= 0);
            const uint32_t rep = bwk ? (uint32_t)__builtin_ctzll(bwk) : (uint32_t)__builtin_ctzll(__ballot(true));
            if (lane == rep) {  // the first lane that is still walking: its cursor, stack and ray
                uint32_t *g = a.guard;
                atomicAdd(&g[0], 1u);
                g[1] = blockIdx.x;
                g[2] = tid >> 6;
                g[3] = (uint32_t)__popcll(bb);
                g[4] = (uint32_t)__popcll(bwk);
                g[5] = 0;
                g[6] = q_empty ? 1u : 0u;
                g[7] = total;
                g[8] = q_in;
                g[9] = 0;
                g[10] = a.depth | (FIRST ? 0x100u : 0u);
                g[11] = K;
                g[12] = c.cur;
                g[13] = c.sp;
                g[14] = c.tos;
                g[15] = rslot;
                g[16] = st.col[0];
                g[17] = st.col[1u << st.shift];
                g[18] = 0;
                g[19] = *ovf.at(0);
                g[20] = *ovf.at(1);
                g[21] = __float_as_uint(o.x);
                g[22] = __float_as_uint(o.y);
                g[23] = __float_as_uint(o.z);
                g[24] = __float_as_uint(d.x);
                g[25] = __float_as_uint(d.y);
                g[26] = __float_as_uint(d.z);
                g[27] = __float_as_uint(best);
                g[28] = st.n_rows;
                g[29] = st.shift;
                g[30] = a.sc.n_nodes;
            }
            break;
        }
        turns = (uint32_t)__builtin_amdgcn_readfirstlane((int)(turns + 1u));
        WF_PROBE(4, 1);
        WF_PROBE(6, __builtin_popcountll(__builtin_amdgcn_ballot_w64(c.cur != WF_IDLE)));
        // ---- retire
        if (c.cur == BVH_SENT) {
            const bool found = hid != 0xffffffffu;
            if (rslot & WF_SHADOW) {
                const float vis = found ? 0.0f : 1.0f;
                if (rslot & WF_DEAD)
                    reinterpret_cast<float *>(a.shd_in + 2u * (size_t)a.cap + (rslot & 0x3fffffffu))[3] = vis;  // q2.w of the record
                else
                    reinterpret_cast<float *>(a.st_in + a.vis_q * (size_t)a.cap + (rslot & 0x3fffffffu))[3] = vis;   // .w of the state's visibility plane
            } else {
                a.hit_id[rslot] = hid;

            }
            c.cur = WF_IDLE;
        }
        // ---- idle lanes take the next rays of the queue.  (No prefetch of the records: at 8 waves per SIMD the stream is bound
        // by VALU issue, the other waves cover the load, and the 9 registers of a record in flight would spill.)
        const unsigned long long bw = __builtin_amdgcn_ballot_w64(c.cur == WF_IDLE);
        const uint32_t nw = (uint32_t)__builtin_popcountll(bw);
        if (!q_empty && (nw >= WF_REFILL_MIN || nw == 64u)) {
            uint32_t got = 0;
            if (lane == 0) got = atomicAdd(&q_in, nw);
            const uint32_t i0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)got);
            if (i0 + nw >= total) q_empty = true;
            const uint32_t i = i0 + __builtin_amdgcn_mbcnt_hi((uint32_t)(bw >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bw, 0u));
            if (i0 < total) {
                while (s_hint + 1u < 3u * K && i0 >= cum[s_hint + 1u]) ++s_hint;  // uniform
            }
            WF_PROBE(5, __builtin_popcountll(__builtin_amdgcn_ballot_w64(c.cur == WF_IDLE && i < total)));
            if (c.cur == WF_IDLE && i < total) {
                uint32_t s = s_hint;
                while (i >= cum[s + 1u]) ++s;
                const uint32_t first = seg0[s];
                const uint32_t slot = (first & 0x3fffffffu) + (i - cum[s]);
                if (FIRST) {
                    uint32_t ka, kb;
                    wf_camera_ray(a, slot, &o, &d, &best, &ka, &kb);
                    rslot = slot;
                } else {
                    const float4 *rec = ((first & WF_SHADOW) ? a.shd_in : a.st_in) + slot;
                    const float4 q0 = rec[0], q1 = rec[a.cap];
                    o = {q0.x, q0.y, q0.z};
                    d = {q1.x, q1.y, q1.z};
                    if (first & WF_SHADOW) {
                        best = q0.w;
                        rslot = WF_SHADOW | __float_as_uint(q1.w);  // where the visibility goes
                    } else {
                        best = (a.key_mode == 1 && a.depth == 0) ? q0.w : K_INF;  // caller rays carry tmax in the eta slot
                        rslot = slot;
                    }
                }
                hid = 0xffffffffu;
                br = make_box_ray(o, d);
                c.cur = 0;
                c.tos = BVH_SENT;
                c.sp = 0;
            }
        }
        if (__builtin_amdgcn_ballot_w64(c.cur != WF_IDLE) == 0ull) {
            if (q_empty) break;
            continue;
        }
        // ---- one turn of the walk: inner nodes until every lane with a ray holds a leaf or has finished (or, while new rays can
        // be had, until fewer than WF_WALK_MIN lanes are still walking), then the held leaves
        const uint32_t walk_min = q_empty ? 1u : WF_WALK_MIN;  // wave-uniform
        for (;;) {
            const bool walking = (int32_t)c.cur >= 0;
            const uint32_t n_walk = (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(walking));
            if (n_walk < walk_min) break;
#ifdef WF_WALK_ADAPT  // A/B (round 5): also stop walking once the lanes that wait with a leaf outnumber the walkers WF_WALK_ADAPT : 1
            if (n_walk * WF_WALK_ADAPT < (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(c.cur - 0x80000000u < 0x7ffffffeu))) break;
#endif
            WF_PROBE(0, 1);
            WF_PROBE(1, __builtin_popcountll(__builtin_amdgcn_ballot_w64(walking)));
            WF_PROBE(7, __builtin_popcountll(__builtin_amdgcn_ballot_w64(c.cur - 0x80000000u < 0x7ffffffeu)));  // lanes that hold a leaf through this visit
            turns = (uint32_t)__builtin_amdgcn_readfirstlane((int)(turns + WF_WALK_UNROLL));
            if (turns > WF_GUARD_TURNS) break;  // (the main loop's guard reports)
            if (walking) bvh_visit(tr, st, c, ovf, br, best);
#pragma unroll
            for (int j = 1; j < WF_WALK_UNROLL; ++j)
                if ((int32_t)c.cur >= 0) bvh_visit(tr, st, c, ovf, br, best);
        }
        // the held leaves: a wave-uniform loop over the primitive index, every lane with a leaf of more than k primitives takes
        // part; the closest-hit update is a select (a hit at the distance of the best one so far wins with the lower primitive
        // index; hid = 0xffffffff loses against any).  An any-hit ray (shadow ray) ends with its first hit.
        const bool leaf = c.cur - 0x80000000u < 0x7ffffffeu;  // bit 31 set, neither BVH_SENT nor WF_IDLE
        const bool any = (rslot & WF_SHADOW) != 0u;
        const uint32_t first = c.cur & 0x07ffffffu;
        uint32_t count = leaf ? (c.cur >> 27) & 15u : 0u;
        for (uint32_t k = 0;; ++k) {
            const bool on = k < count;
            if (__builtin_amdgcn_ballot_w64(on) == 0ull) break;
            WF_PROBE(2, 1);
            WF_PROBE(3, __builtin_popcountll(__builtin_amdgcn_ballot_w64(on)));
            if (on) {
                float t, u, v;
                uint32_t id;
                const bool hit = lprim_hit<CURVED>(tr.leaf(first + k), a.sc.prims, o, d, best, &t, &u, &v, &id);
                const bool better = hit && (t < best || (t == best && id < hid) || any);
                best = better ? t : best;
                hu = better ? u : hu;
                hv = better ? v : hv;
                hid = better ? id : hid;
                count = (hit && any) ? 0u : count;
            }
        }
        if (leaf) c.cur = (any && hid != 0xffffffffu) ? BVH_SENT : bvh_pop(tr, st, c, ovf);
    }
#ifdef PBRT_WF_PROBE
    if (lane == 0)
        for (int i = 0; i < 8; ++i) atomicAdd(reinterpret_cast<unsigned long long *>(a.guard + 32) + i, probe[i]);
#endif
}

// ---- k_trace_primary ---------------------------------------------------------------------------------------------------------
// The closest hits of the CAMERA rays (bounce 0 of a render): 64 consecutive paths are the 8 x 8 pixel tile of one sample
// (path_key), so a wave walks the tree once for all of them (device_scene.h bvh_packet_closest) instead of 64 times with 64
// stacks.  Same grid and region walk as k_trace (workgroup w: regions w, w + G, ...); the waves of a workgroup take the 64-path
// tiles of those regions from a counter in LDS.  Writes hit_id like k_trace; same hits, bit for bit.
// dynamic LDS: the image (ACCEL_K_BVH_LDS).
template <int ACCEL, bool CURVED>
__global__ __launch_bounds__(1024, WF_TRACE_WAVES_PER_EU) void k_trace_primary(const WfArgs a) {
    static_assert(ACCEL == ACCEL_K_BVH_GLOBAL || ACCEL == ACCEL_K_BVH_LDS, "k_trace_primary: BVH scenes");
    extern __shared__ __attribute__((aligned(16))) uint32_t dyn_lds[];
    __shared__ uint32_t q_in;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, G = gridDim.x;
    const uint32_t K = (a.n_regions - blockIdx.x + G - 1u) / G;  // regions of this workgroup
    if (tid == 0) q_in = 0;
    typename WfTree<ACCEL>::type tr;
    if constexpr (ACCEL == ACCEL_K_BVH_LDS) {
        tr = stage_tree_lds(a.sc, dyn_lds);  // ends with a barrier
    } else {
        tr = TreeGlobal{a.sc.nodes, a.sc.lprims};
        __syncthreads();
    }
    constexpr uint32_t TILES = WF_REGION / 64u;
    const uint32_t n_tiles = K * TILES;
    for (;;) {
        uint32_t got = 0;
        if (lane == 0) got = atomicAdd(&q_in, 1u);
        const uint32_t t = (uint32_t)__builtin_amdgcn_readfirstlane((int)got);
        if (t >= n_tiles) break;
        const uint32_t r = a.region0 + blockIdx.x + (t / TILES) * G;
        const uint32_t slot0 = r * WF_REGION + (t % TILES) * 64u;
        if (slot0 >= a.n_paths) continue;  // uniform: beyond the paths of this pass
        const uint32_t slot = slot0 + lane;
        const bool alive = slot < a.n_paths;
        V3 o = {0, 0, 0}, d = {0, 0, 1};
        float best = -1.0f, hu = 0.0f, hv = 0.0f;
        uint32_t hid = 0xffffffffu;
        if (alive) {
            uint32_t ka, kb;
            wf_camera_ray(a, slot, &o, &d, &best, &ka, &kb);
        }
        // the representative ray: the middle of the tile (lane = 8 x + y), or the first lane that has a ray
        const unsigned long long ba = __ballot(alive);
        const uint32_t rep = ((ba >> 27) & 1ull) ? 27u : (uint32_t)__builtin_ctzll(ba);
        const bool found = bvh_packet_closest<CURVED>(tr, a.sc.prims, o, d, rep, best, hu, hv, hid);
        if (alive) {
            a.hit_id[slot] = found ? hid : 0xffffffffu;

        }
    }
}

// ---- k_shade ---------------------------------------------------------------------------------------------------------------
struct WfShadow {
    V3 so, sdir, A, B;
    float tmax;
    bool on;
};
// the 64-byte record of the primitive that was hit, in one batch of loads (so that the compiler cannot split it into
// dependent pieces: type first, then the geometry of that type, then material / emitter)
DEV pbrt_prim wf_load_prim(const pbrt_prim *p) {
    typedef uint32_t __attribute__((ext_vector_type(4))) u32x4;
    struct Raw {
        u32x4 q[4];
    };
    const u32x4 *q = reinterpret_cast<const u32x4 *>(p);
    Raw r = {{q[0], q[1], q[2], q[3]}};
    return __builtin_bit_cast(pbrt_prim, r);
}
struct WfVn {
    float n[9];
};
DEV WfVn wf_load_vn(const float *vn, uint32_t slot) {  // vn != nullptr
    typedef float __attribute__((ext_vector_type(3))) f3;
    struct Raw {
        f3 a, b, c;
    };
    const float *r = vn + 9u * slot;
    WfVn o;
#pragma unroll
    for (int k = 0; k < 9; ++k) o.n[k] = r[k];
    return o;
}
// make_si on the preloaded record (same statements as device_scene.h make_si / shading_normal)
DEV SI wf_make_si(const pbrt_prim &P, V3 o, V3 d, float t, float u, float v, bool has_vn, const WfVn &vn) {
    SI si;
    if (P.type == PBRT_PRIM_SPHERE) {
        V3 c = g3(P, 0);
        V3 p = madd(d, t, o);
        si.n = normalize(p - c);
        si.p = madd(si.n, P.g[3], c);
    } else if (P.type == PBRT_PRIM_CONE) {
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
    si.ns = si.n;
    if (has_vn && (P.type == PBRT_PRIM_TRIANGLE || P.type == PBRT_PRIM_PARALLELOGRAM)) {
        const V3 n0 = {vn.n[0], vn.n[1], vn.n[2]}, n1 = {vn.n[3], vn.n[4], vn.n[5]}, n2 = {vn.n[6], vn.n[7], vn.n[8]};
        if (dot(n0, n0) + dot(n1, n1) + dot(n2, n2) > 0.0f) {
            if (P.type == PBRT_PRIM_PARALLELOGRAM) {
                si.ns = normalize(n0);
            } else {
                const float b0 = 1.0f - u - v;
                si.ns = normalize(madd(n0, b0, madd(n1, u, n2 * v)));
            }
        }
    }
    return si;
}

// bounce_step from the shading on, with the shadow ray handed out instead of traced (same statements, same order)
template <int ACCEL>
DEV bool wf_shade_step(const WfArgs &a, const Tables &tb, uint32_t depth, uint32_t ka, uint32_t kb, const Hit &h, const pbrt_prim &P,
                       bool has_vn, const WfVn &vn, V3 &o, V3 &d, V3 &thr, V3 &L, float &eta, float &prev_pdf, WfShadow &sh) {
    bool survive = false;
    sh.on = false;
    const uint32_t nE = a.sc.n_emitters;
    SI si = wf_make_si(P, o, d, h.t, h.u, h.v, has_vn, vn);
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

// LDS copies of the small shading tables of a BVH scene (materials, emitters, and the records of the LIGHT primitives only:
// the table of all primitives stays in global memory and is read once per hit).  Layout [light prims | mats | emitters | cdf];
// light_prims becomes the identity, so sample_emitter runs unchanged.  Used when every table has <= TAB_MAX entries.
#define WF_TAB_DW (TAB_MAX * 16 + TAB_MAX * 8 + TAB_MAX * 12 + TAB_MAX + TAB_MAX)
DEV Tables wf_tables_lds(const DevScene &sc, uint32_t *lds, uint32_t n_threads) {
    uint32_t *p_prims = lds, *p_mats = lds + TAB_MAX * 16, *p_emit = p_mats + TAB_MAX * 8, *p_lp = p_emit + TAB_MAX * 12;
    float *p_lc = reinterpret_cast<float *>(p_lp + TAB_MAX);
    for (uint32_t t = threadIdx.x; t < TAB_MAX * 16; t += n_threads) {
        if (t < sc.n_light_prims * 16) p_prims[t] = reinterpret_cast<const uint32_t *>(sc.prims_by_id + sc.light_prims[t >> 4])[t & 15u];
        if (t < sc.n_mats * 8) p_mats[t] = reinterpret_cast<const uint32_t *>(sc.mats)[t];
        if (t < sc.n_emitters * 12) p_emit[t] = reinterpret_cast<const uint32_t *>(sc.emitters)[t];
        if (t < sc.n_light_prims) {
            p_lp[t] = t;
            p_lc[t] = sc.light_cdf[t];
        }
    }
    Tables tb;
    tb.prims_by_slot = sc.prims;  // the primitive that was hit: global memory
    tb.prims_by_id = reinterpret_cast<const pbrt_prim *>(p_prims);
    tb.mats = reinterpret_cast<const pbrt_material *>(p_mats);
    tb.emitters = reinterpret_cast<const pbrt_emitter *>(p_emit);
    tb.light_prims = p_lp;
    tb.light_cdf = p_lc;
    tb.n_emitters = sc.n_emitters;
    return tb;
}

// TABS: the small tables fit LDS (wf_tables_lds), else everything is read from global memory
template <bool FIRST, bool TABS>
__global__ __launch_bounds__(WF_SHADE_THREADS, WF_SHADE_WAVES_PER_EU) void k_shade(const WfArgs a) {
    constexpr uint32_t T = WF_SHADE_THREADS, W = T / 64;
    constexpr int NCH = WF_SHADE_CHUNKS;  // chunks of 64 hit indices a wave reads per step
    // per wave: the paths that hit something and wait for a full wave -- slot within the region, and the primitive that was hit
    __shared__ uint32_t wlist[W][64 * (NCH + 1)], wprim[W][64 * (NCH + 1)];
    // ... and (bounces >= 1) its radiance so far with the pending shadow contribution folded in, and its home: the chunk phase streams
    // the L / A / B planes of EVERY path of the chunk anyway (a path that missed needs them to end), so a path that hit keeps what
    // they amount to -- L = fma(A, B, L), home -- on the list instead of gathering the three planes again in the shading step:
    // 48 of the 96 gathered bytes per hit, at 128-byte lines for 16-byte records (round 4: traffic 1.25 x the model)
    __shared__ float4 wlh[FIRST ? 1 : W][FIRST ? 1 : 64 * (NCH + 1)];
    __shared__ uint32_t q_out, q_shd, q_dead, q_done;
    __shared__ uint32_t tab_lds[TABS ? WF_TAB_DW : 1];
    const uint32_t r = a.region0 + xcd_swizzle(blockIdx.x, gridDim.x), base = r * WF_REGION;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wid = tid >> 6;
    const uint32_t cnt_in = FIRST ? (a.n_paths > base ? min(a.n_paths - base, WF_REGION) : 0u) : a.seg_in[r];
    const uint32_t n_dead = (!FIRST && a.nsh_in) ? a.nsh_in[r] >> 16 : 0u;
    if (cnt_in == 0 && n_dead == 0) {  // uniform
        if (tid == 0) {
            a.seg_out[r] = 0;
            a.nsh_out[r] = 0;
        }
        return;
    }
    if (tid == 0) {
        q_out = 0;
        q_shd = 0;
        q_dead = 0;
        q_done = 0;
    }
    const Tables tb = TABS ? wf_tables_lds(a.sc, tab_lds, T) : global_tables(a.sc);
    __syncthreads();  // the only barrier
    float4 *Lh = reinterpret_cast<float4 *>(a.Lhome);
    // ---- shadow rays of paths that ended at the previous bounce: L = fma(A, B, L) on the radiance record
    for (uint32_t k = tid; k < n_dead; k += T) {
        const float4 *rec = a.shd_in + (base + WF_REGION - n_dead + k);
        const float4 A = rec[2u * (size_t)a.cap], B = rec[3u * (size_t)a.cap];
        if (A.w != 0.0f) {
            float4 *Lp = Lh + __float_as_uint(B.w);
            float4 Lv = *Lp;
            Lv.x = fma_(A.x, B.x, Lv.x);
            Lv.y = fma_(A.y, B.y, Lv.y);
            Lv.z = fma_(A.z, B.z, Lv.z);
            *Lp = Lv;
        }
    }
    uint32_t list_n = 0;             // wave-uniform: entries on this wave's list
    uint32_t n_seg_w = 0, n_shd_w = 0;
    uint32_t c0 = wid * 64u;         // the wave's next chunk of the region
    for (;;) {
        if (c0 < cnt_in) {
            // ---- WF_SHADE_CHUNKS chunks of hit records: paths whose ray left the scene end here, the others go on the list.  One
            // batch of loads: the hit indices and (bounces >= 1) the three state planes a path that ends needs.
            uint32_t hidv[NCH];
            float4 q3v[NCH], q4v[NCH], q5v[NCH];
#pragma unroll
            for (int j = 0; j < NCH; ++j) {
                const uint32_t s = c0 + (uint32_t)j * (W * 64u) + lane;
                hidv[j] = 0xffffffffu;
                q3v[j] = q4v[j] = q5v[j] = float4{0, 0, 0, 0};
                if (s < cnt_in) {
                    hidv[j] = a.hit_id[base + s];
                    if (!FIRST) {
                        const float4 *stp = a.st_in + (base + s);
                        const size_t cp = a.cap;
                        q3v[j] = stp[3u * cp];
                        q4v[j] = stp[4u * cp];
                        q5v[j] = stp[5u * cp];
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < NCH; ++j) {
                const uint32_t s = c0 + (uint32_t)j * (W * 64u) + lane;
                const bool valid = s < cnt_in;
                const uint32_t hid = hidv[j];
                const float4 q3 = q3v[j], q4 = q4v[j], q5 = q5v[j];
                const bool is_hit = hid != 0xffffffffu;
                float4 Lv = q3;
                if (!FIRST && q4.w != 0.0f) {  // its shadow ray of the previous bounce got through
                    Lv.x = fma_(q4.x, q5.x, Lv.x);
                    Lv.y = fma_(q4.y, q5.y, Lv.y);
                    Lv.z = fma_(q4.z, q5.z, Lv.z);
                }
                if (valid && !is_hit) {
                    float4 rec = Lv;
                    rec.w = 0.0f;
                    Lh[FIRST ? base + s : __float_as_uint(q3.w)] = rec;
                }
                const unsigned long long bh = __ballot(is_hit);
                if (is_hit) {
                    const uint32_t e = list_n + __builtin_amdgcn_mbcnt_hi((uint32_t)(bh >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bh, 0u));
                    wlist[wid][e] = s;
                    wprim[wid][e] = hid;
                    if (!FIRST) wlh[wid][e] = Lv;  // (L with the pending contribution, home in .w)
                }
                list_n += (uint32_t)__popcll(bh);
            }
            __builtin_amdgcn_wave_barrier();  // other lanes of the wave read these entries below (LDS operations of a wave stay in order)
            c0 += (uint32_t)NCH * W * 64u;
        } else if (list_n == 0) {
            break;
        }
        if (list_n < 64u && c0 < cnt_in) continue;
        // ---- shade 64 listed paths (or what is left at the end) with every lane busy; the list is emptied below 64 entries before
        // the wave reads its next chunks (the list holds 64 x (WF_SHADE_CHUNKS + 1))
        for (;;) {
        const uint32_t take = min(list_n, 64u);
        const bool act = lane < take;
        list_n -= take;
        bool survive = false;
        WfShadow sh;
        sh.on = false;
        V3 o = {0, 0, 0}, d = {0, 0, 1}, thr = {1, 1, 1}, L = {0, 0, 0};
        float eta = 1.0f, prev_pdf = -1.0f;
        uint32_t home = 0;
        if (act) {
            const uint32_t s = wlist[wid][list_n + lane];
            Hit h;
            h.prim = wprim[wid][list_n + lane];
            h.slot = h.prim;
            // one batch of loads: the primitive's record, its vertex normals, the path state.  (t, u, v) of the hit are not carried
            // through memory: k_trace hands over the primitive it found, and the test of THAT primitive against the ray is repeated
            // here -- the same arithmetic on the same operands (a leaf record is the first nine floats of this record), so the same
            // bits, for 60 VALU instructions of a kernel that waits on HBM instead of a 16-byte record written scattered (a 32-byte
            // sector each) and gathered back (a 128-byte line each at the later bounces).
            const pbrt_prim P = wf_load_prim(tb.prims_by_slot + h.slot);
            const bool has_vn = a.sc.vnormals != nullptr;  // uniform
            WfVn vn;
            if (has_vn) vn = wf_load_vn(a.sc.vnormals, h.slot);
            uint32_t ka, kb;
            if (FIRST) {
                float tm;
                home = base + s;
                wf_camera_ray(a, home, &o, &d, &tm, &ka, &kb);
            } else {
                const float4 *stp = a.st_in + (base + s);
                const size_t cp = a.cap;
                const float4 q0 = stp[0], q1 = stp[cp], q2 = stp[2u * cp];
                const float4 lh = wlh[FIRST ? 0 : wid][FIRST ? 0 : list_n + lane];  // L (pending shadow contribution included), home
                o = {q0.x, q0.y, q0.z};
                d = {q1.x, q1.y, q1.z};
                thr = {q2.x, q2.y, q2.z};
                L = {lh.x, lh.y, lh.z};
                eta = (a.key_mode == 1 && a.depth == 0) ? 1.0f : q0.w;  // caller rays carry tmax in the eta slot
                prev_pdf = q1.w;
                home = __float_as_uint(lh.w);
                uint32_t px, py;
                const RadArgs ra = wf_key_args(a);
                path_key<true>(ra, home, &ka, &kb, &px, &py);
            }
            // (t, u, v) of the hit k_trace found: the same test on the same operands (a leaf record is the first nine floats of P).
            // Should the repetition ever disagree (other build flags, another code path for the primitive) the call fails with
            // PBRT_E_DEVICE instead of shading with whatever the registers held: guard word WF_GUARD_REHIT counts the cases.
            h.t = K_INF;
            h.u = h.v = 0.0f;
            if (!prim_hit(P, o, d, K_INF, &h.t, &h.u, &h.v)) atomicAdd(a.guard + WF_GUARD_REHIT, 1u);
            survive = wf_shade_step<ACCEL_K_BVH_GLOBAL>(a, tb, a.depth, ka, kb, h, P, has_vn, vn, o, d, thr, L, eta, prev_pdf, sh);
        }
        n_seg_w += take;
        // survivors -> front of the region of the `out` state
        const unsigned long long bs = __ballot(survive);
        uint32_t out_slot = 0;
        if (bs) {
            uint32_t got = 0;
            if (lane == 0) got = atomicAdd(&q_out, (uint32_t)__popcll(bs));
            const uint32_t off = (uint32_t)__builtin_amdgcn_readfirstlane((int)got);
            out_slot = base + off + __builtin_amdgcn_mbcnt_hi((uint32_t)(bs >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bs, 0u));
        }
        const bool shd_live = sh.on && survive, shd_dead = sh.on && !survive;
        if (survive) {
            float4 *stp = a.st_out + out_slot;
            const size_t cp = a.cap;
            const float4 q0 = {o.x, o.y, o.z, eta}, q1 = {d.x, d.y, d.z, prev_pdf}, q2 = {thr.x, thr.y, thr.z, 0.0f},
                         q3 = {L.x, L.y, L.z, __uint_as_float(home)};
            stp[0] = q0;
            stp[cp] = q1;
            stp[2u * cp] = q2;
            stp[3u * cp] = q3;
            const float4 q4 = {shd_live ? sh.A.x : 0.0f, shd_live ? sh.A.y : 0.0f, shd_live ? sh.A.z : 0.0f, 0.0f},
                         q5 = {shd_live ? sh.B.x : 0.0f, shd_live ? sh.B.y : 0.0f, shd_live ? sh.B.z : 0.0f, 0.0f};
            stp[4u * cp] = q4;
            stp[5u * cp] = q5;
        } else if (act) {  // the path ends (a pending shadow ray is added to this record by the next k_shade)
            const float4 rec = {L.x, L.y, L.z, 0.0f};
            Lh[home] = rec;
        }
        const unsigned long long bl = __ballot(shd_live);
        if (bl) {
            uint32_t got = 0;
            if (lane == 0) got = atomicAdd(&q_shd, (uint32_t)__popcll(bl));
            const uint32_t off = (uint32_t)__builtin_amdgcn_readfirstlane((int)got);
            n_shd_w += (uint32_t)__popcll(bl);
            if (shd_live) {
                const uint32_t k = base + off + __builtin_amdgcn_mbcnt_hi((uint32_t)(bl >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bl, 0u));
                float4 *rec = a.shd_out + k;
                const float4 q0 = {sh.so.x, sh.so.y, sh.so.z, sh.tmax}, q1 = {sh.sdir.x, sh.sdir.y, sh.sdir.z, __uint_as_float(out_slot)};
                rec[0] = q0;
                rec[a.cap] = q1;
            }
        }
        const unsigned long long bd = __ballot(shd_dead);
        if (bd) {  // rare: Russian roulette ended a path that had just sent a shadow ray
            uint32_t got = 0;
            if (lane == 0) got = atomicAdd(&q_dead, (uint32_t)__popcll(bd));
            const uint32_t off = (uint32_t)__builtin_amdgcn_readfirstlane((int)got);
            n_shd_w += (uint32_t)__popcll(bd);
            if (shd_dead) {
                const uint32_t k =
                    base + WF_REGION - 1u - (off + __builtin_amdgcn_mbcnt_hi((uint32_t)(bd >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bd, 0u)));
                float4 *rec = a.shd_out + k;
                const size_t cp = a.cap;
                const float4 q0 = {sh.so.x, sh.so.y, sh.so.z, sh.tmax},
                             q1 = {sh.sdir.x, sh.sdir.y, sh.sdir.z, __uint_as_float(WF_DEAD | k)},
                             q2 = {sh.A.x, sh.A.y, sh.A.z, 0.0f}, q3 = {sh.B.x, sh.B.y, sh.B.z, __uint_as_float(home)};
                rec[0] = q0;
                rec[cp] = q1;
                rec[2u * cp] = q2;
                rec[3u * cp] = q3;
            }
        }
        if (list_n < 64u && c0 < cnt_in) break;  // room for the next chunks
        if (list_n == 0u) break;
        }  // shading steps
    }
    if (lane == 0) {
        unsigned long long *row = a.stats + (size_t)r * W + wid;  // per-wave statistics rows
        const size_t stride = a.stat_stride;
        row[0] += n_seg_w;
        row[stride] += n_shd_w;
        row[(HIT_ROW0 + min(a.depth, (uint32_t)MAX_DEPTH_STATS - 1)) * stride] += n_seg_w;  // hits of this depth (byte model)
        if (wid == 0) row[(2 + min(a.depth, (uint32_t)MAX_DEPTH_STATS - 1)) * stride] += cnt_in;
        // the last wave to finish publishes the region's counts (LDS atomics of one CU are ordered)
        if (atomicAdd(&q_done, 1u) == W - 1) {
            a.seg_out[r] = atomicAdd(&q_out, 0u);
            a.nsh_out[r] = atomicAdd(&q_shd, 0u) | (atomicAdd(&q_dead, 0u) << 16);
        }
    }
}

// upload of caller rays for Integrator.sample(): o, d [3][n] SoA + tmax -> path-state records (tmax rides in the eta slot)
__global__ __launch_bounds__(256) void k_init_rays_wf(float4 *st, uint32_t cap, uint32_t *seg_cnt, uint32_t n_regions, uint32_t n,
                                                      const float *o, const float *d, const float *tmax) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_regions) seg_cnt[i] = n > i * WF_REGION ? min(n - i * WF_REGION, WF_REGION) : 0u;
    if (i >= n) return;
    float4 *s = st + i;
    const size_t cp = cap;
    const float4 q0 = {o[i], o[n + i], o[2 * n + i], tmax[i]}, q1 = {d[i], d[n + i], d[2 * n + i], -1.0f},
                 q2 = {1.0f, 1.0f, 1.0f, 0.0f}, q3 = {0.0f, 0.0f, 0.0f, __uint_as_float(i)}, z = {0.0f, 0.0f, 0.0f, 0.0f};
    s[0] = q0;
    s[cp] = q1;
    s[2u * cp] = q2;
    s[3u * cp] = q3;
    s[4u * cp] = z;
    s[5u * cp] = z;
}

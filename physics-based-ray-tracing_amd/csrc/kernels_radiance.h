// kernels_radiance.h -- radiance-mode wavefront kernels (gfx950).
//
// Layout in HBM (DESIGN.md "Data layout"): path state is a tiled structure of arrays (state_voff below), 15 dwords
// per slot
//   [0..2] ray origin  [3..5] ray direction  [6..8] throughput rgb  [9..11] radiance rgb
//   [12] eta  [13] pdf of the previous BSDF sample (< 0: previous event was a delta lobe / camera)
//   [14] home slot (uint32): index of the path's radiance accumulator and of its pixel/sample
// Slots are grouped in segments of SEG = workgroup size.  One bounce = one launch: workgroup g reads
// the live prefix of region g of the `in` state, advances every path by one bounce (closest hit,
// emission + MIS, next-event estimation with its shadow ray, BSDF sample, Russian roulette) and
// writes the survivors, compacted to the front of region g of the `out` state: brute-force scenes with a wave
// ballot + mbcnt prefix and a scan across the workgroup's waves through LDS (region = one segment), BVH scenes with
// a chunk queue and a slot reservation in LDS (region = REGION_SEGS_BVH segments).  No global atomics are on
// the path-state data path; paths that end write their radiance once to Lhome[home].
#pragma once
#include "device_scene.h"

// slots per compaction segment == threads per workgroup (seg_threads(ACCEL) below).  Measured on MI355X, cbox
// 512^2 x 256 spp (current kernel): 128 / 256 / 512 / 1024 -> 10.5 / 9.0 / 8.6 / 9.9 ms; the LDS-staged BVH wants
// the largest workgroup (ring: 39 ms at 1024, 78 ms at 256 when first measured).
#ifndef SEG_BRUTE
#define SEG_BRUTE 512
#endif
#ifndef SEG_BVH
#define SEG_BVH 1024
#endif
#ifndef SEG_WAVES_PER_EU
#define SEG_WAVES_PER_EU 8  // __launch_bounds__ hint for the brute-force kernels: 64 VGPRs, 8 waves per SIMD
#endif
// A workgroup owns a REGION of region_segs * SEG slots and walks its live prefix in chunks.
// Measured on MI355X: walking several chunks amortises the LDS staging of BVH scenes
// (ring: 47 -> 39 ms; best at 4 segments with the chunk queue) and the block prologue of the ultrasound kernel
// (3.9 -> 3.1 ms), but any loop around the bounce body pushes the brute-force radiance kernel over its 64-VGPR
// budget (2 / 4 segments: 13.6 / 12.0 ms against 8.2), so that variant keeps one chunk per workgroup.
#ifndef REGION_SEGS_BVH
#define REGION_SEGS_BVH 4
#endif
#ifndef REGION_SEGS_BRUTE
#define REGION_SEGS_BRUTE 1
#endif
#ifndef REGION_SEGS_US
#define REGION_SEGS_US 8
#endif
__host__ __device__ constexpr uint32_t rad_region_segs(int accel);
// (emitter primary rays, D15: every first-bounce echo has its own arrival time, and the workgroup's echo table catches more of them
// the more paths of a ray it sees before it is flushed -- 8 / 16 / 32 segments: 24.5 - 25.8 / 21.0 - 21.5 / 20.9 - 21.5 ms; with the
// integrator's own rays, whose first bounce is table-driven, 16 segments cost 2 - 4 %: 10.8 - 10.9 -> 11.0 - 11.2 ms)
#ifndef REGION_SEGS_US_EMIT
#define REGION_SEGS_US_EMIT 16
#endif
__host__ __device__ constexpr uint32_t us_region_segs(int, bool emit = false) { return emit ? REGION_SEGS_US_EMIT : REGION_SEGS_US; }
#define N_STATE 15
#define MAX_DEPTH_STATS 62
#define MAX_CHAIN 6  // bounces one launch of the multi-bounce kernels walks at most (the per-bounce counts are packed 10 bits each)
#define HIT_ROW0 (2 + MAX_DEPTH_STATS)  // k_bounce_pool: statistics rows HIT_ROW0 + d = rays of depth d that hit something

// ACCEL_K_BRUTE      uniform primitive loop (scalar loads) + shading tables staged in LDS; every table <= 32 entries,
//                    no analytic cones (device_scene.h brute_intersect CONES)
// ACCEL_K_BVH_GLOBAL stackless BVH, nodes / primitives through the vector caches (scene larger than LDS)
// ACCEL_K_BVH_LDS    stackless BVH, nodes + primitives + ids staged in LDS per workgroup
// ACCEL_K_BRUTE_BIG  uniform primitive loop, tables in global memory (brute force forced on a large scene)
enum { ACCEL_K_BRUTE = 0, ACCEL_K_BVH_GLOBAL = 1, ACCEL_K_BVH_LDS = 2, ACCEL_K_BRUTE_BIG = 3 };
__host__ __device__ constexpr uint32_t rad_region_segs(int accel) {
    return (accel == ACCEL_K_BVH_GLOBAL || accel == ACCEL_K_BVH_LDS) ? REGION_SEGS_BVH : REGION_SEGS_BRUTE;
}
__host__ __device__ constexpr uint32_t seg_threads(int accel) {
    return (accel == ACCEL_K_BVH_GLOBAL || accel == ACCEL_K_BVH_LDS) ? SEG_BVH : SEG_BRUTE;
}
#ifndef BIG_WAVES_PER_EU
#define BIG_WAVES_PER_EU 8
#endif
__host__ __device__ constexpr uint32_t seg_waves_per_eu(int accel) {
    // _BIG (tables in global memory, cone code): round 1 gave these kernels the 128-register budget (4 spilled VGPRs at 64 then).
    // Built without SLP vectorisation the one-bounce variants fit 58 - 64 VGPRs and the two-bounce ones spill 4 - 8 at 64;
    // four 512-thread workgroups per CU instead of three: cone_room 512^2 x 256 10.21 -> 9.82 ms
    return (accel == ACCEL_K_BVH_GLOBAL || accel == ACCEL_K_BVH_LDS) ? SEG_BVH / 256 : (accel == ACCEL_K_BRUTE_BIG ? BIG_WAVES_PER_EU : SEG_WAVES_PER_EU);
}
// BVH kernels compact per WAVE: every wave owns REGION / (SEG / 64) slots of its workgroup's region, walks its own
// live prefix 64 paths at a time and packs its survivors with ballot + mbcnt alone -- no barrier after the scene is
// staged.  Traversal time varies a lot from wave to wave (measured: 2.7 of 4 possible waves per SIMD active with the
// per-chunk workgroup barrier), so the 16 waves of a workgroup must not wait for each other.  (The brute-force
// kernels keep the workgroup scan: there the barrier is cheap and private regions cost registers, DESIGN.md.)
__host__ __device__ constexpr bool rad_wave_private(int accel) {
#ifdef PBRT_BVH_WG_COMPACT  // diagnostic builds only (A/B against the workgroup-level scan)
    return false && accel;
#elif defined(PBRT_BRUTE_DYN)  // diagnostic builds only: chunk queue for the brute-force kernels as well
    return accel >= 0;
#else
    return accel == ACCEL_K_BVH_GLOBAL || accel == ACCEL_K_BVH_LDS;
#endif
}
// Radiance BVH kernels go one step further: the waves of a workgroup take their 64-path chunks from a queue in LDS
// (one returning ds_add per chunk) and reserve the slots of their survivors with another, so the region is one
// compaction domain again and the waves of a workgroup finish together -- with fixed 512-slot shares a wave with
// expensive rays kept its workgroup (and the 114 KB LDS image) alive while the other 15 had long left
// (measured: 2.3 of 4 possible waves per SIMD).
__host__ __device__ constexpr bool rad_dynamic(int accel) {
#ifdef PBRT_BVH_STATIC_WAVES  // diagnostic builds only (A/B against fixed per-wave shares)
    return false && accel;
#else
    return rad_wave_private(accel);
#endif
}
// live-path counters per region of the radiance kernels (one per workgroup, or one per wave with fixed shares)
__host__ __device__ constexpr uint32_t rad_owners_per_region(int accel) {
    return (rad_wave_private(accel) && !rad_dynamic(accel)) ? seg_threads(accel) / 64 : 1;
}
// statistics rows per region (per workgroup, or per wave)
__host__ __device__ constexpr uint32_t rad_rows_per_region(int accel) {
    return rad_wave_private(accel) ? seg_threads(accel) / 64 : 1;
}
#define TAB_MAX 32
#define TAB_DW (TAB_MAX * 16 + TAB_MAX * 8 + TAB_MAX * 12 + TAB_MAX + TAB_MAX)

struct RadArgs {
    DevScene sc;
    pbrt_camera cam;
    const float *in;   // [cap / 64][N_STATE][64] tiled SoA (state_voff)
    float *out;        // same layout
    float *Lhome;      // [cap] float4 records (r, g, b, 0) indexed by home
    const uint32_t *seg_in;
    uint32_t *seg_out;
    unsigned long long *stats;  // [k][stat_stride] per-segment rows: k = 0 segments, 1 shadow rays, 2 + d live paths entering depth d
    uint32_t stat_stride;
    uint32_t cap;        // slots of Lhome
    uint32_t state_cap;  // slots of the `in` / `out` state (<= cap: a pass whose bounces all run in one launch needs none)
    uint32_t n_paths;  // paths generated by the first bounce of this pass
    uint32_t depth, max_depth, rr_depth, seed;
    uint32_t nb;  // bounces this launch walks (the multi-bounce variants k_bounce<.., 2>; 2 .. MAX_CHAIN)
    uint32_t repack_mask;  // chain launches: bit b = after the launch's b-th bounce the workgroup packs its live paths to its first lanes (k_bounce)
    uint32_t merge_at;  // k_chain_pair: the bounce from which a wave walks the survivors of its two tiles together (1 .. max_depth - 1)
    // key mode 0 (render): home -> (region pixel, local sample)
    uint32_t key_mode;
    uint32_t rx0, ry0, rw, npix_r, s_first, film_w, film_h;
    uint32_t tile_rows;  // rows of the region in 8-row bands (region_index); 0 for the brute-force kernels
    FastDiv div_npix, div_rw;  // exact home / npix_r and pr / rw without the 20-instruction variable udiv
    // key mode 1 (Integrator.sample on caller rays): key = (index_offset + home, sample_index)
    uint32_t index_offset, sample_index;
    uint32_t lds_bytes;  // ACCEL_K_BVH_LDS: bytes of nodes + prims + ids staged per workgroup
};

// Path state is tiled SoA: 64 consecutive slots (one wave) form a tile of N_STATE rows x 64 lanes = 3840 contiguous
// bytes, row k of a slot at byte  tile(slot) + k * 256 + lane * 4.  Accessed through buffer descriptors:
// `buffer_load_dword v, v_off, s[rsrc], 0 offen offset:k*256` folds the row into the 12-bit immediate, so the
// 15 + 15 state accesses of a bounce need no 64-bit VALU address arithmetic (the kernel is VALU-bound) and no
// SGPR per row (the kernel sits at the 80-SGPR occupancy limit).  A wave reads one contiguous 3840-byte tile.
// Requires N_STATE * cap * 4 < 4 GiB (checked by the host) and cap % 64 == 0.
#define STATE_TILE_BYTES (64u * N_STATE * 4u)
#define STATE_ROW_BYTES 256u
DEV uint32_t state_voff(uint32_t slot) { return (slot >> 6) * STATE_TILE_BYTES + (slot & 63u) * 4u; }
typedef __amdgpu_buffer_rsrc_t Rsrc;
DEV Rsrc make_rsrc(const void *p, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, bytes, 0x00020000);
}
DEV float bld(Rsrc r, uint32_t voff, uint32_t soff) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}
DEV void bst(Rsrc r, uint32_t voff, uint32_t soff, float v) {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, v), r, voff, soff, 0);
}

// Pixel order inside the rendered region (index k of a pixel = its slot within one sample's block of `Lhome` and
// of the first-bounce state).  BVH kernels (TILED): the region is cut into bands of 8 rows and a band is walked
// column by column, so the 64 consecutive paths of a wave are an 8 x 8 pixel tile, not a 64 x 1 strip -- their
// primary rays visit fewer distinct nodes (every lane of a wave pays for the union).  The rh % 8 rows below the last
// full band keep the row-major order; tile_rows = rh & ~7 (0: plain row-major everywhere).
DEV uint32_t region_index(uint32_t xx, uint32_t yy, uint32_t rw, uint32_t tile_rows) {
    return yy < tile_rows ? (yy >> 3) * (8u * rw) + (xx << 3) + (yy & 7u) : yy * rw + xx;
}

// Workgroup b runs on XCD b % 8 (round-robin dispatch), and consecutive regions are consecutive pixels of a row: on a film
// that is 8 segments wide (4096 px) XCD k would render column stripe k of EVERY row -- the XCDs with the spheres in their stripe
// finish last (cbox 4096^2: the later bounces 18-40 % slower than on the 512^2 film with the same paths).  Rotating the regions
// inside every aligned group of 8 by the sum of the base-8 digits of the group index deals every stripe to every XCD in turn.
// A permutation of [0, n): the last, partial group is left alone.
DEV uint32_t xcd_swizzle(uint32_t b, uint32_t n) {
#ifdef PBRT_NO_XCD_SWIZZLE
    return b;
#else
    if ((b | 7u) >= n) return b;
    uint32_t g = b >> 3, rot = 0;
    while (g) {
        rot += g;
        g >>= 3;
    }
    return (b & ~7u) | ((b + rot) & 7u);
#endif
}

template <bool TILED>
DEV void path_key(const RadArgs &a, uint32_t home, uint32_t *ka, uint32_t *kb, uint32_t *px, uint32_t *py) {
    if (a.key_mode == 0) {
        uint32_t sl = udiv_fast(home, a.div_npix), pr = home - sl * a.npix_r;
        uint32_t rx, ry;
        if (TILED) {  // inverse of region_index: one division either way
            const bool tiled = pr < a.tile_rows * a.rw;
            const uint32_t num = tiled ? pr >> 3 : pr;
            const uint32_t q = udiv_fast(num, a.div_rw);
            rx = num - q * a.rw;
            ry = tiled ? (q << 3) + (pr & 7u) : q;
        } else {
            ry = udiv_fast(pr, a.div_rw);
            rx = pr - ry * a.rw;
        }
        *px = a.rx0 + rx;
        *py = a.ry0 + ry;
        *ka = *py * a.film_w + *px;
        *kb = a.s_first + sl;
    } else {
        *ka = a.index_offset + home;
        *kb = a.sample_index;
        *px = *py = 0;
    }
}

// LDS image of the BVH scene (device_scene.h TreeLds: node planes | leaf records), and the workgroup's traversal stacks (BvhStack)
struct LdsScene {
    TreeLds tree;
    BvhStack stk;
};
#define NO_TREE_LDS TreeLds{nullptr, 0u, 0u}
#ifndef BVH_STK_ROWS
#define BVH_STK_ROWS 4  // LDS rows of the traversal stacks of the fused kernels; deeper entries go to scratch memory
#endif
#define BVH_STK_DW(threads) (BVH_STK_ROWS * (threads))
__host__ __device__ constexpr uint32_t ilog2_c(uint32_t v) { return v <= 1u ? 0u : 1u + ilog2_c(v >> 1); }
// threads: the workgroup size, a power of two
#define MAKE_BVH_STACK(lds, threads) BvhStack{(LDS_AS uint32_t *)(lds) + threadIdx.x, ilog2_c(threads) + 0u, BVH_STK_ROWS}
// static LDS of a kernel that may walk a BVH: its traversal stacks (one dword for the brute-force variants)
#define BVH_STACK_LDS(ACCEL, THREADS)                                                                              \
    static_assert(((THREADS) & ((THREADS)-1)) == 0, "BVH stack rows are addressed by a shift");                    \
    __shared__ uint32_t bvh_stk_lds[((ACCEL) == ACCEL_K_BVH_GLOBAL || (ACCEL) == ACCEL_K_BVH_LDS) ? BVH_STK_DW(THREADS) : 1]
#define NO_LDS_SCENE {NO_TREE_LDS, BvhStack{nullptr, 0u, 0u}}

// SEGMENT: the ray is a segment between two points of the scene (next-event shadow ray): brute-force
// scenes then only walk the primitives that can occlude such a segment (DevScene::occ_prims).
template <int ACCEL, bool ANY, bool SEGMENT = false>
DEV bool scene_intersect(const DevScene &sc, const LdsScene &ls, V3 o, V3 d, float tmax, Hit *h) {
#ifdef PBRT_BRUTE_PAIRS
    if (ACCEL == ACCEL_K_BRUTE && !ANY) return brute_closest_pairs(sc, o, d, tmax, h);
#endif
    if (ACCEL == ACCEL_K_BRUTE || ACCEL == ACCEL_K_BRUTE_BIG)
        return brute_intersect<ANY, SEGMENT, ACCEL != ACCEL_K_BRUTE>(sc, o, d, tmax, h);
    if (ACCEL == ACCEL_K_BVH_GLOBAL) return bvh_intersect<ANY>(TreeGlobal{sc.nodes, sc.lprims}, sc.prims, ls.stk, o, d, tmax, h);
    return bvh_intersect<ANY>(ls.tree, sc.prims, ls.stk, o, d, tmax, h);
}

// ACCEL_K_BRUTE: copy the (<= 32-entry) shading tables into LDS; layout [prims | mats | emitters | light_prims | light_cdf]
// fill: issued AFTER the workgroup's path-state loads so that both round trips overlap; ends with the barrier
// n_threads: the threads of the workgroup that take part (a multiple of 64; waves that left early do not)
DEV void fill_tables_lds(const DevScene &sc, uint32_t *lds, uint32_t n_threads) {
    uint32_t *p_prims = lds, *p_mats = lds + TAB_MAX * 16, *p_emit = p_mats + TAB_MAX * 8, *p_lp = p_emit + TAB_MAX * 12;
    float *p_lc = reinterpret_cast<float *>(p_lp + TAB_MAX);
    for (uint32_t t = threadIdx.x; t < TAB_MAX * 16; t += n_threads) {
        if (t < sc.n_prims * 16) p_prims[t] = reinterpret_cast<const uint32_t *>(sc.prims)[t];
        if (t < sc.n_mats * 8) p_mats[t] = reinterpret_cast<const uint32_t *>(sc.mats)[t];
        if (t < sc.n_emitters * 12) p_emit[t] = reinterpret_cast<const uint32_t *>(sc.emitters)[t];
        if (t < sc.n_light_prims) {
            p_lp[t] = sc.light_prims[t];
            p_lc[t] = sc.light_cdf[t];
        }
    }
    __syncthreads();
}
DEV Tables stage_tables_lds(const DevScene &sc, uint32_t *lds) {
    uint32_t *p_prims = lds, *p_mats = lds + TAB_MAX * 16, *p_emit = p_mats + TAB_MAX * 8, *p_lp = p_emit + TAB_MAX * 12;
    float *p_lc = reinterpret_cast<float *>(p_lp + TAB_MAX);
    Tables tb;
    tb.prims_by_slot = reinterpret_cast<const pbrt_prim *>(p_prims);
    tb.prims_by_id = tb.prims_by_slot;
    tb.mats = reinterpret_cast<const pbrt_material *>(p_mats);
    tb.emitters = reinterpret_cast<const pbrt_emitter *>(p_emit);
    tb.light_prims = p_lp;
    tb.light_cdf = p_lc;
    tb.n_emitters = sc.n_emitters;
    return tb;
}

template <int ACCEL>
DEV Tables make_tables(const DevScene &sc, const LdsScene &ls, uint32_t *tab_lds) {
    if (ACCEL == ACCEL_K_BRUTE) return stage_tables_lds(sc, tab_lds);
    return global_tables(sc);  // BVH: Hit::slot is the caller's index, the hit primitive's full record comes from global memory
}

// Stage the tree into LDS (cooperative): the 64-byte node records of the global image become the planes of TreeLds (quad k of
// node n -> plane k, 16 bytes per lane per step; the fourth quad keeps its first 8 bytes), the 40-byte leaf records follow as
// they are.  Ends with a barrier.
DEV TreeLds stage_tree_lds(const DevScene &sc, uint32_t *lds) {
    const uint32_t n = sc.n_nodes, plane = n * 16u, leaf_off = n * LDS_IMAGE_NODE_BYTES, prim_dw = sc.n_prims * 10u;
    LDS_AS char *img = reinterpret_cast<LDS_AS char *>((LDS_AS uint32_t *)lds);
    const u32x4 *src_n = reinterpret_cast<const u32x4 *>(sc.nodes);
    for (uint32_t i = threadIdx.x; i < n * 4u; i += blockDim.x) {
        const u32x4 q = src_n[i];
        const uint32_t node = i >> 2, k = i & 3u;
        if (k < 3u) {
            *reinterpret_cast<LDS_AS u32x4 *>(img + k * plane + node * 16u) = q;
        } else {
            const u32x2 h = {q.x, q.y};
            *reinterpret_cast<LDS_AS u32x2 *>(img + 3u * plane + node * 8u) = h;
        }
    }
    const u32x2 *src_p = reinterpret_cast<const u32x2 *>(sc.lprims);
    LDS_AS u32x2 *dst_p = reinterpret_cast<LDS_AS u32x2 *>(img + leaf_off);
    for (uint32_t i = threadIdx.x; i < prim_dw / 2u; i += blockDim.x) dst_p[i] = src_p[i];
    __syncthreads();
    return TreeLds{reinterpret_cast<const LDS_AS uint32_t *>(img), plane, leaf_off};
}

// One bounce of one path (the body shared by k_bounce and k_walk): closest hit, emission + MIS, next-event estimation with
// its shadow segment, BSDF sample, Russian roulette.  Returns whether the path goes on; a path that ends writes its
// radiance to Lhome[home].
// HAVE_HIT: the closest hit was found earlier (k_bounce_pool) and comes in through h_in; the step starts at the shading.
template <int ACCEL, bool HAVE_HIT = false>
DEV bool bounce_step(const RadArgs &a, const Tables &tb, const LdsScene &ls, Rsrc r_L, uint32_t depth, uint32_t ka, uint32_t kb,
                     uint32_t home, float tmax, V3 &o, V3 &d, V3 &thr, V3 &L, float &eta, float &prev_pdf, bool &did_seg,
                     bool &did_shadow, const Hit *h_in = nullptr) {
    bool survive = false;
        const uint32_t nE = a.sc.n_emitters;
        Hit h;
        if (HAVE_HIT) h = *h_in;
        if (HAVE_HIT || scene_intersect<ACCEL, false>(a.sc, ls, o, d, tmax, &h)) {
            did_seg = true;
            const pbrt_prim &P = tb.prims_by_slot[h.slot];
            SI si = make_si<ACCEL != ACCEL_K_BRUTE>(P, o, d, h.t, h.u, h.v, a.sc.vnormals, h.slot);
            const int32_t emitter = P.emitter;
            const uint32_t mat_id = P.material;
            // ---- direct emission (one-sided area emitters), MIS against emitter sampling
            if (emitter >= 0) {
                const pbrt_emitter &E = tb.emitters[emitter];
                float cosl = -dot(si.ns, d);  // Frame::cos_theta(si.wi): the shading frame (== n on emitters)
                if (cosl > 0.0f) {
                    float w = 1.0f;
                    if (prev_pdf >= 0.0f) {
                        float pdf_em = (h.t * h.t) / (cosl * E.area * (float)nE);
                        w = mis_weight(prev_pdf, pdf_em);
                    }
                    L = {fma_(thr.x * E.radiance[0], w, L.x), fma_(thr.y * E.radiance[1], w, L.y),
                         fma_(thr.z * E.radiance[2], w, L.z)};
                }
            }
            if (depth + 1 < a.max_depth) {
                const pbrt_material M = tb.mats[mat_id];
                Frame fr = make_frame(si.ns);
                V3 wi = to_local(fr, -d);
                // ---- emitter sampling (next-event estimation) + shadow ray
#ifdef PBRT_ABLATE_NEE  // diagnostic builds only (tools/ablate.sh): never defined in the shipped library
                if (false) {
#else
                if (M.type == PBRT_MAT_DIFFUSE && nE > 0) {
#endif
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
                            did_shadow = true;
                            Hit hs;
#ifdef PBRT_ABLATE_SHADOW
                            if (sd > 0.0f) {
#else
                            if (!scene_intersect<ACCEL, true, true>(a.sc, ls, so, sdir, sd * (1.0f - K_SHADOW_EPS), &hs)) {
#endif
                                float mis = es.delta ? 1.0f : mis_weight(es.pdf, bpdf);
                                L = {fma_(thr.x * f.x, es.weight.x * mis, L.x), fma_(thr.y * f.y, es.weight.y * mis, L.y),
                                     fma_(thr.z * f.z, es.weight.z * mis, L.z)};
                            }
                        }
                    }
                }
                // ---- BSDF sampling, continuation ray, Russian roulette
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
        }
        if (!survive) {
            // one 16-byte record per finished path: three 4-byte row stores at a scattered `home` cost three
            // 32-byte HBM sectors (measured: k_bounce WRITE_SIZE 1.23 x algorithmic), one dwordx4 store costs one
            typedef uint32_t __attribute__((ext_vector_type(4))) u32x4;
            const u32x4 rec = {__float_as_uint(L.x), __float_as_uint(L.y), __float_as_uint(L.z), 0u};
            __builtin_amdgcn_raw_buffer_store_b128(rec, r_L, home * 16u, 0, 0);
        }
    return survive;
}

// NB = 2: the multi-bounce variant (brute-force kernels only): the launch walks a.nb (2 .. MAX_CHAIN) bounces.  A path that
// survives a bounce of the launch goes straight on in registers -- no state write, no compaction, no state read in between -- and the
// lanes whose paths ended idle through the second bounce (87 % / 83 % of the lanes stay busy at depths 0 / 2 of the
// Cornell box, 20 % at depth 4 -- where the second bounce is the last one and only looks for emitters).  Same arithmetic
// per bounce, same RNG keys: the film does not change.  Which depths start a two-bounce launch is the host's fuse plan
// (pbrt_api.hip PBRT_DEFAULT_FUSE_PLAN: every pair).
template <bool FIRST, int ACCEL, int NB = 1>
// two-bounce variants of ACCEL_K_BRUTE at the 64-register budget: 8 waves per SIMD with ONE spilled VGPR (a 4-byte scratch store
// and load per bounce).  Without the budget the kernel takes 67 VGPRs = 7 waves, i.e. three 512-thread workgroups per CU instead of
// four: 6.99 - 7.03 -> 6.82 - 6.85 ms on the Cornell box.  (Before -fno-slp-vectorize the same budget spilled 5 VGPRs and wrote
// 700 MB of scratch per launch for +0.8 %: not taken then.)
#ifndef FUSED_WAVES_PER_EU
#define FUSED_WAVES_PER_EU 8
#endif
__global__ __launch_bounds__(seg_threads(ACCEL), NB > 1 ? (ACCEL == ACCEL_K_BRUTE ? FUSED_WAVES_PER_EU : BIG_WAVES_PER_EU) : seg_waves_per_eu(ACCEL)) void k_bounce(const RadArgs a) {
    static_assert(NB == 1 || ACCEL == ACCEL_K_BRUTE || ACCEL == ACCEL_K_BRUTE_BIG, "fused bounces: brute-force kernels only");
    // the per-bounce survivor counts of a chain are packed 10 bits each per WORKGROUP and summed by thread 0: one segment per region,
    // workgroup-level compaction (a -DREGION_SEGS_BRUTE / -DPBRT_BRUTE_DYN build would silently lose them)
    static_assert(NB == 1 || (rad_region_segs(ACCEL) == 1 && !rad_wave_private(ACCEL) && seg_threads(ACCEL) <= 1023),
                  "chain launches: REGION == SEG, workgroup scan");
    constexpr uint32_t SEG = seg_threads(ACCEL);
    extern __shared__ __attribute__((aligned(16))) uint32_t dyn_lds[];
    __shared__ uint32_t wave_tot[2][SEG / 64];  // double-buffered across the chunk loop: one barrier per chunk
    __shared__ uint32_t wave_seg[2][SEG / 64];
    __shared__ uint32_t wave_shd[2][SEG / 64];
    // multi-bounce launches: paths that went on to the launch's 2nd .. 6th bounce, 10 bits each (a workgroup has <= 512):
    // bounces 2 - 4 in wave_mid, 5 - 6 in wave_mid_hi
    __shared__ uint32_t wave_mid[2][NB > 1 ? SEG / 64 : 1], wave_mid_hi[2][NB > 1 ? SEG / 64 : 1];
    // Repack inside a chain launch (round 4, build switch -DPBRT_CHAIN_REPACK: measured, lost): in the Cornell
    // box 100 / 87 / 67 / 56 / 47 / 9 % of the lanes carry a path at bounces 0 .. 5.  After the bounces the host names
    // (a.repack_mask, PBRT_CHAIN_REPACK=mask in the environment) the workgroup packs its live paths to its first lanes through LDS
    // -- 15 dwords each, [row][thread] so that both sides are conflict-free -- and the waves left without a path retire (s_endpgm;
    // the barriers do not wait for them).  Same paths, same arithmetic, other lanes: the film does not change.  What it gives is
    // 2 % (a wave with few live lanes skips most of a bounce anyway); what its code costs the kernel at the 64-register budget is
    // 8 % (20 spilled VGPRs, 30 KB of LDS per workgroup): 6.33 -> 6.84 ms with the switch compiled in and no bounce named, 6.69 at best.
#ifdef PBRT_CHAIN_REPACK
    constexpr bool REPACK = NB > 1 && ACCEL == ACCEL_K_BRUTE;
#else
    constexpr bool REPACK = false;
#endif
    __shared__ uint32_t rp_cnt[REPACK ? SEG / 64 : 1];
    __shared__ float rp_state[REPACK ? N_STATE : 1][REPACK ? SEG : 1];

    const uint32_t seg = xcd_swizzle(blockIdx.x, gridDim.x);  // region index
    const uint32_t tid = threadIdx.x;
    constexpr uint32_t REGION = rad_region_segs(ACCEL) * SEG;
    // 8 x 8 pixel tiles per wave (path_key): coherent rays for the BVH kernels; for the brute-force kernels, whose primitive loop does
    // not care, it keeps the long paths of a chain launch (the pixels of the glass and the mirror sphere) together in the same
    // waves instead of one or two lanes in every wave of a row: Cornell box, one launch per pass, 6.41 -> 6.25 ms
    constexpr bool TILED = true;
    constexpr bool DYN = rad_dynamic(ACCEL);                  // chunk queue + slot reservation in LDS (BVH kernels)
    constexpr bool WP = rad_wave_private(ACCEL) && !DYN;      // fixed per-wave shares (diagnostic fallback)
    constexpr bool PERWAVE = WP || DYN;                       // the waves walk 64-path chunks on their own
    constexpr uint32_t W = SEG / 64, WREG = REGION / W;       // waves per workgroup, slots owned by one wave (WP)
    constexpr uint32_t CH = PERWAVE ? 64u : SEG;              // paths per chunk of the walk
    const uint32_t lane_c = PERWAVE ? (tid & 63u) : tid;      // position inside the chunk
    const uint32_t own = WP ? seg * W + (tid >> 6) : seg;     // live counter of this wave / workgroup
    const uint32_t row_id = PERWAVE ? seg * W + (tid >> 6) : seg;  // statistics row
    const uint32_t base = WP ? seg * REGION + (tid >> 6) * WREG : seg * REGION;
    __shared__ uint32_t q_in, q_out, q_done;                  // DYN: next chunk, next free output slot, finished waves
    if (DYN && tid == 0) {
        q_in = 0;
        q_out = 0;
        q_done = 0;
    }
    uint32_t cnt_in;
    if (FIRST) {
        cnt_in = a.n_paths > base ? min(a.n_paths - base, WP ? WREG : REGION) : 0u;
    } else {
        cnt_in = a.seg_in[own];
    }
    if (WP) {
        // the workgroup stages the scene together: leave only if no wave of it has work (same answer in every wave)
        cnt_in = (uint32_t)__builtin_amdgcn_readfirstlane((int)cnt_in);
        uint32_t c = 0;
        if ((tid & 63u) < W) {
            const uint32_t b2 = seg * REGION + (tid & 63u) * WREG;
            c = FIRST ? (a.n_paths > b2 ? 1u : 0u) : a.seg_in[seg * W + (tid & 63u)];
        }
        if (__ballot(c != 0) == 0) {
            if ((tid & 63u) == 0) a.seg_out[own] = 0;
            return;
        }
    } else if (cnt_in == 0) {  // uniform across the workgroup
        if (tid == 0) a.seg_out[seg] = 0;
        return;
    }
    if (DYN) cnt_in = (uint32_t)__builtin_amdgcn_readfirstlane((int)cnt_in);
    // Waves without a live path leave at once instead of idling until the compaction barrier: their wave slots
    // are free for the next workgroup's waves (late bounces run at 10-50 % fill).  They publish a zero survivor
    // count first; s_barrier does not wait for terminated waves.
#ifndef PBRT_ABLATE_EARLY_EXIT
    constexpr bool early_exit = ACCEL == ACCEL_K_BRUTE && !FIRST && rad_region_segs(ACCEL) == 1 && !PERWAVE;
#else
    constexpr bool early_exit = false;
#endif
    if (early_exit) {
        if ((tid & 63u) == 0) {  // every wave; the live ones overwrite their entries before the compaction barrier
            wave_tot[0][tid >> 6] = 0;
            wave_seg[0][tid >> 6] = 0;
            wave_shd[0][tid >> 6] = 0;
            if (NB > 1) {
                wave_mid[0][tid >> 6] = 0;
                wave_mid_hi[0][tid >> 6] = 0;
            }
        }
        // s_endpgm behind the compiler's back keeps the kernel single-exit for the structurizer
        const uint32_t wave_first = (uint32_t)__builtin_amdgcn_readfirstlane((int)tid) & ~63u;
        asm volatile("s_cmp_lt_u32 %0, %1\n\t"
                     "s_cbranch_scc1 .Llive_%=\n\t"
                     "s_waitcnt lgkmcnt(0)\n\t"
                     "s_endpgm\n"
                     ".Llive_%=:" ::"s"(wave_first), "s"(cnt_in) : "scc", "memory");
    }
    const uint32_t live_threads = early_exit ? (min(cnt_in, SEG) + 63u) & ~63u : SEG;  // waves still present
    BVH_STACK_LDS(ACCEL, SEG);
    LdsScene ls = {NO_TREE_LDS, MAKE_BVH_STACK(bvh_stk_lds, SEG)};
    if (ACCEL == ACCEL_K_BVH_LDS) ls.tree = stage_tree_lds(a.sc, dyn_lds);  // ends with a barrier
    if (DYN && ACCEL != ACCEL_K_BVH_LDS) __syncthreads();               // publishes the queue words
    __shared__ uint32_t tab_lds[ACCEL == ACCEL_K_BRUTE ? TAB_DW : 1];
    const Tables tb = make_tables<ACCEL>(a.sc, ls, tab_lds);
    if (ACCEL == ACCEL_K_BRUTE && (FIRST || PERWAVE)) fill_tables_lds(a.sc, tab_lds, SEG);  // (PERWAVE: no barrier in the walk)

    const uint32_t cap = a.cap;
    const Rsrc r_in = make_rsrc(a.in, a.state_cap * (N_STATE * 4u)), r_out = make_rsrc(a.out, a.state_cap * (N_STATE * 4u));
    const Rsrc r_L = make_rsrc(a.Lhome, cap * 16u);
    uint32_t out_off = 0;      // survivors written so far (front of this region of the `out` state)
    uint32_t ns_acc = 0, nh_acc = 0, live_acc = 0, mid_acc = 0, mid_acc_hi = 0;
    const uint32_t nb_run = NB > 1 ? min(a.nb, (uint32_t)MAX_CHAIN) : 1u;  // bounces this launch walks
    // the region's live paths sit compacted at its front: walk them SEG at a time; dead slots cost nothing
    for (uint32_t it0 = 0; it0 < (REGION > SEG ? cnt_in : 1u); it0 += CH) {  // single trip when REGION == SEG
    if (DYN) {  // take the next 64-path chunk of the region from the workgroup's queue
        uint32_t nxt = 0;
        if ((tid & 63u) == 0) nxt = atomicAdd(&q_in, 64u);
        it0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)nxt);
        if (it0 >= cnt_in) break;
    }
    const uint32_t buf = (it0 / SEG) & 1u;
    const bool alive = it0 + lane_c < cnt_in;
    const uint32_t slot = base + it0 + lane_c;
    bool survive = false;
    bool did_seg = false, did_shadow = false;
    V3 o, d, thr, L;
    float eta, prev_pdf, tmax;
    uint32_t home = slot;
    uint32_t ka = 0, kb = 0, px = 0, py = 0;
    if (alive) {
        if (FIRST) {
            home = slot;
            path_key<TILED>(a, home, &ka, &kb, &px, &py);
            F4 uj = rng4(ka, kb, 0, a.seed);
            float fx = (float)px + uj.x, fy = (float)py + uj.y;
            camera_ray(a.cam, fx / (float)a.film_w, fy / (float)a.film_h, &o, &d, &tmax);
            thr = {1, 1, 1};
            L = {0, 0, 0};
            eta = 1.0f;
            prev_pdf = -1.0f;
        } else {
            const uint32_t v4 = state_voff(slot);
            constexpr uint32_t row = STATE_ROW_BYTES;
            o = {bld(r_in, v4 + 0 * row, 0), bld(r_in, v4 + 1 * row, 0), bld(r_in, v4 + 2 * row, 0)};
            d = {bld(r_in, v4 + 3 * row, 0), bld(r_in, v4 + 4 * row, 0), bld(r_in, v4 + 5 * row, 0)};
            thr = {bld(r_in, v4 + 6 * row, 0), bld(r_in, v4 + 7 * row, 0), bld(r_in, v4 + 8 * row, 0)};
            L = {bld(r_in, v4 + 9 * row, 0), bld(r_in, v4 + 10 * row, 0), bld(r_in, v4 + 11 * row, 0)};
            eta = bld(r_in, v4 + 12 * row, 0);
            prev_pdf = bld(r_in, v4 + 13 * row, 0);
            home = __float_as_uint(bld(r_in, v4 + 14 * row, 0));
            tmax = (a.key_mode == 1 && a.depth == 0) ? eta : K_INF;  // caller rays carry tmax in the eta slot
            if (a.key_mode == 1 && a.depth == 0) eta = 1.0f;
        }
    }
    // shading tables -> LDS, behind the state loads so that the two memory round trips overlap
    if (ACCEL == ACCEL_K_BRUTE && !FIRST && !PERWAVE && it0 == 0) fill_tables_lds(a.sc, tab_lds, live_threads);
    if (alive && !FIRST) path_key<TILED>(a, home, &ka, &kb, &px, &py);
    bool live = alive;                        // the lane still carries a path
    uint32_t nseg_w = 0, nshd_w = 0, nmid_w = 0, nmid_hi_w = 0;  // wave-uniform counts over the launch's bounces
#pragma unroll 1
    for (uint32_t bounce = 0; bounce < nb_run; ++bounce) {
    const uint32_t depth = a.depth + bounce;
    if (NB > 1 && bounce > 0) {
        if (depth >= a.max_depth) break;      // uniform
        const uint32_t n_on = (uint32_t)__popcll(__ballot(live));
        if (bounce <= 3)
            nmid_w += n_on << (10u * (bounce - 1u));
        else
            nmid_hi_w += n_on << (10u * (bounce - 4u));
        tmax = K_INF;
    }
    did_seg = false;
    did_shadow = false;
    survive = false;
    if (live) {
#ifdef PBRT_PROBE_EXTRA_VALU  // diagnostic builds only: N dependent full-rate VALU instructions per live wave-bounce
        {
            float probe = o.x;
#pragma unroll
            for (int k = 0; k < PBRT_PROBE_EXTRA_VALU; ++k) asm volatile("v_add_f32 %0, %0, %0" : "+v"(probe));
            if (probe == 12345.678f) o.x = probe;  // never true; keeps the chain alive
        }
#endif
        survive = bounce_step<ACCEL>(a, tb, ls, r_L, depth, ka, kb, home, tmax, o, d, thr, L, eta, prev_pdf, did_seg, did_shadow);
    }
    live = survive;
    nseg_w += (uint32_t)__popcll(__ballot(did_seg));
    nshd_w += (uint32_t)__popcll(__ballot(did_shadow));
    if (REPACK && ((a.repack_mask >> bounce) & 1u) && bounce + 1 < nb_run && depth + 1 < a.max_depth) {  // uniform
        const uint32_t wid_r = tid >> 6;
        const unsigned long long bl = __ballot(live);
        const uint32_t pre = __builtin_amdgcn_mbcnt_hi((uint32_t)(bl >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bl, 0u));
        if ((tid & 63u) == 0) rp_cnt[wid_r] = (uint32_t)__popcll(bl);
        __syncthreads();
        const uint32_t c_lane = rp_cnt[tid & (SEG / 64 - 1)];
        const uint32_t wid_s = (uint32_t)__builtin_amdgcn_readfirstlane((int)wid_r);
        uint32_t off_r = 0, total_r = 0;
#pragma unroll
        for (uint32_t w = 0; w < SEG / 64; ++w) {
            const uint32_t t = (uint32_t)__builtin_amdgcn_readlane((int)c_lane, (int)w);
            off_r += (w < wid_s) ? t : 0u;
            total_r += t;
        }
        if (live) {
            const uint32_t k = off_r + pre;
            rp_state[0][k] = o.x;
            rp_state[1][k] = o.y;
            rp_state[2][k] = o.z;
            rp_state[3][k] = d.x;
            rp_state[4][k] = d.y;
            rp_state[5][k] = d.z;
            rp_state[6][k] = thr.x;
            rp_state[7][k] = thr.y;
            rp_state[8][k] = thr.z;
            rp_state[9][k] = L.x;
            rp_state[10][k] = L.y;
            rp_state[11][k] = L.z;
            rp_state[12][k] = eta;
            rp_state[13][k] = prev_pdf;
            rp_state[14][k] = __uint_as_float(home);
        }
        __syncthreads();
        live = tid < total_r;
        if (live) {
            o = {rp_state[0][tid], rp_state[1][tid], rp_state[2][tid]};
            d = {rp_state[3][tid], rp_state[4][tid], rp_state[5][tid]};
            thr = {rp_state[6][tid], rp_state[7][tid], rp_state[8][tid]};
            L = {rp_state[9][tid], rp_state[10][tid], rp_state[11][tid]};
            eta = rp_state[12][tid];
            prev_pdf = rp_state[13][tid];
            home = __float_as_uint(rp_state[14][tid]);
            path_key<TILED>(a, home, &ka, &kb, &px, &py);
        }
        // a wave left without a path retires: it publishes what the end of the kernel expects from it (its counts of this launch,
        // no survivors) and ends; the waves that go on never wait for it.  (The next repack must see it with no live lane: rp_cnt.)
        // s_endpgm behind the compiler's back keeps the kernel single-exit for the structurizer (as the early exit above).
        const uint32_t wave_first_r = (uint32_t)__builtin_amdgcn_readfirstlane((int)tid) & ~63u;
        if ((tid & 63u) == 0 && wave_first_r >= total_r && wave_first_r != 0u) {  // (wave 0 stays: thread 0 writes the region's counters)
            rp_cnt[wid_r] = 0;
            wave_tot[buf][wid_r] = 0;
            wave_seg[buf][wid_r] = nseg_w;
            wave_shd[buf][wid_r] = nshd_w;
            wave_mid[buf][wid_r] = nmid_w;
            wave_mid_hi[buf][wid_r] = nmid_hi_w;
        }
        const uint32_t keep = (uint32_t)__builtin_amdgcn_readfirstlane((int)((wave_first_r < total_r || wave_first_r == 0u) ? 1u : 0u));
        asm volatile("s_cmp_lg_u32 %0, 0\n\t"
                     "s_cbranch_scc1 .Lstay_%=\n\t"
                     "s_waitcnt lgkmcnt(0)\n\t"
                     "s_endpgm\n"
                     ".Lstay_%=:" ::"s"(keep) : "scc", "memory");
    }
    }  // bounces of this launch
    // ---- segment-local stream compaction: ballot + mbcnt inside the wave, LDS scan across waves
    const uint32_t wid = tid >> 6;
    const unsigned long long bal = __ballot(survive);
    const uint32_t prefix = __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u));
    uint32_t off = 0, total = 0;
    if (DYN) {  // reserve the survivors' slots in the region with one returning LDS atomic per wave and chunk
        const uint32_t cnt_w = (uint32_t)__popcll(bal);
        uint32_t got = 0;
        if ((tid & 63u) == 0 && cnt_w) got = atomicAdd(&q_out, cnt_w);
        off = (uint32_t)__builtin_amdgcn_readfirstlane((int)got);
        ns_acc += nseg_w;
        nh_acc += nshd_w;
        live_acc += (uint32_t)__popcll(__ballot(alive));
    } else if (WP) {  // the wave packs its own survivors behind its own cursor: no LDS, no barrier
        total = (uint32_t)__popcll(bal);
        ns_acc += nseg_w;
        nh_acc += nshd_w;
    } else {
    if ((tid & 63) == 0) {
        wave_tot[buf][wid] = (uint32_t)__popcll(bal);
        wave_seg[buf][wid] = nseg_w;
        wave_shd[buf][wid] = nshd_w;
        if (NB > 1) {
            wave_mid[buf][wid] = nmid_w;
            wave_mid_hi[buf][wid] = nmid_hi_w;
        }
    }
    __syncthreads();
    // exclusive scan over the waves' survivor counts on the scalar unit: one LDS read per lane, then
    // v_readlane + s_add per wave (the per-lane form cost 40 VALU per wave-bounce)
        const uint32_t t_lane = wave_tot[buf][tid & (SEG / 64 - 1)];
        const uint32_t wid_s = (uint32_t)__builtin_amdgcn_readfirstlane((int)wid);
#pragma unroll
        for (uint32_t w = 0; w < SEG / 64; ++w) {
            const uint32_t t = (uint32_t)__builtin_amdgcn_readlane((int)t_lane, (int)w);
            off += (w < wid_s) ? t : 0u;
            total += t;
        }
    }
    if (survive) {
        const uint32_t v4 = state_voff(base + out_off + off + prefix);
        constexpr uint32_t row = STATE_ROW_BYTES;
        bst(r_out, v4 + 0 * row, 0, o.x);
        bst(r_out, v4 + 1 * row, 0, o.y);
        bst(r_out, v4 + 2 * row, 0, o.z);
        bst(r_out, v4 + 3 * row, 0, d.x);
        bst(r_out, v4 + 4 * row, 0, d.y);
        bst(r_out, v4 + 5 * row, 0, d.z);
        bst(r_out, v4 + 6 * row, 0, thr.x);
        bst(r_out, v4 + 7 * row, 0, thr.y);
        bst(r_out, v4 + 8 * row, 0, thr.z);
        bst(r_out, v4 + 9 * row, 0, L.x);
        bst(r_out, v4 + 10 * row, 0, L.y);
        bst(r_out, v4 + 11 * row, 0, L.z);
        bst(r_out, v4 + 12 * row, 0, eta);
        bst(r_out, v4 + 13 * row, 0, prev_pdf);
        bst(r_out, v4 + 14 * row, 0, __uint_as_float(home));
    }
    out_off += total;
    if (!PERWAVE && tid == 0) {
        for (uint32_t w = 0; w < SEG / 64; ++w) {
            ns_acc += wave_seg[buf][w];
            nh_acc += wave_shd[buf][w];
            if (NB > 1) {
                mid_acc += wave_mid[buf][w];
                mid_acc_hi += wave_mid_hi[buf][w];
            }
        }
    }
    }  // chunk loop
    if (DYN) {
        if ((tid & 63u) == 0) {
            unsigned long long *row = a.stats + row_id;  // per-wave statistics rows
            const size_t stride = a.stat_stride;
            row[0] += ns_acc;
            row[stride] += nh_acc;
            row[(2 + min(a.depth, (uint32_t)MAX_DEPTH_STATS - 1)) * stride] += live_acc;
            // the last wave to finish publishes the region's survivor count (LDS atomics of one CU are ordered)
            if (atomicAdd(&q_done, 1u) == W - 1) a.seg_out[seg] = atomicAdd(&q_out, 0u);
        }
        return;
    }
    if (WP ? (tid & 63u) == 0 : tid == 0) {
        a.seg_out[own] = out_off;
        // per-region statistics rows (plain read-modify-write by the owning workgroup; launches of a
        // call are ordered on the stream).  NOT global atomics: 3 same-line atomics per workgroup
        // serialise at ~12 ns each and were the whole kernel time (DESIGN.md "What did not work").
        unsigned long long *row = a.stats + row_id;
        const size_t stride = a.stat_stride;
        row[0] += ns_acc;
        row[stride] += nh_acc;
        row[(2 + min(a.depth, (uint32_t)MAX_DEPTH_STATS - 1)) * stride] += cnt_in;
        if (NB > 1)
            for (uint32_t k = 1; k < nb_run; ++k) {
                const uint32_t n_on = k <= 3 ? (mid_acc >> (10u * (k - 1u))) & 1023u : (mid_acc_hi >> (10u * (k - 4u))) & 1023u;
                row[(2 + min(a.depth + k, (uint32_t)MAX_DEPTH_STATS - 1)) * stride] += n_on;
            }
    }
}

#ifdef PBRT_DIAG  // launch structures that lost their A/B: kept for the diagnostic build only (make diag)
// ---- k_chain_pair: the whole path in one launch, TWO 64-path tiles per wave ---------------------------------------------------
// A chain launch keeps a wave busy until the longest of its 64 paths has ended: in the Cornell box 100 / 87 / 67 / 56 / 47 / 9 % of
// the lanes carry a path at bounces 0 .. 5, and a bounce costs a wave the same whether 64 or 6 of its lanes are live.  Here a wave
// walks tile A up to bounce a.merge_at, parks the survivors in a wave-private LDS strip (N_STATE dwords each, packed), walks tile B
// up to the same bounce, deals the parked paths to B's idle lanes and walks the rest of both tiles ONCE.  Parked paths that find no
// idle lane (more than 64 survivors in the pair) are walked afterwards in a turn of their own.  Every path does the arithmetic it
// did before with the same RNG keys (path_key of its home), so the film does not change.  The strip is only touched by its own wave:
// LDS operations of one wave complete in order, no barrier.
// Measured (Cornell box 512^2 x 256, one launch per pass): k_bounce chain 6.02 - 6.07 ms; this kernel with merge bounce 3 / 4 / 5:
// 6.37 / 6.43 - 6.55 / 6.48 ms.  Merging at bounce 5 merges nothing that costs (the last bounce only looks for emitters), so 6.48 is
// the price of the structure itself (12 spilled VGPRs and 50 spilled SGPRs at the 64-register budget, against 1 and 13), and the
// merge buys 0.1 ms of the 0.6 - 0.8 ms that the count of wave-bounces promises: a wave with half of its lanes idle skips the
// branches that none of its paths takes, a full wave of paths from two tiles takes them all.  Diagnostic build only
// (PBRT_PAIR_MERGE=k picks the merge bounce).
// Host: max_depth <= MAX_CHAIN (every path ends inside the launch: no state goes out), 1 <= merge_at < max_depth,
// grid = ceil(n_paths / (2 * SEG_BRUTE)).
template <int ACCEL>
__global__ __launch_bounds__(SEG_BRUTE, ACCEL == ACCEL_K_BRUTE ? FUSED_WAVES_PER_EU : BIG_WAVES_PER_EU) void k_chain_pair(const RadArgs a) {
    static_assert(ACCEL == ACCEL_K_BRUTE || ACCEL == ACCEL_K_BRUTE_BIG, "k_chain_pair: brute-force kernels");
    constexpr uint32_t SEG = SEG_BRUTE, W = SEG / 64;
    __shared__ uint32_t tab_lds[ACCEL == ACCEL_K_BRUTE ? TAB_DW : 1];
    __shared__ uint32_t park[W][N_STATE][64];
    __shared__ uint32_t wave_cnt[W][2 + MAX_CHAIN];  // per wave: segments, shadow rays, paths entering bounce b (lane 0 adds)
    const uint32_t region = xcd_swizzle(blockIdx.x, gridDim.x);
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wid = tid >> 6;
    const LdsScene ls = NO_LDS_SCENE;
    const Tables tb = make_tables<ACCEL>(a.sc, ls, tab_lds);
    if (ACCEL == ACCEL_K_BRUTE) fill_tables_lds(a.sc, tab_lds, SEG);  // ends with a barrier
    const Rsrc r_L = make_rsrc(a.Lhome, a.cap * 16u);
    uint32_t(*pk)[64] = park[wid];
    const uint32_t tile0 = (region * W + wid) * 2u;
    const uint32_t m = a.merge_at;
    // four turns (wave-uniform): 0 = tile A up to the merge bounce, 1 = tile B likewise, 2 = both from there on, 3 = parked paths
    // that found no lane
    uint32_t n_park = 0, left0 = 0, n_left = 0;
    if (lane < 2u + MAX_CHAIN) wave_cnt[wid][lane] = 0u;
    V3 o = {0, 0, 0}, d = {0, 0, 1}, thr = {0, 0, 0}, L = {0, 0, 0};
    float eta = 1.0f, prev_pdf = -1.0f, tmax = K_INF;
    uint32_t home = 0, ka = 0, kb = 0;
    bool live = false;
#pragma unroll 1
    for (uint32_t turn = 0; turn < 4u; ++turn) {
        uint32_t b_first = 0, b_end = m;
        if (turn < 2u) {  // the camera paths of the tile
            const uint32_t slot = (tile0 + turn) * 64u + lane;
            live = slot < a.n_paths;
            home = slot;
            if (live) {
                uint32_t px, py;
                path_key<true>(a, home, &ka, &kb, &px, &py);
                F4 uj = rng4(ka, kb, 0, a.seed);
                float fx = (float)px + uj.x, fy = (float)py + uj.y;
                camera_ray(a.cam, fx / (float)a.film_w, fy / (float)a.film_h, &o, &d, &tmax);
                thr = {1, 1, 1};
                L = {0, 0, 0};
                eta = 1.0f;
                prev_pdf = -1.0f;
            }
        } else {
            // turn 2: the parked paths go to the idle lanes of tile B, in order; turn 3: those that found none, from lane 0 on
            // (every lane is idle by then)
            if (turn == 3u && n_left == 0u) break;
            const unsigned long long idle = ~__ballot(live);
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(idle >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idle, 0u));
            uint32_t n_take;
            if (turn == 2u) {
                n_take = min(n_park, (uint32_t)__popcll(idle));
                left0 = n_take;
                n_left = n_park - n_take;
            } else {
                n_take = n_left;
            }
            const uint32_t r = (turn == 2u ? 0u : left0) + rank;
            if (!live && rank < n_take) {
                o = {__uint_as_float(pk[0][r]), __uint_as_float(pk[1][r]), __uint_as_float(pk[2][r])};
                d = {__uint_as_float(pk[3][r]), __uint_as_float(pk[4][r]), __uint_as_float(pk[5][r])};
                thr = {__uint_as_float(pk[6][r]), __uint_as_float(pk[7][r]), __uint_as_float(pk[8][r])};
                L = {__uint_as_float(pk[9][r]), __uint_as_float(pk[10][r]), __uint_as_float(pk[11][r])};
                eta = __uint_as_float(pk[12][r]);
                prev_pdf = __uint_as_float(pk[13][r]);
                home = pk[14][r];
                uint32_t px, py;
                path_key<true>(a, home, &ka, &kb, &px, &py);
                live = true;
            }
            b_first = m;
            b_end = a.max_depth;
        }
#pragma unroll 1
        for (uint32_t bounce = b_first; bounce < b_end; ++bounce) {
            const uint32_t n_on = (uint32_t)__popcll(__ballot(live));
            if (n_on == 0u) break;
            if (lane == 0u) atomicAdd(&wave_cnt[wid][2u + bounce], n_on);
            if (bounce > 0u) tmax = K_INF;
            bool did_seg = false, did_shadow = false, survive = false;
            if (live) survive = bounce_step<ACCEL>(a, tb, ls, r_L, bounce, ka, kb, home, tmax, o, d, thr, L, eta, prev_pdf, did_seg, did_shadow);
            live = survive;
            const uint32_t n_sg = (uint32_t)__popcll(__ballot(did_seg)), n_sh = (uint32_t)__popcll(__ballot(did_shadow));
            if (lane == 0u) {
                atomicAdd(&wave_cnt[wid][0], n_sg);
                atomicAdd(&wave_cnt[wid][1], n_sh);
            }
        }
        if (turn == 0u) {  // park the survivors of tile A
            const unsigned long long bal = __ballot(live);
            n_park = (uint32_t)__popcll(bal);
            if (live) {
                const uint32_t k = __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u));
                pk[0][k] = __float_as_uint(o.x);
                pk[1][k] = __float_as_uint(o.y);
                pk[2][k] = __float_as_uint(o.z);
                pk[3][k] = __float_as_uint(d.x);
                pk[4][k] = __float_as_uint(d.y);
                pk[5][k] = __float_as_uint(d.z);
                pk[6][k] = __float_as_uint(thr.x);
                pk[7][k] = __float_as_uint(thr.y);
                pk[8][k] = __float_as_uint(thr.z);
                pk[9][k] = __float_as_uint(L.x);
                pk[10][k] = __float_as_uint(L.y);
                pk[11][k] = __float_as_uint(L.z);
                pk[12][k] = __float_as_uint(eta);
                pk[13][k] = __float_as_uint(prev_pdf);
                pk[14][k] = home;
            }
        }
    }
    __syncthreads();
    if (tid == 0) {
        unsigned long long *row = a.stats + region;
        const size_t stride = a.stat_stride;
        for (uint32_t k = 0; k < 2u + a.max_depth; ++k) {
            uint32_t n = 0;
            for (uint32_t w = 0; w < W; ++w) n += wave_cnt[w][k];
            row[k * stride] += n;
        }
    }
}

// ---- k_walk: ONE launch walks every remaining bounce (brute-force kernels) -----------------------------------------------
// Compaction is local to the segment, so nothing forces a grid-wide barrier between bounces: the workgroup that owns a
// segment carries its survivors from bounce to bounce on its own, ping-ponging between the two state buffers (release
// fence + workgroup barrier between bounces).  What it saves over one launch per bounce: the drain and ramp of every
// launch, the launches of the late depths in which most workgroups only find an empty segment, and most of the state
// READS -- the survivors a workgroup wrote a few microseconds ago are still in its XCD's L2.  Waves whose slots lie
// beyond the live prefix can never get work again (the prefix only shrinks) and leave for good.  The first trip of the
// loop may walk NB0 = 2 bounces in registers like k_bounce<.., 2>.  Same arithmetic, same keys: the film does not change.
#ifndef WALK_WAVES_PER_EU
#define WALK_WAVES_PER_EU 4
#endif
template <bool FIRST, int ACCEL, int NB0>
__global__ __launch_bounds__(SEG_BRUTE, WALK_WAVES_PER_EU) void k_walk(const RadArgs a) {
    static_assert(ACCEL == ACCEL_K_BRUTE || ACCEL == ACCEL_K_BRUTE_BIG, "k_walk: brute-force kernels only");
    constexpr uint32_t SEG = SEG_BRUTE, W = SEG / 64;
    __shared__ uint32_t wave_tot[2][W], wave_seg[2][W], wave_shd[2][W], wave_mid[2][W];  // double-buffered over the bounces
    __shared__ uint32_t tab_lds[ACCEL == ACCEL_K_BRUTE ? TAB_DW : 1];
    const uint32_t seg = xcd_swizzle(blockIdx.x, gridDim.x), tid = threadIdx.x, wid = tid >> 6, base = seg * SEG;
    uint32_t cnt_in = FIRST ? (a.n_paths > base ? min(a.n_paths - base, SEG) : 0u) : a.seg_in[seg];
    if (cnt_in == 0) {  // uniform across the workgroup
        if (tid == 0) a.seg_out[seg] = 0;
        return;
    }
    const uint32_t wave_first = (uint32_t)__builtin_amdgcn_readfirstlane((int)tid) & ~63u;
    // a wave that leaves has published zero counts in both buffers first; s_barrier does not wait for terminated waves,
    // and it does wait for a wave that has neither arrived nor terminated, so the zeros are in place before anybody scans
    auto leave_if_idle = [&](uint32_t cnt) {
        if ((tid & 63u) == 0 && wave_first >= cnt) {
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                wave_tot[b][wid] = 0;
                wave_seg[b][wid] = 0;
                wave_shd[b][wid] = 0;
                wave_mid[b][wid] = 0;
            }
        }
        asm volatile("s_cmp_lt_u32 %0, %1\n\t"
                     "s_cbranch_scc1 .Lstay_%=\n\t"
                     "s_waitcnt lgkmcnt(0)\n\t"
                     "s_endpgm\n"
                     ".Lstay_%=:" ::"s"(wave_first), "s"(cnt) : "scc", "memory");
    };
    leave_if_idle(cnt_in);
    const uint32_t live_threads = (min(cnt_in, SEG) + 63u) & ~63u;  // waves present at the first trip
    const LdsScene ls = NO_LDS_SCENE;
    const Tables tb = make_tables<ACCEL>(a.sc, ls, tab_lds);
    if (ACCEL == ACCEL_K_BRUTE && FIRST) fill_tables_lds(a.sc, tab_lds, live_threads);
    const uint32_t cap = a.cap;
    Rsrc r_in = make_rsrc(a.in, a.state_cap * (N_STATE * 4u)), r_out = make_rsrc(a.out, a.state_cap * (N_STATE * 4u));
    const Rsrc r_L = make_rsrc(a.Lhome, cap * 16u);
    uint32_t depth = a.depth, trip = 0, left = 0;
    uint32_t ns_acc = 0, nh_acc = 0;
    for (;;) {
        const uint32_t buf = trip & 1u;
        const bool first = FIRST && trip == 0;
        const bool alive = tid < cnt_in;
        const uint32_t slot = base + tid;
        bool survive = false, did_seg = false, did_shadow = false;
        V3 o = {0, 0, 0}, d = {0, 0, 1}, thr = {1, 1, 1}, L = {0, 0, 0};
        float eta = 1.0f, prev_pdf = -1.0f, tmax = K_INF;
        uint32_t home = slot, ka = 0, kb = 0, px = 0, py = 0;
        if (alive) {
            if (first) {
                path_key<true>(a, home, &ka, &kb, &px, &py);
                F4 uj = rng4(ka, kb, 0, a.seed);
                float fx = (float)px + uj.x, fy = (float)py + uj.y;
                camera_ray(a.cam, fx / (float)a.film_w, fy / (float)a.film_h, &o, &d, &tmax);
            } else {
                const uint32_t v4 = state_voff(slot);
                constexpr uint32_t row = STATE_ROW_BYTES;
                o = {bld(r_in, v4 + 0 * row, 0), bld(r_in, v4 + 1 * row, 0), bld(r_in, v4 + 2 * row, 0)};
                d = {bld(r_in, v4 + 3 * row, 0), bld(r_in, v4 + 4 * row, 0), bld(r_in, v4 + 5 * row, 0)};
                thr = {bld(r_in, v4 + 6 * row, 0), bld(r_in, v4 + 7 * row, 0), bld(r_in, v4 + 8 * row, 0)};
                L = {bld(r_in, v4 + 9 * row, 0), bld(r_in, v4 + 10 * row, 0), bld(r_in, v4 + 11 * row, 0)};
                eta = bld(r_in, v4 + 12 * row, 0);
                prev_pdf = bld(r_in, v4 + 13 * row, 0);
                home = __float_as_uint(bld(r_in, v4 + 14 * row, 0));
            }
        }
        // shading tables -> LDS, behind the state loads of the first trip so that the two memory round trips overlap
        if (ACCEL == ACCEL_K_BRUTE && !FIRST && trip == 0) fill_tables_lds(a.sc, tab_lds, live_threads);
        if (alive && !first) path_key<true>(a, home, &ka, &kb, &px, &py);
        const uint32_t nb = (trip == 0) ? (uint32_t)NB0 : 1u;
        bool live = alive;
        uint32_t nseg_w = 0, nshd_w = 0, nmid_w = 0;
#pragma unroll 1
        for (uint32_t bounce = 0; bounce < nb; ++bounce) {
            const uint32_t dd = depth + bounce;
            if (bounce > 0) {
                if (dd >= a.max_depth) break;  // uniform
                nmid_w += (uint32_t)__popcll(__ballot(live));
                tmax = K_INF;
            }
            did_seg = false;
            did_shadow = false;
            survive = false;
            if (live) survive = bounce_step<ACCEL>(a, tb, ls, r_L, dd, ka, kb, home, tmax, o, d, thr, L, eta, prev_pdf, did_seg, did_shadow);
            live = survive;
            nseg_w += (uint32_t)__popcll(__ballot(did_seg));
            nshd_w += (uint32_t)__popcll(__ballot(did_shadow));
        }
        // ---- segment-local stream compaction (as in k_bounce)
        const unsigned long long bal = __ballot(survive);
        const uint32_t prefix = __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u));
        if ((tid & 63u) == 0) {
            wave_tot[buf][wid] = (uint32_t)__popcll(bal);
            wave_seg[buf][wid] = nseg_w;
            wave_shd[buf][wid] = nshd_w;
            wave_mid[buf][wid] = nmid_w;
        }
        __syncthreads();
        uint32_t off = 0, total = 0;
        {
            const uint32_t t_lane = wave_tot[buf][tid & (W - 1)];
            const uint32_t wid_s = (uint32_t)__builtin_amdgcn_readfirstlane((int)wid);
#pragma unroll
            for (uint32_t w = 0; w < W; ++w) {
                const uint32_t t = (uint32_t)__builtin_amdgcn_readlane((int)t_lane, (int)w);
                off += (w < wid_s) ? t : 0u;
                total += t;
            }
        }
        if (survive) {
            const uint32_t v4 = state_voff(base + off + prefix);
            constexpr uint32_t row = STATE_ROW_BYTES;
            bst(r_out, v4 + 0 * row, 0, o.x);
            bst(r_out, v4 + 1 * row, 0, o.y);
            bst(r_out, v4 + 2 * row, 0, o.z);
            bst(r_out, v4 + 3 * row, 0, d.x);
            bst(r_out, v4 + 4 * row, 0, d.y);
            bst(r_out, v4 + 5 * row, 0, d.z);
            bst(r_out, v4 + 6 * row, 0, thr.x);
            bst(r_out, v4 + 7 * row, 0, thr.y);
            bst(r_out, v4 + 8 * row, 0, thr.z);
            bst(r_out, v4 + 9 * row, 0, L.x);
            bst(r_out, v4 + 10 * row, 0, L.y);
            bst(r_out, v4 + 11 * row, 0, L.z);
            bst(r_out, v4 + 12 * row, 0, eta);
            bst(r_out, v4 + 13 * row, 0, prev_pdf);
            bst(r_out, v4 + 14 * row, 0, __uint_as_float(home));
        }
        if (tid == 0) {  // wave 0 stays as long as the segment has a live path
            uint32_t mid = 0;
            for (uint32_t w = 0; w < W; ++w) {
                ns_acc += wave_seg[buf][w];
                nh_acc += wave_shd[buf][w];
                mid += wave_mid[buf][w];
            }
            unsigned long long *row = a.stats + seg;
            const size_t stride = a.stat_stride;
            row[(2 + min(depth, (uint32_t)MAX_DEPTH_STATS - 1)) * stride] += cnt_in;
            if (nb > 1 && depth + 1 < a.max_depth) row[(2 + min(depth + 1, (uint32_t)MAX_DEPTH_STATS - 1)) * stride] += mid;
        }
        left = total;
        depth += nb;
        if (total == 0 || depth >= a.max_depth) break;  // uniform: `total` is the same in every wave
        // the survivors just written are the next bounce's input: stores complete (release at workgroup scope; the waves
        // of a workgroup share the CU's vector L1), and everybody has finished reading the old input
        __threadfence_block();
        __syncthreads();
        const Rsrc t_r = r_in;
        r_in = r_out;
        r_out = t_r;
        cnt_in = total;
        ++trip;
        leave_if_idle(cnt_in);
    }
    if (tid == 0) {
        a.seg_out[seg] = left;
        unsigned long long *row = a.stats + seg;
        row[0] += ns_acc;
        row[a.stat_stride] += nh_acc;
    }
}

// ---- k_regen: persistent waves with path regeneration (brute-force kernels) -------------------------------------------
// The uniform primitive loop of the brute-force kernels does not care which paths share a wave, so nothing forces the
// wavefront organisation on them: here a lane carries ONE path from the camera to its end in registers, and a lane whose
// path ended takes the next unstarted path of its wave (one ballot + mbcnt per trip: the wave hands out the homes of its
// share in order).  No path state in memory at all, no compaction, no barrier after the table staging, one launch per
// pass; what a path leaves behind is its 16-byte Lhome record, and the film gather reads those in the same fixed order as
// before, so the film does not change by a bit.  The grid is the number of waves the GPU holds at once (host:
// regen_grid); wave w of W takes the 64-home groups w, w + W, w + 2 W, ... of the pass, so every wave samples the whole
// film and the waves finish together (no queue in memory: same-word returning atomics cost 0.3 us each here).
// Statistics: segments and shadow rays are counted per trip (wave-uniform popcounts); the depth rows come from a
// histogram of the paths' final depths (one LDS atomic per finished path), live[d] = paths with more than d bounces.
#ifndef REGEN_WG
#define REGEN_WG 256
#endif
#ifndef REGEN_WAVES_PER_EU
#define REGEN_WAVES_PER_EU 4
#endif
template <int ACCEL>
__global__ __launch_bounds__(REGEN_WG, REGEN_WAVES_PER_EU) void k_regen(const RadArgs a) {
    static_assert(ACCEL == ACCEL_K_BRUTE || ACCEL == ACCEL_K_BRUTE_BIG, "path regeneration: brute-force kernels only");
    constexpr uint32_t W = REGEN_WG / 64;
    __shared__ uint32_t tab_lds[ACCEL == ACCEL_K_BRUTE ? TAB_DW : 1];
    __shared__ uint32_t hist[W][MAX_DEPTH_STATS + 2];  // per wave: paths that ended after d + 1 bounces
    const uint32_t tid = threadIdx.x, wid = tid >> 6, lane = tid & 63u;
    const LdsScene ls = NO_LDS_SCENE;
    const Tables tb = make_tables<ACCEL>(a.sc, ls, tab_lds);
    hist[wid][lane] = 0;  // MAX_DEPTH_STATS + 2 == 64
    if (ACCEL == ACCEL_K_BRUTE)
        fill_tables_lds(a.sc, tab_lds, REGEN_WG);  // ends with the only barrier of the kernel
    else
        __syncthreads();
    const uint32_t gw = (uint32_t)__builtin_amdgcn_readfirstlane((int)(blockIdx.x * W + wid)), GW = gridDim.x * W;
    const Rsrc r_L = make_rsrc(a.Lhome, a.cap * 16u);
    uint32_t q = 0;  // wave-uniform: homes of this wave's share handed out so far
    bool live = false;
    V3 o = {0, 0, 0}, d = {0, 0, 1}, thr = {0, 0, 0}, L = {0, 0, 0};
    float eta = 1.0f, prev_pdf = -1.0f, tmax = K_INF;
    uint32_t home = 0, ka = 0, kb = 0, depth = 0;
    uint32_t nseg_w = 0, nshd_w = 0;
#pragma unroll 1
    for (;;) {
        const unsigned long long need = __ballot(!live);
        if (need) {  // uniform: hand the next homes of the share to the idle lanes
            const uint32_t v = q + __builtin_amdgcn_mbcnt_hi((uint32_t)(need >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)need, 0u));
            const uint32_t h0 = ((v >> 6) * GW + gw) * 64u + (v & 63u);
            if (!live && h0 < a.n_paths) {
                uint32_t px, py;
                home = h0;
                path_key<true>(a, home, &ka, &kb, &px, &py);
                F4 uj = rng4(ka, kb, 0, a.seed);
                float fx = (float)px + uj.x, fy = (float)py + uj.y;
                camera_ray(a.cam, fx / (float)a.film_w, fy / (float)a.film_h, &o, &d, &tmax);
                thr = {1, 1, 1};
                L = {0, 0, 0};
                eta = 1.0f;
                prev_pdf = -1.0f;
                depth = 0;
                live = true;
            }
            q += (uint32_t)__popcll(need);
        }
        if (__ballot(live) == 0) break;  // the share is exhausted and every path of the wave has ended
        bool did_seg = false, did_shadow = false;
        if (live) {
            const bool survive =
                bounce_step<ACCEL>(a, tb, ls, r_L, depth, ka, kb, home, tmax, o, d, thr, L, eta, prev_pdf, did_seg, did_shadow);
            if (!survive) atomicAdd(&hist[wid][min(depth, (uint32_t)MAX_DEPTH_STATS)], 1u);
            live = survive;
            ++depth;
            tmax = K_INF;
        }
        nseg_w += (uint32_t)__popcll(__ballot(did_seg));
        nshd_w += (uint32_t)__popcll(__ballot(did_shadow));
    }
    // statistics row of this wave: live[d] = paths that entered depth d = paths that ended after more than d bounces
    unsigned long long *row = a.stats + gw;
    const size_t stride = a.stat_stride;
    if (lane == 0) {
        row[0] += nseg_w;
        row[stride] += nshd_w;
    }
    if (lane < min(a.max_depth, (uint32_t)MAX_DEPTH_STATS)) {
        uint32_t n = 0;
        for (uint32_t k = lane; k <= MAX_DEPTH_STATS; ++k) n += hist[wid][k];  // own wave's atomics: program order
        row[(2 + lane) * stride] += n;
    }
}

#endif  // PBRT_DIAG

// column sums of the per-segment statistics: out[k] += sum_seg stats[k][seg].  grid (rows, REDUCE_SLICES): every block sums
// one slice of a row and adds it to the row's total with one 64-bit atomic (out is zeroed by the caller; a single block
// per row took 55 us for the 32 Ki rows of a 16 Mi-path pass, on the host's critical path of every call)
#define REDUCE_SLICES 32
__global__ __launch_bounds__(256) void k_reduce_stats(const unsigned long long *stats, uint32_t nseg, size_t stride,
                                                      unsigned long long *out) {
    __shared__ unsigned long long part[256];
    const unsigned long long *row = stats + (size_t)blockIdx.x * stride;
    const uint32_t per = (nseg + gridDim.y - 1) / gridDim.y, lo = min(blockIdx.y * per, nseg), hi = min(lo + per, nseg);
    unsigned long long s = 0;
    for (uint32_t i = lo + threadIdx.x; i < hi; i += 256) s += row[i];
    part[threadIdx.x] = s;
    __syncthreads();
    for (uint32_t w = 128; w > 0; w >>= 1) {
        if (threadIdx.x < w) part[threadIdx.x] += part[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0 && part[0]) atomicAdd(&out[blockIdx.x], part[0]);
}

// The ultrasound acquisition's form (round 5): slice s of row r leaves its partial sum in out[r * REDUCE_SLICES + s] -- `out` is the
// context's pinned host page, written by the kernel itself, the host adds the slices -- and ZEROES the counters it has read, so that
// the next acquisition finds its rows clean.  No fill command in front of the counters and no copy command behind them: at one path
// per ray (USMain.py:36) those two were 9 of the 60 us an acquisition kept the device.
__global__ __launch_bounds__(256) void k_us_reduce_stats(unsigned long long *stats, uint32_t nseg, size_t stride,
                                                         unsigned long long *out) {
    __shared__ unsigned long long part[4];
    unsigned long long *row = stats + (size_t)blockIdx.x * stride;
    const uint32_t per = (nseg + gridDim.y - 1) / gridDim.y, lo = min(blockIdx.y * per, nseg), hi = min(lo + per, nseg);
    unsigned long long s = 0;
    for (uint32_t i = lo + threadIdx.x; i < hi; i += 256) {
        s += row[i];
        row[i] = 0ull;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    if ((threadIdx.x & 63u) == 0u) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) out[(size_t)blockIdx.x * gridDim.y + blockIdx.y] = part[0] + part[1] + part[2] + part[3];
}

// ---- film: deterministic gather of the pass's samples through the reconstruction filter -----------
struct FilmArgs {
    const float *Lhome;  // [cap] float4 records (r, g, b, 0)
    float *acc;          // [4][cw*ch] running sums (w*r, w*g, w*b, w)
    uint32_t cap;
    uint32_t cx0, cy0, cw, ch;     // crop
    uint32_t rx0, ry0, rw, rh;     // rendered region (crop + halo, clipped to the film)
    uint32_t npix_r, s_first, s_count, film_w, film_h, filter, seed;
    uint32_t tile_rows;            // pixel order of Lhome (region_index)
};

DEV float filter_1d(uint32_t f, float x) {
    if (f == PBRT_FILTER_TENT) return fmaxf(0.0f, 1.0f - fabsf(x));
    const float alpha = -2.0f;
    return fmaxf(0.0f, expf(alpha * x * x) - expf(alpha * 4.0f));
}

__global__ __launch_bounds__(256) void k_film_accum(const FilmArgs a) {
    const uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= a.cw * a.ch) return;
    const uint32_t iy = idx / a.cw, ix = idx - iy * a.cw;
    const int x = (int)(a.cx0 + ix), y = (int)(a.cy0 + iy);
    const uint32_t npx = a.cw * a.ch;
    float ar = a.acc[idx], ag = a.acc[npx + idx], ab = a.acc[2 * npx + idx], aw = a.acc[3 * npx + idx];
    const int R = a.filter == PBRT_FILTER_BOX ? 0 : (a.filter == PBRT_FILTER_TENT ? 1 : 2);
    const float ccx = (float)x + 0.5f, ccy = (float)y + 0.5f;
    for (uint32_t sl = 0; sl < a.s_count; ++sl) {
        const uint32_t s_idx = a.s_first + sl;
        const float4 *Ls = reinterpret_cast<const float4 *>(a.Lhome) + (size_t)sl * a.npix_r;
        for (int ny = y - R; ny <= y + R; ++ny)
            for (int nx = x - R; nx <= x + R; ++nx) {
                if (nx < 0 || ny < 0 || nx >= (int)a.film_w || ny >= (int)a.film_h) continue;
                float w = 1.0f;
                if (a.filter != PBRT_FILTER_BOX) {
                    F4 uj = rng4((uint32_t)ny * a.film_w + (uint32_t)nx, s_idx, 0, a.seed);
                    float px = (float)nx + uj.x, py = (float)ny + uj.y;
                    w = filter_1d(a.filter, ccx - px) * filter_1d(a.filter, ccy - py);
                }
                if (w > 0.0f) {
                    const uint32_t k = region_index((uint32_t)(nx - (int)a.rx0), (uint32_t)(ny - (int)a.ry0), a.rw, a.tile_rows);
                    const float4 Lk = Ls[k];
                    ar = fma_(w, Lk.x, ar);
                    ag = fma_(w, Lk.y, ag);
                    ab = fma_(w, Lk.z, ab);
                    aw += w;
                }
            }
    }
    a.acc[idx] = ar;
    a.acc[npx + idx] = ag;
    a.acc[2 * npx + idx] = ab;
    a.acc[3 * npx + idx] = aw;
}

// Tiled form for the tent / gaussian filters: a TILE x TILE pixel tile per workgroup (16; 8 when the crop has too
// few 16 x 16 tiles to fill the GPU -- the sample loop of a tile is sequential, and e.g. one rank's 512 x 64 band
// of an 8-GPU job at 2048 spp spent 4.9 of 13.5 ms in 128 workgroups).  Per sample index the
// workgroup evaluates the jitter hash ONCE per pixel of the haloed tile (not once per gathering
// neighbour: 1.3 instead of 9 / 25 hashes per thread) and stages (film x, film y, r, g, b) in LDS; every
// thread then gathers its (2R+1)^2 neighbourhood from LDS in the same fixed order as k_film_accum, so the
// sums are bit-identical.  Double-buffered: one barrier per sample.
#define FILM_RMAX 2
template <int FILM_TILE, int FILTER>  // the filter is a template parameter: radius and tap loops are compile-time
__global__ __launch_bounds__(FILM_TILE *FILM_TILE) void k_film_accum_tiled(const FilmArgs a) {
    constexpr int FILM_TW = FILM_TILE + 2 * FILM_RMAX;
    // LDS image of the haloed tile, one plane per component, rows padded to a stride that puts the rows a half-wave
    // reads at once (32 lanes: 2 rows of 16 / 4 rows of 8) on disjoint banks (AoS [entry][5] measured 39 % of the LDS
    // cycles in bank conflicts)
    constexpr int LS = FILM_TILE == 16 ? 48 : 40;
    static_assert(LS >= FILM_TW, "row stride covers the haloed row");
    __shared__ float tile[2][5][FILM_TW * LS];
    constexpr int R = FILTER == PBRT_FILTER_TENT ? 1 : 2;
    constexpr int TW = FILM_TILE + 2 * R;
    const uint32_t tid = threadIdx.y * FILM_TILE + threadIdx.x;
    const int tx0 = (int)(a.cx0 + blockIdx.x * FILM_TILE), ty0 = (int)(a.cy0 + blockIdx.y * FILM_TILE);  // film coords of the tile
    const int x = tx0 + (int)threadIdx.x, y = ty0 + (int)threadIdx.y;
    const bool inside = x < (int)(a.cx0 + a.cw) && y < (int)(a.cy0 + a.ch);
    const uint32_t npx = a.cw * a.ch;
    const uint32_t idx = inside ? (uint32_t)(y - (int)a.cy0) * a.cw + (uint32_t)(x - (int)a.cx0) : 0u;
    float ar = 0, ag = 0, ab = 0, aw = 0;
    if (inside) {
        ar = a.acc[idx];
        ag = a.acc[npx + idx];
        ab = a.acc[2 * npx + idx];
        aw = a.acc[3 * npx + idx];
    }
    const float ccx = (float)x + 0.5f, ccy = (float)y + 0.5f;
    // staging duty of this thread: haloed-tile entries tid, tid + NT, ... (NST of them); everything that does not
    // depend on the sample index is computed once, and the radiance record of sample sl + 1 is requested before
    // sample sl is staged and gathered, so the HBM / L2 round trip is off the per-sample critical path
    constexpr int NT = FILM_TILE * FILM_TILE, NST = (FILM_TW * FILM_TW + NT - 1) / NT;
    bool st_in[NST], st_valid[NST];
    uint32_t st_key[NST], st_k[NST];
    float st_x[NST], st_y[NST];
    int st_lds[NST];
#pragma unroll
    for (int j = 0; j < NST; ++j) {
        const int i = (int)tid + j * NT;
        const int nx = tx0 - R + i % TW, ny = ty0 - R + i / TW;
        st_lds[j] = (i / TW) * LS + i % TW;
        st_in[j] = i < TW * TW;
        st_valid[j] = st_in[j] && nx >= (int)a.rx0 && ny >= (int)a.ry0 && nx < (int)(a.rx0 + a.rw) && ny < (int)(a.ry0 + a.rh);
        st_key[j] = (uint32_t)ny * a.film_w + (uint32_t)nx;
        st_k[j] = st_valid[j] ? region_index((uint32_t)(nx - (int)a.rx0), (uint32_t)(ny - (int)a.ry0), a.rw, a.tile_rows) : 0u;
        st_x[j] = (float)nx;
        st_y[j] = (float)ny;
    }
    const float4 *Lrec = reinterpret_cast<const float4 *>(a.Lhome);
    const float4 zero4 = {0.0f, 0.0f, 0.0f, 0.0f};
    // radiance records of the next PF samples, requested PF iterations ahead: one iteration is shorter than an HBM / L2
    // round trip (with a distance of one the loop ran at 2.3 us per sample whatever the workgroup count)
#ifndef FILM_PF
#define FILM_PF 4
#endif
    constexpr int PF = FILM_PF;
    float4 Lq[PF][NST];
#pragma unroll
    for (int u = 0; u < PF; ++u)
#pragma unroll
        for (int j = 0; j < NST; ++j)
            Lq[u][j] = (st_valid[j] && (uint32_t)u < a.s_count) ? Lrec[(size_t)u * a.npix_r + st_k[j]] : zero4;
    for (uint32_t sl0 = 0; sl0 < a.s_count; sl0 += PF) {
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            const uint32_t sl = sl0 + (uint32_t)u;
            if (sl >= a.s_count) break;
            const uint32_t s_idx = a.s_first + sl;
            float(*T)[FILM_TW * LS] = tile[sl & 1];
#pragma unroll
            for (int j = 0; j < NST; ++j) {
                if (!st_in[j]) continue;
                const int i = st_lds[j];
                float px = 1e30f, py = 1e30f;  // outside the film / rendered region: weight 0
                if (st_valid[j]) {
                    F4 uj = rng4(st_key[j], s_idx, 0, a.seed);
                    px = st_x[j] + uj.x;
                    py = st_y[j] + uj.y;
                }
                T[0][i] = px;
                T[1][i] = py;
                T[2][i] = Lq[u][j].x;
                T[3][i] = Lq[u][j].y;
                T[4][i] = Lq[u][j].z;
            }
#pragma unroll
            for (int j = 0; j < NST; ++j)
                Lq[u][j] = (st_valid[j] && sl + PF < a.s_count) ? Lrec[(size_t)(sl + PF) * a.npix_r + st_k[j]] : zero4;
            __syncthreads();
            if (inside) {
                const int e0 = (int)threadIdx.y * LS + (int)threadIdx.x;
#pragma unroll
                for (int dy = 0; dy <= 2 * R; ++dy)
#pragma unroll
                    for (int dx = 0; dx <= 2 * R; ++dx) {
                        const int e = e0 + dy * LS + dx;
                        float w = filter_1d(FILTER, ccx - T[0][e]) * filter_1d(FILTER, ccy - T[1][e]);
                        if (w > 0.0f) {
                            ar = fma_(w, T[2][e], ar);
                            ag = fma_(w, T[3][e], ag);
                            ab = fma_(w, T[4][e], ab);
                            aw += w;
                        }
                    }
            }
        }
    }
    if (inside) {
        a.acc[idx] = ar;
        a.acc[npx + idx] = ag;
        a.acc[2 * npx + idx] = ab;
        a.acc[3 * npx + idx] = aw;
    }
}

__global__ __launch_bounds__(256) void k_film_resolve(const float *acc, float *out, uint32_t npx, uint32_t raw) {
    const uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= npx) return;
    float r = acc[idx], g = acc[npx + idx], b = acc[2 * npx + idx], w = acc[3 * npx + idx];
    if (raw) {
        out[4 * idx] = r;
        out[4 * idx + 1] = g;
        out[4 * idx + 2] = b;
        out[4 * idx + 3] = w;
    } else {
        float inv = w > 0.0f ? 1.0f / w : 0.0f;
        out[3 * idx] = r * inv;
        out[3 * idx + 1] = g * inv;
        out[3 * idx + 2] = b * inv;
    }
}

// upload of caller rays for Integrator.sample(): o,d [3][n] SoA + tmax -> state slots
// region: slots per live counter (a region, or the part of it one wave owns); n_cnt counters
__global__ __launch_bounds__(256) void k_init_rays(float *st, uint32_t *seg_cnt, uint32_t n_cnt, uint32_t region, uint32_t n,
                                                   const float *o, const float *d, const float *tmax) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_cnt) seg_cnt[i] = n > i * region ? min(n - i * region, region) : 0u;
    if (i >= n) return;
    float *s = st + (state_voff(i) >> 2);  // tiled SoA: row k at +64 * k floats
    s[0 * 64] = o[i];
    s[1 * 64] = o[n + i];
    s[2 * 64] = o[2 * n + i];
    s[3 * 64] = d[i];
    s[4 * 64] = d[n + i];
    s[5 * 64] = d[2 * n + i];
    s[6 * 64] = 1.0f;
    s[7 * 64] = 1.0f;
    s[8 * 64] = 1.0f;
    s[9 * 64] = 0.0f;
    s[10 * 64] = 0.0f;
    s[11 * 64] = 0.0f;
    s[12 * 64] = tmax[i];  // consumed as tmax by the first bounce
    s[13 * 64] = -1.0f;
    s[14 * 64] = __uint_as_float(i);
}

#ifdef PBRT_DIAG  // the fused BVH bounce kernels (k_bounce<.., BVH>) and their repack pass: diagnostic build only
// ---- repack (BVH kernels, depth >= 2): deal the live paths evenly to as few workgroups as fill the GPU ----------
// Per-wave compaction keeps every path inside the 512 slots its wave owns.  When few paths are left (open scenes:
// 22 % after two bounces of the ring scene, 1 % after five) every workgroup still stages the whole scene in LDS for
// a handful of paths and runs with mostly empty waves: 330-1500 ps per path instead of 160.  k_scan_owners turns the
// owners' live counts into exclusive offsets (one workgroup; <= a few 10^4 counts) and picks the new shape: the
// paths go to G = clamp(ceil(total / REGION), min_wg, n_regions) workgroups (min_wg = one per CU: fewer would leave
// CUs idle, more would stage the scene more often), `quota` paths per wave.  k_repack_copy moves path j of owner i
// to its new slot in the spare state buffer (120 B per live path, a fraction of a BVH bounce).  Slot order changes,
// results do not (the film gathers by `home`).
__global__ __launch_bounds__(1024) void k_scan_owners(const uint32_t *cnt, uint32_t n_own, uint32_t wreg, uint32_t owners_per_wg,
                                                     uint32_t min_wg, uint32_t *offs, uint32_t *cnt_new, uint32_t *quota_out) {
    __shared__ uint32_t part[1024];
    const uint32_t t = threadIdx.x, per = (n_own + 1023u) / 1024u;
    const uint32_t lo = min(t * per, n_own), hi = min(lo + per, n_own);
    uint32_t s = 0;
    for (uint32_t i = lo; i < hi; ++i) s += cnt[i];
    part[t] = s;
    __syncthreads();
    for (uint32_t w = 1; w < 1024; w <<= 1) {  // Hillis-Steele inclusive scan
        const uint32_t v = t >= w ? part[t - w] : 0u;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    const uint32_t total = part[1023];
    uint32_t run = part[t] - s;  // exclusive prefix of this thread's block of owners
    for (uint32_t i = lo; i < hi; ++i) {
        offs[i] = run;
        run += cnt[i];
    }
    const uint32_t n_wg = n_own / owners_per_wg, region = wreg * owners_per_wg;
    const uint32_t g = min(max((total + region - 1u) / region, min(min_wg, n_wg)), n_wg);
    const uint32_t act = g * owners_per_wg;                       // owners that receive paths
    const uint32_t quota = max((total + act - 1u) / act, 1u);     // <= wreg because g * region >= total
    if (t == 0) *quota_out = quota;
    for (uint32_t i = lo; i < hi; ++i) cnt_new[i] = total > i * quota ? min(total - i * quota, quota) : 0u;
}

// one workgroup per source owner
__global__ __launch_bounds__(256) void k_repack_copy(const float *__restrict__ in, float *__restrict__ out,
                                                     const uint32_t *__restrict__ cnt, const uint32_t *__restrict__ offs,
                                                     const uint32_t *__restrict__ quota_p, uint32_t n_own, uint32_t wreg) {
    const uint32_t own = blockIdx.x;
    if (own >= n_own) return;
    const uint32_t n = cnt[own], dense0 = offs[own], src0 = own * wreg, quota = *quota_p;
    for (uint32_t j = threadIdx.x; j < n; j += 256u) {
        const uint32_t dense = dense0 + j, o2 = dense / quota, p2 = dense - o2 * quota;
        const float *s = in + (state_voff(src0 + j) >> 2);
        float *d = out + (state_voff(o2 * wreg + p2) >> 2);
        float v[N_STATE];
#pragma unroll
        for (int k = 0; k < N_STATE; ++k) v[k] = s[k * 64];
#pragma unroll
        for (int k = 0; k < N_STATE; ++k) d[k * 64] = v[k];
    }
}
#endif  // PBRT_DIAG

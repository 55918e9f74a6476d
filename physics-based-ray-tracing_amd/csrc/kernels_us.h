// kernels_us.h -- ultrasound-mode wavefront kernels + the batched leaf operators (gfx950).
//
// Acquisition = UltraIntegrator.simulate_acquisition_parallel (CustomIntegrator.py:235-405) with
// P independent Monte-Carlo paths per (angle, element) primary ray.  Same segmented wavefront as
// radiance mode; state is 11 dwords per slot:
//   [0..2] origin [3..5] direction [6] amp [7] atten [8] tof [9] geo_len [10] home (uint32)
// home = ray_id * paths_this_pass + k_local, ray_id = angle * n_elements + element.
// Echoes go to channel_buf[(angle * n_elements + recv) * T + t_idx] (CustomIntegrator.py:351-354): summed per
// workgroup in a small LDS table keyed by that index, then one f32 global atomic per used bin.
#pragma once
#include "kernels_radiance.h"

#define N_USTATE 11

struct UsArgs {
    DevScene sc;
    pbrt_us_params p;
    const float *in;
    float *out;
    const uint32_t *seg_in;
    uint32_t *seg_out;
    unsigned long long *stats;
    float *channel;       // [n_angles * n_elements * time_samples]
    const float *tx;      // [n_angles * n_elements] transmission delays
    const float *dir0;    // [n_angles][3] primary directions (world)
    const float *elem_x;  // [n_elements]
    float tn[3];          // transducer normal (world)
    float am, ac, cos_min, katt, two_pi_f, inv_c;
    uint32_t cap, n_paths, depth, seed;
    uint32_t ppr_pass, path_first;  // paths per ray in this pass, global index of local path 0
    uint32_t lds_bytes;
    uint32_t stat_stride;
    uint32_t fuse;  // bounces a.depth .. max_depth - 1 in this launch (k_us_bounce<false>)
    uint32_t blk_mul;  // 0: workgroup b walks region b; else region (b * blk_mul) mod gridDim.x (coprime: a permutation) -- emitter rays
    FastDiv div_ppr, div_ne;  // exact home / ppr_pass and ray_id / n_elements without the 20-instruction variable udiv
    // First-bounce tables (k_us_first): the primary ray of an (angle, element) pair is deterministic
    // (CustomIntegrator.py:270-273), so all P paths of a ray share the first hit, and the occlusion of the ray to a
    // receive element, the directivity weight, the echo time and its carrier depend on (ray, element) only.  Computed
    // once per acquisition with the same statements; the first bounce of every path looks them up instead of walking
    // the scene twice and evaluating sqrtf / acosf / sinf / rintf.  Null: every path computes them (same result).
    const float4 *first_hit;  // [n_rays]: t, u, v, slot as bits (0xffffffff: miss)
    const float4 *first_rx;   // [n_rays][n_elements]: fd = directivity * w_o, carrier (sin(phase) or 1), channel index as
                              // bits (0xffffffff: occluded or outside the time window), unused
};

DEV float directivity_weight_i(V3 sec_dir, V3 tn, float am, float ac) {  // CustomIntegrator.py:289-304
    V3 w = -sec_dir;
    float dt = dot(tn, w);
    float alpha = fabsf(acosf(dt));
    float mid = (ac - alpha) / (ac - am);
    return alpha <= am ? 1.0f : (alpha <= ac ? mid : 0.0f);
}

// One ray of CustomEmitter.sample_ray (CustomEmmitter.py:51-107): element pick (:56-57), element centre and normal of the linear
// (:33-38) or convex (:41-47) array, jitter inside the element (:64-68), steering angle (:85-87), direction (:90), steering delay
// (:93-94), cosine weight (:97-98).  Shared by the leaf operator and by the acquisition's emitter-primary mode.
struct EmitRay {
    V3 o, d;
    float time, weight, pdf_pos;
};
// sin / cos of a steering or element angle: up to 45 degrees (every array the reference describes) the fixed polynomial the CPU
// restatement shares bit for bit (device_math.h sincos_pi4, ~1 ulp), beyond that the library routines.  Why it matters: the ray this
// feeds is the input of the path's first bounce, whose pdf 1 / (4 |wi . m|) is unbounded (and MULTIPLIES the amplitude, quirk B9) --
// an input that differs in its last bit between ocml and libm moves the few near-grazing echoes that dominate a channel buffer by
// per cents (measured: rel. L2 2.6e-2 against the CPU restatement at 32 768 paths per ray, 2e-7 with identical inputs).
DEV void emit_sincos(float x, float *s, float *c) {
    if (fabsf(x) <= K_PI_OVER_4) {
        sincos_pi4(x, s, c);
    } else {
        *s = sinf(x);
        *c = cosf(x);
    }
}
DEV EmitRay us_emitter_ray(const pbrt_us_emitter &e, float time, float s1, float s2x, float s2y, float s3) {
    const float N = (float)e.number_of_elements;
    const float total_rays = (float)(e.number_of_elements * e.number_of_rays_per_element);  // :17
    float idx = fminf(floorf(s1 * N), N - 1.0f);                                            // :56-57
    V3 c, nrm;
    if (e.radius == 0.0f) {                                                                 // :33-38
        float lo = -(N - 1.0f) / 2.0f * e.pitch, hi = (N - 1.0f) / 2.0f * e.pitch;
        float x = N > 1.0f ? fma_(idx, (hi - lo) / (N - 1.0f), lo) : lo;
        c = {x, 0.0f, 0.0f};
        nrm = {0.0f, 0.0f, 1.0f};
    } else {                                                                                // :41-47
        float span = e.opening_angle * (K_PI / 180.0f);
        float lo = -span / 2.0f, hi = span / 2.0f;
        float th = N > 1.0f ? fma_(idx, (hi - lo) / (N - 1.0f), lo) : lo;
        float sth, cth;
        emit_sincos(th, &sth, &cth);
        c = {e.radius * sth, 0.0f, e.radius * cth};
        nrm = normalize(v3(sth, 0.0f, cth));                                                // :49
    }
    float dx = (s2x - 0.5f) * e.element_width, dy = (s2y - 0.5f) * e.element_height;        // :64-65
    EmitRay r;
    r.o = c + v3(dx, dy, 0.0f);                                                             // :68
    r.pdf_pos = 1.0f / (N * e.element_width * e.element_height);                            // :77
    float pmin = e.steering_angle_min * (K_PI / 180.0f), pmax = e.steering_angle_max * (K_PI / 180.0f);
    float psi = fma_(s3, pmax - pmin, pmin);                                                // :85-87
    float spsi, cpsi;
    emit_sincos(psi, &spsi, &cpsi);
    r.d = {spsi, 0.0f, cpsi};                                                               // :90
    float delay = -(r.o.x * spsi) / e.speed_of_sound;                                       // :93
    r.time = time + delay;                                                                  // :94
    float fd = fmaxf(0.0f, dot(r.d, nrm));                                                  // :97
    r.weight = fd / total_rays;                                                             // :98
    return r;
}

// PBRT_US_PRIMARY_EMITTER (include/pbrt_hip.h, DESIGN D15): the primary ray of path k of the (angle, element) pair -- its own draw
// from CustomEmitter.sample_ray, the acquisition grid stratifying the element pick and the steering angle, RNG block 0x80000000.
// -> origin and direction in the world (the sensor transform, as :272-273 do for the integrator's own ray), the initial time of
// flight (the emitter's ray time: the element's steering delay) and the ray's weight, which multiplies every echo the path deposits
// (Mitsuba's render loop multiplies what the integrator returns by the ray weight; the path's own amplitude starts at 1, :276).
#define US_EMIT_BLOCK 0x80000000u
DEV EmitRay us_emitter_path_ray(const pbrt_us_params &p, uint32_t ray_id, uint32_t k, uint32_t ang, uint32_t el, uint32_t seed) {
    const F4 ue = rng4(ray_id, k, US_EMIT_BLOCK, seed);
    const float s1 = ((float)el + 0.5f) / (float)p.n_elements, s3 = ((float)ang + ue.z) / (float)p.n_angles;
    return us_emitter_ray(p.emitter, 0.0f, s1, ue.x, ue.y, s3);
}
DEV float us_emitter_primary(const pbrt_us_params &p, const float *M, uint32_t ray_id, uint32_t k, uint32_t ang, uint32_t el,
                             uint32_t seed, V3 *o, V3 *d, float *tof) {
    const EmitRay r = us_emitter_path_ray(p, ray_id, k, ang, el, seed);
    *o = xf_point(M, r.o);
    *d = normalize(xf_vec(M, r.d));
    *tof = r.time;
    return r.weight;
}
// waves per SIMD the register allocator aims for: the ultrasound bounce (GGX sampling, expf / sinf / acosf) needs
// about 95 VGPRs: 4 waves per SIMD run it without spills.  (While same-word global atomics dominated the kernel the
// spilling 8-wave build was the fastest -- 2.83 / 2.98 / 3.06 ms at 8 / 6 / 4 waves; with the echoes summed in LDS
// it is the other way round: 1.17 / 1.10 / 1.11 ms on Sphere_Box 5 x 64 x 65536.)
// Round 2 (-fno-slp-vectorize, uniforms in LDS): what counts is whole 512-thread workgroups per CU, 2 waves per SIMD each --
// 96 VGPRs (5 waves) still means two workgroups, 80 VGPRs with 4 of them spilled means three: config 3
// 13.2 / 13.2 / 11.9 / 14.4 ms at 4 / 5 / 6 / 8 waves (8: 30 spilled).
#define US_N_STATE 11  // origin, direction, amp, atten, tof, geo_len, home
DEV uint32_t us_state_voff(uint32_t slot, uint32_t rows = US_N_STATE) { return (slot >> 6) * (64u * rows * 4u) + (slot & 63u) * 4u; }
// bins of a workgroup's echo table (k_us_bounce, k_us_shade).  Round 5: 256 -> 2048.  The table lives as long as the workgroup,
// and a bin claimed by one channel index sends every other index that hashes there to the global atomic; 256 bins are enough while
// a workgroup's first-bounce echoes land on <= n_elements words (the table-driven first bounce), but with CustomEmitter primary
// rays (D15) every path has its own arrival time and 4096 first-bounce echoes spread over ~10^3 words that a hundred other
// workgroups of the same ray hit too: bounce 0 took 1.62 ms per 16 Mi-path pass against 0.41 ms for bounce 1 with as many paths
// (tools/us_per_depth.sh).  256 / 1024 / 2048 / 4096 bins: emitter rays 31.4 - 33.2 / 26.1 / 24.5 - 24.7 / 24.0 - 24.2 ms;
// config 3 11.2 - 11.4 / 10.8 - 11.0 / 10.7 - 10.9 / 10.8 ms; the ring phantom 36.1 - 36.5 / 35.8 - 36.0 / 35.1 - 35.5 ms
// (profiles/r05_us_echo_table_ab.txt).  A second, third, fourth bin tried before the global atomic (US_AGG_PROBES): flat.
#ifndef US_AGG_LOG2
#define US_AGG_LOG2 11
#endif
#define US_AGG_BINS (1u << US_AGG_LOG2)
#ifndef US_AGG_LOG2_EMIT
#define US_AGG_LOG2_EMIT 12   // the EMIT instances of k_us_bounce (16-segment regions: 21.0 - 21.5 -> 20.5 ms)
#endif
#ifndef US_AGG_PROBES
#define US_AGG_PROBES 1
#endif
#ifndef US_WAVES_PER_EU
#define US_WAVES_PER_EU 6
#endif
__host__ __device__ constexpr uint32_t us_waves_per_eu(int accel) {
    return (accel == ACCEL_K_BVH_GLOBAL || accel == ACCEL_K_BVH_LDS) ? seg_waves_per_eu(accel) : US_WAVES_PER_EU;
}

// live counters / statistics rows per region of the ultrasound kernels (per wave for the BVH variants)
__host__ __device__ constexpr uint32_t us_owners_per_region(int accel) {
    return rad_wave_private(accel) ? seg_threads(accel) / 64 : 1;
}

// EMIT: the instances of PBRT_US_PRIMARY_EMITTER -- the first bounce draws every path's primary ray from CustomEmitter.sample_ray,
// every echo is multiplied by the ray's weight; instances of their own so that the deterministic-ray kernels keep their register
// allocation (they sit at the 6-wave budget).
// Q: the behaviour switches (pbrt_us_params.quirks) as a compile-time constant, or US_Q_RUNTIME to read them from the arguments.
// The switches are launch-uniform, and read at run time the compiler hoists every `quirks & BIT` out of the bounce loop and keeps it
// as a 64-bit lane mask: 17 such masks were 34 of the 50 scalar registers this kernel spilled into VGPR lanes (round 4: 106
// v_readlane / v_writelane in 1 720 VALU instructions).  The library's default set (PBRT_USQ_REFERENCE, with and without the
// carrier) gets instances with the switches folded away; any other set runs the generic instance (same results:
// tests/test_gpu_ultrasound.py runs both on the same job; PBRT_US_GENERIC_KERNEL=1 forces the generic one).
// TAB: first-bounce tables present (1) / absent (0) / decided at run time (-1), for the same reason.
#define US_Q_RUNTIME 0xffffffffu
template <bool FIRST, int ACCEL, bool EMIT = false, uint32_t Q = US_Q_RUNTIME, int TAB = -1>
__global__ __launch_bounds__(seg_threads(ACCEL), us_waves_per_eu(ACCEL)) void k_us_bounce(const UsArgs a) {
    const uint32_t quirks = Q == US_Q_RUNTIME ? a.p.quirks : Q;
    const bool have_hit_tab = TAB < 0 ? a.first_hit != nullptr : TAB == 1;
    const bool have_rx_tab = TAB < 0 ? a.first_rx != nullptr : TAB == 1;
    constexpr uint32_t SEG = seg_threads(ACCEL);
    extern __shared__ __attribute__((aligned(16))) uint32_t dyn_lds[];
    __shared__ uint32_t wave_tot[2][SEG / 64];
    __shared__ uint32_t wave_seg[2][SEG / 64];
    constexpr uint32_t REGION = us_region_segs(ACCEL, EMIT) * SEG;
    constexpr bool WP = rad_wave_private(ACCEL);  // BVH scenes: per-wave compaction, no barrier per chunk (see k_bounce)
    constexpr uint32_t W = SEG / 64, WREG = REGION / W, CH = WP ? 64u : SEG;
    // seg: region index (see k_bounce).  Emitter rays: consecutive workgroups take regions a stride apart, so that the workgroups that
    // run -- and flush their echo tables -- side by side do not belong to the same ray (the six regions of a ray's paths land on the
    // same ~10^3 channel words): 20.3 -> 18.4 ms.  Any stride of a ray or more does (7 .. 1229 of 2048 regions: 18.5 - 19.0 ms); the
    // host takes regions / n_angles.  With the integrator's own rays (table-driven first bounce, <= 64 words per ray) it changes nothing
    const uint32_t seg = a.blk_mul ? (uint32_t)(((unsigned long long)blockIdx.x * a.blk_mul) % gridDim.x) : blockIdx.x;
    const uint32_t tid = threadIdx.x;
    const uint32_t lane_c = WP ? (tid & 63u) : tid;
    const uint32_t own = WP ? seg * W + (tid >> 6) : seg;
    const uint32_t base = WP ? seg * REGION + (tid >> 6) * WREG : seg * REGION;
    uint32_t cnt_in = FIRST ? (a.n_paths > base ? min(a.n_paths - base, WP ? WREG : REGION) : 0u) : a.seg_in[own];
    if (WP) {
        cnt_in = (uint32_t)__builtin_amdgcn_readfirstlane((int)cnt_in);
        uint32_t c = 0;
        if ((tid & 63u) < W) {
            const uint32_t b2 = seg * REGION + (tid & 63u) * WREG;
            c = FIRST ? (a.n_paths > b2 ? 1u : 0u) : a.seg_in[seg * W + (tid & 63u)];
        }
        if (__ballot(c != 0) == 0) {  // no wave of the workgroup has work (same answer in every wave)
            if ((tid & 63u) == 0) a.seg_out[own] = 0;
            return;
        }
    } else if (cnt_in == 0) {
        if (tid == 0) a.seg_out[seg] = 0;
        return;
    }
    BVH_STACK_LDS(ACCEL, SEG);
    LdsScene ls = {NO_TREE_LDS, MAKE_BVH_STACK(bvh_stk_lds, SEG)};
    if (ACCEL == ACCEL_K_BVH_LDS) ls.tree = stage_tree_lds(a.sc, dyn_lds);
    __shared__ uint32_t tab_lds[ACCEL == ACCEL_K_BRUTE ? TAB_DW : 1];
    const Tables tb = make_tables<ACCEL>(a.sc, ls, tab_lds);
    // Echo aggregation: the paths of a workgroup belong to few (angle, element) rays -- at the first bounce to ONE
    // ray whose paths all hit the same point, so 4096 echoes land on <= n_elements channel-buffer words.  Global
    // float atomics on the same word serialise in L2 (measured: 64 % of the kernel).  Echoes are therefore summed
    // in a small LDS table keyed by the channel index (ds_cmpst claims a bin, ds_add_f32 adds); a bin owned by
    // another index falls back to the global atomic; one global atomic per used bin when the workgroup is done.
    constexpr uint32_t AGG_LOG2 = EMIT ? US_AGG_LOG2_EMIT : US_AGG_LOG2, AGG_BINS = 1u << AGG_LOG2;
    __shared__ uint32_t agg_idx[AGG_BINS];
    __shared__ float agg_sum[AGG_BINS];
    for (uint32_t t = threadIdx.x; t < AGG_BINS; t += blockDim.x) {
        agg_idx[t] = 0xffffffffu;
        agg_sum[t] = 0.0f;
    }
    // the launch-uniform floats of the bounce are read from LDS (broadcast ds_read) instead of living in SGPRs: the kernel wants
    // more scalars than the 102 it has and spills them into VGPR lanes (v_writelane / v_readlane).  Staging these 23 took the
    // SGPR spills 105 -> 73 and config 3 14.16 -> 13.85 ms; staging the integer uniforms as well (64 spills) was slower, 13.92 ms.
    __shared__ float uni[24];
    if (threadIdx.x < 12) uni[threadIdx.x] = a.p.sensor_to_world[threadIdx.x];
    if (threadIdx.x == 12) {
        uni[12] = a.tn[0]; uni[13] = a.tn[1]; uni[14] = a.tn[2]; uni[15] = a.am; uni[16] = a.ac; uni[17] = a.cos_min;
        uni[18] = a.katt; uni[19] = a.two_pi_f; uni[20] = a.inv_c; uni[21] = a.p.fs; uni[22] = a.p.max_path_len;
    }
#define U_M uni
#define U_TN(k) uni[12 + (k)]
#define U_AM uni[15]
#define U_AC uni[16]
#define U_COSMIN uni[17]
#define U_KATT uni[18]
#define U_2PIF uni[19]
#define U_INVC uni[20]
#define U_FS uni[21]
#define U_MAXLEN uni[22]
    if (ACCEL == ACCEL_K_BRUTE)
        fill_tables_lds(a.sc, tab_lds, blockDim.x);  // ends with the barrier that also publishes the empty bins
    else
        __syncthreads();
    const uint32_t cap = a.cap;
    const uint32_t NE = a.p.n_elements, T = a.p.time_samples;
    // ALL bounces of a pass run in ONE launch (a.fuse; the FIRST kernel goes on with the later bounces): compaction is
    // local to the region (to the wave for BVH scenes), so the owner carries its survivors from bounce to bounce on its
    // own, ping-ponging between the two state buffers -- no grid-wide barrier per bounce, the survivors are re-read
    // while they are still in L2, and no launches for the bounces that find nothing alive (ultrasound paths die
    // fast: Sphere_Box has 20 % of them left after the first bounce and none after the second; the 8 empty launches up
    // to max_depth cost 5 us each per pass).  Config 3: 10.1 ms with one launch per bounce, 9.1 with bounces >= 1
    // fused, 8.6 with all of them.
    // path state: tiles of 64 slots x 11 rows like the radiance kernels' (kernels_radiance.h state_voff), read and written through
    // buffer descriptors: the row offset k * 256 is an immediate of the instruction, no 64-bit address arithmetic and no
    // pointer pair per array in SGPRs (this kernel spills scalars)
    // (the EMIT instances carry a twelfth row: the weight of the path's primary ray, a factor of every echo it deposits)
    constexpr uint32_t ROWS = EMIT ? US_N_STATE + 1u : US_N_STATE;
    Rsrc r_in = make_rsrc(a.in, cap * (ROWS * 4u)), r_out = make_rsrc(a.out, cap * (ROWS * 4u));
    uint32_t depth = a.depth;
    uint32_t out_off, ns_acc;
    for (;;) {  // bounce loop: a single trip unless a.fuse
    const bool first = FIRST && depth == 0;  // FIRST kernels continue with the later bounces when a.fuse
    out_off = 0;
    ns_acc = 0;
    for (uint32_t it0 = 0; it0 < cnt_in; it0 += CH) {
    const uint32_t buf = (it0 / SEG) & 1u;
    const bool alive = it0 + lane_c < cnt_in;
    const uint32_t slot = base + it0 + lane_c;
    bool survive = false, did_seg = false;
    V3 o, d;
    float amp, atten, tof, geo_len, w_ray = 1.0f;
    uint32_t home = slot;
    if (alive) {
#ifdef PBRT_PROBE_EXTRA_VALU  // diagnostic builds only (see k_bounce)
        {
            float probe = __uint_as_float(slot);
#pragma unroll
            for (int kk = 0; kk < PBRT_PROBE_EXTRA_VALU; ++kk) asm volatile("v_add_f32 %0, %0, %0" : "+v"(probe));
            if (probe == 12345.678f) home = 0;  // never true; keeps the chain alive
        }
#endif
        uint32_t ray_id, k;
        if (first) {
            ray_id = udiv_fast(home, a.div_ppr);
            k = a.path_first + (home - ray_id * a.ppr_pass);
            const uint32_t ang = udiv_fast(ray_id, a.div_ne), el = ray_id - ang * NE;
            o = xf_point(U_M, v3(a.elem_x[el], 0.0f, 0.0f));           // :270,273
            d = v3(a.dir0[3 * ang], a.dir0[3 * ang + 1], a.dir0[3 * ang + 2]);         // :271,273
            amp = 1.0f;
            atten = 1.0f;
            tof = 0.0f;
            geo_len = 0.0f;                                                            // :276-279
            if (EMIT) w_ray = us_emitter_primary(a.p, U_M, ray_id, k, ang, el, a.seed, &o, &d, &tof);  // (a.tx is all zero then)
        } else {
            const uint32_t v4 = us_state_voff(slot, ROWS);
            constexpr uint32_t row = STATE_ROW_BYTES;
            if (EMIT) w_ray = bld(r_in, v4 + 11 * row, 0);
            o = {bld(r_in, v4 + 0 * row, 0), bld(r_in, v4 + 1 * row, 0), bld(r_in, v4 + 2 * row, 0)};
            d = {bld(r_in, v4 + 3 * row, 0), bld(r_in, v4 + 4 * row, 0), bld(r_in, v4 + 5 * row, 0)};
            amp = bld(r_in, v4 + 6 * row, 0);
            atten = bld(r_in, v4 + 7 * row, 0);
            tof = bld(r_in, v4 + 8 * row, 0);
            geo_len = bld(r_in, v4 + 9 * row, 0);
            home = __float_as_uint(bld(r_in, v4 + 10 * row, 0));
            ray_id = udiv_fast(home, a.div_ppr);
            k = a.path_first + (home - ray_id * a.ppr_pass);
        }
        const uint32_t ang = udiv_fast(ray_id, a.div_ne);
        const V3 tn = {U_TN(0), U_TN(1), U_TN(2)};
        Hit h;
        bool hit;
        if (first && have_hit_tab) {  // shared first hit of the ray (k_us_first)
            const float4 r = a.first_hit[ray_id];
            h.t = r.x;
            h.u = r.y;
            h.v = r.z;
            h.slot = __float_as_uint(r.w);
            h.prim = 0;
            hit = h.slot != 0xffffffffu;
        } else {
            hit = scene_intersect<ACCEL, false>(a.sc, ls, o, d, K_INF, &h);             // :309-312
        }
        if (hit) {
            did_seg = true;
            const pbrt_prim &P = tb.prims_by_slot[h.slot];
            SI si = make_si<ACCEL != ACCEL_K_BRUTE>(P, o, d, h.t, h.u, h.v, a.sc.vnormals, h.slot);
            const float distance = h.t;                                                // :314
            geo_len += distance;                                                       // :315
            const bool no_acc = (quirks & PBRT_USQ_NO_TOF_ACCUM) != 0;
            if (!no_acc) tof += distance * U_INVC;                                    // :316
            // B1 (Dr.Jit variant): the draws are constants of the traced loop body -- every bounce reuses block 0
            const uint32_t block = (quirks & PBRT_USQ_FROZEN_DRAWS) ? 0u : depth;
            F4 u = rng4(ray_id, k, block, a.seed);
            uint32_t recv = min((uint32_t)(u.x * (float)NE), NE - 1);                  // :319
            const bool tab = first && have_rx_tab;  // (ray, receive element) record of k_us_first
            float4 rx = {0.0f, 0.0f, 0.0f, 0.0f};
            V3 sec_dir = {0.0f, 0.0f, 0.0f};
            bool visible = false;
            float total_time = 0.0f, phase = 0.0f;
            if (tab) {
                rx = a.first_rx[(size_t)ray_id * NE + recv];
            } else {
                V3 target = xf_point(U_M, v3(a.elem_x[recv], 0.0f, 0.0f)); // :320-321
                V3 tv = target - si.p;
                float dist_recv = sqrtf(dot(tv, tv));
                sec_dir = tv * (1.0f / dist_recv);                                     // :322
                Hit hs;
                visible = !scene_intersect<ACCEL, true>(a.sc, ls, offset_origin(si.p, si.n, sec_dir), sec_dir, K_INF,
                                                        &hs);                           // :324-325
                float tof_hit = no_acc ? tof + distance * U_INVC : tof;
                total_time = a.tx[ray_id] + tof_hit + dist_recv * U_INVC;             // :329
                phase = U_2PIF * total_time;                                       // :330
            }
            atten *= expf(U_KATT * distance / 8.686f);                                 // :328
            const pbrt_material M = tb.mats[P.material];
            // si.sh_frame as Mitsuba builds it (from dp_du, not coordinate_system(n)): si.wi, si.to_local, si.to_world
            const Frame fr = make_sh_frame(si.ns, si_dp_du<ACCEL != ACCEL_K_BRUTE>(P, si));
            V3 wi = to_local(fr, -d);                                                  // si.wi (CustomBSDF.py:90)
            float a_resp, bpdf;
            V3 new_dir;
            bool ok = true;
            if (M.type == PBRT_MAT_ULTRA) {
                // intent arithmetic (no diagonal broadcast, A2 off): the micro-normal's second variate comes from a second
                // block of the path's stream -- u.w also decides the roulette below and must not steer the facet as well
                const float s1b = (quirks & PBRT_USQ_DIAG_SAMPLE) ? u.w : rng4(ray_id, k, block | 0x40000000u, a.seed).x;
                UltraOut uo = ultra_core(M, quirks, wi, si.n, si.ns, u.y, u.z, s1b); // :338
                a_resp = uo.amp;
                bpdf = uo.pdf;
                new_dir = to_world(fr, to_local(fr, uo.chosen));                       // CustomBSDF.py:165 + :358
            } else {
                BSample bs = bsdf_sample(M, quirks, wi, si.n, si.ns, fr, u.y, u.z, u.w);
                ok = bs.valid;
                a_resp = bs.weight.x;
                bpdf = bs.pdf;
                new_dir = to_world(fr, bs.wo);
            }
            if (ok) {
                float cos_theta = dot(si.ns, -d);                                      // :340 (si.sh_frame.n)
                amp *= a_resp * cos_theta * fmaxf(bpdf, 1e-6f);                        // :341
                float fd = 0.0f, carrier = 0.0f;
                uint32_t ci = 0xffffffffu;
                if (tab) {
                    fd = rx.x;
                    carrier = rx.y;
                    ci = __float_as_uint(rx.z);
                } else {
                    float tf = rintf(total_time * U_FS);                             // :351-352
                    if (quirks & PBRT_USQ_CLAMP_TIME) tf = fminf(fmaxf(tf, 0.0f), (float)(T - 1));
                    if (tf >= 0.0f && tf < (float)T && visible) {                      // :353
                        ci = (ang * NE + recv) * T + (uint32_t)tf;                     // :354 (host checks it fits 32 bits)
                        // the echo's weight and carrier only where an echo is deposited: acosf and sinf are a tenth of the
                        // bounce, and e.g. every second bounce of the Sphere_Box phantom runs inside the sphere, unseen
                        float w_o = dot(d, si.ns) / (float)(a.p.n_angles * NE);        // :286-287,345 (si.sh_frame.n)
#if defined(PBRT_ABLATE_US_DEPOSIT) && PBRT_ABLATE_US_DEPOSIT == 2
                        fd = w_o;
                        carrier = phase * 1e-9f;
#else
                        fd = directivity_weight_i(sec_dir, tn, U_AM, U_AC) * w_o;      // :345
                        // f-3 pulse model: plain amplitude here, the carrier is applied by k_apply_pulse afterwards
                        carrier = (quirks & PBRT_USQ_NO_CARRIER) ? 1.0f : sinf(phase);
#endif
                    }
                }
                float pressure = atten * amp * fd * carrier;                           // :348
#ifdef PBRT_ABLATE_US_DEPOSIT  // timing probe only (wrong channel buffer): no echo is deposited; =2: the deposit stays, its weight is 1
                if (PBRT_ABLATE_US_DEPOSIT == 1) ci = pressure != 12345.0f ? 0xffffffffu : ci;
#endif
                if (ci != 0xffffffffu) {
                    if (EMIT) pressure *= w_ray;  // the weight of the path's primary ray (DESIGN D15)
#ifdef PBRT_ABLATE_US_AGG  // diagnostic builds only
                    atomicAdd(&a.channel[ci], pressure);
#else
                    uint32_t bin = (ci * 2654435761u) >> (32 - AGG_LOG2);
                    bool mine = false;
#pragma unroll
                    for (uint32_t pr = 0; pr < US_AGG_PROBES; ++pr) {  // (a bin owned by another index: the next one, US_AGG_PROBES tries)
                        if (!mine) {
                            const uint32_t owner = atomicCAS(&agg_idx[bin], 0xffffffffu, ci);
                            mine = owner == 0xffffffffu || owner == ci;
                            if (!mine) bin = (bin + 1u) & (AGG_BINS - 1u);
                        }
                    }
                    if (mine)
                        __hip_atomic_fetch_add(&agg_sum[bin], pressure, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    else
                        atomicAdd(&a.channel[ci], pressure);
#endif
                }
                d = normalize(new_dir);                                                // :358-359
                o = offset_origin(si.p, si.n, d);
                bool surv;
                if (quirks & PBRT_USQ_SIGNED_RR) {                                 // Dr.Jit variant :219-224
                    const float rr_prob = fminf(atten * amp, 1.0f);
                    surv = u.w < rr_prob;
                    atten = surv ? atten / rr_prob : 0.0f;
                } else {
                    const float rr_prob = fminf(fabsf(atten * amp), 1.0f);             // :364
                    surv = !(u.w > rr_prob);                                           // :365-366
                    atten /= rr_prob;                                                  // :367
                }
                bool within = dot(d, tn) >= U_COSMIN;                                 // :371
                survive = within && (geo_len < U_MAXLEN) && (depth + 1 < a.p.max_depth) && surv;  // :372-376
            }
        }
    }
    const uint32_t wid = tid >> 6;
    const unsigned long long bal = __ballot(survive);
    const uint32_t prefix = __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u));
    const unsigned long long bseg = __ballot(did_seg);
    uint32_t off = 0, total = 0;
    if (WP) {  // the wave packs its own survivors behind its own cursor
        total = (uint32_t)__popcll(bal);
        ns_acc += (uint32_t)__popcll(bseg);
    } else {
        if ((tid & 63) == 0) {
            wave_tot[buf][wid] = (uint32_t)__popcll(bal);
            wave_seg[buf][wid] = (uint32_t)__popcll(bseg);
        }
        __syncthreads();
#pragma unroll
        for (uint32_t w = 0; w < SEG / 64; ++w) {
            uint32_t t = wave_tot[buf][w];
            off += (w < wid) ? t : 0u;
            total += t;
        }
    }
    if (survive) {
        const uint32_t v4 = us_state_voff(base + out_off + off + prefix, ROWS);
        constexpr uint32_t row = STATE_ROW_BYTES;
        if (EMIT) bst(r_out, v4 + 11 * row, 0, w_ray);
        bst(r_out, v4 + 0 * row, 0, o.x);
        bst(r_out, v4 + 1 * row, 0, o.y);
        bst(r_out, v4 + 2 * row, 0, o.z);
        bst(r_out, v4 + 3 * row, 0, d.x);
        bst(r_out, v4 + 4 * row, 0, d.y);
        bst(r_out, v4 + 5 * row, 0, d.z);
        bst(r_out, v4 + 6 * row, 0, amp);
        bst(r_out, v4 + 7 * row, 0, atten);
        bst(r_out, v4 + 8 * row, 0, tof);
        bst(r_out, v4 + 9 * row, 0, geo_len);
        bst(r_out, v4 + 10 * row, 0, __uint_as_float(home));
    }
    out_off += total;
    if (!WP && tid == 0)
        for (uint32_t w = 0; w < SEG / 64; ++w) ns_acc += wave_seg[buf][w];
    }  // chunk loop
    if (WP ? (tid & 63u) == 0 : tid == 0) {
        unsigned long long *row = a.stats + own;  // per-region / per-wave rows, see k_bounce
        const size_t stride = a.stat_stride;
        row[0] += ns_acc;
        row[stride] += ns_acc;  // one occlusion ray per shaded segment
        row[(2 + min(depth, (uint32_t)MAX_DEPTH_STATS - 1)) * stride] += cnt_in;
    }
    if (!a.fuse || out_off == 0 || depth + 1 >= a.p.max_depth) break;  // uniform over the owner
    // the survivors just written are the next bounce's input: stores complete (release at workgroup scope; the waves
    // of a workgroup share the CU's vector L1, so no invalidate), then everybody has finished reading the old input
    __threadfence_block();
    if (!WP) __syncthreads();
    const Rsrc nxt_in = r_out;
    r_out = r_in;
    r_in = nxt_in;
    cnt_in = out_off;
    ++depth;
    }  // bounce loop
    __syncthreads();  // all echoes of the workgroup are in the bins
    for (uint32_t t = tid; t < AGG_BINS; t += SEG) {
        const uint32_t ci = agg_idx[t];
        if (ci != 0xffffffffu) atomicAdd(&a.channel[ci], agg_sum[t]);
    }
    if (WP ? (tid & 63u) == 0 : tid == 0) a.seg_out[own] = out_off;
}
#undef U_M
#undef U_TN
#undef U_AM
#undef U_AC
#undef U_COSMIN
#undef U_KATT
#undef U_2PIF
#undef U_INVC
#undef U_FS
#undef U_MAXLEN

// PBRT_US_PRIMARY_EMITTER, brute-force scenes, PBRT_US_EMIT_FUSED=0: the primary rays of a pass written into the (twelve-row) path
// state, so that k_us_bounce<false, ., EMIT> can walk every bounce from depth 0.  The shipped path draws the ray inside the
// first-bounce instance instead (k_us_bounce<true, ., EMIT>: no state traffic at depth 0, at the price of 17 spilled VGPRs at its
// 80-register budget -- the emitter's sincos on top of the table-less first bounce): level while the echo deposit of bounce 0 was
// the bound (28.7 - 34.6 against 30.2 - 33.7 ms), 7 % faster since (17.1 against 18.4 ms, profiles/r05_us_emitter_ab.txt).
__global__ __launch_bounds__(256) void k_us_emit_init(const UsArgs a, uint32_t region, uint32_t n_regions) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_regions) a.seg_out[i] = a.n_paths > i * region ? min(a.n_paths - i * region, region) : 0u;
    if (i >= a.n_paths) return;
    const uint32_t ray_id = udiv_fast(i, a.div_ppr);
    const uint32_t ang = udiv_fast(ray_id, a.div_ne), el = ray_id - ang * a.p.n_elements;
    V3 o, d;
    float tof;
    const float w_ray = us_emitter_primary(a.p, a.p.sensor_to_world, ray_id, a.path_first + (i - ray_id * a.ppr_pass), ang, el, a.seed, &o, &d, &tof);
    float *st = a.out + us_state_voff(i, US_N_STATE + 1u) / 4u;
    constexpr uint32_t row = STATE_ROW_BYTES / 4u;
    st[0 * row] = o.x;
    st[1 * row] = o.y;
    st[2 * row] = o.z;
    st[3 * row] = d.x;
    st[4 * row] = d.y;
    st[5 * row] = d.z;
    st[6 * row] = 1.0f;   // amp   :276
    st[7 * row] = 1.0f;   // atten :277
    st[8 * row] = tof;    // the ray's emission time (CustomEmmitter.py:93-94)
    st[9 * row] = 0.0f;   // geo_len
    st[10 * row] = __uint_as_float(i);
    st[11 * row] = w_ray;
}

// First-bounce tables, one thread per (ray, receive element): the primary ray, its closest hit, and the occlusion test
// towards the element -- the statements of k_us_bounce<FIRST> up to `visible`, once instead of once per path.
// ACCEL: ACCEL_K_BRUTE_BIG or ACCEL_K_BVH_GLOBAL (no LDS image needed for n_rays * n_elements rays; same primitive
// order / same tree, so the same hit).
template <int ACCEL>
__global__ __launch_bounds__(256) void k_us_first(const UsArgs a, uint32_t n_rays, float4 *first_hit, float4 *first_rx) {
    const uint32_t NE = a.p.n_elements;
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_rays * NE) return;
    const uint32_t ray_id = i / NE, recv = i - ray_id * NE;
    const uint32_t ang = ray_id / NE, el = ray_id - ang * NE;
    const V3 o = xf_point(a.p.sensor_to_world, v3(a.elem_x[el], 0.0f, 0.0f));            // :270,273
    const V3 d = v3(a.dir0[3 * ang], a.dir0[3 * ang + 1], a.dir0[3 * ang + 2]);          // :271,273
    BVH_STACK_LDS(ACCEL, 256);
    LdsScene ls = {NO_TREE_LDS, MAKE_BVH_STACK(bvh_stk_lds, 256)};
    Hit h;
    const bool hit = scene_intersect<ACCEL, false>(a.sc, ls, o, d, K_INF, &h);
    if (recv == 0) {
        float4 r = {0.0f, 0.0f, 0.0f, __uint_as_float(0xffffffffu)};
        if (hit) r = {h.t, h.u, h.v, __uint_as_float(h.slot)};
        first_hit[ray_id] = r;
    }
    float4 rx = {0.0f, 0.0f, __uint_as_float(0xffffffffu), 0.0f};
    if (hit) {
        const uint32_t T = a.p.time_samples;
        const V3 tn = {a.tn[0], a.tn[1], a.tn[2]};
        const pbrt_prim &P = a.sc.prims[h.slot];
        SI si = make_si(P, o, d, h.t, h.u, h.v, a.sc.vnormals, h.slot);
        const float distance = h.t;                                                      // :314
        const bool no_acc = (a.p.quirks & PBRT_USQ_NO_TOF_ACCUM) != 0;
        float tof = 0.0f;                                                                // :278
        if (!no_acc) tof += distance * a.inv_c;                                          // :316
        V3 target = xf_point(a.p.sensor_to_world, v3(a.elem_x[recv], 0.0f, 0.0f));       // :320-321
        V3 tv = target - si.p;
        float dist_recv = sqrtf(dot(tv, tv));
        V3 sec_dir = tv * (1.0f / dist_recv);                                            // :322
        Hit hs;
        const bool visible = !scene_intersect<ACCEL, true>(a.sc, ls, offset_origin(si.p, si.n, sec_dir), sec_dir, K_INF, &hs);
        float tof_hit = no_acc ? tof + distance * a.inv_c : tof;
        float total_time = a.tx[ray_id] + tof_hit + dist_recv * a.inv_c;                 // :329
        float phase = a.two_pi_f * total_time;                                           // :330
        float w_o = dot(d, si.ns) / (float)(a.p.n_angles * NE);                          // :286-287,345 (si.sh_frame.n)
        rx.x = directivity_weight_i(sec_dir, tn, a.am, a.ac) * w_o;                      // :345
        rx.y = (a.p.quirks & PBRT_USQ_NO_CARRIER) ? 1.0f : sinf(phase);
        float tf = rintf(total_time * a.p.fs);                                           // :351-352
        if (a.p.quirks & PBRT_USQ_CLAMP_TIME) tf = fminf(fmaxf(tf, 0.0f), (float)(T - 1));
        if (tf >= 0.0f && tf < (float)T && visible) rx.z = __uint_as_float((ang * NE + recv) * T + (uint32_t)tf);  // :353-354
    }
    first_rx[i] = rx;
}

__global__ __launch_bounds__(256) void k_scale(float *buf, size_t n, float s) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) buf[i] *= s;
}

// ================================================================================================
// leaf operators: one thread per element, SoA in / SoA out
// ================================================================================================
template <int ACCEL>
__global__ __launch_bounds__(256) void k_ray_intersect(DevScene sc, uint32_t n, const float *o, const float *d,
                                                       const float *tmax, float *t, uint32_t *prim, float *u, float *v) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    BVH_STACK_LDS(ACCEL, 256);
    LdsScene ls = {NO_TREE_LDS, MAKE_BVH_STACK(bvh_stk_lds, 256)};
    Hit h;
    bool f = scene_intersect<ACCEL, false>(sc, ls, v3(o[i], o[n + i], o[2 * n + i]), v3(d[i], d[n + i], d[2 * n + i]),
                                           tmax[i], &h);
    t[i] = f ? h.t : K_INF;
    prim[i] = f ? h.prim : 0xffffffffu;
    u[i] = f ? h.u : 0.0f;
    v[i] = f ? h.v : 0.0f;
}

template <int ACCEL>
__global__ __launch_bounds__(256) void k_ray_test(DevScene sc, uint32_t n, const float *o, const float *d,
                                                  const float *tmax, uint8_t *hit) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    BVH_STACK_LDS(ACCEL, 256);
    LdsScene ls = {NO_TREE_LDS, MAKE_BVH_STACK(bvh_stk_lds, 256)};
    Hit h;
    hit[i] = scene_intersect<ACCEL, true>(sc, ls, v3(o[i], o[n + i], o[2 * n + i]), v3(d[i], d[n + i], d[2 * n + i]),
                                          tmax[i], &h)
                 ? 1
                 : 0;
}

__global__ __launch_bounds__(256) void k_bsdf_sample(pbrt_material m, uint32_t quirks, uint32_t n, const float *wi,
                                                     const float *ng, const float *ns, const float *shs, const float *s1,
                                                     const float *s2, float *wo, float *pdf, float *weight, uint32_t *sampled) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    V3 g = ng ? v3(ng[i], ng[n + i], ng[2 * n + i]) : v3(0, 0, 1);
    V3 s = ns ? v3(ns[i], ns[n + i], ns[2 * n + i]) : v3(0, 0, 1);
    // si.sh_frame: its tangent when the caller has one (dp_du / sh_frame.s), else coordinate_system(n_sh)
    const Frame shf = shs ? make_sh_frame(s, v3(shs[i], shs[n + i], shs[2 * n + i])) : make_frame(s);
    BSample b = bsdf_sample(m, quirks, v3(wi[i], wi[n + i], wi[2 * n + i]), g, s, shf, s1[i], s2[i], s2[n + i]);
    wo[i] = b.wo.x;
    wo[n + i] = b.wo.y;
    wo[2 * n + i] = b.wo.z;
    pdf[i] = b.pdf;
    weight[i] = b.weight.x;
    weight[n + i] = b.weight.y;
    weight[2 * n + i] = b.weight.z;
    sampled[i] = b.valid ? b.lobe : 0xffffffffu;
}

__global__ __launch_bounds__(256) void k_bsdf_eval_pdf(pbrt_material m, uint32_t n, const float *wi, const float *wo,
                                                       float *f, float *pdf) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    V3 fv;
    float p;
    bsdf_eval_pdf(m, v3(wi[i], wi[n + i], wi[2 * n + i]), v3(wo[i], wo[n + i], wo[2 * n + i]), &fv, &p);
    f[i] = fv.x;
    f[n + i] = fv.y;
    f[2 * n + i] = fv.z;
    pdf[i] = p;
}

__global__ __launch_bounds__(256) void k_emitter_sample(DevScene sc, uint32_t n, const float *p, const float *u, float *d,
                                                        float *dist, float *pdf, float *weight, float *q,
                                                        uint32_t *emitter) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    ESample e = sample_emitter(global_tables(sc), v3(p[i], p[n + i], p[2 * n + i]), F4{u[i], u[n + i], u[2 * n + i], u[3 * n + i]});
    d[i] = e.d.x;
    d[n + i] = e.d.y;
    d[2 * n + i] = e.d.z;
    q[i] = e.q.x;
    q[n + i] = e.q.y;
    q[2 * n + i] = e.q.z;
    dist[i] = e.dist;
    pdf[i] = e.valid ? e.pdf : 0.0f;
    weight[i] = e.valid ? e.weight.x : 0.0f;
    weight[n + i] = e.valid ? e.weight.y : 0.0f;
    weight[2 * n + i] = e.valid ? e.weight.z : 0.0f;
    emitter[i] = e.emitter;
}

__global__ __launch_bounds__(256) void k_sensor_sample_ray(pbrt_camera cam, uint32_t n, const float *pos, float *o,
                                                           float *d, float *tmax) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    V3 oo, dd;
    float tm;
    camera_ray(cam, pos[i], pos[n + i], &oo, &dd, &tm);
    o[i] = oo.x;
    o[n + i] = oo.y;
    o[2 * n + i] = oo.z;
    d[i] = dd.x;
    d[n + i] = dd.y;
    d[2 * n + i] = dd.z;
    tmax[i] = tm;
}

// UltraSensor.sample_ray (bytecode-only class)
__global__ __launch_bounds__(256) void k_us_sensor_sample_ray(pbrt_us_sensor s, int hemi, uint32_t n, const float *time,
                                                              const float *wl, const float *pos, const float *ap,
                                                              float *o, float *d, float *weight) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float N = (float)s.num_elements;
    float px = pos[i], py = pos[n + i], ax = ap[i], ay = ap[n + i];
    float ei = fminf(floorf(px * N), N - 1.0f);
    float ex, ez;
    if (__builtin_isinf(s.radius)) {
        ex = fma_(ei, s.pitch, -((N - 1.0f) * s.pitch) / 2.0f);
        ez = 0.0f;
    } else {
        float th = (ei - N / 2.0f) * (s.pitch / s.radius);
        ex = s.radius * sinf(th);
        ez = s.radius * (1.0f - cosf(th));
    }
    float offx = (ax - 0.5f) * s.element_width, offy = (ay - 0.5f) * s.element_height;
    V3 ol = {ex + offx, offy, ez};
    V3 dl;
    if (hemi) {
        dl = square_to_uniform_hemisphere(ax, ay);
    } else {
        float phi = 2.0f * K_PI * py, ct = wl[i];
        float st = sqrtf(fmaxf(0.0f, 1.0f - ct * ct));
        dl = {st * cosf(phi), st * sinf(phi), ct};
    }
    V3 ow = xf_point(s.to_world, ol);
    V3 dw = normalize(xf_vec(s.to_world, dl));
    float dweight = fabsf(dl.z) * s.directivity;
    weight[i] = cosf(2.0f * K_PI * s.center_frequency * time[i]) * dweight;
    o[i] = ow.x;
    o[n + i] = ow.y;
    o[2 * n + i] = ow.z;
    d[i] = dw.x;
    d[n + i] = dw.y;
    d[2 * n + i] = dw.z;
}

// CustomEmitter.sample_position + sample_ray (CustomEmmitter.py:30-107)
__global__ __launch_bounds__(256) void k_us_emitter_sample_ray(pbrt_us_emitter e, uint32_t n, const float *time,
                                                               const float *s1, const float *s2, const float *s3, float *o,
                                                               float *d, float *ray_time, float *weight, float *pdf_pos) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const EmitRay r = us_emitter_ray(e, time[i], s1[i], s2[i], s2[n + i], s3[i]);
    pdf_pos[i] = r.pdf_pos;
    ray_time[i] = r.time;
    weight[i] = r.weight;
    o[i] = r.o.x;
    o[n + i] = r.o.y;
    o[2 * n + i] = r.o.z;
    d[i] = r.d.x;
    d[n + i] = r.d.y;
    d[2 * n + i] = r.d.z;
}

// CustomSensor.put_data (CustomSensor.py:29-59)
__global__ __launch_bounds__(256) void k_us_put_data(pbrt_us_receiver r, uint32_t n, const float *ox, const float *time,
                                                     const float *d, const float *amplitude, float *buf) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double idx = rint((double)ox[i] / (double)r.pitch + (double)r.number_of_elements / 2.0);  // :36
    double ti = rint((double)time[i] * (double)r.sample_rate);                                // :43
    V3 dir = normalize(-v3(d[i], d[n + i], d[2 * n + i]));                                    // :46
    float gain = fmaxf(0.0f, dot(dir, v3(0, 0, 1)));                                          // :51
    float amp = amplitude[i] * gain;                                                          // :53
    if (idx >= 0 && idx < (double)r.number_of_elements && ti >= 0 && ti < (double)r.time_samples)  // :58
        atomicAdd(&buf[(size_t)idx * r.time_samples + (size_t)ti], amp);                      // :59
}

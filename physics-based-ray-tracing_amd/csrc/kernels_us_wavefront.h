// kernels_us_wavefront.h -- ultrasound mode on BVH scenes (tessellated phantoms, meshes) as streams (gfx950).
//
// The fused ultrasound bounce (kernels_us.h k_us_bounce<.., BVH>) walks the tree for the closest hit, shades, and walks it again
// for the unbounded occlusion ray towards the receive element (CustomIntegrator.py:324-325), all in one kernel: 125 - 128 VGPRs,
// four waves per SIMD, and a wave stays in the tree until its last lane has left it.  Here a bounce is the two launches of the
// radiance streams (kernels_wavefront.h):
//   k_trace      unchanged: the continuation rays of the live paths (closest hit) and the occlusion rays the previous bounce
//                emitted (any hit, unbounded), 64 VGPRs, eight waves per SIMD
//   k_us_shade   every wave on its own: paths whose ray left the scene end, the others are listed and shaded 64 at a time --
//                the statements of k_us_bounce from the hit on (:314-376) -- except that the occlusion ray is handed out instead of
//                traced: the echo (channel index, pressure) rides in the path state as PENDING and is deposited by the next
//                k_us_shade once k_trace has written its visibility (a path that ended meanwhile leaves a record of its own).
//                A flush (k_trace + k_us_shade on the records alone) follows the last bounce.
// Same arithmetic per path, same RNG keys; the echoes of a bin are summed in another order (f32 atomics, as before).
// With first-bounce tables (k_us_first) depth 0 needs no tracing at all: k_us_shade<true> reads the ray's shared hit and the
// (ray, receive element) record, visibility included, and deposits at once.
//
// Path state, 64 B in float4 planes:  q0 = (o, amp)  q1 = (d, atten)  q2 = (tof, geo_len, home, pending channel index | ~0)
//                                     q3 = (pending pressure, weight of the primary ray, -, visibility: written 0 here, set by k_trace)
// Occlusion-ray records as in kernels_wavefront.h: q0 = (origin, tmax = inf), q1 = (direction, dest); records of ended paths also
// q2 = (pressure, channel index, -, visibility).
#pragma once
#include "kernels_us.h"
#include "kernels_wavefront.h"

#define US_WF_STATE_Q 4u
#define US_WF_VIS_Q 3u   // plane whose .w receives the visibility (WfArgs::vis_q)

struct UsWfArgs {
    UsArgs u;                       // scene, acquisition parameters, tables, channel buffer, statistics rows
    float4 *st_in, *st_out;         // [4][cap]
    uint32_t *hit_id;
    float4 *shd_in, *shd_out;       // [4][cap]
    const uint32_t *seg_in, *nsh_in;
    uint32_t *seg_out, *nsh_out;
    uint32_t region0, n_regions;
    uint32_t *guard;                // the context's guard words (kernels_wavefront.h WfArgs::guard)
};

// primary rays of a pass into the path state (depth 0 without first-bounce tables): CustomIntegrator.py:270-279
__global__ __launch_bounds__(256) void k_us_init_wf(const UsArgs a, float4 *st, uint32_t *seg_cnt, uint32_t n_regions) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_regions) seg_cnt[i] = a.n_paths > i * WF_REGION ? min(a.n_paths - i * WF_REGION, WF_REGION) : 0u;
    if (i >= a.n_paths) return;
    const uint32_t ray_id = udiv_fast(i, a.div_ppr);
    const uint32_t ang = udiv_fast(ray_id, a.div_ne), el = ray_id - ang * a.p.n_elements;
    V3 o = xf_point(a.p.sensor_to_world, v3(a.elem_x[el], 0.0f, 0.0f));
    V3 d = v3(a.dir0[3 * ang], a.dir0[3 * ang + 1], a.dir0[3 * ang + 2]);
    float tof = 0.0f, w_ray = 1.0f;
    if (a.p.primary == PBRT_US_PRIMARY_EMITTER)  // the path's own ray from CustomEmitter.sample_ray (kernels_us.h us_emitter_primary)
        w_ray = us_emitter_primary(a.p, a.p.sensor_to_world, ray_id, a.path_first + (i - ray_id * a.ppr_pass), ang, el, a.seed, &o, &d, &tof);
    const size_t cp = a.cap;
    const float4 q0 = {o.x, o.y, o.z, 1.0f}, q1 = {d.x, d.y, d.z, 1.0f}, q2 = {tof, 0.0f, __uint_as_float(i), __uint_as_float(0xffffffffu)},
                 q3 = {0.0f, w_ray, 0.0f, 0.0f};  // q3.y: the weight of the path's primary ray (1 for the integrator's own)
    st[i] = q0;
    st[cp + i] = q1;
    st[2u * cp + i] = q2;
    st[3u * cp + i] = q3;
}

// TAB: depth 0 with the first-bounce tables (the paths are generated from their index, nothing is read but the tables)
template <bool TAB>
__global__ __launch_bounds__(WF_SHADE_THREADS, WF_SHADE_WAVES_PER_EU) void k_us_shade(const UsWfArgs w) {
    constexpr uint32_t T_ = WF_SHADE_THREADS, W = T_ / 64;
    constexpr int NCH = WF_SHADE_CHUNKS;
    const UsArgs &a = w.u;
    __shared__ uint32_t wlist[W][64 * (NCH + 1)], wprim[W][64 * (NCH + 1)];
    __shared__ uint32_t q_out, q_shd, q_dead, q_done;
    __shared__ uint32_t agg_idx[US_AGG_BINS];
    __shared__ float agg_sum[US_AGG_BINS];
    __shared__ float uni[24];
    const uint32_t r = w.region0 + xcd_swizzle(blockIdx.x, gridDim.x), base = r * WF_REGION;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wid = tid >> 6;
    const uint32_t cnt_in = TAB ? (a.n_paths > base ? min(a.n_paths - base, WF_REGION) : 0u) : w.seg_in[r];
    const uint32_t n_dead = (!TAB && w.nsh_in) ? w.nsh_in[r] >> 16 : 0u;
    if (cnt_in == 0 && n_dead == 0) {  // uniform
        if (tid == 0) {
            w.seg_out[r] = 0;
            w.nsh_out[r] = 0;
        }
        return;
    }
    if (tid == 0) {
        q_out = 0;
        q_shd = 0;
        q_dead = 0;
        q_done = 0;
    }
    for (uint32_t t = tid; t < US_AGG_BINS; t += T_) {
        agg_idx[t] = 0xffffffffu;
        agg_sum[t] = 0.0f;
    }
    // the launch-uniform floats through LDS (broadcast reads), as in k_us_bounce: the kernel wants more scalars than a wave has
    if (tid < 12) uni[tid] = a.p.sensor_to_world[tid];
    if (tid == 12) {
        uni[12] = a.tn[0]; uni[13] = a.tn[1]; uni[14] = a.tn[2]; uni[15] = a.am; uni[16] = a.ac; uni[17] = a.cos_min;
        uni[18] = a.katt; uni[19] = a.two_pi_f; uni[20] = a.inv_c; uni[21] = a.p.fs; uni[22] = a.p.max_path_len;
    }
    __syncthreads();
    // an echo into the workgroup's table (kernels_us.h: ds_cmpst claims a bin, ds_add_f32 adds, foreign bins go to the global atomic)
    auto deposit = [&](uint32_t ci, float pressure) {
        const uint32_t bin = (ci * 2654435761u) >> (32 - US_AGG_LOG2);
        const uint32_t owner = atomicCAS(&agg_idx[bin], 0xffffffffu, ci);
        if (owner == 0xffffffffu || owner == ci)
            __hip_atomic_fetch_add(&agg_sum[bin], pressure, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        else
            atomicAdd(&a.channel[ci], pressure);
    };
    const uint32_t NE = a.p.n_elements, T = a.p.time_samples;
    const size_t cp = a.cap;
    // ---- occlusion rays of paths that ended at the previous bounce: their echo, if the ray got through
    for (uint32_t k = tid; k < n_dead; k += T_) {
        const float4 rec = w.shd_in[2u * cp + (base + WF_REGION - n_dead + k)];
        if (rec.w != 0.0f) deposit(__float_as_uint(rec.y), rec.x);
    }
    uint32_t list_n = 0, n_seg_w = 0;
    uint32_t c0 = wid * 64u;
    for (;;) {
        if (c0 < cnt_in) {
            // WF_SHADE_CHUNKS chunks per step, their loads in one batch (kernels_wavefront.h k_shade)
            uint32_t hidv[NCH];
            float4 q2v[NCH], q3v[NCH];
#pragma unroll
            for (int j = 0; j < NCH; ++j) {
                const uint32_t s = c0 + (uint32_t)j * (W * 64u) + lane;
                hidv[j] = 0xffffffffu;
                q2v[j] = q3v[j] = float4{0.0f, 0.0f, 0.0f, 0.0f};
                if (s < cnt_in) {
                    if (TAB) {
                        const uint32_t ray_id = udiv_fast(base + s, a.div_ppr);
                        hidv[j] = __float_as_uint(a.first_hit[ray_id].w);  // the primitive the ray's shared first hit lies on
                    } else {
                        hidv[j] = w.hit_id[base + s];
                        q2v[j] = w.st_in[2u * cp + base + s];
                        q3v[j] = w.st_in[3u * cp + base + s];
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < NCH; ++j) {
                const uint32_t s = c0 + (uint32_t)j * (W * 64u) + lane;
                const uint32_t hid = hidv[j];
                // the pending echo of the previous bounce (every path, whether it goes on or not)
                if (!TAB && s < cnt_in && q3v[j].w != 0.0f && __float_as_uint(q2v[j].w) != 0xffffffffu) deposit(__float_as_uint(q2v[j].w), q3v[j].x);
                const bool is_hit = hid != 0xffffffffu;
                const unsigned long long bh = __ballot(is_hit);
                if (is_hit) {
                    const uint32_t e = list_n + __builtin_amdgcn_mbcnt_hi((uint32_t)(bh >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bh, 0u));
                    wlist[wid][e] = s;
                    wprim[wid][e] = hid;
                }
                list_n += (uint32_t)__popcll(bh);
            }
            __builtin_amdgcn_wave_barrier();
            c0 += (uint32_t)NCH * W * 64u;
        } else if (list_n == 0) {
            break;
        }
        if (list_n < 64u && c0 < cnt_in) continue;
        // ---- shade 64 listed paths: kernels_us.h k_us_bounce from the hit on, statement for statement (the list is emptied below 64
        // entries before the wave reads its next chunks)
        for (;;) {
        const uint32_t take = min(list_n, 64u);
        const bool act = lane < take;
        list_n -= take;
        bool survive = false, pend = false;
        V3 o = {0, 0, 0}, d = {0, 0, 1}, so = {0, 0, 0}, sdir = {0, 0, 1};
        float amp = 1.0f, atten = 1.0f, tof = 0.0f, geo_len = 0.0f, pressure = 0.0f, w_ray = 1.0f;
        uint32_t home = 0, ci = 0xffffffffu;
        if (act) {
            const uint32_t s = wlist[wid][list_n + lane];
            Hit h;
            h.slot = wprim[wid][list_n + lane];
            h.prim = h.slot;
            uint32_t ray_id, k, ang;
            float4 rx = {0.0f, 0.0f, 0.0f, 0.0f};
            if (TAB) {
                home = base + s;
                ray_id = udiv_fast(home, a.div_ppr);
                k = a.path_first + (home - ray_id * a.ppr_pass);
                ang = udiv_fast(ray_id, a.div_ne);
                const uint32_t el = ray_id - ang * NE;
                o = xf_point(uni, v3(a.elem_x[el], 0.0f, 0.0f));                         // :270,273
                d = v3(a.dir0[3 * ang], a.dir0[3 * ang + 1], a.dir0[3 * ang + 2]);        // :271,273
                const float4 fh = a.first_hit[ray_id];
                h.t = fh.x;
                h.u = fh.y;
                h.v = fh.z;
            } else {
                const float4 q0 = w.st_in[base + s], q1 = w.st_in[cp + base + s], q2 = w.st_in[2u * cp + base + s];
                // the weight of the path's primary ray rides in q3.y (k_us_init_wf); it is 1 unless the rays come from the emitter, so
                // only that mode reads it
                if (a.p.primary == PBRT_US_PRIMARY_EMITTER) w_ray = w.st_in[3u * cp + base + s].y;
                o = {q0.x, q0.y, q0.z};
                amp = q0.w;
                d = {q1.x, q1.y, q1.z};
                atten = q1.w;
                tof = q2.x;
                geo_len = q2.y;
                home = __float_as_uint(q2.z);
                ray_id = udiv_fast(home, a.div_ppr);
                k = a.path_first + (home - ray_id * a.ppr_pass);
                ang = udiv_fast(ray_id, a.div_ne);
            }
            const pbrt_prim P = wf_load_prim(a.sc.prims + h.slot);
            const bool has_vn = a.sc.vnormals != nullptr;  // uniform
            WfVn vn;
            if (has_vn) vn = wf_load_vn(a.sc.vnormals, h.slot);
            if (!TAB) {  // (t, u, v) of the hit k_trace found; a repetition that disagrees fails the call (kernels_wavefront.h k_shade)
                h.t = K_INF;
                h.u = h.v = 0.0f;
                if (!prim_hit(P, o, d, K_INF, &h.t, &h.u, &h.v)) atomicAdd(w.guard + WF_GUARD_REHIT, 1u);
            }
            const uint32_t depth = a.depth;
            const V3 tn = {uni[12], uni[13], uni[14]};
            const SI si = wf_make_si(P, o, d, h.t, h.u, h.v, has_vn, vn);
            const float distance = h.t;                                                   // :314
            geo_len += distance;                                                          // :315
            const bool no_acc = (a.p.quirks & PBRT_USQ_NO_TOF_ACCUM) != 0;
            if (!no_acc) tof += distance * uni[20];                                       // :316
            const uint32_t block = (a.p.quirks & PBRT_USQ_FROZEN_DRAWS) ? 0u : depth;
            const F4 u = rng4(ray_id, k, block, a.seed);
            const uint32_t recv = min((uint32_t)(u.x * (float)NE), NE - 1);               // :319
            float total_time = 0.0f, phase = 0.0f;
            if (TAB) {
                rx = a.first_rx[(size_t)ray_id * NE + recv];
            } else {
                const V3 target = xf_point(uni, v3(a.elem_x[recv], 0.0f, 0.0f));          // :320-321
                const V3 tv = target - si.p;
                const float dist_recv = sqrtf(dot(tv, tv));
                sdir = tv * (1.0f / dist_recv);                                           // :322
                so = offset_origin(si.p, si.n, sdir);                                     // :324 (the ray k_trace walks)
                const float tof_hit = no_acc ? tof + distance * uni[20] : tof;
                total_time = a.tx[ray_id] + tof_hit + dist_recv * uni[20];                // :329
                phase = uni[19] * total_time;                                             // :330
            }
            atten *= expf(uni[18] * distance / 8.686f);                                   // :328
            const pbrt_material M = a.sc.mats[P.material];
            const Frame fr = make_sh_frame(si.ns, si_dp_du<true>(P, si));
            const V3 wi = to_local(fr, -d);
            float a_resp, bpdf;
            V3 new_dir;
            bool ok = true;
            if (M.type == PBRT_MAT_ULTRA) {
                const float s1b = (a.p.quirks & PBRT_USQ_DIAG_SAMPLE) ? u.w : rng4(ray_id, k, block | 0x40000000u, a.seed).x;
                const UltraOut uo = ultra_core(M, a.p.quirks, wi, si.n, si.ns, u.y, u.z, s1b);  // :338
                a_resp = uo.amp;
                bpdf = uo.pdf;
                new_dir = to_world(fr, to_local(fr, uo.chosen));                          // CustomBSDF.py:165 + :358
            } else {
                const BSample bs = bsdf_sample(M, a.p.quirks, wi, si.n, si.ns, fr, u.y, u.z, u.w);
                ok = bs.valid;
                a_resp = bs.weight.x;
                bpdf = bs.pdf;
                new_dir = to_world(fr, bs.wo);
            }
            if (ok) {
                const float cos_theta = dot(si.ns, -d);                                   // :340
                amp *= a_resp * cos_theta * fmaxf(bpdf, 1e-6f);                           // :341
                float fd = 0.0f, carrier = 0.0f;
                if (TAB) {
                    fd = rx.x;
                    carrier = rx.y;
                    ci = __float_as_uint(rx.z);                                           // (visibility included)
                } else {
                    float tf = rintf(total_time * uni[21]);                               // :351-352
                    if (a.p.quirks & PBRT_USQ_CLAMP_TIME) tf = fminf(fmaxf(tf, 0.0f), (float)(T - 1));
                    if (tf >= 0.0f && tf < (float)T) {                                    // :353, `visible` comes from k_trace
                        ci = (ang * NE + recv) * T + (uint32_t)tf;                        // :354
                        const float w_o = dot(d, si.ns) / (float)(a.p.n_angles * NE);     // :286-287,345
                        fd = directivity_weight_i(sdir, tn, uni[15], uni[16]) * w_o;      // :345
                        carrier = (a.p.quirks & PBRT_USQ_NO_CARRIER) ? 1.0f : sinf(phase);
                    }
                }
                pressure = atten * amp * fd * carrier * w_ray;                            // :348 (x 1, or the emitter ray's weight: D15)
                if (ci != 0xffffffffu) {
                    if (TAB) {
                        deposit(ci, pressure);
                        ci = 0xffffffffu;
                    } else {
                        pend = true;  // an occlusion ray decides
                    }
                }
                d = normalize(new_dir);                                                   // :358-359
                o = offset_origin(si.p, si.n, d);
                bool surv;
                if (a.p.quirks & PBRT_USQ_SIGNED_RR) {                                    // Dr.Jit variant :219-224
                    const float rr_prob = fminf(atten * amp, 1.0f);
                    surv = u.w < rr_prob;
                    atten = surv ? atten / rr_prob : 0.0f;
                } else {
                    const float rr_prob = fminf(fabsf(atten * amp), 1.0f);                // :364
                    surv = !(u.w > rr_prob);                                              // :365-366
                    atten /= rr_prob;                                                     // :367
                }
                const bool within = dot(d, tn) >= uni[17];                                // :371
                survive = within && (geo_len < uni[22]) && (depth + 1 < a.p.max_depth) && surv;  // :372-376
            }
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
        const bool shd_live = pend && survive, shd_dead = pend && !survive;
        if (survive) {
            const float4 q0 = {o.x, o.y, o.z, amp}, q1 = {d.x, d.y, d.z, atten},
                         q2 = {tof, geo_len, __uint_as_float(home), __uint_as_float(shd_live ? ci : 0xffffffffu)},
                         q3 = {shd_live ? pressure : 0.0f, w_ray, 0.0f, 0.0f};
            w.st_out[out_slot] = q0;
            w.st_out[cp + out_slot] = q1;
            w.st_out[2u * cp + out_slot] = q2;
            w.st_out[3u * cp + out_slot] = q3;
        }
        const unsigned long long bl = __ballot(shd_live);
        if (bl) {
            uint32_t got = 0;
            if (lane == 0) got = atomicAdd(&q_shd, (uint32_t)__popcll(bl));
            const uint32_t off = (uint32_t)__builtin_amdgcn_readfirstlane((int)got);
            if (shd_live) {
                const uint32_t k = base + off + __builtin_amdgcn_mbcnt_hi((uint32_t)(bl >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bl, 0u));
                const float4 q0 = {so.x, so.y, so.z, K_INF}, q1 = {sdir.x, sdir.y, sdir.z, __uint_as_float(out_slot)};
                w.shd_out[k] = q0;
                w.shd_out[cp + k] = q1;
            }
        }
        const unsigned long long bd = __ballot(shd_dead);
        if (bd) {  // the path ended at this bounce; its echo still waits for its occlusion ray
            uint32_t got = 0;
            if (lane == 0) got = atomicAdd(&q_dead, (uint32_t)__popcll(bd));
            const uint32_t off = (uint32_t)__builtin_amdgcn_readfirstlane((int)got);
            if (shd_dead) {
                const uint32_t k =
                    base + WF_REGION - 1u - (off + __builtin_amdgcn_mbcnt_hi((uint32_t)(bd >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bd, 0u)));
                const float4 q0 = {so.x, so.y, so.z, K_INF}, q1 = {sdir.x, sdir.y, sdir.z, __uint_as_float(WF_DEAD | k)},
                             q2 = {pressure, __uint_as_float(ci), 0.0f, 0.0f};
                w.shd_out[k] = q0;
                w.shd_out[cp + k] = q1;
                w.shd_out[2u * cp + k] = q2;
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
        row[stride] += n_seg_w;  // one occlusion ray per shaded segment (the reference traces one per bounce, :324)
        if (wid == 0) row[(2 + min(a.depth, (uint32_t)MAX_DEPTH_STATS - 1)) * stride] += cnt_in;
        if (atomicAdd(&q_done, 1u) == W - 1) {
            w.seg_out[r] = atomicAdd(&q_out, 0u);
            w.nsh_out[r] = atomicAdd(&q_shd, 0u) | (atomicAdd(&q_dead, 0u) << 16);
        }
    }
    __syncthreads();  // all echoes of the workgroup are in the bins (every wave gets here: no exit after the set-up)
    for (uint32_t t = tid; t < US_AGG_BINS; t += T_) {
        const uint32_t ci = agg_idx[t];
        if (ci != 0xffffffffu) atomicAdd(&a.channel[ci], agg_sum[t]);
    }
}

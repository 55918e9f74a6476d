// kernels_beamform.h -- image formation behind the ultrasound hot path: delay-and-sum
// beamforming of the channel buffer onto a GridScan, envelope (modulus of the analytic signal along z) and log
// compression.  The reference delegates these to the third-party `ultraspy` package (USMain.py:126-221, absent
// here): the arithmetic below is this build's own definition (include/pbrt_hip.h), restated in oracle/beamform.py.
#pragma once
#include "../../include/pbrt_hip.h"
#include "device_math.h"

// One thread per pixel, z fastest: neighbouring lanes read neighbouring samples of the same (angle, element)
// trace, so the gathers coalesce; the 12.8 MB channel buffer stays in L2 / MALL.  Sample positions in f64 (a
// position of 10^4 samples leaves f32 only 10 bits of fraction), samples and sums in f32.
__global__ __launch_bounds__(256) void k_das_beamform(pbrt_das_params p, const float *__restrict__ data,
                                                      const float *__restrict__ tx, const float *__restrict__ elem_x,
                                                      const float *__restrict__ gx, const float *__restrict__ gz,
                                                      float *__restrict__ out) {
    const uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= p.nx * p.nz) return;
    const uint32_t ix = idx / p.nz, iz = idx - ix * p.nz;
    const double x = (double)gx[ix], z = (double)gz[iz];
    const double inv_c = 1.0 / (double)p.sound_speed, fs = (double)p.fs, t0 = (double)p.t0;
    const uint32_t A = p.n_angles, E = p.n_elements, T = p.time_samples;
    const double half_ap = p.f_number > 0.0f ? z / (2.0 * (double)p.f_number) : 1e300;
    float acc = 0.0f;
    for (uint32_t a = 0; a < A; ++a) {
        // first arrival of the emitted wavefront at the pixel
        double t_tx = 1e300;
        for (uint32_t e = 0; e < E; ++e) {
            const double dx = x - (double)elem_x[e];
            t_tx = fmin(t_tx, (double)tx[a * E + e] + sqrt(dx * dx + z * z) * inv_c);
        }
        const float *trace = data + (size_t)a * E * T;
        for (uint32_t e = 0; e < E; ++e) {
            const double dx = x - (double)elem_x[e];
            if (fabs(dx) > half_ap) continue;
            const double s = (t_tx + sqrt(dx * dx + z * z) * inv_c - t0) * fs;
            if (p.interpolation == PBRT_DAS_NEAREST) {
                const double r = rint(s);
                if (r >= 0.0 && r <= (double)(T - 1)) acc += trace[(size_t)e * T + (uint32_t)r];
            } else {
                const double f = floor(s);
                if (f >= 0.0 && f < (double)(T - 1)) {
                    const uint32_t i0 = (uint32_t)f;
                    const float w = (float)(s - f);
                    const float v0 = trace[(size_t)e * T + i0], v1 = trace[(size_t)e * T + i0 + 1];
                    acc += fma_(w, v1 - v0, v0);
                } else if (s == (double)(T - 1)) {
                    acc += trace[(size_t)e * T + (T - 1)];
                }
            }
        }
    }
    out[idx] = p.compound_mean ? acc / (float)A : acc;
}

// Envelope: one workgroup per image column (nz samples along z).  Analytic signal by the DFT definition
// (scipy.signal.hilbert): X = DFT(x); X[0] and X[N/2] (N even) kept, positive frequencies doubled, negative
// frequencies zeroed; y = IDFT(X); env = |y|.  O(N^2) with an exact twiddle table (sincospi of 2 k / N, index
// reduced mod N in integers), N <= 4096: 650 columns x 400^2 is 0.1 G complex MACs -- not worth an FFT.
#define ENV_MAX_N 4096
__global__ __launch_bounds__(256) void k_hilbert_env(uint32_t nz, const float *__restrict__ rf, float *__restrict__ env) {
    extern __shared__ __attribute__((aligned(16))) float lds_env[];
    float *xs = lds_env;              // [nz]
    float *wc = xs + nz, *ws = wc + nz;  // twiddles cos / sin (2 pi j / N)
    float *Xr = ws + nz, *Xi = Xr + nz;  // spectrum with the analytic-signal weights applied
    const uint32_t N = nz, col = blockIdx.x;
    for (uint32_t j = threadIdx.x; j < N; j += blockDim.x) {
        xs[j] = rf[(size_t)col * N + j];
        float sn, cs;
        sincospif(2.0f * (float)j / (float)N, &sn, &cs);
        wc[j] = cs;
        ws[j] = sn;
    }
    __syncthreads();
    for (uint32_t k = threadIdx.x; k < N; k += blockDim.x) {
        float re = 0.0f, im = 0.0f;
        uint32_t j = 0;  // (k * n) mod N
        for (uint32_t n = 0; n < N; ++n) {
            re = fma_(xs[n], wc[j], re);
            im = fma_(-xs[n], ws[j], im);  // e^{-i 2 pi k n / N}
            j += k;
            if (j >= N) j -= N;
        }
        float h;
        if (k == 0 || (2 * k == N))
            h = 1.0f;
        else if (2 * k < N)
            h = 2.0f;
        else
            h = 0.0f;
        Xr[k] = re * h;
        Xi[k] = im * h;
    }
    __syncthreads();
    const float inv_n = 1.0f / (float)N;
    for (uint32_t n = threadIdx.x; n < N; n += blockDim.x) {
        float re = 0.0f, im = 0.0f;
        uint32_t j = 0;
        const uint32_t kmax = N / 2 + 1;  // the weights vanish above N / 2
        for (uint32_t k = 0; k < kmax; ++k) {
            // (Xr + i Xi) * (cos + i sin)
            re = fma_(Xr[k], wc[j], re);
            re = fma_(-Xi[k], ws[j], re);
            im = fma_(Xr[k], ws[j], im);
            im = fma_(Xi[k], wc[j], im);
            j += n;
            if (j >= N) j -= N;
        }
        re *= inv_n;
        im *= inv_n;
        env[(size_t)col * N + n] = sqrtf(fma_(re, re, im * im));
    }
}

// Log compression (USMain.py:210-218).  Pass 1: maximum of the (non-negative) envelope; the order-preserving
// uint view of non-negative floats lets atomicMax do it.  Pass 2: map.
__global__ __launch_bounds__(256) void k_env_max(uint32_t n, const float *__restrict__ env, uint32_t *mx) {
    __shared__ float part[256];
    float m = 0.0f;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) m = fmaxf(m, env[i]);
    part[threadIdx.x] = m;
    __syncthreads();
    for (uint32_t w = 128; w > 0; w >>= 1) {
        if (threadIdx.x < w) part[threadIdx.x] = fmaxf(part[threadIdx.x], part[threadIdx.x + w]);
        __syncthreads();
    }
    if (threadIdx.x == 0) atomicMax(mx, __float_as_uint(fmaxf(part[0], 0.0f)));
}
__global__ __launch_bounds__(256) void k_log_compress(uint32_t n, const float *__restrict__ env, const uint32_t *mx, float dr,
                                                      float *__restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float max_db = 20.0f * log10f(__uint_as_float(*mx) + 1e-12f);
    const float min_db = max_db - dr;
    float db = 20.0f * log10f(env[i] + 1e-12f);
    db = fminf(fmaxf(db, min_db), max_db);
    out[i] = (db - min_db) / dr;
}

// Pulse model (SURVEY f-3; RayTracingV0.py:194-204): every trace convolved with the Gaussian-windowed carrier
// h[k] = sin(2 pi fc k / fs) * exp(-(k / fs)^2 / sigma^2), |k| <= K.  One workgroup per 256 output samples of one
// trace: taps and the 256 + 2K input samples staged in LDS.  HBM-bound (4 B in, 4 B out per sample).
#define PULSE_MAX_K 1024
__global__ __launch_bounds__(256) void k_apply_pulse(uint32_t T, uint32_t K, float fs, float fc, float sigma,
                                                     const float *__restrict__ in, float *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float lds_pulse[];
    float *h = lds_pulse;            // [2K + 1], h[K + k]
    float *x = lds_pulse + 2 * K + 1;  // [256 + 2K]
    const uint32_t tr = blockIdx.y, n0 = blockIdx.x * 256u;
    const float *src = in + (size_t)tr * T;
    for (uint32_t i = threadIdx.x; i < 2 * K + 1; i += 256u) {
        const float t = ((float)i - (float)K) / fs;
        // phase reduced per cycle: sin(2 pi fc t) = sinpi(2 fc t)
        h[i] = sinpif(2.0f * fc * t) * expf(-(t * t) / (sigma * sigma));
    }
    for (uint32_t i = threadIdx.x; i < 256u + 2 * K; i += 256u) {
        const int64_t n = (int64_t)n0 + (int64_t)i - (int64_t)K;
        x[i] = (n >= 0 && n < (int64_t)T) ? src[n] : 0.0f;
    }
    __syncthreads();
    const uint32_t n = n0 + threadIdx.x;
    if (n >= T) return;
    float acc = 0.0f;
    // out[n] = sum_k in[n - k] h[k]: x index of in[n - k] is threadIdx.x + K - k
    for (uint32_t j = 0; j < 2 * K + 1; ++j) acc = fma_(x[threadIdx.x + 2 * K - j], h[j], acc);
    out[(size_t)tr * T + n] = acc;
}

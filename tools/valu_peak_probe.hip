// What does one SIMD of this GPU sustain in plain f32 VALU instructions, and at which clock?  Independent v_fma_f32 chains, w waves per
// SIMD; s_memtime (shader clock) against s_memrealtime (100 MHz) gives the clock the kernel actually ran at.
// build + run on the GPU box: hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_peak tools/valu_peak_probe.hip && /tmp/valu_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int CHAINS>
__global__ __launch_bounds__(256) void k_fma(float *sink, unsigned long long *clk, int iters) {
    unsigned long long t0, r0, t1, r1;
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(r0));
    float a[CHAINS];
    for (int k = 0; k < CHAINS; ++k) a[k] = threadIdx.x * 1e-3f + k;
    const float b = 1.0001f, c = 0.5f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int k = 0; k < CHAINS; ++k) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c));
    }
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1));
    float s = 0;
    for (int k = 0; k < CHAINS; ++k) s += a[k];
    if (s == 123.456f) sink[0] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        clk[0] = t1 - t0;
        clk[1] = r1 - r0;
    }
}
// same loop with v_pk_fma_f32 (two f32 FMAs per lane per instruction)
template <int CHAINS>
__global__ __launch_bounds__(256) void k_pk(float *sink, unsigned long long *clk, int iters) {
    typedef float __attribute__((ext_vector_type(2))) f2;
    f2 a[CHAINS];
    for (int k = 0; k < CHAINS; ++k) a[k] = f2{threadIdx.x * 1e-3f + k, 1.0f};
    const f2 b = {1.0001f, 1.0002f}, c = {0.5f, 0.25f};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int k = 0; k < CHAINS; ++k) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c));
    }
    float s = 0;
    for (int k = 0; k < CHAINS; ++k) s += a[k].x + a[k].y;
    if (s == 123.456f) sink[0] = s;
}
template <int CHAINS>
void run_pk(int waves_per_simd, int n_cu) {
    float *sink;
    hipMalloc(&sink, 4);
    const int iters = 4000, nb = n_cu * waves_per_simd;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(k_pk<CHAINS>, dim3(nb), dim3(256), 0, 0, sink, nullptr, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k_pk<CHAINS>, dim3(nb), dim3(256), 0, 0, sink, nullptr, iters);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    printf("v_pk_fma_f32 chains %d waves/SIMD %d: %.3f ms for %.0f instructions per SIMD\n", CHAINS, waves_per_simd, ms,
           (double)iters * 8 * CHAINS * waves_per_simd);
}
template <int CHAINS>
void run(int waves_per_simd, int n_cu) {
    float *sink;
    unsigned long long *clk, h[2];
    hipMalloc(&sink, 4);
    hipMalloc(&clk, 16);
    const int iters = 4000, nb = n_cu * waves_per_simd;  // one 256-thread block = one wave per SIMD of a CU
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(k_fma<CHAINS>, dim3(nb), dim3(256), 0, 0, sink, clk, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k_fma<CHAINS>, dim3(nb), dim3(256), 0, 0, sink, clk, iters);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    const double instr_per_simd = (double)iters * 8 * CHAINS * waves_per_simd;
    const double mhz = (double)h[0] / (double)h[1] * 100.0;
    printf("chains %d waves/SIMD %d: %.3f ms, shader clock %.0f MHz (memtime/memrealtime), %.2f shader cycles per wave64 v_fma_f32 per SIMD\n",
           CHAINS, waves_per_simd, ms, mhz, (double)h[0] / instr_per_simd);
}
int main() {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    printf("%s, %d CUs, clockRate %d kHz\n", p.name, p.multiProcessorCount, p.clockRate);
    for (int w : {1, 2, 4, 8}) run<1>(w, p.multiProcessorCount);
    for (int w : {1, 2, 4, 8}) run<4>(w, p.multiProcessorCount);
    for (int w : {1, 2, 8}) run<8>(w, p.multiProcessorCount);
    for (int w : {1, 8}) run_pk<4>(w, p.multiProcessorCount);
    for (int w : {1, 8}) run_pk<8>(w, p.multiProcessorCount);
    return 0;
}

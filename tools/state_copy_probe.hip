// state_copy_probe.hip -- calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on the access pattern of k_bounce's path
// state (MI355X_MICROARCH.md "HBM": FETCH_SIZE is calibrated for 16-B-per-lane streaming reads only -- "calibrate on a
// known byte count in your own access pattern").  The probe copies a tiled-SoA state buffer exactly as k_bounce reads
// and writes it: one wave = one 3840-byte tile = 15 rows of 64 lanes x 4 B, each row one buffer_load_dword /
// buffer_store_dword with the row offset in the instruction's immediate (kernels_radiance.h state_voff, bld, bst).
// Known traffic per launch: cap * 60 B read + cap * 60 B written, cap = 16 Mi slots (1.0 GB each way, 4 x the 256 MB
// Infinity Cache).  Build + run (GPU box):  hipcc --offload-arch=gfx950 -O3 -o /tmp/state_copy_probe tools/state_copy_probe.hip
//   rocprofv3 --pmc FETCH_SIZE --kernel-trace -d out -- /tmp/state_copy_probe      (then --pmc WRITE_SIZE)
// tools/pmc_calibrate.sh does both and prints  factor = known bytes / (counter * 1024).
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

#define N_STATE 15
typedef __amdgpu_buffer_rsrc_t Rsrc;
__device__ __forceinline__ Rsrc make_rsrc(const void *p, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, bytes, 0x00020000);
}

__global__ __launch_bounds__(512) void k_state_copy(const float *in, float *out, uint32_t cap) {
    const uint32_t slot = blockIdx.x * 512u + threadIdx.x;
    if (slot >= cap) return;
    const Rsrc r_in = make_rsrc(in, cap * (N_STATE * 4u)), r_out = make_rsrc(out, cap * (N_STATE * 4u));
    const uint32_t v4 = (slot >> 6) * (64u * N_STATE * 4u) + (slot & 63u) * 4u;
    uint32_t v[N_STATE];
#pragma unroll
    for (int k = 0; k < N_STATE; ++k) v[k] = __builtin_amdgcn_raw_buffer_load_b32(r_in, v4 + k * 256u, 0, 0);
#pragma unroll
    for (int k = 0; k < N_STATE; ++k) __builtin_amdgcn_raw_buffer_store_b32(v[k] + 1u, r_out, v4 + k * 256u, 0, 0);
}

// the 16-byte-per-lane streaming copy the guide's factor was measured on, for the same byte count (control)
__global__ __launch_bounds__(512) void k_wide_copy(const uint4 *in, uint4 *out, uint32_t n16) {
    const uint32_t i = blockIdx.x * 512u + threadIdx.x;
    if (i < n16) {
        uint4 v = in[i];
        v.x += 1u;
        out[i] = v;
    }
}

int main() {
    const uint32_t cap = 16u << 20;
    const size_t bytes = (size_t)cap * N_STATE * 4;
    float *a = nullptr, *b = nullptr;
    if (hipMalloc(&a, bytes) != hipSuccess || hipMalloc(&b, bytes) != hipSuccess) return 1;
    hipMemset(a, 0, bytes);
    hipMemset(b, 0, bytes);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_state_copy, dim3(cap / 512), dim3(512), 0, 0, a, b, cap);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        printf("k_state_copy: %zu B read + %zu B written in %.3f ms = %.0f GB/s\n", bytes, bytes, ms, 2.0 * bytes / ms / 1e6);
    }
    const uint32_t n16 = (uint32_t)(bytes / 16);
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_wide_copy, dim3((n16 + 511) / 512), dim3(512), 0, 0, (const uint4 *)a, (uint4 *)b, n16);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        printf("k_wide_copy:  %zu B read + %zu B written in %.3f ms = %.0f GB/s\n", bytes, bytes, ms, 2.0 * bytes / ms / 1e6);
    }
    printf("KNOWN_BYTES_EACH_WAY %zu\n", bytes);
    hipFree(a);
    hipFree(b);
    return 0;
}

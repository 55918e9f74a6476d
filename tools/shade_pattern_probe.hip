// shade_pattern_probe.hip -- calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on the access patterns of the BVH stream kernels
// (kernels_wavefront.h), which tools/state_copy_probe.hip does not cover: that one copies k_bounce's 4-B-per-lane tiled rows.
// MI355X_MICROARCH.md "HBM": FETCH_SIZE = TCC_EA0_RDREQ x 64 B, a 128-B request tallied as 64 -- exact x 1/2 for wide streaming
// reads, "other access widths are uncalibrated".  k_trace / k_shade read
//   (a) float4 PLANES, 16 B per lane, a wave = 1 KiB contiguous per plane           -> k_plane_copy   (known: 96 B per record each way)
//   (b) 16-B records GATHERED by a list of slots in increasing order with gaps      -> k_gather_sorted (density 0.95: the paths that
//       hit something at bounce 0 of the ring scene; 0.25: at bounce 1), and, as the worst case, by a random permutation
//                                                                                   -> k_gather_perm
// For the gathers the useful bytes are 16 per record (+ 4 for the index); what the counter should see depends on the request size
// the L2 issues, which is the question: the probe prints the useful bytes, tools/pmc_calibrate.sh divides the counters by them.
// Every buffer is >= 1 GB (4 x the 256 MB Infinity Cache) and read once.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <numeric>
#include <random>
#include <vector>

__global__ __launch_bounds__(256) void k_plane_copy(const float4 *in, float4 *out, uint32_t cap) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= cap) return;
    float4 q[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) q[k] = in[(size_t)k * cap + i];
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        q[k].x += 1.0f;
        out[(size_t)k * cap + i] = q[k];
    }
}
// out[j] = in[idx[j]]: the read is the gather, the write streams
__global__ __launch_bounds__(256) void k_gather(const float4 *in, const uint32_t *idx, float4 *out, uint32_t m) {
    const uint32_t j = blockIdx.x * 256u + threadIdx.x;
    if (j >= m) return;
    float4 v = in[idx[j]];
    v.x += 1.0f;
    out[j] = v;
}

static float run(const char *name, size_t useful_rd, size_t useful_wr, void (*launch)()) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        launch();
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        best = std::min(best, ms);
    }
    printf("PATTERN %s useful_read %zu useful_write %zu best_ms %.3f GBps %.0f\n", name, useful_rd, useful_wr, best,
           (useful_rd + useful_wr) / best / 1e6);
    return best;
}

static float4 *g_in, *g_out;
static uint32_t *g_idx;
static uint32_t g_cap, g_m;

int main() {
    const uint32_t cap = 16u << 20;   // planes: 16 Mi records x 96 B = 1.6 GB each way
    const uint32_t n = 64u << 20;     // gathers: a table of 64 Mi 16-B records = 1 GB
    if (hipMalloc(&g_in, (size_t)cap * 96) != hipSuccess || hipMalloc(&g_out, (size_t)cap * 96) != hipSuccess ||
        hipMalloc(&g_idx, (size_t)n * 4) != hipSuccess)
        return 1;
    hipMemset(g_in, 0, (size_t)cap * 96);
    hipMemset(g_out, 0, (size_t)cap * 96);
    g_cap = cap;
    run("planes_16B_per_lane", (size_t)cap * 96, (size_t)cap * 96,
        [] { hipLaunchKernelGGL(k_plane_copy, dim3(g_cap / 256), dim3(256), 0, 0, g_in, g_out, g_cap); });
    std::mt19937 rng(12345);
    std::vector<uint32_t> idx;
    for (double density : {0.95, 0.25}) {  // sorted subsets of the table
        idx.clear();
        std::bernoulli_distribution keep(density);
        for (uint32_t i = 0; i < n; ++i)
            if (keep(rng)) idx.push_back(i);
        g_m = (uint32_t)idx.size();
        hipMemcpy(g_idx, idx.data(), (size_t)g_m * 4, hipMemcpyHostToDevice);
        char name[64];
        snprintf(name, sizeof name, "gather_sorted_density_%.2f", density);
        run(name, (size_t)g_m * 20, (size_t)g_m * 16,
            [] { hipLaunchKernelGGL(k_gather, dim3((g_m + 255) / 256), dim3(256), 0, 0, g_in, g_idx, g_out, g_m); });
    }
    idx.resize(n);
    std::iota(idx.begin(), idx.end(), 0u);
    std::shuffle(idx.begin(), idx.end(), rng);
    g_m = n;
    hipMemcpy(g_idx, idx.data(), (size_t)n * 4, hipMemcpyHostToDevice);
    run("gather_random_permutation", (size_t)n * 20, (size_t)n * 16,
        [] { hipLaunchKernelGGL(k_gather, dim3((g_m + 255) / 256), dim3(256), 0, 0, g_in, g_idx, g_out, g_m); });
    hipFree(g_in);
    hipFree(g_out);
    hipFree(g_idx);
    return 0;
}

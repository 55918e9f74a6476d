// Which CUs does a hipExtStreamCreateWithCUMask mask enable, and what does a fixed amount of VALU work cost on them?
// build: hipcc --offload-arch=gfx950 -O3 -o gpurun_out/cu_mask_probe tools/cu_mask_probe.hip ; run on the GPU box.
// Per mask: distinct (xcc, se, cu) triples that ran a workgroup, and the time of a pure-FMA grid.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <set>
#include <vector>
__global__ void k_where(uint32_t *out, float *sink, int iters) {
    uint32_t hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    float a = threadIdx.x * 1e-3f, b = 1.0001f;
    for (int i = 0; i < iters; ++i) {
        a = __builtin_fmaf(a, b, 0.5f);
        a = __builtin_fmaf(a, b, -0.5f);
    }
    if (a == 123.456f) sink[0] = a;
    if (threadIdx.x == 0) {
        out[2 * blockIdx.x] = hw;
        out[2 * blockIdx.x + 1] = xcc;
    }
}
int main() {
    const int nb = 256 * 32, iters = 20000;
    uint32_t *d;
    float *sink;
    hipMalloc(&d, nb * 8);
    hipMalloc(&sink, 4);
    std::vector<uint32_t> h(nb * 2);
    const char *names[] = {"all", "low", "altcu", "altcu2", "altse"};
    for (const char *m : names) {
        uint32_t mask[8] = {0};
        for (uint32_t b = 0; b < 256; ++b) {
            bool on = !strcmp(m, "even") ? (b & 1u) == 0 : !strcmp(m, "odd") ? (b & 1u) == 1 : !strcmp(m, "low") ? b < 128
                      : !strcmp(m, "altcu") ? ((b >> 5) & 1u) == 0 : !strcmp(m, "altse") ? ((b >> 3) & 1u) == 0 : !strcmp(m, "altcu2") ? ((b >> 6) & 1u) == 0 : !strcmp(m, "pairs") ? (b & 2u) == 0 : !strcmp(m, "quads") ? (b & 4u) == 0 : true;
            if (on) mask[b >> 5] |= 1u << (b & 31u);
        }
        hipStream_t st;
        if (hipExtStreamCreateWithCUMask(&st, 8, mask) != hipSuccess) { printf("%s: stream failed\n", m); continue; }
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        hipLaunchKernelGGL(k_where, dim3(nb), dim3(256), 0, st, d, sink, 10);
        hipStreamSynchronize(st);
        hipEventRecord(e0, st);
        hipLaunchKernelGGL(k_where, dim3(nb), dim3(256), 0, st, d, sink, iters);
        hipEventRecord(e1, st);
        hipStreamSynchronize(st);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(h.data(), d, nb * 8, hipMemcpyDeviceToHost);
        std::set<uint32_t> cus, xccs, ses;
        for (int i = 0; i < nb; ++i) {
            const uint32_t hw = h[2 * i], xcc = h[2 * i + 1] & 0xf;
            const uint32_t cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
            cus.insert((xcc << 16) | (se << 8) | (sh << 4) | cu);
            xccs.insert(xcc);
            ses.insert((xcc << 4) | se);
        }
        printf("%-6s %8.3f ms  distinct CUs %3zu  XCCs %zu  (xcc,se) %zu   first blocks:", m, ms, cus.size(), xccs.size(), ses.size());
        for (int i = 0; i < 8; ++i) printf(" x%u.se%u.cu%u", h[2 * i + 1] & 0xf, (h[2 * i] >> 13) & 7, (h[2 * i] >> 8) & 0xf);
        printf("\n");
        hipStreamDestroy(st);
    }
    return 0;
}

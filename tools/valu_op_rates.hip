// Shader cycles a SIMD of this GPU spends per wave64 VALU instruction, opcode by opcode (8 waves per SIMD, 4 independent chains
// per wave, so neither latency nor issue arbitration between waves limits it): which instructions run at the v_fma_f32 rate and
// which at half or a quarter of it?  The BVH stream kernels are bound by VALU issue and most of their instructions are integer /
// select / convert work, so SQ_INSTS_VALU alone does not say how busy the VALUs are.
// Result (profiles/r04_valu_op_rates.txt): every plain opcode the kernels use -- f32 arithmetic, min / max / med3, integer, logic,
// shifts, bit-field, byte converts, DPP moves -- costs the same as v_fma_f32; compares 1.1 x, v_rcp / v_sqrt 1.9 x.  The VOP2 select
// on a VCC that was not written by the v_cmp right in front of it reads 4 - 6 x HERE, but re-assembling all of the library's
// VOP2 selects as VOP3 changed no kernel's time: an artefact of the probe's instruction stream, not a property to design around.
// (An inline-asm probe must not touch SCC: the compiler keeps the loop condition there.)
// build + run on the GPU box: hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_ops tools/valu_op_rates.hip && /tmp/valu_ops
#include <hip/hip_runtime.h>
#include <cstdio>
#define PROBE(NAME, ASM)                                                                                              \
    __global__ __launch_bounds__(256) void k_##NAME(unsigned *sink, unsigned long long *clk, int iters) {            \
        unsigned a[4], b = threadIdx.x * 2654435761u | 1u, c = 0x3f800001u + threadIdx.x;                              \
        for (int k = 0; k < 4; ++k) a[k] = 0x3f800000u + threadIdx.x * 7u + k;                                         \
        unsigned long long t0, t1;                                                                                    \
        asm volatile("s_mov_b64 s[20:21], 0x5555\n\ts_mov_b64 vcc, 0x3333\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) : : "vcc", "s20", "s21");                                             \
        for (int i = 0; i < iters; ++i) {                                                                             \
            _Pragma("unroll") for (int u = 0; u < 8; ++u) {                                                           \
                _Pragma("unroll") for (int k = 0; k < 4; ++k) asm volatile(ASM : "+v"(a[k]) : "v"(b), "v"(c) : "vcc", "s20", "s21", "s22", "s23"); \
            }                                                                                                         \
        }                                                                                                             \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1));                                             \
        if ((a[0] ^ a[1] ^ a[2] ^ a[3]) == 0x12345678u) sink[0] = a[0];                                               \
        if (blockIdx.x == 0 && threadIdx.x == 0) clk[0] = t1 - t0;                                                    \
    }
PROBE(fma_f32, "v_fma_f32 %0, %0, %1, %2")
PROBE(mul_f32, "v_mul_f32 %0, %0, %1")
PROBE(add_f32, "v_add_f32 %0, %0, %1")
PROBE(max_f32, "v_max_f32 %0, %0, %1")
PROBE(max3_f32, "v_max3_f32 %0, %0, %1, %2")
PROBE(med3_f32, "v_med3_f32 %0, %0, %1, %2")
PROBE(and_b32, "v_and_b32 %0, %0, %1")
PROBE(or_b32, "v_or_b32 %0, %0, %1")
PROBE(xor_b32, "v_xor_b32 %0, %0, %1")
PROBE(lshlrev_b32, "v_lshlrev_b32 %0, 1, %0")
PROBE(lshrrev_b32, "v_lshrrev_b32 %0, 1, %0")
PROBE(add_u32, "v_add_u32 %0, %0, %1")
PROBE(min_u32, "v_min_u32 %0, %0, %1")
PROBE(mov_b32, "v_mov_b32 %0, %1")
PROBE(cndmask_b32, "v_cndmask_b32 %0, %0, %1, vcc")
PROBE(cndmask_e64_sgpr, "v_cndmask_b32_e64 %0, %0, %1, s[20:21]")
PROBE(cndmask_nodep, "v_cndmask_b32 %0, %1, %2, vcc")
PROBE(cndmask_const, "v_cndmask_b32 %0, 0, %0, vcc")
PROBE(cmp_then_cndmask, "v_cmp_le_f32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc")
PROBE(cmp_e64_then_cndmask, "v_cmp_le_f32_e64 s[20:21], %0, %1\n\tv_cndmask_b32_e64 %0, %0, %1, s[20:21]")
PROBE(cmp_then_4cndmask, "v_cmp_le_f32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc\n\tv_cndmask_b32 %0, %0, %2, vcc\n\tv_cndmask_b32 %0, %0, %1, vcc\n\tv_cndmask_b32 %0, %0, %2, vcc")
PROBE(smov_vcc_then_cndmask, "s_mov_b64 vcc, s[20:21]\n\tv_cndmask_b32 %0, %0, %1, vcc")
PROBE(smov_sgpr_then_cndmask, "s_mov_b64 s[22:23], s[20:21]\n\tv_cndmask_b32_e64 %0, %0, %1, s[22:23]")
PROBE(cndmask_e64_vcc, "v_cndmask_b32_e64 %0, %0, %1, vcc")
PROBE(cmp_e64_sgpr, "v_cmp_le_f32_e64 s[20:21], %0, %1")
PROBE(add_co_u32, "v_add_co_u32 %0, vcc, %0, %1")
PROBE(addc_co_u32, "v_addc_co_u32 %0, vcc, %0, %1, vcc")
PROBE(max_u32, "v_max_u32 %0, %0, %1")
PROBE(min_f32, "v_min_f32 %0, %0, %1")
PROBE(cmp_le_f32, "v_cmp_le_f32 vcc, %0, %1")
PROBE(cmp_lt_u32, "v_cmp_lt_u32 vcc, %0, %1")
PROBE(cvt_f32_ubyte0, "v_cvt_f32_ubyte0 %0, %0")
PROBE(cvt_f32_ubyte2, "v_cvt_f32_ubyte2 %0, %0")
PROBE(cvt_f32_u32, "v_cvt_f32_u32 %0, %0")
PROBE(bfe_u32, "v_bfe_u32 %0, %0, 8, 8")
PROBE(and_or_b32, "v_and_or_b32 %0, %0, %1, %2")
PROBE(lshl_or_b32, "v_lshl_or_b32 %0, %0, 1, %2")
PROBE(lshl_add_u32, "v_lshl_add_u32 %0, %0, 1, %2")
PROBE(add3_u32, "v_add3_u32 %0, %0, %1, %2")
PROBE(bfi_b32, "v_bfi_b32 %0, %0, %1, %2")
PROBE(perm_b32, "v_perm_b32 %0, %0, %1, %2")
PROBE(mad_u32_u24, "v_mad_u32_u24 %0, %0, %1, %2")
PROBE(mul_lo_u32, "v_mul_lo_u32 %0, %0, %1")
PROBE(rcp_f32, "v_rcp_f32 %0, %0")
PROBE(sqrt_f32, "v_sqrt_f32 %0, %0")
PROBE(readfirstlane, "v_readfirstlane_b32 s20, %0")
PROBE(mov_dpp, "v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
PROBE(sub_f32_e64_abs, "v_sub_f32_e64 %0, |%0|, %1")
typedef void (*kern_t)(unsigned *, unsigned long long *, int);
static void run(const char *name, kern_t k, int n_cu, unsigned *sink, unsigned long long *clk) {
    const int iters = 2000, waves = 8;
    hipLaunchKernelGGL(k, dim3(n_cu * waves), dim3(256), 0, 0, sink, clk, 10);
    (void)hipDeviceSynchronize();
    hipLaunchKernelGGL(k, dim3(n_cu * waves), dim3(256), 0, 0, sink, clk, iters);
    (void)hipDeviceSynchronize();
    unsigned long long h = 0;
    (void)hipMemcpy(&h, clk, 8, hipMemcpyDeviceToHost);
    // s_memtime counts at a fixed 100 MHz-derived rate on some parts; the v_fma_f32 row calibrates the unit
    printf("%-18s %8.3f s_memtime ticks per wave64 instruction per SIMD\n", name, (double)h / ((double)iters * 32 * waves));
    fflush(stdout);
}
int main() {
    hipDeviceProp_t p;
    (void)hipGetDeviceProperties(&p, 0);
    printf("%s, %d CUs\n", p.name, p.multiProcessorCount);
    unsigned *sink;
    unsigned long long *clk;
    (void)hipMalloc(&sink, 4);
    (void)hipMalloc(&clk, 8);
#define RUN(NAME) run(#NAME, k_##NAME, p.multiProcessorCount, sink, clk)
    RUN(fma_f32); RUN(mul_f32); RUN(add_f32); RUN(max_f32); RUN(max3_f32); RUN(med3_f32); RUN(and_b32); RUN(or_b32); RUN(xor_b32);
    RUN(lshlrev_b32); RUN(lshrrev_b32); RUN(add_u32); RUN(min_u32); RUN(mov_b32); RUN(cndmask_b32); RUN(cndmask_e64_sgpr); RUN(cndmask_nodep); RUN(cndmask_const); RUN(cmp_then_cndmask); RUN(cmp_e64_then_cndmask); RUN(cmp_then_4cndmask); RUN(smov_vcc_then_cndmask); RUN(smov_sgpr_then_cndmask); RUN(cndmask_e64_vcc); RUN(cmp_e64_sgpr); RUN(add_co_u32); RUN(addc_co_u32); RUN(max_u32); RUN(min_f32); RUN(cmp_le_f32); RUN(cmp_lt_u32);
    RUN(cvt_f32_ubyte0); RUN(cvt_f32_ubyte2); RUN(cvt_f32_u32); RUN(bfe_u32); RUN(and_or_b32); RUN(lshl_or_b32); RUN(lshl_add_u32);
    RUN(add3_u32); RUN(bfi_b32); RUN(perm_b32); RUN(mad_u32_u24); RUN(mul_lo_u32); RUN(rcp_f32); RUN(sqrt_f32);
    RUN(readfirstlane); RUN(mov_dpp); RUN(sub_f32_e64_abs);
    return 0;
}

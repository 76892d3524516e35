// Microbenchmark: issue cost of the VALU instructions the SGM kernels are made of (gfx950).
// Every op is an inline-asm instruction (the compiler cannot fold or drop it); each thread runs 8 independent
// dependency chains of the op, W waves per SIMD on every CU.  Prints cycles per wave-instruction per SIMD at
// W = 8, 4, 2 and 1 waves per SIMD (2.0 = one wave64 instruction per 2 cycles = the full rate of a SIMD-32).
//   hipcc --offload-arch=gfx950 -O3 -o valu_rate tools/ubench/valu_rate.hip && ./valu_rate
#include <hip/hip_runtime.h>
#include <stdio.h>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

#define OP1(name, text)                                                                               \
    struct name { static constexpr const char* label = #name;                                         \
        static __device__ __forceinline__ void op(unsigned& a, unsigned b, unsigned c) {              \
            asm volatile(text : "+v"(a) : "v"(b), "v"(c)); } };

// a = op(a, b[, c]); %0 = a (read-write), %1 = b, %2 = c
OP1(v_add_u32,          "v_add_u32 %0, %0, %1")
OP1(v_and_b32,          "v_and_b32 %0, %0, %1")
OP1(v_xor_b32,          "v_xor_b32 %0, %0, %1")
OP1(v_min_u32,          "v_min_u32 %0, %0, %1")
OP1(v_pk_add_u16,       "v_pk_add_u16 %0, %0, %1")
OP1(v_pk_sub_u16,       "v_pk_sub_u16 %0, %0, %1")
OP1(v_pk_min_u16,       "v_pk_min_u16 %0, %0, %1")
OP1(v_pk_max_u16,       "v_pk_max_u16 %0, %0, %1")
OP1(v_alignbit_b32,     "v_alignbit_b32 %0, %0, %1, 16")
OP1(v_perm_b32,         "v_perm_b32 %0, %0, %1, %2")
OP1(v_bcnt_u32_b32,     "v_bcnt_u32_b32 %0, %1, %0")
OP1(v_lshl_add_u32,     "v_lshl_add_u32 %0, %0, 16, %1")
OP1(v_lshl_or_b32,      "v_lshl_or_b32 %0, %0, 16, %1")
OP1(v_and_or_b32,       "v_and_or_b32 %0, %0, %1, %2")
OP1(v_add3_u32,         "v_add3_u32 %0, %0, %1, %2")
OP1(v_min3_u32,         "v_min3_u32 %0, %0, %1, %2")
OP1(v_med3_f32,         "v_med3_f32 %0, %0, %1, %2")
OP1(v_bfe_u32,          "v_bfe_u32 %0, %0, 8, 8")
OP1(v_mad_u32_u24,      "v_mad_u32_u24 %0, %0, %1, %2")
OP1(v_cndmask_b32,      "v_cndmask_b32 %0, %0, %1, vcc")
OP1(v_mov_dpp_row_shr1, "v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf")
OP1(v_mov_dpp_quad_perm,"v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
OP1(v_min_u32_dpp_quad, "v_min_u32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
OP1(v_min_u32_dpp_mirror,"v_min_u32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf")
OP1(v_min_u32_dpp_rowshr,"v_min_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0")
OP1(v_pk_min_u16_dpp_NA, "v_pk_min_u16 %0, %0, %1")   /* VOP3P has no DPP form on gfx950: plain op, for reference */
OP1(v_min_u16_sdwa,     "v_min_u16_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:WORD_0")
OP1(v_add_u32_sdwa_b,   "v_add_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD")
OP1(v_sad_u8,           "v_sad_u8 %0, %0, %1, %2")
OP1(v_permlane16_swap,  "v_permlane16_swap_b32 %0, %1")
OP1(v_permlane32_swap,  "v_permlane32_swap_b32 %0, %1")
OP1(ds_bpermute_b32,    "ds_bpermute_b32 %0, %1, %0\n s_waitcnt lgkmcnt(0)")
OP1(ds_swizzle_b32,     "ds_swizzle_b32 %0, %0 offset:swizzle(SWAP,8)\n s_waitcnt lgkmcnt(0)")

// shader clock held during the loop: s_memtime (shader-clock ticks) against s_memrealtime (100 MHz), MI355X_MICROARCH.md "DVFS"
__device__ unsigned long long g_clk[2];
template <class OP>
__global__ __launch_bounds__(256) void k(unsigned* out, int iters, unsigned seed)
{
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    unsigned a[8];
    for (int i = 0; i < 8; ++i) a[i] = seed * (threadIdx.x + 1) + i * 77;
    unsigned b = seed | 1, c = 0x06040200u ^ (seed & 1);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < 8; ++i) OP::op(a[i], b, c);
    }
    unsigned s = 0;
    for (int i = 0; i < 8; ++i) s ^= a[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (blockIdx.x == gridDim.x / 2 && threadIdx.x == 0) {
        g_clk[0] = __builtin_amdgcn_s_memtime() - c0;
        g_clk[1] = __builtin_amdgcn_s_memrealtime() - r0;
    }
}

template <class OP> int run()
{
    unsigned* d;
    CHK(hipMalloc(&d, 256 * 2048 * 4 * 4));
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    printf("%-24s", OP::label);
    for (int wps : {8, 4, 2, 1}) {
        const int iters = 20000, blocks = 256 * wps;           // wps blocks of 4 waves per CU -> wps waves per SIMD
        hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 10, 12345u);
        CHK(hipDeviceSynchronize());
        CHK(hipEventRecord(e0));
        hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, iters, 12345u);
        CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
        float ms;
        CHK(hipEventElapsedTime(&ms, e0, e1));
        const double winstr = (double)blocks * 4 * iters * 32;             // wave-instructions
        unsigned long long clk[2];
        CHK(hipMemcpyFromSymbol(clk, HIP_SYMBOL(g_clk), sizeof clk));
        const double ghz = clk[1] ? (double)clk[0] / (double)clk[1] * 0.1 : 0.0;   // shader clock the loop ran at
        const double simd_cycles = ms * 1e-3 * ghz * 1e9 * 1024;
        printf("  %d w/SIMD: %5.2f @%.2f GHz", wps, simd_cycles / winstr, ghz);
    }
    printf("   cycles per wave-instr per SIMD at the measured shader clock\n");
    CHK(hipFree(d));
    return 0;
}

int main()
{
#define RUN(n) if (run<n>()) return 1;
    RUN(v_add_u32) RUN(v_and_b32) RUN(v_xor_b32) RUN(v_min_u32) RUN(v_pk_add_u16) RUN(v_pk_sub_u16) RUN(v_pk_min_u16) RUN(v_pk_max_u16)
    RUN(v_alignbit_b32) RUN(v_perm_b32) RUN(v_bcnt_u32_b32) RUN(v_lshl_add_u32) RUN(v_lshl_or_b32) RUN(v_and_or_b32) RUN(v_add3_u32)
    RUN(v_min3_u32) RUN(v_med3_f32) RUN(v_bfe_u32) RUN(v_mad_u32_u24) RUN(v_cndmask_b32)
    RUN(v_mov_dpp_row_shr1) RUN(v_mov_dpp_quad_perm) RUN(v_min_u32_dpp_quad) RUN(v_min_u32_dpp_mirror) RUN(v_min_u32_dpp_rowshr)
    RUN(v_min_u16_sdwa) RUN(v_add_u32_sdwa_b) RUN(v_sad_u8) RUN(v_permlane16_swap) RUN(v_permlane32_swap)
    RUN(ds_bpermute_b32) RUN(ds_swizzle_b32)
    return 0;
}

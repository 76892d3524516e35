// Probe: semantics and issue cost of v_mqsad_pk_u16_u8 used as a "widen 4 bytes to 4 x u16 and accumulate" instruction,
// plus v_cndmask_b32 with a defined condition and dependent chains of packed ops (gfx950).
//   with S1 = 0x000000FF only byte 0 of each 4-byte window takes part (reference bytes equal to 0 are masked), so
//   D.u16[i] = S2.u16[i] + |S0.byte[i] - 255| = S2.u16[i] + 255 - S0.byte[i]      (i = 0..3)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void sem(const uint32_t* in, uint64_t* out, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint64_t acc = 0x0004000300020001ull;
    uint64_t s0 = (uint64_t)in[i] | ((uint64_t)0xDEADBEEFu << 32);
    const uint32_t ref = 0x000000FFu;
    asm volatile("v_mqsad_pk_u16_u8 %0, %1, %2, %0" : "+v"(acc) : "v"(s0), "v"(ref));
    out[i] = acc;
}

template <int OP>
__global__ __launch_bounds__(256) void rate(uint64_t* out, int iters, unsigned seed)
{
    uint64_t a[8];
    unsigned u[8];
    for (int i = 0; i < 8; ++i) { a[i] = seed * (threadIdx.x + 1) + i * 77; u[i] = (unsigned)a[i]; }
    uint64_t s0 = 0x0102030405060708ull * seed;
    unsigned ref = 0xFFu, b = seed | 1;
    asm volatile("v_cmp_gt_u32 vcc, %0, %1" :: "v"(b), "v"(u[0]) : "vcc");
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (OP == 0) asm volatile("v_mqsad_pk_u16_u8 %0, %1, %2, %0" : "+v"(a[i]) : "v"(s0), "v"(ref));
                if (OP == 1) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(u[i]) : "v"(b));
                if (OP == 2) asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"(u[i]) : "v"(b));
                if (OP == 3) asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(u[0]) : "v"(b));          // ONE dependent chain
                if (OP == 4) asm volatile("v_add_u32 %0, %0, %1" : "+v"(u[0]) : "v"(b));             // ONE dependent chain
                if (OP == 5) asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(u[i & 1]) : "v"(b));      // two chains
                if (OP == 6) asm volatile("v_qsad_pk_u16_u8 %0, %1, %2, %0" : "+v"(a[i]) : "v"(s0), "v"(ref));
                if (OP == 7) asm volatile("v_pk_add_u16 %0, %0, %1 op_sel_hi:[1,0]" : "+v"(u[i]) : "v"(b));
                if (OP == 8) asm volatile("v_cmp_gt_u32_e32 vcc, %0, %1" :: "v"(u[i]), "v"(b) : "vcc");
                if (OP == 9) asm volatile("v_cmp_gt_u32_e64 s[20:21], %0, %1" :: "v"(u[i]), "v"(b) : "s20", "s21");
                if (OP == 10) asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(u[i]) : "v"(b), "v"(ref));   // dst not a source
                if (OP == 11) asm volatile("v_addc_co_u32 %0, vcc, %0, %1, vcc" : "+v"(u[i]) : "v"(b) : "vcc");
                if (OP == 12) asm volatile("v_add_co_u32 %0, vcc, %0, %1" : "+v"(u[i]) : "v"(b) : "vcc");
                if (OP == 13) asm volatile("v_add_u32 %0, %0, %1\n s_nop 0" : "+v"(u[i]) : "v"(b));
                if (OP == 14) asm volatile("v_cmp_gt_u32_e32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(u[i]) : "v"(b) : "vcc");
                if (OP == 15) asm volatile("v_cmp_gt_u32_e64 s[20:21], %0, %1\n v_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"(u[i]) : "v"(b) : "s20", "s21");
                if (OP == 16) asm volatile("v_max_i32 %0, %0, %1" : "+v"(u[i]) : "v"(b));
                if (OP == 17) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(u[i]) : "v"(b));
                if (OP == 18) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(u[i]) : "v"(b));
                if (OP == 19) asm volatile("v_or_b32 %0, %0, %1" : "+v"(u[i]) : "v"(b));
                if (OP == 20) asm volatile("v_lshlrev_b32 %0, 3, %0" : "+v"(u[i]));
                if (OP == 21) asm volatile("v_mov_b32 %0, %1" : "=v"(u[i]) : "v"(b));
                if (OP == 22) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0xe4" : "+v"(u[i]) : "v"(b), "v"(ref));
            }
    }
    uint64_t s = 0;
    for (int i = 0; i < 8; ++i) s ^= a[i] ^ u[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int OP> int run(const char* label)
{
    uint64_t* d;
    CHK(hipMalloc(&d, 256 * 2048 * 8));
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    printf("%-34s", label);
    for (int wps : {8, 2, 1}) {
        const int iters = 2000, blocks = 256 * wps;
        hipLaunchKernelGGL(rate<OP>, dim3(blocks), dim3(256), 0, 0, d, 10, 12345u);
        CHK(hipDeviceSynchronize());
        CHK(hipEventRecord(e0));
        hipLaunchKernelGGL(rate<OP>, dim3(blocks), dim3(256), 0, 0, d, iters, 12345u);
        CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
        float ms;
        CHK(hipEventElapsedTime(&ms, e0, e1));
        printf("  %d waves/SIMD: %6.2f", wps, ms * 1e-3 * 2.4e9 * 1024 / ((double)blocks * 4 * iters * 32));
    }
    printf("   cycles per wave-instr per SIMD (at 2.4 GHz)\n");
    CHK(hipFree(d));
    return 0;
}

int main()
{
    const int n = 8;
    uint32_t h_in[n] = {0x00000000u, 0xFFFFFFFFu, 0x04030201u, 0x80FF0001u, 0x18171615u, 0x7F7F7F7Fu, 0x000000FFu, 0xFF000000u};
    uint32_t* d_in; uint64_t* d_out; uint64_t h_out[n];
    CHK(hipMalloc(&d_in, sizeof h_in)); CHK(hipMalloc(&d_out, sizeof h_out));
    CHK(hipMemcpy(d_in, h_in, sizeof h_in, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(sem, dim3(1), dim3(64), 0, 0, d_in, d_out, n);
    CHK(hipMemcpy(h_out, d_out, sizeof h_out, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int i = 0; i < n; ++i) {
        uint64_t want = 0;
        for (int k = 0; k < 4; ++k) want |= (uint64_t)((k + 1) + 255 - ((h_in[i] >> (8 * k)) & 0xFF)) << (16 * k);
        printf("in %08x -> %016llx  want %016llx %s\n", h_in[i], (unsigned long long)h_out[i], (unsigned long long)want,
               h_out[i] == want ? "ok" : "DIFFERENT");
        bad += h_out[i] != want;
    }
    printf("semantics: %s\n", bad ? "NOT as assumed" : "as assumed (acc + 255 - byte, per u16 lane)");
    if (run<0>("v_mqsad_pk_u16_u8")) return 1;
    if (run<6>("v_qsad_pk_u16_u8")) return 1;
    if (run<1>("v_cndmask_b32 (vcc set)")) return 1;
    if (run<2>("v_cndmask_b32_e64 (sgpr pair)")) return 1;
    if (run<3>("v_pk_add_u16, one dependent chain")) return 1;
    if (run<5>("v_pk_add_u16, two chains")) return 1;
    if (run<4>("v_add_u32, one dependent chain")) return 1;
    if (run<7>("v_pk_add_u16 op_sel_hi")) return 1;
    if (run<8>("v_cmp_gt_u32_e32 -> vcc")) return 1;
    if (run<9>("v_cmp_gt_u32_e64 -> sgpr pair")) return 1;
    if (run<10>("v_cndmask_b32 vcc, dst != src")) return 1;
    if (run<11>("v_addc_co_u32 (vcc in/out)")) return 1;
    if (run<12>("v_add_co_u32 (vcc out)")) return 1;
    if (run<13>("v_add_u32 + s_nop 0 (two slots)")) return 1;
    if (run<14>("v_cmp_e32 + v_cndmask vcc (pair)")) return 1;
    if (run<15>("v_cmp_e64 + v_cndmask_e64 (pair)")) return 1;
    if (run<16>("v_max_i32")) return 1;
    if (run<17>("v_mul_u32_u24")) return 1;
    if (run<18>("v_sub_u32")) return 1;
    if (run<19>("v_or_b32")) return 1;
    if (run<20>("v_lshlrev_b32")) return 1;
    if (run<21>("v_mov_b32")) return 1;
    if (run<22>("v_bitop3_b32")) return 1;
    return 0;
}

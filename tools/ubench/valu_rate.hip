// Microbenchmark: issue rate of the VALU ops the aggregation kernel is made of (gfx950).
// Each wave runs independent chains of one op; 8 waves per SIMD, every CU busy.  Prints cycles per
// wave-instruction per SIMD (2.0 = full rate on a SIMD-32).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned short us2 __attribute__((ext_vector_type(2)));
#define AS_U(v) __builtin_bit_cast(unsigned, v)
#define AS_P(v) __builtin_bit_cast(us2, v)
template <int OP>
__global__ __launch_bounds__(256) void k(unsigned* out, int iters, unsigned seed)
{
    unsigned a[8];
    for (int i = 0; i < 8; ++i) a[i] = seed * (threadIdx.x + 1) + i * 77;
    const unsigned c = seed | 1;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (OP == 0) a[i] = a[i] + c;
                if (OP == 1) a[i] = AS_U(AS_P(a[i]) + AS_P(c));
                if (OP == 2) a[i] = AS_U(__builtin_elementwise_min(AS_P(a[i]), AS_P(c + i)));
                if (OP == 3) a[i] = min(a[i], c + i) + 1;            // 2 ops
                if (OP == 4) a[i] = __builtin_amdgcn_alignbit(a[i], c, 16);
                if (OP == 5) a[i] = __builtin_amdgcn_perm(a[i], c, 0x06040200u);
                if (OP == 6) a[i] = (unsigned)__popc(a[i] ^ c) + a[i];   // xor + bcnt(acc) = 2 ops
                if (OP == 7) a[i] = a[i] & c;
                if (OP == 8) a[i] = (unsigned)__builtin_amdgcn_update_dpp((int)c, (int)a[i], 0x111, 0xF, 0xF, false);
                if (OP == 9) a[i] = AS_U(AS_P(a[i]) - AS_P(c));
            }
    }
    unsigned s = 0;
    for (int i = 0; i < 8; ++i) s ^= a[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int OP> void run(const char* name, int ops_per_inner)
{
    unsigned* d; hipMalloc(&d, 256 * 2048 * 4 * 4);
    const int iters = 2000, blocks = 256 * 8;   // 8 blocks of 4 waves per CU -> 8 waves per SIMD
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 10, 12345u);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, iters, 12345u);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double winstr = (double)blocks * 4 * iters * 32 * ops_per_inner;      // wave-instructions
    const double simd_cycles = ms * 1e-3 * 2.4e9 * 1024;
    printf("%-28s %.2f cycles per wave-instr per SIMD (%.3f ms)\n", name, simd_cycles / winstr, ms);
    hipFree(d);
}
int main()
{
    run<0>("v_add_u32", 1); run<1>("v_pk_add_u16", 1); run<2>("v_pk_min_u16", 1); run<3>("v_min_u32+v_add", 2);
    run<4>("v_alignbit_b32", 1); run<5>("v_perm_b32", 1); run<6>("v_xor+v_bcnt", 2); run<7>("v_and_b32", 1);
    run<8>("v_mov_b32_dpp row_shr", 1); run<9>("v_pk_sub_u16", 1);
    return 0;
}

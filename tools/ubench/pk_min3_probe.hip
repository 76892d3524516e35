// Probe (gfx950): is v_pk_minimum3_f16 an exact 3-way minimum of packed UNSIGNED 16-bit integers below 0x7C00?
// Positive finite binary16 values order like their bit patterns, so min(f16) == min(u16) bit for bit -- if the instruction neither
// flushes denormals (patterns 0x0001..0x03FF) nor canonicalises anything.  Checked for every pair of 16-bit patterns below
// 0x7C00 against a third operand sweeping a few values, both halves, and the instruction's issue cost is measured beside
// v_pk_min_u16.   hipcc --offload-arch=gfx950 -O3 -o pk_min3_probe tools/ubench/pk_min3_probe.hip && ./pk_min3_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

static __device__ __forceinline__ unsigned pk_min3_f16(unsigned a, unsigned b, unsigned c)
{
    unsigned r;
    asm volatile("v_pk_minimum3_f16 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
static __device__ __forceinline__ unsigned pk_min_f16(unsigned a, unsigned b)
{
    unsigned r;
    asm volatile("v_pk_min_f16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

__global__ void check(unsigned long long* bad, unsigned* first)
{
    const unsigned a = blockIdx.x * blockDim.x + threadIdx.x;        // 0 .. 0x7BFF
    if (a >= 0x7C00u) return;
    const unsigned cs[8] = {0u, 1u, 0x3FFu, 0x400u, 255u, 265u, 0x7BFFu, 12345u};
    unsigned long long n = 0;
    for (unsigned b = 0; b < 0x7C00u; ++b) {
        for (int k = 0; k < 8; ++k) {
            const unsigned c = cs[k];
            const unsigned want = min(a, min(b, c));
            const unsigned lo_hi = a | (b << 16), x = b | (c << 16), y = c | (a << 16);     // halves carry different triples
            const unsigned got = pk_min3_f16(lo_hi, x, y);
            const unsigned want2 = min(a, min(b, c)) | (min(b, min(c, a)) << 16);
            const unsigned got2 = pk_min_f16(a | (b << 16), b | (c << 16));
            const unsigned want3 = min(a, b) | (min(b, c) << 16);
            if (got != want2 || got2 != want3 || (got & 0xFFFFu) != want) {
                if (n == 0) { first[0] = a; first[1] = b; first[2] = c; first[3] = got; first[4] = want2; first[5] = got2; first[6] = want3; }
                ++n;
            }
        }
    }
    if (n) atomicAdd(bad, n);
}

template <int KIND>
__global__ __launch_bounds__(256) void rate(unsigned* out, int iters, unsigned seed)
{
    unsigned a[8];
    for (int i = 0; i < 8; ++i) a[i] = (seed * (threadIdx.x + 1) + i * 77) & 0x3FFF3FFFu;
    unsigned b = (seed | 1) & 0x3FFF3FFFu, c = 0x01000100u ^ (seed & 1);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (KIND == 0) asm volatile("v_pk_min_u16 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if (KIND == 1) asm volatile("v_pk_min_f16 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if (KIND == 2) asm volatile("v_pk_minimum3_f16 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
                if (KIND == 3) asm volatile("v_min3_u32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
            }
    }
    unsigned s = 0;
    for (int i = 0; i < 8; ++i) s ^= a[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int KIND> static int time_it(const char* name)
{
    unsigned* d;
    CHK(hipMalloc(&d, 256 * 2048 * 4));
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    const int iters = 20000, blocks = 256 * 4;
    hipLaunchKernelGGL(rate<KIND>, dim3(blocks), dim3(256), 0, 0, d, 10, 12345u);
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(e0));
    hipLaunchKernelGGL(rate<KIND>, dim3(blocks), dim3(256), 0, 0, d, iters, 12345u);
    CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
    float ms;
    CHK(hipEventElapsedTime(&ms, e0, e1));
    const double winstr = (double)blocks * 4 * iters * 32;
    printf("%-22s %.2f cycles per wave-instruction per SIMD (4 waves/SIMD, at a nominal 2.4 GHz)\n", name, ms * 1e-3 * 2.4e9 * 1024 / winstr);
    CHK(hipFree(d));
    return 0;
}

int main()
{
    unsigned long long* bad; unsigned* first;
    CHK(hipMalloc(&bad, 8)); CHK(hipMalloc(&first, 32));
    CHK(hipMemset(bad, 0, 8)); CHK(hipMemset(first, 0, 32));
    hipLaunchKernelGGL(check, dim3((0x7C00 + 255) / 256), dim3(256), 0, 0, bad, first);
    CHK(hipDeviceSynchronize());
    unsigned long long nb; unsigned f[8];
    CHK(hipMemcpy(&nb, bad, 8, hipMemcpyDeviceToHost)); CHK(hipMemcpy(f, first, 32, hipMemcpyDeviceToHost));
    printf("v_pk_minimum3_f16 / v_pk_min_f16 as unsigned minima of 16-bit patterns < 0x7C00: %llu mismatches of %llu triples\n", nb,
           (unsigned long long)0x7C00 * 0x7C00 * 8);
    if (nb) printf("  first: a=%#x b=%#x c=%#x  min3 got %#x want %#x; min got %#x want %#x\n", f[0], f[1], f[2], f[3], f[4], f[5], f[6]);
    if (time_it<0>("v_pk_min_u16")) return 1;
    if (time_it<1>("v_pk_min_f16")) return 1;
    if (time_it<2>("v_pk_minimum3_f16")) return 1;
    if (time_it<3>("v_min3_u32")) return 1;
    return nb ? 2 : 0;
}

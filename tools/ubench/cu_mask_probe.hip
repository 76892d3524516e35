// Probe (gfx950): which compute units does a stream created with hipExtStreamCreateWithCUMask really run on?  The library's
// sgm_set_stage_cus assumes bit i of the mask = CU (i / xcds) of XCD (i % xcds), so that bits [first * xcds, (first + count) * xcds)
// are `count` CUs of EVERY XCD (csrc/sgm_runtime.hip).  Every workgroup records its XCC id and its HW_ID (SE / SH / CU); the host
// prints, per mask, how many distinct CUs of each XCD were used.
//   hipcc --offload-arch=gfx950 -O2 -o cu_mask_probe tools/ubench/cu_mask_probe.hip && ./cu_mask_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#include <set>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void where(unsigned* out, int spin)
{
    unsigned xcc, hwid;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    // keep the workgroup resident for a while so that the grid spreads over every CU the queue may use
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    while (__builtin_amdgcn_s_memtime() - t0 < (unsigned long long)spin) { }
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = xcc; out[2 * blockIdx.x + 1] = hwid; }
}

static int run(const char* label, hipStream_t st, unsigned* d_out, unsigned* h_out, int blocks)
{
    CHK(hipMemsetAsync(d_out, 0xFF, blocks * 8, st));
    hipLaunchKernelGGL(where, dim3(blocks), dim3(64), 0, st, d_out, 200000);
    CHK(hipStreamSynchronize(st));
    CHK(hipMemcpy(h_out, d_out, blocks * 8, hipMemcpyDeviceToHost));
    std::set<unsigned> cus[16];
    for (int b = 0; b < blocks; ++b) {
        const unsigned xcc = h_out[2 * b] & 0xF, hw = h_out[2 * b + 1];
        // HW_ID (gfx9): wave_id[3:0] simd_id[5:4] pipe_id[7:6] cu_id[11:8] sh_id[12] se_id[15:13] ...
        cus[xcc].insert((hw >> 8) & 0xFF);                  // cu_id + sh_id + se_id: one value per physical CU of the XCD
    }
    printf("%-34s CUs used per XCD:", label);
    int total = 0;
    for (int x = 0; x < 8; ++x) { printf(" %2zu", cus[x].size()); total += (int)cus[x].size(); }
    printf("   total %d\n", total);
    return 0;
}

int main()
{
    int cus = 0, xcds = 8;
    CHK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
    const int per = cus / xcds, blocks = 8192;
    unsigned *d_out, *h_out = new unsigned[2 * blocks];
    CHK(hipMalloc(&d_out, blocks * 8));
    hipStream_t plain;
    CHK(hipStreamCreate(&plain));
    if (run("ordinary stream", plain, d_out, h_out, blocks)) return 1;
    struct { const char* label; int first, count; } cases[] = {{"bits [0, 8 xcds): 8 CUs per XCD?", 0, 8}, {"bits [8 xcds, 32 xcds): 24 per XCD?", 8, 24},
                                                                {"bits [0, 2 xcds): 2 CUs per XCD?", 0, 2}, {"bits [0, 32 xcds): all", 0, 32}};
    for (auto& c : cases) {
        uint32_t mask[32];
        memset(mask, 0, sizeof mask);
        for (int b = c.first * xcds; b < (c.first + c.count) * xcds; ++b) mask[b / 32] |= 1u << (b % 32);
        hipStream_t s;
        CHK(hipExtStreamCreateWithCUMask(&s, (uint32_t)((per * xcds + 31) / 32), mask));
        if (run(c.label, s, d_out, h_out, blocks)) return 1;
        CHK(hipStreamDestroy(s));
    }
    // the other reading: XCD-major bits (bit i = CU i % per of XCD i / per): the first 64 bits = all of XCD 0 and XCD 1
    return 0;
}

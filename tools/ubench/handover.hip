// Probe: what a per-row, both-ways hand-over between neighbouring workgroups costs on gfx950 -- the synchronisation a
// row-synchronous sweep kernel (NOTES.md section 9: column strips marching down the rows, computing the three downward
// directions together) would need at every row: a strip's first / last column of row r feeds the neighbouring strips' row r+1
// in BOTH directions, so a chain of workgroups moves in lockstep.
//
//   kernel: workgroup b of a chain publishes, per row, one 128-byte edge to each neighbour as 16 data-tagged 8-byte granules
//           {payload, row} (relaxed agent-scope stores, double-buffered by row parity), then polls both neighbours' edges of the
//           same row (agent-scope loads, bounded), __syncthreads, next row.  `work` dependent FMAs per row stand in for the
//           row's arithmetic.  `stride` spaces the chain's blocks in the grid: 1 = neighbours on different XCDs (blocks are
//           dealt round-robin to the 8 XCDs), 8 = the whole chain on one XCD (one L2).
//   output: microseconds per row (whole launch / rows), payload checksum verified.
//
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench/handover.out tools/ubench/handover.hip && tools/ubench/handover.out
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

#define EDGE 16            // granules per edge
#define POLL_LIMIT (1 << 20)

// edges[chain][wg][side 0 = towards wg-1, 1 = towards wg+1][parity][EDGE]
__global__ __launch_bounds__(256) void lockstep_k(uint64_t* edges, int rows, int nwg, int stride, int work, unsigned tag0, unsigned* fail,
                                                   unsigned long long* sums)
{
    if (blockIdx.x % stride) return;
    const int b = (blockIdx.x / stride) % nwg, chain = (blockIdx.x / stride) / nwg;
    uint64_t* mine = edges + ((size_t)chain * nwg + b) * 2 * 2 * EDGE;
    const uint64_t* left = b > 0 ? edges + ((size_t)chain * nwg + b - 1) * 2 * 2 * EDGE + 1 * 2 * EDGE : nullptr;   // its edge towards wg+1
    const uint64_t* right = b + 1 < nwg ? edges + ((size_t)chain * nwg + b + 1) * 2 * 2 * EDGE : nullptr;             // its edge towards wg-1
    const int t = threadIdx.x;
    float acc = (float)(t + b);
    unsigned long long sum = 0;
    __shared__ unsigned s_bad;
    if (t == 0) s_bad = 0;
    __syncthreads();
    for (int r = 1; r <= rows; ++r) {
        for (int k = 0; k < work; ++k) acc = acc * 1.0000001f + 0.5f;                    // the row's arithmetic (dependent chain)
        const unsigned tag = tag0 + (unsigned)r;
        const int par = r & 1;
        if (t < 2 * EDGE) {                                                              // publish both edges
            const int side = t / EDGE, j = t % EDGE;
            const uint64_t g = ((uint64_t)tag << 32) | (uint32_t)(b * 1000 + r + j);
            __hip_atomic_store(mine + (side * 2 + par) * EDGE + j, g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (t >= 64 && t < 64 + 2 * EDGE) {                                              // another wave polls the neighbours' edges
            const int side = (t - 64) / EDGE, j = (t - 64) % EDGE;
            const uint64_t* src = side == 0 ? left : right;
            if (src) {
                uint64_t g = 0;
                int polls = 0;
                for (; polls < POLL_LIMIT; ++polls) {
                    g = __hip_atomic_load(src + par * EDGE + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if ((unsigned)(g >> 32) == tag) break;
                    __builtin_amdgcn_s_sleep(1);
                }
                if (polls == POLL_LIMIT) atomicAdd(&s_bad, 1u);
                sum += (uint32_t)g;
            }
        }
        __syncthreads();
        if (s_bad) break;                                                                // a neighbour never arrived: everybody leaves
    }
    if (t == 0 && s_bad) atomicAdd(fail, 1u);
    if (t >= 64 && t < 64 + 2 * EDGE) atomicAdd(sums, sum);
    if (acc == 12345.678f) sums[1] = 1;                                                  // keeps the arithmetic alive
}

int main()
{
    hipDeviceProp_t prop;
    CHK(hipGetDeviceProperties(&prop, 0));
    printf("# %s, %d CUs; microseconds per row of a lockstep chain of workgroups (both-ways 128-byte edge hand-over per row)\n", prop.gcnArchName,
           prop.multiProcessorCount);
    const int rows = 2000;
    uint64_t* edges;
    unsigned* fail;
    unsigned long long* sums;
    const size_t max_wg = 4096;
    CHK(hipMalloc(&edges, max_wg * 2 * 2 * EDGE * 8));
    CHK(hipMalloc(&fail, 4));
    CHK(hipMalloc(&sums, 16));
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0));
    CHK(hipEventCreate(&e1));
    unsigned tag0 = 0x1000;
    printf("%8s %8s %8s %8s %12s %10s\n", "chains", "wgs", "stride", "work", "us_per_row", "check");
    const int cfgs[][4] = {                                      // chains, workgroups per chain, stride, work
        {1, 2, 1, 0},   {1, 2, 8, 0},   {1, 10, 1, 0},  {1, 10, 8, 0},  {1, 20, 1, 0},  {1, 20, 8, 0},
        {16, 10, 1, 0}, {16, 10, 1, 100}, {16, 10, 1, 400}, {16, 20, 1, 0}, {16, 20, 1, 100}, {16, 20, 1, 400}, {32, 20, 1, 100}, {32, 20, 1, 400},
        {1, 1, 1, 100}, {1, 1, 1, 400},                         // no neighbours: the arithmetic alone
    };
    for (auto& c : cfgs) {
        const int chains = c[0], nwg = c[1], stride = c[2], work = c[3];
        const int grid = chains * nwg * stride;
        if ((size_t)chains * nwg > max_wg || grid > prop.multiProcessorCount * 4) continue;   // every block must be resident at once
        CHK(hipMemset(edges, 0, max_wg * 2 * 2 * EDGE * 8));
        CHK(hipMemset(fail, 0, 4));
        CHK(hipMemset(sums, 0, 16));
        float best = 1e30f;
        for (int rep = 0; rep < 3; ++rep) {
            CHK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(lockstep_k, dim3(grid), dim3(256), 0, 0, edges, rows, nwg, stride, work, tag0, fail, sums);
            CHK(hipEventRecord(e1, 0));
            CHK(hipEventSynchronize(e1));
            float ms;
            CHK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
            tag0 += rows + 16;
        }
        unsigned h_fail;
        unsigned long long h_sum;
        CHK(hipMemcpy(&h_fail, fail, 4, hipMemcpyDeviceToHost));
        CHK(hipMemcpy(&h_sum, sums, 8, hipMemcpyDeviceToHost));
        // expected payload sum: every interior edge read once per row per rep by the neighbour
        unsigned long long want = 0;
        for (int b = 0; b < nwg; ++b)
            for (int side = 0; side < 2; ++side) {
                const int nb = side == 0 ? b - 1 : b + 1;
                if (nb < 0 || nb >= nwg) continue;
                for (int r = 1; r <= rows; ++r)
                    for (int j = 0; j < EDGE; ++j) want += (unsigned)(nb * 1000 + r + j);
            }
        want *= 3ull * chains;
        printf("%8d %8d %8d %8d %12.3f %10s\n", chains, nwg, stride, work, best * 1e3 / rows,
               h_fail ? "TIMEOUT" : (h_sum == want ? "ok" : "MISMATCH"));
    }
    return 0;
}

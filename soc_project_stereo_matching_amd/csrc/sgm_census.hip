#include "sgm_common.hpp"

// ============================================================================================
// census 5x5  (ref :134-159)
// ============================================================================================

// A workgroup computes a 64 x 16 block: the 68 x 20 bytes it needs (2-pixel halo) are staged in LDS once (5.3 global
// byte loads per thread instead of 25 per pixel); a thread then owns one column of four consecutive rows and reads its
// 5 x 8 window bytes from LDS once for all four pixels.
#define CEN_BW 64
#define CEN_BH 16
#define CEN_LD 72                                                       // LDS row stride in bytes (68 used)
// need (row tiles): one byte per 64 x 16 block of the image, 0 = nobody on this GPU reads the block's census words
__global__ __launch_bounds__(256) void sgm_census_k(const uint8_t* __restrict__ left, const uint8_t* __restrict__ right,
                                                    uint32_t* __restrict__ cl, uint32_t* __restrict__ cr, int W, int H,
                                                    const uint8_t* __restrict__ need, int keep_border)
{
    if (need && !need[blockIdx.y * gridDim.x + blockIdx.x]) return;
    __shared__ uint8_t tile[(CEN_BH + 4) * CEN_LD];
    const size_t frame_px = (size_t)(blockIdx.z >> 1) * W * H;         // batch: z = 2 * frame + image
    const uint8_t* img = ((blockIdx.z & 1) ? right : left) + frame_px;
    uint32_t* out = ((blockIdx.z & 1) ? cr : cl) + frame_px;
    const int x0 = blockIdx.x * CEN_BW, y0 = blockIdx.y * CEN_BH;
    // positions outside the image are clamped: only pixels of the 2-pixel border ever see them, and those get 0
    for (int t = threadIdx.x; t < (CEN_BH + 4) * (CEN_BW + 4); t += 256) {
        const int r = t / (CEN_BW + 4), c = t % (CEN_BW + 4);
        const int yy = min(max(y0 + r - 2, 0), H - 1), xx = min(max(x0 + c - 2, 0), W - 1);
        tile[r * CEN_LD + c] = img[(size_t)yy * W + xx];
    }
    __syncthreads();
    const int cx = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int x = x0 + cx;
    if (x >= W) return;
    unsigned v[8][5];
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
        for (int c = 0; c < 5; ++c) v[r][c] = tile[(rg * 4 + r) * CEN_LD + cx + c];
    // border of 2 px is never written by the reference (zero-initialised statics, Q3); also nothing
    // at all is written for images with W <= 5 or H <= 5 (ref :136).  keep_border: exactly that -- those words keep what an
    // earlier frame (of another shape) left at the same linear index; otherwise they are written as 0
    const bool col_ok = W > 5 && H > 5 && x >= 2 && x < W - 2;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int y = y0 + rg * 4 + i;
        if (y >= H) break;
        uint32_t bits = 0;
        const bool interior = col_ok && y >= 2 && y < H - 2;
        if (!interior && keep_border) continue;
        if (interior) {
            const unsigned centre = v[i + 2][2];
#pragma unroll
            for (int r = 0; r < 5; ++r)
#pragma unroll
                for (int c = 0; c < 5; ++c) bits = (bits << 1) | (unsigned)(v[i + r][c] < centre);   // raster order, ref :146-154
        }
        out[(size_t)y * W + x] = bits;
    }
}

// ============================================================================================
// matching cost  (ref :161-196): one thread = 16 consecutive disparities of one pixel
// ============================================================================================

__global__ __launch_bounds__(256) void sgm_cost_k(const uint32_t* __restrict__ cl, const uint32_t* __restrict__ cr,
                                                  uint8_t* __restrict__ cost, int W, int H, int D, int Dp, int dmin)
{
    const int chunks = Dp >> 4;                       // Dp is a multiple of 32
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long total = (long long)W * H * chunks;
    if (t >= total) return;
    cl += (size_t)blockIdx.y * W * H;                                  // batch: y = frame
    cr += (size_t)blockIdx.y * W * H;
    cost += (size_t)blockIdx.y * W * H * Dp;
    const int chunk = (int)(t % chunks);
    const long long pix = t / chunks;
    const int x = (int)(pix % W);
    const uint32_t a = cl[pix];
    const uint32_t* rrow = cr + (pix - x);
    unsigned w[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        unsigned word = 0;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int di = chunk * 16 + q * 4 + b;    // index into the volume
            const int xr = x - (dmin + di);
            unsigned c = 127u;                        // off-image: UINT8_MAX/2 (ref :170-171)
            if (di < D && xr >= 0 && xr < W) c = (unsigned)__popc(a ^ rrow[xr]);
            word |= c << (8 * b);
        }
        w[q] = word;
    }
    *reinterpret_cast<uint4*>(cost + pix * Dp + chunk * 16) = make_uint4(w[0], w[1], w[2], w[3]);
}


// ============================================================================================
// Extension (the reference only has 5x5): census over any odd window cw x ch of at most 64 pixels,
// u64 words, same bit order (raster, first comparison in the highest bit, centre included) and zero border; and the
// Hamming cost of those words as a materialised u8 volume for the volume-fed aggregation kernels.
// ============================================================================================

__global__ __launch_bounds__(256) void sgm_census_window_k(const uint8_t* __restrict__ left, const uint8_t* __restrict__ right,
                                                           unsigned long long* __restrict__ cl, unsigned long long* __restrict__ cr,
                                                           int W, int H, int cw, int ch)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= W || y >= H) return;
    const size_t frame_px = (size_t)(blockIdx.z >> 1) * W * H;
    const uint8_t* img = ((blockIdx.z & 1) ? right : left) + frame_px;
    unsigned long long* out = ((blockIdx.z & 1) ? cr : cl) + frame_px;
    const int rx = cw / 2, ry = ch / 2;
    unsigned long long bits = 0;
    if (W > cw && H > ch && x >= rx && x < W - rx && y >= ry && y < H - ry) {
        const unsigned centre = img[(size_t)y * W + x];
        for (int r = -ry; r <= ry; ++r)
            for (int c = -rx; c <= rx; ++c) bits = (bits << 1) | (unsigned long long)(img[(size_t)(y + r) * W + (x + c)] < centre);
    }
    out[(size_t)y * W + x] = bits;
}

__global__ __launch_bounds__(256) void sgm_cost64_k(const unsigned long long* __restrict__ cl, const unsigned long long* __restrict__ cr,
                                                    uint8_t* __restrict__ cost, int W, int H, int D, int Dp, int dmin)
{
    const int chunks = Dp >> 4;
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long total = (long long)W * H * chunks;
    if (t >= total) return;
    cl += (size_t)blockIdx.y * W * H;
    cr += (size_t)blockIdx.y * W * H;
    cost += (size_t)blockIdx.y * W * H * Dp;
    const int chunk = (int)(t % chunks);
    const long long pix = t / chunks;
    const int x = (int)(pix % W);
    const unsigned long long a = cl[pix];
    const unsigned long long* rrow = cr + (pix - x);
    unsigned w[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        unsigned word = 0;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int di = chunk * 16 + q * 4 + b;
            const int xr = x - (dmin + di);
            unsigned c = 127u;                        // off-image: UINT8_MAX/2 (ref :170-171); padding cells too (never read back)
            if (di < D && xr >= 0 && xr < W) c = (unsigned)__popcll(a ^ rrow[xr]);
            word |= c << (8 * b);
        }
        w[q] = word;
    }
    *reinterpret_cast<uint4*>(cost + pix * Dp + chunk * 16) = make_uint4(w[0], w[1], w[2], w[3]);
}


extern "C" {

int sgmd_census_window(int ord, void* stream, const sgmd_geom* g, int cw, int ch, const void* left, const void* right, void* cl64,
                       void* cr64)
{
    HIP_TRY(hipSetDevice(ord));
    dim3 grid((g->W + 63) / 64, (g->H + 3) / 4, 2 * g->B);
    hipLaunchKernelGGL(sgm_census_window_k, grid, dim3(256), 0, (hipStream_t)stream, (const uint8_t*)left, (const uint8_t*)right,
                       (unsigned long long*)cl64, (unsigned long long*)cr64, g->W, g->H, cw, ch);
    HIP_TRY(hipGetLastError());
    return 0;
}

int sgmd_cost64(int ord, void* stream, const sgmd_geom* g, const void* cl64, const void* cr64, void* cost)
{
    HIP_TRY(hipSetDevice(ord));
    const long long total = (long long)g->W * g->H * (g->Dp / 16);
    dim3 grid((unsigned)((total + 255) / 256), g->B);
    hipLaunchKernelGGL(sgm_cost64_k, grid, dim3(256), 0, (hipStream_t)stream, (const unsigned long long*)cl64,
                       (const unsigned long long*)cr64, (uint8_t*)cost, g->W, g->H, g->D, g->Dp, g->dmin);
    HIP_TRY(hipGetLastError());
    return 0;
}

size_t sgmd_census_slack(const sgmd_geom* g)
{
    // lowest census-right index read is p - (dmin + Dp - 1) with p >= 0; round up to 256 B
    return (((size_t)g->dmin + g->Dp + 8) * sizeof(uint32_t) + 255) & ~(size_t)255;
}

void sgmd_census_blocks(const sgmd_geom* g, int* blocks_x, int* blocks_y)
{
    *blocks_x = (g->W + CEN_BW - 1) / CEN_BW;
    *blocks_y = (g->H + CEN_BH - 1) / CEN_BH;
}

int sgmd_census(int ord, void* stream, const sgmd_geom* g, const void* left, const void* right, void* cl, void* cr, const void* need,
                int keep_border)
{
    HIP_TRY(hipSetDevice(ord));
    dim3 grid((g->W + CEN_BW - 1) / CEN_BW, (g->H + CEN_BH - 1) / CEN_BH, 2 * g->B);
    hipLaunchKernelGGL(sgm_census_k, grid, dim3(256), 0, (hipStream_t)stream, (const uint8_t*)left,
                       (const uint8_t*)right, (uint32_t*)cl, (uint32_t*)cr, g->W, g->H, (const uint8_t*)need, keep_border);
    HIP_TRY(hipGetLastError());
    return 0;
}

int sgmd_cost(int ord, void* stream, const sgmd_geom* g, const void* cl, const void* cr, void* cost)
{
    HIP_TRY(hipSetDevice(ord));
    const long long total = (long long)g->W * g->H * (g->Dp / 16);
    dim3 grid((unsigned)((total + 255) / 256), g->B);
    hipLaunchKernelGGL(sgm_cost_k, grid, dim3(256), 0, (hipStream_t)stream, (const uint32_t*)cl, (const uint32_t*)cr,
                       (uint8_t*)cost, g->W, g->H, g->D, g->Dp, g->dmin);
    HIP_TRY(hipGetLastError());
    return 0;
}

}  // extern "C"

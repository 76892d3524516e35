#include "sgm_common.hpp"

// ============================================================================================
// speckle removal  (ref :585-642): connected-component labelling by union-find.  The reference's
// breadth-first flood defines components of the symmetric relation "8-neighbours, both valid,
// |delta| <= diff", so the result does not depend on traversal order.
//
// Two levels keep global atomics and pointer chasing rare: (A) every 64x16 tile is labelled
// entirely in LDS and leaves one root per tile-local component, with its pixel count; (B) only
// pixels on tile borders union roots across tiles in global memory; (C) every tile-local root adds
// its count to its final root; (D) pixels whose component total is < min_area become +INF.
// ============================================================================================

#define SPK_TW 64
// tile height is a template parameter: 16 rows for one frame per launch (more tiles = more workgroups), 32 for
// batches (fewer tile-border pixels for the global union pass; measured 0.25 / 0.23 / 0.26 ms per 8 frames at 16 / 32 / 64)

template <typename P>
static __device__ __forceinline__ int uf_find(P lab, int x)
{
    int p = __hip_atomic_load(lab + x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    while (p != x) { x = p; p = __hip_atomic_load(lab + x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    return x;
}
template <typename P>
static __device__ __forceinline__ void uf_union(P lab, int a, int b)
{
    for (;;) {
        a = uf_find(lab, a);
        b = uf_find(lab, b);
        if (a == b) return;
        if (a < b) { const int t = a; a = b; b = t; }            // hook the larger root under the smaller
        const int old = atomicMin(lab + a, b);
        if (old == a) return;
        a = old;
    }
}

static __device__ __forceinline__ bool spk_linked(float u, float v, float diff)
{
    const float inf = __builtin_inff();
    return u != inf && v != inf && fabs((double)(u - v)) <= (double)diff;       // ref :622-624
}

// (A) label[p] = global pixel index of the tile-local root (or -1 for invalid pixels);
//     local_size[p] = pixel count of the tile-local component for roots, 0 elsewhere; total[p] = 0
// A tile row is exactly one wave (64 px): horizontal runs are labelled with a wave prefix-max (no atomics), so the
// union-find only has to join RUNS of adjacent rows, and only where the link is not already implied by the
// pixel to the left -- a flat 64x16 tile needs ~16 unions instead of ~3000 on contended roots.
template <int SPK_TH>
__global__ __launch_bounds__(256) void sgm_speckle_tile_k(const float* __restrict__ disp, int* __restrict__ label,
                                                          int* __restrict__ local_size, int* __restrict__ total, int W,
                                                          int H, float diff)
{
    static_assert(SPK_TW == 64, "one tile row = one wave");
    constexpr int SPK_N = SPK_TW * SPK_TH;
    __shared__ float tile[SPK_N];
    __shared__ int lab[SPK_N];
    __shared__ int cnt[SPK_N];
    __shared__ unsigned char left_link[SPK_N];             // pixel linked to its left neighbour (same run)
    const int tx0 = blockIdx.x * SPK_TW, ty0 = blockIdx.y * SPK_TH;
    const float inf = __builtin_inff();
    {
        const size_t frame_px = (size_t)blockIdx.z * W * H;            // batch: z = frame (labels are per-frame pixel indices)
        disp += frame_px; label += frame_px; local_size += frame_px; total += frame_px;
    }
    const int lx = threadIdx.x & 63, wv = threadIdx.x >> 6;
    // ---- rows wv, wv+4, ...: load, link to the left, label runs with their first pixel ----
    for (int ly = wv; ly < SPK_TH; ly += 4) {
        const int i = ly * SPK_TW + lx;
        const int x = tx0 + lx, y = ty0 + ly;
        const float v = (x < W && y < H) ? disp[(size_t)y * W + x] : inf;
        const float vl = __shfl_up(v, 1);
        const bool linkl = lx > 0 && spk_linked(vl, v, diff);
        int start = linkl ? -1 : lx;                       // run starts where the link to the left is broken
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {                 // inclusive prefix max over the wave
            const int o = __shfl_up(start, d);
            if (lx >= d) start = max(start, o);
        }
        tile[i] = v;
        left_link[i] = linkl ? 1 : 0;
        lab[i] = (v == inf) ? -1 : ly * SPK_TW + start;
        cnt[i] = 0;
    }
    __syncthreads();
    // ---- join runs of adjacent rows ----
    for (int i = threadIdx.x; i < SPK_N; i += 256) {
        const float v = tile[i];
        const int ly = i / SPK_TW;
        if (v == inf || ly == 0) continue;
        const int up = i - SPK_TW;
        const bool l_up = spk_linked(tile[up], v, diff);
        // (x,y)-(x,y-1): already joined by the pixel to the left if both rows continue their runs there and
        // the left pair is linked as well
        if (l_up) {
            const bool implied = lx > 0 && left_link[i] && left_link[up] && spk_linked(tile[up - 1], tile[i - 1], diff);
            if (!implied) uf_union(lab, lab[i], lab[up]);
        }
        // diagonals: implied when the pixel straight above is linked to us and continues into the diagonal one
        if (lx > 0 && spk_linked(tile[up - 1], v, diff) && !(l_up && left_link[up])) uf_union(lab, lab[i], lab[up - 1]);
        if (lx < SPK_TW - 1 && spk_linked(tile[up + 1], v, diff) && !(l_up && left_link[up + 1])) uf_union(lab, lab[i], lab[up + 1]);
    }
    __syncthreads();
    // ---- pixel counts per tile-local root: one LDS atomic per run (its last pixel knows the run length) ----
    for (int i = threadIdx.x; i < SPK_N; i += 256) {
        if (lab[i] < 0) continue;
        const bool last_of_run = (lx == SPK_TW - 1) || !left_link[i + 1];
        if (last_of_run) {
            const int first = left_link[i] ? lab[i] : i;   // entries of non-first pixels still name the run's first pixel
            atomicAdd(&cnt[uf_find(lab, first)], i - first + 1);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < SPK_N; i += 256) {
        const int x = tx0 + lx, y = ty0 + (i / SPK_TW);
        if (x >= W || y >= H) continue;
        const size_t p = (size_t)y * W + x;
        int g = -1;
        if (lab[i] >= 0) {
            const int r = uf_find(lab, lab[i]);
            g = (ty0 + r / SPK_TW) * W + tx0 + (r & (SPK_TW - 1));
        }
        label[p] = g;
        local_size[p] = cnt[i];
        total[p] = 0;
    }
}

// (B) unions across tile borders (only pixels in the first row / first or last column of a tile have
//     an already-scanned neighbour in another tile).  Along a straight tile edge most of these unions join the
//     same two tile-local components again and again, all of them chasing the same global roots; a link is
//     therefore skipped when it is implied by the union its left (or upper) neighbour pair makes plus links
//     INSIDE the two tiles, which pass (A) has already joined -- the same rule (A) uses between rows.
template <int SPK_TH>
__global__ __launch_bounds__(256) void sgm_speckle_border_k(const float* __restrict__ disp, int* __restrict__ label,
                                                            int W, int H, float diff)
{
    const int x = blockIdx.x * 256 + threadIdx.x;
    const int y = blockIdx.y;
    if (x >= W) return;
    const int lx = x & (SPK_TW - 1), ly = y & (SPK_TH - 1);
    if (ly != 0 && lx != 0 && lx != SPK_TW - 1) return;
    disp += (size_t)blockIdx.z * W * H;                                 // batch: z = frame
    label += (size_t)blockIdx.z * W * H;
    const int p = y * W + x;
    const float v = disp[p];
    const float inf = __builtin_inff();
    if (v == inf) return;
    // the 3x3 neighbourhood's already-scanned half (inf = outside the image, never linked)
    const bool has_up = y > 0, has_l = x > 0, has_r = x < W - 1;
    const float up = has_up ? disp[p - W] : inf;
    const float ul = (has_up && has_l) ? disp[p - W - 1] : inf;
    const float ur = (has_up && has_r) ? disp[p - W + 1] : inf;
    const float lf = has_l ? disp[p - 1] : inf;
    const bool l_up = spk_linked(up, v, diff), l_ul = spk_linked(ul, v, diff), l_ur = spk_linked(ur, v, diff);
    const bool l_lf = spk_linked(lf, v, diff);
    if (ly == 0) {
        // the three upper neighbours lie in the tiles above
        if (l_up) {
            // implied by the left pair: p ~ left and up ~ up-left inside their tiles, left ~ up-left across the edge
            const bool implied = lx > 0 && l_lf && spk_linked(ul, up, diff) && spk_linked(ul, lf, diff);
            if (!implied) uf_union(label, p, p - W);
        }
        if (l_ul && !(lx > 0 && l_up && spk_linked(ul, up, diff))) uf_union(label, p, p - W - 1);
        if (l_ur && !(lx < SPK_TW - 1 && l_up && spk_linked(ur, up, diff))) uf_union(label, p, p - W + 1);
        if (lx == 0 && l_lf) {                                           // left neighbour: the tile to the left
            uf_union(label, p, p - 1);
        }
    } else if (lx == 0) {
        // left and up-left neighbours lie in the tile to the left; `up` is in this tile
        if (l_lf) {
            const bool implied = l_up && spk_linked(ul, lf, diff) && spk_linked(ul, up, diff);
            if (!implied) uf_union(label, p, p - 1);
        }
        if (l_ul && !(l_lf && spk_linked(ul, lf, diff))) uf_union(label, p, p - W - 1);
    }
    if (lx == SPK_TW - 1 && ly != 0) {
        // up-right neighbour lies in the tile to the right; implied via `up` (this tile), which that neighbour's
        // own left link joins with it
        if (l_ur && !(l_up && spk_linked(ur, up, diff))) uf_union(label, p, p - W + 1);
    }
}

// (C) every tile-local root adds its pixel count to the component's final root
__global__ __launch_bounds__(256) void sgm_speckle_total_k(const int* __restrict__ label, const int* __restrict__ local_size,
                                                           int* __restrict__ total, int n)
{
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= n) return;
    label += (size_t)blockIdx.y * n;                                    // batch: y = frame
    local_size += (size_t)blockIdx.y * n;
    total += (size_t)blockIdx.y * n;
    const int c = local_size[p];
    if (c == 0) return;
    int r = p;
    for (int q = label[r]; q != r; q = label[r]) r = q;
    atomicAdd(total + r, c);
}

// (D) ref :633
__global__ __launch_bounds__(256) void sgm_speckle_apply_k(float* __restrict__ disp, const int* __restrict__ label,
                                                           const int* __restrict__ total, int n, unsigned min_area)
{
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= n) return;
    disp += (size_t)blockIdx.y * n;                                     // batch: y = frame
    label += (size_t)blockIdx.y * n;
    total += (size_t)blockIdx.y * n;
    int r = label[p];
    if (r < 0) return;
    for (int q = label[r]; q != r; q = label[r]) r = q;
    if ((unsigned)total[r] < min_area) disp[p] = __builtin_inff();
}

// ============================================================================================
// in-place 3x3 median  (ref :525-557 with in == out, .c:120 -> raster-order recurrence, Q13)
//
// Output (y,x) is the 5th smallest of: filtered (y-1,x-1..x+1) and (y,x-1), originals (y,x),
// (y,x+1), (y+1,x-1..x+1).  The recurrence is serial along x and y, so the kernel is built around
// its critical path.  With the five originals pre-sorted (e0..e4, fully parallel pre-pass) and the
// three values of the row above sorted (q0..q2), ranks 3 and 4 of those eight values are
//   s3 = max(min(e3,q0), min(e2,q1), min(e1,q2), e0),  s4 = max(min(e4,q0), min(e3,q1), min(e2,q2), e1)
// and the median of all nine is med3(s3, out(y,x-1), s4): ONE v_med3_f32 on the serial chain.
// Rows map to lanes (64 rows per wave) skewed by 3 columns per row, so the value from the row
// above is two steps old when it is needed and moves down one lane with a single DPP wave_shr.
// Waves are decoupled: the last row of a wave feeds the first row of the next through an LDS ring
// with progress counters.  The pre-pass stores its output time-skewed and lane-interleaved
// ([band][t/4][e][lane][t%4], t = x + 3*lane), so every load of the serial kernel is a coalesced 1 KiB.
// ============================================================================================

#define MED_PF 4                 // batches (4 steps each) of pre-sorted inputs kept in flight per wave
#define MED_NE 6                 // float4 planes per time slot: 5 pre-sorted originals + the row above a band
#ifndef MED_SKEW
#define MED_SKEW 3
#endif
#define MED_LAG (MED_SKEW * 63)
#define MED_RING 128
#define MED_WAVES 8                // waves (64 rows each) per workgroup: 512 threads leave 256 VGPRs per lane

static __device__ __forceinline__ void cswapf(float& a, float& b)
{
    const float lo = fminf(a, b), hi = fmaxf(a, b);
    a = lo; b = hi;
}

// number of float4 time slots per band
// (the time axis is padded to whole groups of MED_PF batches so the serial loop has no tail conditions:
// hipcc's s_waitcnt insertion merges over every CFG path, and a skippable batch would force vmcnt(0))
static inline int med_tq(int W) { return ((W + MED_LAG + 4 * MED_PF - 1) / (4 * MED_PF)) * MED_PF; }

__global__ __launch_bounds__(64) void sgm_median_prep_k(const float* __restrict__ disp, float4* __restrict__ P, int W, int H,
                                                        int Tq, int* __restrict__ ticket)
{
    const int l = threadIdx.x, tq = blockIdx.x, g = blockIdx.y;
    if (ticket && tq == 0 && g == 0 && l == 0) ticket[blockIdx.z] = 0;     // band tickets of the chained kernel (next in the stream)
    const int y = 1 + 64 * g + l;
    disp += (size_t)blockIdx.z * W * H;                                 // batch: z = frame
    P += (size_t)blockIdx.z * gridDim.y * Tq * MED_NE * 64;
    float e[MED_NE][4];
#pragma unroll
    for (int k = 0; k < MED_NE; ++k)
#pragma unroll
        for (int j = 0; j < 4; ++j) e[k][j] = 0.f;
    if (y <= H - 2) {
        const float* r0 = disp + (size_t)y * W;
        const float* r1 = r0 + W;
        const float* rm = r0 - W;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int x = 4 * tq + j - MED_SKEW * l;
            // plane 5: ORIGINAL value of (y-1, x+1).  Only lane 0 of the wave at the top of a band reads
            // it (row 0 of the image is never modified; later bands overwrite it, see the serial kernel).
            if (x + 1 >= 0 && x + 1 <= W - 1) e[5][j] = rm[x + 1];
            if (x < 0 || x > W - 1) continue;
            if (x == 0 || x == W - 1) {
                // border column: all five equal -> s3 = s4 = the original, the serial kernel passes it through
                const float v = r0[x];
                e[0][j] = e[1][j] = e[2][j] = e[3][j] = e[4][j] = v;
                continue;
            }
            float v0 = r0[x], v1 = r0[x + 1], v2 = r1[x - 1], v3 = r1[x], v4 = r1[x + 1];
            cswapf(v0, v1); cswapf(v3, v4); cswapf(v2, v4); cswapf(v2, v3); cswapf(v1, v4);
            cswapf(v0, v3); cswapf(v0, v2); cswapf(v1, v3); cswapf(v1, v2);
            e[0][j] = v0; e[1][j] = v1; e[2][j] = v2; e[3][j] = v3; e[4][j] = v4;
        }
    }
#pragma unroll
    for (int k = 0; k < MED_NE; ++k)
        P[(((size_t)g * Tq + tq) * MED_NE + k) * 64 + l] = make_float4(e[k][0], e[k][1], e[k][2], e[k][3]);
}

__global__ __launch_bounds__(64 * MED_WAVES) void sgm_median_serial_k(const float* __restrict__ disp, float4* __restrict__ P,
                                                                      float4* __restrict__ O, int W, int H, int Tq)
{
    __shared__ __attribute__((aligned(16))) float ring[MED_WAVES][MED_RING];
    __shared__ int prog[MED_WAVES];      // last column the wave's lane 63 has put into its ring
    __shared__ int cons[MED_WAVES];      // last column the wave has taken from the ring of the wave above
    __shared__ float ring_dummy[64];
    const int wv = threadIdx.x >> 6, l = threadIdx.x & 63;
    const int rows = H - 2;
    if (rows <= 0 || W <= 2) return;
    const int groups = (rows + 63) / 64;
    disp += (size_t)blockIdx.x * W * H;                              // batch: one workgroup per frame
    P += (size_t)blockIdx.x * groups * Tq * MED_NE * 64;
    O += (size_t)blockIdx.x * groups * Tq * 64;
    const int t_end = 4 * Tq;                                        // >= W + MED_LAG: lane 63 reaches column W-1 at t = W-1+MED_LAG

    for (int gbase = 0; gbase < groups; gbase += MED_WAVES) {
        if (threadIdx.x < MED_WAVES) { prog[threadIdx.x] = 0; cons[threadIdx.x] = 0; }
        if (gbase > 0) {
            // the row above this band is the previous band's finished last row: put it where lane 0 of the
            // band's first wave expects the row above (plane 5), replacing the originals of the pre-pass
            // (results live in the time-skewed buffer O until the un-skew kernel: row 64*gbase is lane 63 of
            // group gbase-1, column c sits at time slot c + MED_LAG)
            const float* const above = reinterpret_cast<const float*>(O + (size_t)(gbase - 1) * Tq * 64);
            for (int tq = threadIdx.x; tq < Tq; tq += blockDim.x) {
                float v[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int t = min(4 * tq + j + 1, W - 1) + MED_LAG;
                    v[j] = __hip_atomic_load(above + ((size_t)(t >> 2) * 64 + 63) * 4 + (t & 3), __ATOMIC_RELAXED,
                                             __HIP_MEMORY_SCOPE_AGENT);
                }
                P[(((size_t)gbase * Tq + tq) * MED_NE + 5) * 64] = make_float4(v[0], v[1], v[2], v[3]);
            }
            __threadfence();
        }
        __syncthreads();
        const int g = gbase + wv;
        if (g < groups) {                                            // wave-uniform
            const int y = 1 + 64 * g + l;
            const bool valid = y <= H - 2;
            const bool feeds_next = (wv + 1 < MED_WAVES) && (g + 1 < groups);
            const bool top_from_ring = wv > 0;
            const int yr = valid ? y : H - 2;
            float4* const Og = O + (size_t)g * Tq * 64 + l;
            const float* const top_row = disp + (size_t)(yr - 1) * W;
            const float4* Pg = P + (size_t)g * Tq * MED_NE * 64 + l;

            float o1 = 0.f;                                          // own output of the last step (in-row chain)
            float oh[MED_SKEW - 1];                                  // own outputs of the last MED_SKEW - 1 steps, oh[0] = o1
#pragma unroll
            for (int k = 0; k < MED_SKEW - 1; ++k) oh[k] = 0.f;
            float T0 = 0.f;                                          // out(y-1, x-1)
            float T1 = top_row[0];                                   // out(y-1, x): column 0 is border, never modified
            asm volatile("" : "+v"(T1));                             // retire this load here, not at its first use inside the loop
            __builtin_amdgcn_sched_barrier(0);
            // pre-sorted neighbourhoods are read-only input: keep MED_PF batches (4 steps each) in flight
            float4 evr[MED_PF][MED_NE];
            auto load_batch = [&](float4 (&dst)[MED_NE], int t0) {
                const int tq = min(t0 >> 2, Tq - 1);
#pragma unroll
                for (int k = 0; k < MED_NE; ++k) dst[k] = Pg[((size_t)tq * MED_NE + k) * 64];
            };
            // issue the prologue groups strictly in order: the loop's s_waitcnt counts are the minimum over the
            // prologue path and the back edge, and vmcnt retires in issue order
#pragma unroll
            for (int u = 0; u < MED_PF; ++u) {
                load_batch(evr[u], 4 * u);
                __builtin_amdgcn_sched_barrier(0);
            }

            // lane 63 feeds the next wave through the ring; every other lane writes to a dummy slot instead of
            // branching (slots more than ~100 columns behind the consumer are free, so the out-of-range
            // columns lane 63 writes before/after its row are harmless)
            float* const ring_dst = (l == 63) ? &ring[wv][0] : &ring_dummy[l];
            const unsigned ring_mask = (l == 63) ? (MED_RING - 1) : 0u;

            auto run_batch = [&](const float4 (&ev)[MED_NE], int t0, const float (&tv)[4]) {
                float res[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int x = t0 + j - MED_SKEW * l;
                    const float e0 = (&ev[0].x)[j], e1 = (&ev[1].x)[j], e2 = (&ev[2].x)[j], e3 = (&ev[3].x)[j],
                                e4 = (&ev[4].x)[j];
                    // out(y-1, x+1): produced by the lane above MED_SKEW - 1 steps ago; lane 0 takes the top row
                    const float b = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(tv[j]), __float_as_int(oh[MED_SKEW - 2]),
                                                                               0x138 /* wave_shr:1 */, 0xF, 0xF, false));
                    const float lo = fminf(fminf(T0, T1), b), hi = fmaxf(fmaxf(T0, T1), b);
                    const float mid = __builtin_amdgcn_fmed3f(T0, T1, b);
                    const float s3 = fmaxf(fmaxf(fminf(e3, lo), fminf(e2, mid)), fmaxf(fminf(e1, hi), e0));
                    const float s4 = fmaxf(fmaxf(fminf(e4, lo), fminf(e3, mid)), fmaxf(fminf(e2, hi), e1));
                    const float outv = __builtin_amdgcn_fmed3f(s3, o1, s4);   // border columns: s3 == s4 == original
                    res[j] = outv;
                    T0 = T1; T1 = b;
#pragma unroll
                    for (int k = MED_SKEW - 2; k > 0; --k) oh[k] = oh[k - 1];
                    oh[0] = o1 = outv;
                    ring_dst[(unsigned)(x - 1) & ring_mask] = outv;
                }
                // results go to the time-skewed, lane-interleaved buffer O (one coalesced 1 KiB store per batch;
                // storing straight into the image would touch 64 different rows per instruction); the un-skew
                // kernel moves them into the image afterwards
                Og[(size_t)(t0 >> 2) * 64] = make_float4(res[0], res[1], res[2], res[3]);
            };

            for (int tb = 0; tb < t_end; tb += 4 * MED_PF) {
                // ---- flow control between waves, once per MED_PF batches (LDS only) ----
                if (top_from_ring) {
                    const int need = min(tb + 4 * MED_PF, W - 1);
                    while (__hip_atomic_load(&prog[wv - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < need)
                        __builtin_amdgcn_s_sleep(1);
                }
                if (feeds_next) {
                    const int last = tb + 4 * MED_PF - 1 - MED_LAG;  // last column lane 63 writes in this group
                    while (__hip_atomic_load(&cons[wv + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < last - (MED_RING - 4 * MED_PF - 8))
                        __builtin_amdgcn_s_sleep(1);
                }
                asm volatile("" ::: "memory");
                float tvr[MED_PF][4];                                // lane 0: out(y-1, tb+1 .. tb+16)
                if (top_from_ring) {
#pragma unroll
                    for (int u = 0; u < MED_PF; ++u) {
                        const float4 r = *reinterpret_cast<const float4*>(&ring[wv - 1][(tb + 4 * u) & (MED_RING - 1)]);
                        tvr[u][0] = r.x; tvr[u][1] = r.y; tvr[u][2] = r.z; tvr[u][3] = r.w;
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    if (l == 0) __hip_atomic_store(&cons[wv], tb + 4 * MED_PF, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
#pragma unroll
                for (int u = 0; u < MED_PF; ++u) {
                    const int t0 = tb + 4 * u;
                    if (!top_from_ring) {
                        // top of a band: the row above comes with the pre-pass data (plane 5 of lane 0)
                        tvr[u][0] = evr[u][5].x; tvr[u][1] = evr[u][5].y; tvr[u][2] = evr[u][5].z; tvr[u][3] = evr[u][5].w;
                    }
                    run_batch(evr[u], t0, tvr[u]);
                    load_batch(evr[u], t0 + 4 * MED_PF);
                }
                if (feeds_next) {
                    const int done = min(tb + 4 * MED_PF - 1 - MED_LAG, W - 1);
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    if (l == 63 && done >= 1) __hip_atomic_store(&prog[wv], done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            }
            if (l == 0) __hip_atomic_store(&cons[wv], 0x7FFFFFF0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        __syncthreads();                                             // band finished and stored before the next one reads it
    }
}

// Bands as a CHAIN of workgroups (frames taller than one band, e.g. 2160 rows): every band is a workgroup of its own on
// its own CU, all bands of a frame run at the same time, skewed like the waves inside a band.  The last row of a band
// reaches the band below through global memory as 8-byte granules {value, tag} -- one write-through (agent-scope)
// store per column by the producing lane, polled with agent-scope loads by the consumer; the tag is a per-launch
// generation number, so a granule is either this launch's value or not yet there, no fence and no flag needed
// (MI355X_MICROARCH.md: data-tagged granules).  A granule row holds a whole image row: no back-pressure.  Bands are
// handed out by an atomic ticket, so a band's producer has always started before its consumer (no reliance on the
// dispatch order); polls are bounded.  One band after the other in one workgroup (the kernel above) took 3.9 ms at
// 3840 x 2160; the chain takes the time of ONE skewed pass.
#define MED_GPAD 256               // granule slots in front of column 0 (lane 63 runs 189 columns behind the time axis)
template <int WAVES>
__global__ __launch_bounds__(64 * WAVES) void sgm_median_chain_k(const float* __restrict__ disp, float4* __restrict__ P,
                                                                 float4* __restrict__ O, unsigned long long* __restrict__ G,
                                                                 int* __restrict__ ticket, unsigned long long* __restrict__ sink,
                                                                 int W, int H, int Tq, unsigned gen, int* __restrict__ status)
{
#undef MED_WAVES
#define MED_WAVES WAVES
    __shared__ __attribute__((aligned(16))) float ring[MED_WAVES][MED_RING];
    __shared__ int prog[MED_WAVES];      // last column the wave's lane 63 has put into its ring
    __shared__ int cons[MED_WAVES];      // last column the wave has taken from the ring of the wave above
    __shared__ float ring_dummy[64];
    const int wv = threadIdx.x >> 6, l = threadIdx.x & 63;
    const int rows = H - 2;
    if (rows <= 0 || W <= 2) return;
    const int groups = (rows + 63) / 64;
    const int frame = blockIdx.y, nbands = gridDim.x;
    disp += (size_t)frame * W * H;                                   // batch: y = frame
    P += (size_t)frame * groups * Tq * MED_NE * 64;
    O += (size_t)frame * groups * Tq * 64;
    const int t_end = 4 * Tq;                                        // >= W + MED_LAG: lane 63 reaches column W-1 at t = W-1+MED_LAG
    const int gstride = MED_GPAD + 4 * Tq + 64;                      // granules per band boundary
    __shared__ int band_s;
    if (threadIdx.x == 0) band_s = atomicAdd(&ticket[frame], 1);     // bands start in ticket order
    __syncthreads();
    const int band = band_s;
    G += ((size_t)frame * nbands) * gstride;

    {
        const int gbase = band * MED_WAVES;
        if (threadIdx.x < MED_WAVES) { prog[threadIdx.x] = 0; cons[threadIdx.x] = 0; }
        __syncthreads();
        const int g = gbase + wv;
        if (g < groups) {                                            // wave-uniform
            const int y = 1 + 64 * g + l;
            const bool valid = y <= H - 2;
            const bool feeds_next = (wv + 1 < MED_WAVES) && (g + 1 < groups);
            const bool feeds_band = (wv + 1 == MED_WAVES) && (g + 1 < groups);   // this wave's last row goes to the band below
            const bool top_from_band = (wv == 0) && (band > 0);
            const bool top_from_ring = wv > 0;
            // lane 63 of a band's last wave publishes its outputs as granules; everybody else writes to a sink
            unsigned long long* const gout = (feeds_band && l == 63) ? G + (size_t)band * gstride + MED_GPAD
                                                                     : sink + ((size_t)(frame * nbands + band) * 64 + l) * 4;
            const unsigned long long* const gin = G + (size_t)(band > 0 ? band - 1 : 0) * gstride + MED_GPAD;
            const bool publish = feeds_band && l == 63;
            const int yr = valid ? y : H - 2;
            float4* const Og = O + (size_t)g * Tq * 64 + l;
            const float* const top_row = disp + (size_t)(yr - 1) * W;
            const float4* Pg = P + (size_t)g * Tq * MED_NE * 64 + l;

            float o1 = 0.f;                                          // own output of the last step (in-row chain)
            float oh[MED_SKEW - 1];                                  // own outputs of the last MED_SKEW - 1 steps, oh[0] = o1
#pragma unroll
            for (int k = 0; k < MED_SKEW - 1; ++k) oh[k] = 0.f;
            float T0 = 0.f;                                          // out(y-1, x-1)
            float T1 = top_row[0];                                   // out(y-1, x): column 0 is border, never modified
            asm volatile("" : "+v"(T1));                             // retire this load here, not at its first use inside the loop
            __builtin_amdgcn_sched_barrier(0);
            // pre-sorted neighbourhoods are read-only input: keep MED_PF batches (4 steps each) in flight
            float4 evr[MED_PF][MED_NE];
            auto load_batch = [&](float4 (&dst)[MED_NE], int t0) {
                const int tq = min(t0 >> 2, Tq - 1);
#pragma unroll
                for (int k = 0; k < MED_NE; ++k) dst[k] = Pg[((size_t)tq * MED_NE + k) * 64];
            };
            // issue the prologue groups strictly in order: the loop's s_waitcnt counts are the minimum over the
            // prologue path and the back edge, and vmcnt retires in issue order
#pragma unroll
            for (int u = 0; u < MED_PF; ++u) {
                load_batch(evr[u], 4 * u);
                __builtin_amdgcn_sched_barrier(0);
            }

            // lane 63 feeds the next wave through the ring; every other lane writes to a dummy slot instead of
            // branching (slots more than ~100 columns behind the consumer are free, so the out-of-range
            // columns lane 63 writes before/after its row are harmless)
            float* const ring_dst = (l == 63) ? &ring[wv][0] : &ring_dummy[l];
            const unsigned ring_mask = (l == 63) ? (MED_RING - 1) : 0u;

            auto run_batch = [&](const float4 (&ev)[MED_NE], int t0, const float (&tv)[4]) {
                float res[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int x = t0 + j - MED_SKEW * l;
                    const float e0 = (&ev[0].x)[j], e1 = (&ev[1].x)[j], e2 = (&ev[2].x)[j], e3 = (&ev[3].x)[j],
                                e4 = (&ev[4].x)[j];
                    // out(y-1, x+1): produced by the lane above MED_SKEW - 1 steps ago; lane 0 takes the top row
                    const float b = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(tv[j]), __float_as_int(oh[MED_SKEW - 2]),
                                                                               0x138 /* wave_shr:1 */, 0xF, 0xF, false));
                    const float lo = fminf(fminf(T0, T1), b), hi = fmaxf(fmaxf(T0, T1), b);
                    const float mid = __builtin_amdgcn_fmed3f(T0, T1, b);
                    const float s3 = fmaxf(fmaxf(fminf(e3, lo), fminf(e2, mid)), fmaxf(fminf(e1, hi), e0));
                    const float s4 = fmaxf(fmaxf(fminf(e4, lo), fminf(e3, mid)), fmaxf(fminf(e2, hi), e1));
                    const float outv = __builtin_amdgcn_fmed3f(s3, o1, s4);   // border columns: s3 == s4 == original
                    res[j] = outv;
                    T0 = T1; T1 = b;
#pragma unroll
                    for (int k = MED_SKEW - 2; k > 0; --k) oh[k] = oh[k - 1];
                    oh[0] = o1 = outv;
                    ring_dst[(unsigned)(x - 1) & ring_mask] = outv;
                }
                // results go to the time-skewed, lane-interleaved buffer O (one coalesced 1 KiB store per batch;
                // storing straight into the image would touch 64 different rows per instruction); the un-skew
                // kernel moves them into the image afterwards
                Og[(size_t)(t0 >> 2) * 64] = make_float4(res[0], res[1], res[2], res[3]);
                if (feeds_band) {                                    // wave-uniform
                    const int x0 = publish ? t0 - MED_SKEW * l : 0;  // column of res[0] (>= -MED_LAG: the pad in front takes it)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const bool real = publish && x0 + j >= 0 && x0 + j <= W - 1;
                        const unsigned long long gr = (unsigned long long)__float_as_uint(res[j]) | ((unsigned long long)(real ? gen : 0u) << 32);
                        __hip_atomic_store(gout + (publish ? x0 + j : j), gr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
            };

            for (int tb = 0; tb < t_end; tb += 4 * MED_PF) {
                // ---- flow control between waves, once per MED_PF batches (LDS only) ----
                if (top_from_ring) {
                    const int need = min(tb + 4 * MED_PF, W - 1);
                    while (__hip_atomic_load(&prog[wv - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < need)
                        __builtin_amdgcn_s_sleep(1);
                }
                if (feeds_next) {
                    const int last = tb + 4 * MED_PF - 1 - MED_LAG;  // last column lane 63 writes in this group
                    while (__hip_atomic_load(&cons[wv + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < last - (MED_RING - 4 * MED_PF - 8))
                        __builtin_amdgcn_s_sleep(1);
                }
                asm volatile("" ::: "memory");
                float tvr[MED_PF][4];                                // lane 0: out(y-1, tb+1 .. tb+16)
                if (top_from_band) {
                    // the band above publishes its last row as granules: lanes 0..15 poll one column each
                    const int col = min(tb + 1 + (l & 15), W - 1);
                    unsigned long long gr = 0;
                    int spin = 0;
                    for (; spin < (1 << 18); ++spin) {               // bounded: a lost producer must not hang the GPU
                        gr = __hip_atomic_load(gin + col, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (__all((unsigned)(gr >> 32) == gen)) break;
                        __builtin_amdgcn_s_sleep(2);
                    }
                    // gave up: the rows below are computed from stale values.  Tell the host (a word of pinned host memory it
                    // reads after the next synchronisation, sgm_host.c sync_streams): the match fails instead of returning them
                    if (spin == (1 << 18) && status && l == 0) __hip_atomic_store(status, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    const unsigned vbits = (unsigned)gr;
#pragma unroll
                    for (int u = 0; u < MED_PF; ++u)
#pragma unroll
                        for (int j = 0; j < 4; ++j) tvr[u][j] = __uint_as_float(__builtin_amdgcn_readlane(vbits, 4 * u + j));
                }
                if (top_from_ring) {
#pragma unroll
                    for (int u = 0; u < MED_PF; ++u) {
                        const float4 r = *reinterpret_cast<const float4*>(&ring[wv - 1][(tb + 4 * u) & (MED_RING - 1)]);
                        tvr[u][0] = r.x; tvr[u][1] = r.y; tvr[u][2] = r.z; tvr[u][3] = r.w;
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    if (l == 0) __hip_atomic_store(&cons[wv], tb + 4 * MED_PF, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
#pragma unroll
                for (int u = 0; u < MED_PF; ++u) {
                    const int t0 = tb + 4 * u;
                    if (!top_from_ring && !top_from_band) {
                        // top of the frame: the row above comes with the pre-pass data (plane 5 of lane 0)
                        tvr[u][0] = evr[u][5].x; tvr[u][1] = evr[u][5].y; tvr[u][2] = evr[u][5].z; tvr[u][3] = evr[u][5].w;
                    }
                    run_batch(evr[u], t0, tvr[u]);
                    load_batch(evr[u], t0 + 4 * MED_PF);
                }
                if (feeds_next) {
                    const int done = min(tb + 4 * MED_PF - 1 - MED_LAG, W - 1);
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    if (l == 63 && done >= 1) __hip_atomic_store(&prog[wv], done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            }
            if (l == 0) __hip_atomic_store(&cons[wv], 0x7FFFFFF0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
#undef MED_WAVES
}
#define MED_WAVES 8


// results of the serial kernel: O[group][t/4][lane][t%4] with t = x + MED_SKEW*lane  ->  disp[y][x], interior only
__global__ __launch_bounds__(256) void sgm_median_unskew_k(const float* __restrict__ O, float* __restrict__ disp, int W, int H,
                                                           int Tq)
{
    const int x = 1 + blockIdx.x * 256 + threadIdx.x;
    const int y = 1 + blockIdx.y;
    if (x > W - 2) return;
    const int groups = (H - 2 + 63) / 64;
    O += (size_t)blockIdx.z * groups * Tq * 64 * 4;                    // batch: z = frame
    disp += (size_t)blockIdx.z * W * H;
    const int g = (y - 1) >> 6, l = (y - 1) & 63;
    const int t = x + MED_SKEW * l;
    disp[(size_t)y * W + x] = O[(((size_t)g * Tq + (t >> 2)) * 64 + l) * 4 + (t & 3)];
}


// ============================================================================================
// Test-platform arithmetic next to the hot path, on device buffers: disparity -> depth in mm and the
// scores the reference's server computes for a returned depth image (HostScript_Server/depth_image.py:138-165, 276-319;
// restated from reading -- that module imports cv2, so no reference vectors exist: "parity unpinned").
// ============================================================================================

// (this translation unit is compiled with -fno-honor-nans: the hot path has no NaN.  These two kernels do see NaN, so they
// test and make it with integer operations on the bit patterns, which no floating-point option can fold away)
static __device__ __forceinline__ bool finite_bits(float v) { return (__float_as_uint(v) & 0x7F800000u) != 0x7F800000u; }

// depth = float32(fx * baseline) / (disparity + doffs); a non-finite or zero denominator (invalid disparity = +INF) gives NaN
__global__ __launch_bounds__(256) void sgm_depth_k(const float* __restrict__ disp, float* __restrict__ depth, size_t n, float fb,
                                                   float doffs)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float denom = disp[i] + doffs;
    const bool ok = finite_bits(denom) && (__float_as_uint(denom) << 1) != 0u;      // finite and not +-0
    depth[i] = ok ? fb / denom : __uint_as_float(0x7FC00000u);
}

// per-block partial sums over the pixels finite in both images: sum of squared differences (double), count, count of
// |difference| > abs_thresh; the host adds the partials in block order (deterministic)
struct ScorePartial { double sumsq; unsigned long long n, bad; };
#define SCORE_PER_THREAD 16
__global__ __launch_bounds__(256) void sgm_score_k(const float* __restrict__ gt, const float* __restrict__ test, size_t n,
                                                   float abs_thresh, ScorePartial* __restrict__ out)
{
    __shared__ double s_sum[256];
    __shared__ unsigned s_n[256], s_bad[256];
    double sum = 0.0;
    unsigned cnt = 0, bad = 0;
    const size_t base = (size_t)blockIdx.x * 256 * SCORE_PER_THREAD + threadIdx.x;
    for (int k = 0; k < SCORE_PER_THREAD; ++k) {
        const size_t i = base + (size_t)k * 256;
        if (i < n) {
            const float a = test[i], b = gt[i];
            if (finite_bits(a) && finite_bits(b)) {
                const float diff = a - b;                            // float32, as numpy on float32 arrays
                sum += (double)diff * (double)diff;
                ++cnt;
                bad += fabsf(diff) > abs_thresh;
            }
        }
    }
    s_sum[threadIdx.x] = sum; s_n[threadIdx.x] = cnt; s_bad[threadIdx.x] = bad;
    __syncthreads();
    for (int step = 128; step > 0; step >>= 1) {
        if ((int)threadIdx.x < step) {
            s_sum[threadIdx.x] += s_sum[threadIdx.x + step];
            s_n[threadIdx.x] += s_n[threadIdx.x + step];
            s_bad[threadIdx.x] += s_bad[threadIdx.x + step];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) { out[blockIdx.x].sumsq = s_sum[0]; out[blockIdx.x].n = s_n[0]; out[blockIdx.x].bad = s_bad[0]; }
}

// grey = (wr r + 150 g + 29 b) >> 8 of three byte planes B, G, R (stereo_matching.c:18-25: wr = 76; stb_image.h:1746-1749: 77);
// four pixels per lane when the planes are dword-aligned, else one
template <bool VEC>
__global__ __launch_bounds__(256) void sgm_gray_planes_k(const uint8_t* __restrict__ bgr, size_t n, unsigned wr, uint8_t* __restrict__ gray)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (VEC) {
        if (i * 4 >= n) return;
        const uint32_t b = ((const uint32_t*)bgr)[i], g = ((const uint32_t*)(bgr + n))[i], r = ((const uint32_t*)(bgr + 2 * n))[i];
        uint32_t out = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t v = wr * ((r >> (8 * k)) & 255u) + 150u * ((g >> (8 * k)) & 255u) + 29u * ((b >> (8 * k)) & 255u);
            out |= ((v >> 8) & 255u) << (8 * k);
        }
        ((uint32_t*)gray)[i] = out;
    } else {
        if (i >= n) return;
        gray[i] = (uint8_t)((wr * bgr[2 * n + i] + 150u * bgr[n + i] + 29u * bgr[i]) >> 8);
    }
}

extern "C" {

int sgmd_gray_planes(int ord, void* stream, const void* bgr, size_t n, int weight_r, void* gray)
{
    HIP_TRY(hipSetDevice(ord));
    if (n == 0) return 0;
    const bool vec = n % 4 == 0 && ((uintptr_t)bgr | (uintptr_t)gray) % 4 == 0;
    if (vec)
        hipLaunchKernelGGL(sgm_gray_planes_k<true>, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                           (const uint8_t*)bgr, n, (unsigned)weight_r, (uint8_t*)gray);
    else
        hipLaunchKernelGGL(sgm_gray_planes_k<false>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                           (const uint8_t*)bgr, n, (unsigned)weight_r, (uint8_t*)gray);
    HIP_TRY(hipGetLastError());
    return 0;
}

int sgmd_depth(int ord, void* stream, const void* disp, size_t n, float fx, float baseline, float doffs, void* depth)
{
    HIP_TRY(hipSetDevice(ord));
    if (n == 0) return 0;
    const float fb = (float)((double)fx * (double)baseline);
    hipLaunchKernelGGL(sgm_depth_k, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const float*)disp,
                       (float*)depth, n, fb, doffs);
    HIP_TRY(hipGetLastError());
    return 0;
}

/* blocking: launches the reduction, copies the per-block partials back and adds them in block order */
int sgmd_score(int ord, void* stream, const void* gt, const void* test, size_t n, float abs_thresh, double* sumsq,
               unsigned long long* n_valid, unsigned long long* n_bad)
{
    HIP_TRY(hipSetDevice(ord));
    *sumsq = 0.0; *n_valid = 0; *n_bad = 0;
    if (n == 0) return 0;
    const size_t per_block = 256 * SCORE_PER_THREAD;
    const unsigned blocks = (unsigned)((n + per_block - 1) / per_block);
    ScorePartial* d_part = nullptr;
    HIP_TRY(hipMalloc(&d_part, sizeof(ScorePartial) * blocks));
    hipLaunchKernelGGL(sgm_score_k, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const float*)gt, (const float*)test, n,
                       abs_thresh, d_part);
    ScorePartial* h = new ScorePartial[blocks];
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(h, d_part, sizeof(ScorePartial) * blocks, hipMemcpyDeviceToHost, (hipStream_t)stream);
    if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)stream);
    if (e == hipSuccess)
        for (unsigned b = 0; b < blocks; ++b) { *sumsq += h[b].sumsq; *n_valid += h[b].n; *n_bad += h[b].bad; }
    delete[] h;
    (void)hipFree(d_part);
    HIP_TRY(e);
    return 0;
}

int sgmd_speckle(int ord, void* stream, const sgmd_geom* g, void* disp, float diff, unsigned min_area, void* labels,
                 void* sizes, void* totals)
{
    HIP_TRY(hipSetDevice(ord));
    hipStream_t st = (hipStream_t)stream;
    const int n = g->W * g->H;
    const dim3 lin((n + 255) / 256, g->B), b(256);
    const char* th_env = getenv("SGM_SPECKLE_TILE_ROWS");               // tuning / test knob
    const int th = (th_env && *th_env) ? atoi(th_env) : (g->B >= 2 ? 32 : 16);
#define SPK_LAUNCH(TH)                                                                                                     \
    do {                                                                                                                   \
        hipLaunchKernelGGL(sgm_speckle_tile_k<TH>, dim3((g->W + SPK_TW - 1) / SPK_TW, (g->H + TH - 1) / TH, g->B), b, 0,   \
                           st, (const float*)disp, (int*)labels, (int*)sizes, (int*)totals, g->W, g->H, diff);            \
        hipLaunchKernelGGL(sgm_speckle_border_k<TH>, dim3((g->W + 255) / 256, g->H, g->B), b, 0, st, (const float*)disp,   \
                           (int*)labels, g->W, g->H, diff);                                                                \
    } while (0)
    if (th >= 64) SPK_LAUNCH(64);
    else if (th >= 32) SPK_LAUNCH(32);
    else SPK_LAUNCH(16);
#undef SPK_LAUNCH
    hipLaunchKernelGGL(sgm_speckle_total_k, lin, b, 0, st, (const int*)labels, (const int*)sizes, (int*)totals, n);
    hipLaunchKernelGGL(sgm_speckle_apply_k, lin, b, 0, st, (float*)disp, (const int*)labels, (const int*)totals, n,
                       min_area);
    HIP_TRY(hipGetLastError());
    return 0;
}

// rows per band of the chained kernel: 4 waves (256 rows, one wave per SIMD of the band's CU)
#define MED_CHAIN_WAVES 4
static inline int med_chain(const sgmd_geom* g, int groups)
{
    static const int want = getenv("SGM_MEDIAN_CHAIN") ? atoi(getenv("SGM_MEDIAN_CHAIN")) : 1;   // 0: one band after the other
    return want && groups > MED_WAVES;
}
static inline size_t med_granules(const sgmd_geom* g, int groups)      // granule rows + sink + tickets, in 8-byte words
{
    const int nbands = (groups + MED_CHAIN_WAVES - 1) / MED_CHAIN_WAVES;
    const size_t gstride = MED_GPAD + 4 * (size_t)med_tq(g->W) + 64;
    return (size_t)g->B * nbands * (gstride + 64 * 4) + (size_t)g->B + 8;
}

size_t sgmd_median_scratch_bytes(const sgmd_geom* g)
{
    const int groups = (g->H - 2 + 63) / 64;
    if (groups <= 0) return 16;
    // inputs (MED_NE planes) + results, then (tall frames) the granule rows between bands
    return (size_t)g->B * groups * med_tq(g->W) * (MED_NE + 1) * 64 * sizeof(float4) + med_granules(g, groups) * 8;
}

int sgmd_median(int ord, void* stream, const sgmd_geom* g, void* disp, void* scratch, void* status)
{
    HIP_TRY(hipSetDevice(ord));
    const int rows = g->H - 2;
    if (rows <= 0 || g->W <= 2) return 0;                            // no interior pixel: the filter is a no-op
    const int groups = (rows + 63) / 64;
    const int Tq = med_tq(g->W);
    hipStream_t st = (hipStream_t)stream;
    float4* const results = (float4*)scratch + (size_t)g->B * groups * Tq * MED_NE * 64;
    unsigned long long* const G = (unsigned long long*)(results + (size_t)g->B * groups * Tq * 64);
    const int nbands = (groups + MED_CHAIN_WAVES - 1) / MED_CHAIN_WAVES;
    const size_t gstride = MED_GPAD + 4 * (size_t)Tq + 64;
    unsigned long long* const sink = G + (size_t)g->B * nbands * gstride;
    int* const ticket = (int*)(sink + (size_t)g->B * nbands * 64 * 4);
    const bool chain = med_chain(g, groups);
    hipLaunchKernelGGL(sgm_median_prep_k, dim3(Tq, groups, g->B), dim3(64), 0, st, (const float*)disp, (float4*)scratch, g->W,
                       g->H, Tq, chain ? ticket : (int*)nullptr);
    if (chain) {
        // per-launch generation number of the granules (never 0; stale granules of earlier launches never match)
        static unsigned generation = 0x5A5A0000u;
        unsigned gen = __atomic_add_fetch(&generation, 1u, __ATOMIC_RELAXED);
        if (gen == 0) gen = __atomic_add_fetch(&generation, 1u, __ATOMIC_RELAXED);
        hipLaunchKernelGGL(sgm_median_chain_k<MED_CHAIN_WAVES>, dim3(nbands, g->B), dim3(64 * MED_CHAIN_WAVES), 0, st, (const float*)disp,
                           (float4*)scratch, results, G, ticket, sink, g->W, g->H, Tq, gen, (int*)status);
    } else {
        const int waves = groups < MED_WAVES ? groups : MED_WAVES;
        hipLaunchKernelGGL(sgm_median_serial_k, dim3(g->B), dim3(64 * waves), 0, st, (const float*)disp, (float4*)scratch, results,
                           g->W, g->H, Tq);
    }
    hipLaunchKernelGGL(sgm_median_unskew_k, dim3((g->W - 2 + 255) / 256, g->H - 2, g->B), dim3(256), 0, st,
                       (const float*)results, (float*)disp, g->W, g->H, Tq);
    HIP_TRY(hipGetLastError());
    return 0;
}

}  // extern "C"

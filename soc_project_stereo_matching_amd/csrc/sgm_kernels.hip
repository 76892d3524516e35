// sgm_kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels of the SGM hot path and their
// extern "C" launchers (interface: sgm_device.h).  No MFMA: the path is integer min-plus and
// byte streaming, bounded by HBM bandwidth and by the length of the serial path recurrences.
//
// Data layout in HBM (all row-major, disparity fastest, Dp = padded disparity stride):
//   census  u32 [H][W]            cost   u8  [H][W][Dp]
//   planes  u8  [dir][H][W][Dp]   (per-direction path cost L_r, written once, never RMW)
//   extras  u8  [4][H][Dp]        (L_r of the 4 anomalous diagonal lines, step-major)
//   S       u16 [H][W][Dp]        disparity maps f32 [H][W]
//
// Reference for every stage: /root/reference/SemiGlobalMatching/SemiGlobalMatching/SemiGlobalMatching.c
// (line numbers in the comments below refer to that file).

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <math.h>
#include <string.h>
#include <type_traits>

#include "sgm_device.h"

#define HIP_TRY(expr)                                                                         \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess) {                                                               \
            fprintf(stderr, "sgm_mi355x: %s failed: %s (%s:%d)\n", #expr, hipGetErrorString(e_), \
                    __FILE__, __LINE__);                                                      \
            return (int)e_;                                                                   \
        }                                                                                     \
    } while (0)

// ============================================================================================
// small device helpers
// ============================================================================================

typedef unsigned short us2 __attribute__((ext_vector_type(2)));   // two u16 in one VGPR (v_pk_*_u16)

static __device__ __forceinline__ unsigned as_u(us2 v) { return __builtin_bit_cast(unsigned, v); }
static __device__ __forceinline__ us2 as_p(unsigned v) { return __builtin_bit_cast(us2, v); }
static __device__ __forceinline__ us2 pk_min(us2 a, us2 b) { return __builtin_elementwise_min(a, b); }
static __device__ __forceinline__ us2 splat(unsigned v) { return as_p((v & 0xFFFFu) * 0x00010001u); }

// DPP cross-lane moves inside a 16-lane row (one VALU op, no LDS).  Lanes whose source lane does
// not exist keep `old` (bound_ctrl = 0), which is how the 255 sentinels of ref :260-263 appear.
template <int CTRL>
static __device__ __forceinline__ unsigned dpp_mov(unsigned old, unsigned src)
{
    return (unsigned)__builtin_amdgcn_update_dpp((int)old, (int)src, CTRL, 0xF, 0xF, false);
}
enum : int {
    DPP_QUAD_XOR1 = 0xB1,        // quad_perm [1,0,3,2]
    DPP_QUAD_XOR2 = 0x4E,        // quad_perm [2,3,0,1]
    DPP_ROW_SHL1 = 0x101,        // lane i <- lane i+1 (within the row)
    DPP_ROW_SHR1 = 0x111,        // lane i <- lane i-1
    DPP_ROW_MIRROR = 0x140,      // lane i <- lane 15-i
    DPP_ROW_HALF_MIRROR = 0x141  // lane i <- lane 7-i (within each 8)
};

// min over the 16 lanes of a row, result in every lane of the row (4 DPP steps; every source lane
// exists for these permutations, so no `old` operand is needed and the move folds into v_min_u32_dpp)
template <int CTRL>
static __device__ __forceinline__ unsigned dpp_perm(unsigned src)
{
    return (unsigned)__builtin_amdgcn_mov_dpp((int)src, CTRL, 0xF, 0xF, true);
}
template <int LPP = 16>
static __device__ __forceinline__ unsigned row_allmin(unsigned v)
{
    v = min(v, dpp_perm<DPP_QUAD_XOR1>(v));
    v = min(v, dpp_perm<DPP_QUAD_XOR2>(v));
    v = min(v, dpp_perm<DPP_ROW_HALF_MIRROR>(v));                 // all 8 lanes of a half row agree
    if (LPP >= 16) v = min(v, dpp_perm<DPP_ROW_MIRROR>(v));       // all 16 lanes of the row agree
    if (LPP == 32) {                                              // rows 0|1 and 2|3: v_permlane16_swap exchanges them
        const auto sw = __builtin_amdgcn_permlane16_swap(v, v, false, false);
        v = min(sw[0], sw[1]);
    }
    if (LPP == 64) {                                              // one pixel per wave: the result is wave-uniform (SGPR)
        const unsigned a = __builtin_amdgcn_readlane(v, 0), b = __builtin_amdgcn_readlane(v, 16);
        const unsigned c = __builtin_amdgcn_readlane(v, 32), d = __builtin_amdgcn_readlane(v, 48);
        v = min(min(a, b), min(c, d));
    }
    return v;
}

// ============================================================================================
// census 5x5  (ref :134-159)
// ============================================================================================

__global__ __launch_bounds__(256) void sgm_census_k(const uint8_t* __restrict__ left, const uint8_t* __restrict__ right,
                                                    uint32_t* __restrict__ cl, uint32_t* __restrict__ cr, int W, int H)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= W || y >= H) return;
    const size_t frame_px = (size_t)(blockIdx.z >> 1) * W * H;         // batch: z = 2 * frame + image
    const uint8_t* img = ((blockIdx.z & 1) ? right : left) + frame_px;
    uint32_t* out = ((blockIdx.z & 1) ? cr : cl) + frame_px;
    uint32_t bits = 0;
    // border of 2 px is never written by the reference (zero-initialised statics, Q3); also nothing
    // at all is written for images with W <= 5 or H <= 5 (ref :136)
    if (W > 5 && H > 5 && x >= 2 && x < W - 2 && y >= 2 && y < H - 2) {
        const unsigned centre = img[(size_t)y * W + x];
#pragma unroll
        for (int r = -2; r <= 2; ++r)
#pragma unroll
            for (int c = -2; c <= 2; ++c) bits = (bits << 1) | (unsigned)(img[(size_t)(y + r) * W + (x + c)] < centre);
    }
    out[(size_t)y * W + x] = bits;
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
// path aggregation  (ref :198-372)
//
// One wave = 4 path lines (one per 16-lane DPP row); lane `sub` of a row owns DPL consecutive
// disparities, kept as DPL/2 packed u16 pairs.  Per step and line:
//   L(d) = u8( C(d) + min( Lp(d), Lp(d-1)+P1, Lp(d+1)+P1, minPrev + pen ) - minPrev )
// with Lp(-1) = Lp(D) = 255 (ref :260-263), all sums truncated to u16 as in the C (ref :332-335)
// and the result truncated to u8 (ref :343, Q7).  d+-1 neighbours of the lane-edge elements come
// from DPP row shifts; min over d is an in-lane tree plus a 4-step DPP all-reduce.  All
// directions run in one launch; every line walks the image with the reference's own pointer
// state machine (ref :281-323, 359-367), so wrap-around diagonals (Q5) need no special casing.
// ============================================================================================

struct AggArgs {
    const uint8_t* img;
    const uint32_t* census_l;
    const uint32_t* census_r;   // the allocation has >= dmin + Dp dwords of slack in front (reads left of column 0)
    int dmin;
    const uint16_t* lut;        // (uint16) max(P1, P2 / (|dg| + 1)), 256 entries (ref :335)
    uint8_t* planes;
    size_t plane_bytes;
    uint8_t* extras;
    int W, H, D, Dp;
    int row_begin, row_end;     // rows of the frame this launch covers (a row tile of a multi-GPU run; [0,H) normally)
    int run_anom;               // 1: also run the four anomalous diagonal lines (whole frame)
    int B;                      // frames per launch; frame f uses img/census + f*W*H, planes + f*8*plane_bytes, extras + f*4*H*Dp
    int p1;
    int ndirs;
    int dx[8], dy[8];
    int anom_line[8];
    int block_begin[9];
    int ghost_zero;
};

// per-frame base pointers of a batched launch (kept apart from the kernel-argument struct so that struct
// stays in the scalar kernarg segment)
struct AggFrame {
    const uint8_t* img;
    const uint32_t* census_l;
    const uint32_t* census_r;
    uint8_t* planes;
    uint8_t* extras;
};

template <int DPL> struct CellVec { unsigned w[(DPL + 3) / 4]; };

template <int DPL>
static __device__ __forceinline__ void load_cells(const uint8_t* p, CellVec<DPL>& v)
{
    if constexpr (DPL == 2) {
        v.w[0] = *reinterpret_cast<const unsigned short*>(p);
    } else if constexpr (DPL == 4) {
        v.w[0] = *reinterpret_cast<const unsigned*>(p);
    } else if constexpr (DPL == 8) {
        const uint2 t = *reinterpret_cast<const uint2*>(p);
        v.w[0] = t.x; v.w[1] = t.y;
    } else if constexpr (DPL == 12) {
        struct __attribute__((packed, aligned(4))) u3 { unsigned a, b, c; };
        const u3 t = *reinterpret_cast<const u3*>(p);
        v.w[0] = t.a; v.w[1] = t.b; v.w[2] = t.c;
    } else if constexpr (DPL == 16) {
        const uint4 t = *reinterpret_cast<const uint4*>(p);
        v.w[0] = t.x; v.w[1] = t.y; v.w[2] = t.z; v.w[3] = t.w;
    } else {
        static_assert(DPL == 32, "unsupported DPL");
        const uint4 t = *reinterpret_cast<const uint4*>(p);
        const uint4 u = *reinterpret_cast<const uint4*>(p + 16);
        v.w[0] = t.x; v.w[1] = t.y; v.w[2] = t.z; v.w[3] = t.w;
        v.w[4] = u.x; v.w[5] = u.y; v.w[6] = u.z; v.w[7] = u.w;
    }
}

// L_r planes are written once and read once by the sum kernel: stream them past the caches (nt) so the
// cost volume, which all eight directions re-read, keeps its place in L2 / Infinity Cache.
// read-once variant (non-temporal) of load_cells
template <int DPL>
static __device__ __forceinline__ void load_cells_nt(const uint8_t* p, CellVec<DPL>& v)
{
    typedef unsigned v2u __attribute__((ext_vector_type(2)));
    typedef unsigned v4u __attribute__((ext_vector_type(4)));
    if constexpr (DPL == 2) {
        v.w[0] = __builtin_nontemporal_load(reinterpret_cast<const unsigned short*>(p));
    } else if constexpr (DPL == 4) {
        v.w[0] = __builtin_nontemporal_load(reinterpret_cast<const unsigned*>(p));
    } else if constexpr (DPL == 8) {
        const v2u t = __builtin_nontemporal_load(reinterpret_cast<const v2u*>(p));
        v.w[0] = t.x; v.w[1] = t.y;
    } else if constexpr (DPL == 12) {
        const v2u t = __builtin_nontemporal_load(reinterpret_cast<const v2u*>(p));
        v.w[0] = t.x; v.w[1] = t.y;
        v.w[2] = __builtin_nontemporal_load(reinterpret_cast<const unsigned*>(p + 8));
    } else if constexpr (DPL == 16) {
        const v4u t = __builtin_nontemporal_load(reinterpret_cast<const v4u*>(p));
        v.w[0] = t.x; v.w[1] = t.y; v.w[2] = t.z; v.w[3] = t.w;
    } else {
        const v4u t = __builtin_nontemporal_load(reinterpret_cast<const v4u*>(p));
        const v4u u = __builtin_nontemporal_load(reinterpret_cast<const v4u*>(p + 16));
        v.w[0] = t.x; v.w[1] = t.y; v.w[2] = t.z; v.w[3] = t.w;
        v.w[4] = u.x; v.w[5] = u.y; v.w[6] = u.z; v.w[7] = u.w;
    }
}

template <int DPL>
static __device__ __forceinline__ void store_cells(uint8_t* p, const CellVec<DPL>& v)
{
    typedef unsigned v2u __attribute__((ext_vector_type(2)));
    typedef unsigned v4u __attribute__((ext_vector_type(4)));
    if constexpr (DPL == 2) {
        __builtin_nontemporal_store((unsigned short)v.w[0], reinterpret_cast<unsigned short*>(p));
    } else if constexpr (DPL == 4) {
        __builtin_nontemporal_store(v.w[0], reinterpret_cast<unsigned*>(p));
    } else if constexpr (DPL == 8) {
        v2u t = {v.w[0], v.w[1]};
        __builtin_nontemporal_store(t, reinterpret_cast<v2u*>(p));
    } else if constexpr (DPL == 12) {
        v2u t = {v.w[0], v.w[1]};
        __builtin_nontemporal_store(t, reinterpret_cast<v2u*>(p));
        __builtin_nontemporal_store(v.w[2], reinterpret_cast<unsigned*>(p + 8));
    } else if constexpr (DPL == 16) {
        v4u t = {v.w[0], v.w[1], v.w[2], v.w[3]};
        __builtin_nontemporal_store(t, reinterpret_cast<v4u*>(p));
    } else {
        v4u t = {v.w[0], v.w[1], v.w[2], v.w[3]};
        v4u u = {v.w[4], v.w[5], v.w[6], v.w[7]};
        __builtin_nontemporal_store(t, reinterpret_cast<v4u*>(p));
        __builtin_nontemporal_store(u, reinterpret_cast<v4u*>(p + 16));
    }
}

// packed u16 pairs -> bytes (v_perm_b32)
template <int DPL>
static __device__ __forceinline__ void pack_cells(const us2 (&pr)[DPL / 2], CellVec<DPL>& v)
{
    if constexpr (DPL == 2) {
        v.w[0] = __builtin_amdgcn_perm(0u, as_u(pr[0]), 0x0c0c0200u);
    } else {
#pragma unroll
        for (int k = 0; k < DPL / 4; ++k)
            v.w[k] = __builtin_amdgcn_perm(as_u(pr[2 * k + 1]), as_u(pr[2 * k]), 0x06040200u);
    }
}

// The matching cost is recomputed here from the two census images instead of being read from a
// materialised cost volume: C(p,d) = popcount(cl[y][x] ^ cr[y][x-d]), 127 where x-d is left of the image
// (ref :161-196).  The census images (2 x 1.9 MB at KITTI) stay in L2, so the 8 directions no longer
// stream the 60 MB volume from HBM eight times.
//
// CensusVec holds, for the DPL disparities of a lane, the census-right words in ASCENDING ADDRESS order:
// r[j] = cr[y][x - dmin - lane_off - (DPL-1) + j], i.e. r[DPL-1-i] belongs to the lane's i-th disparity.
template <int DPL> struct CensusVec { unsigned r[DPL]; };

template <int DPL>
static __device__ __forceinline__ void load_census(const uint32_t* p, CensusVec<DPL>& v)
{
    if constexpr (DPL == 2) {
        struct __attribute__((packed, aligned(4))) u2 { unsigned a, b; };
        const u2 t = *reinterpret_cast<const u2*>(p);
        v.r[0] = t.a; v.r[1] = t.b;
    } else {
        struct __attribute__((packed, aligned(4))) u4 { unsigned a, b, c, d; };
#pragma unroll
        for (int q = 0; q < DPL / 4; ++q) {
            const u4 t = *reinterpret_cast<const u4*>(p + 4 * q);
            v.r[4 * q] = t.a; v.r[4 * q + 1] = t.b; v.r[4 * q + 2] = t.c; v.r[4 * q + 3] = t.d;
        }
    }
}

// packed u16 cost pairs of a lane.  `lim` = x - dmin - lane_off: disparity i of the lane is inside the
// image iff i <= lim; `masked` (wave-uniform) says whether any lane of the wave needs the test at all.
template <int DPL>
static __device__ __forceinline__ void census_costs(unsigned cl, const CensusVec<DPL>& cv, int lim, bool masked,
                                                    us2 (&C)[DPL / 2])
{
#pragma unroll
    for (int j = 0; j < DPL / 2; ++j) {
        const unsigned hi = (unsigned)__popc(cl ^ cv.r[DPL - 2 - 2 * j]) << 16;
        C[j] = as_p((unsigned)__popc(cl ^ cv.r[DPL - 1 - 2 * j]) + hi);
    }
    if (masked) {
#pragma unroll
        for (int j = 0; j < DPL / 2; ++j) {
            const unsigned m = (2 * j > lim ? 0xFFFFu : 0u) | (2 * j + 1 > lim ? 0xFFFF0000u : 0u);
            C[j] = as_p((as_u(C[j]) & ~m) | (0x007F007Fu & m));           // UINT8_MAX/2 (ref :170-171)
        }
    }
}

// One aggregation step for the 4 lines of a wave: returns the new packed L_r in Ln and the new
// row minimum; Lp/min_prev are the previous pixel's (ref :329-353).
template <int DPL, bool PAD, int LPP>
static __device__ __forceinline__ unsigned agg_step(const us2 (&C)[DPL / 2], us2 (&Lp)[DPL / 2], unsigned min_prev,
                                                    unsigned pen16, us2 p1v, const us2 (&padmask)[DPL / 2],
                                                    bool first_lane, bool last_lane, CellVec<DPL>& packed_out)
{
    constexpr int NP = DPL / 2;
    const unsigned l4u = (min_prev + pen16) & 0xFFFFu;                     // ref :335, truncated to u16
    const us2 l4 = as_p(l4u | (l4u << 16));
    const us2 mp = as_p(min_prev | (min_prev << 16));
    // d-1 / d+1 neighbours across the lane boundary; 255 where there is none (ref :260-263)
    unsigned from_left, from_right;
    if (LPP >= 32) {                                    // a pixel spans several DPP rows: shift across the whole wave
        from_left = dpp_mov<0x138 /* wave_shr:1 */>(0x00FF00FFu, as_u(Lp[NP - 1]));
        from_right = dpp_mov<0x130 /* wave_shl:1 */>(0x00FF00FFu, as_u(Lp[0]));
    } else {
        from_left = dpp_mov<DPP_ROW_SHR1>(0x00FF00FFu, as_u(Lp[NP - 1]));
        from_right = dpp_mov<DPP_ROW_SHL1>(0x00FF00FFu, as_u(Lp[0]));
    }
    if (LPP != 16) {                                    // pixel boundaries that are not DPP row boundaries: cut the shift there
        from_left = first_lane ? 0x00FF00FFu : from_left;
        from_right = last_lane ? 0x00FF00FFu : from_right;
    }
    us2 Ln[NP];
#pragma unroll
    for (int j = 0; j < NP; ++j) {
        const unsigned below = (j == 0) ? from_left : as_u(Lp[j - 1]);
        const unsigned above = (j == NP - 1) ? from_right : as_u(Lp[j + 1]);
        const us2 dm1 = as_p(__builtin_amdgcn_alignbit(as_u(Lp[j]), below, 16));   // (Lp(d-1), Lp(d))   pairs
        const us2 dp1 = as_p(__builtin_amdgcn_alignbit(above, as_u(Lp[j]), 16));   // (Lp(d+1), Lp(d+2))
        us2 m = pk_min(dm1 + p1v, dp1 + p1v);           // l2, l3 (ref :333-334), each truncated to u16
        m = pk_min(m, Lp[j]);                           // l1
        m = pk_min(m, l4);
        const us2 wide = (C[j] - mp) + m;               // mod 2^16 == the C's int arithmetic mod 2^16
        unsigned r = as_u(wide) & 0x00FF00FFu;          // ref :343 uint8 truncation (Q7)
        if (PAD) r |= as_u(padmask[j]);
        Ln[j] = as_p(r);
    }
    us2 m = Ln[0];
#pragma unroll
    for (int j = 1; j < NP; ++j) m = pk_min(m, Ln[j]);
#pragma unroll
    for (int j = 0; j < NP; ++j) Lp[j] = Ln[j];
    pack_cells<DPL>(Ln, packed_out);
    return row_allmin<LPP>(min(as_u(m) & 0xFFFFu, as_u(m) >> 16));        // ref :347,353
}

enum { AGG_H = 0, AGG_V = 1, AGG_D = 2 };

// Regular lines of one direction kind.  All addressing is 32-bit offsets from the volume bases
// (the host guarantees W*H*Dp < 2^32); the walk is the reference's (ref :281-323, 359-367) with the
// row test dropped (a regular line is never in the last row before its final step) and the two
// edge tests turned into selects.
template <int DPL, bool PAD, int LPP, int KIND>
static __device__ __forceinline__ void agg_regular(const AggArgs& a, const AggFrame& fr, const unsigned short* lut_s,
                                                   int dir, int grp)
{
    constexpr int NP = DPL / 2;
    constexpr int PF = (LPP >= 32) ? 4 : 2;                              // prefetch depth (steps): 2 keeps the 8-lines-per-wave kernel at 8 waves/SIMD; the latency-critical 32-lane lines look further ahead
    const int lane = threadIdx.x;
    const int dx = a.dx[dir], dy = a.dy[dir];
    const int W = a.W, H = a.H, Dp = a.Dp;
    const bool fwd = (dx == 1 && dy == 0) || (dx == 0 && dy == 1) || (dx == 1 && dy == 1) || (dx == -1 && dy == 1);  // ref :232
    const int s = fwd ? 1 : -1;
    // Row tile [row_begin, row_end): horizontal lines are the tile's rows; a vertical / diagonal line enters the
    // tile with the path state of its previous pixel, read from the row just outside the tile in this direction's
    // plane (written by the neighbouring GPU and copied in), or starts with L = C where the tile touches the frame
    // edge the direction starts from.  [0,H) = the whole frame = the reference's walk.
    const int rows = a.row_end - a.row_begin;
    const int skip = (KIND == AGG_H) ? 0 : (fwd ? a.row_begin : H - a.row_end);    // rows between that edge and the tile
    const bool import_state = skip > 0;
    const int nlines = (KIND == AGG_H) ? rows : W;                         // ref :238
    const int nsteps = (KIND == AGG_H) ? W - 1 : (import_state ? rows : rows - 1);   // ref :281
    if (KIND == AGG_D && W < 2) return;                                    // the only line is the anomalous one

    constexpr int LPW = 64 / LPP;                                          // path lines per wave
    const int sub = lane & (LPP - 1);
    const bool first_lane = (sub == 0), last_lane = (sub == LPP - 1);
    int line = grp * LPW + lane / LPP;
    bool store_ok = line < nlines;
    if (!store_ok) line = nlines - 1;                                      // keep the wave convergent; stores are masked
    if (KIND == AGG_D && line == a.anom_line[dir]) {                       // handled by agg_anomalous()
        store_ok = false;
        line = (line == 0) ? 1 : line - 1;
    }
    const unsigned lane_off = (unsigned)(sub * DPL);
    uint8_t* const plane = fr.planes + (size_t)dir * a.plane_bytes;

    us2 padmask[NP];
#pragma unroll
    for (int j = 0; j < NP; ++j) {
        const unsigned lo = ((int)lane_off + 2 * j >= a.D) ? 0x00FFu : 0u;
        const unsigned hi = ((int)lane_off + 2 * j + 1 >= a.D) ? 0x00FF0000u : 0u;
        padmask[j] = as_p(lo | hi);
    }

    // fetch cursor: pixel index p (grey value, census), true column x and byte offset of this lane's cells
    unsigned p, off;
    int x;
    unsigned rowpix = 0, rowoff = 0, pcol = 0, col = 0;                    // AGG_D only
    int dstep_p = 0, dstep_off = 0;
    if (KIND == AGG_H) {
        x = fwd ? 0 : W - 1;
        p = (unsigned)((a.row_begin + line) * W + x);
        dstep_p = s; dstep_off = s * Dp;
    } else if (KIND == AGG_V) {
        x = line;
        // first row of the tile in walking order, or (import) the row before it
        const int r = fwd ? a.row_begin - (import_state ? 1 : 0) : a.row_end - 1 + (import_state ? 1 : 0);
        p = (unsigned)(r * W + line);
        dstep_p = s * W; dstep_off = s * W * Dp;
    } else {
        rowpix = (unsigned)(fwd ? 0 : (H - 1) * W);
        rowoff = rowpix * (unsigned)Dp;
        pcol = col = (unsigned)line;
        x = line;
        p = rowpix + pcol;
        dstep_p = s * W; dstep_off = s * W * Dp;
    }
    off = p * (unsigned)Dp + lane_off;
    const int col_step = (dx == dy) ? s : -s;                              // ref :360-367
    // census-right words of this lane's disparities start (ascending addresses) at pixel p - back
    const int back = a.dmin + (int)lane_off + DPL - 1;
    const int lim_bias = a.dmin + (int)lane_off;                           // disparity i of the lane is in the image iff i <= x - lim_bias

    auto advance = [&]() {
        if (KIND == AGG_H) {
            p += (unsigned)dstep_p;
            off += (unsigned)dstep_off;
            x += s;
        } else if (KIND == AGG_V) {
            p += (unsigned)dstep_p;
            off += (unsigned)dstep_off;
        } else {
            const bool wr = (col == (unsigned)(W - 1));                    // ref :297 (tracker, not true column)
            const bool wl = !wr && (col == 0);                             // ref :304
            pcol = wr ? 0u : (wl ? (unsigned)(W - 1) : pcol + (unsigned)col_step);
            col = ((wr ? 0u : (wl ? (unsigned)(W - 1) : col)) + (unsigned)col_step) & 0xFFFFu;
            rowpix += (unsigned)dstep_p;
            rowoff += (unsigned)dstep_off;
            p = rowpix + pcol;
            x = (int)pcol;
            off = rowoff + __umul24(pcol, (unsigned)Dp) + lane_off;
        }
    };
    auto fetch = [&](CensusVec<DPL>& cv, unsigned& cl, int& g) {
        load_census<DPL>(fr.census_r + ((long long)p - back), cv);
        cl = fr.census_l[p];
        g = fr.img[p];
    };

    // a diagonal line of a row tile: replay the walk from the frame edge up to the pixel before the tile (cheap
    // register arithmetic; it reproduces the tracker state exactly, early wraps included)
    if (KIND == AGG_D)
        for (int i = 0; i + 1 < skip; ++i) advance();

    us2 Lp[NP];
    unsigned min_prev;
    int g_prev;
    if (import_state) {
        // ---- state of the previous pixel: its L_r from the plane, its grey value, min over d ----
        CellVec<DPL> c0;
        load_cells<DPL>(plane + off, c0);
        g_prev = fr.img[p];
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            const unsigned w = c0.w[j >> 1];
            Lp[j] = as_p(__builtin_amdgcn_perm(w, w, (j & 1) ? 0x0c030c02u : 0x0c010c00u));   // bytes -> u16 pairs
        }
        us2 m = Lp[0];
#pragma unroll
        for (int j = 1; j < NP; ++j) m = pk_min(m, Lp[j]);
        min_prev = row_allmin<LPP>(min(as_u(m) & 0xFFFFu, as_u(m) >> 16));
    } else {
        // ---- first pixel of the line: L = C (ref :266-275) ----
        CensusVec<DPL> cv;
        unsigned cl;
        fetch(cv, cl, g_prev);
        const int lim = x - lim_bias;
        census_costs<DPL>(cl, cv, lim, true, Lp);
        if (PAD) {
#pragma unroll
            for (int j = 0; j < NP; ++j) Lp[j] = as_p(as_u(Lp[j]) | as_u(padmask[j]));
        }
        us2 m = Lp[0];
#pragma unroll
        for (int j = 1; j < NP; ++j) m = pk_min(m, Lp[j]);
        min_prev = row_allmin<LPP>(min(as_u(m) & 0xFFFFu, as_u(m) >> 16));
        if (store_ok) {
            CellVec<DPL> o;
            pack_cells<DPL>(Lp, o);
            store_cells<DPL>(plane + off, o);
        }
    }

    // ---- prefetch ring: census-right words, census-left word, grey value, offset, in-image limit ----
    CensusVec<DPL> cb[PF];
    unsigned clb[PF];
    int gb[PF], limb[PF];
    unsigned ob[PF];
#pragma unroll
    for (int u = 0; u < PF; ++u) {
        gb[u] = 0; ob[u] = off; clb[u] = 0; limb[u] = 0;
#pragma unroll
        for (int i = 0; i < DPL; ++i) cb[u].r[i] = 0;
        if (1 + u <= nsteps) {
            advance();
            ob[u] = off;
            limb[u] = x - lim_bias;
            fetch(cb[u], clb[u], gb[u]);
        }
    }
    const us2 p1v = splat((unsigned)a.p1);

    // one step on ring slot u; `refill` = also fetch step k + PF into the slot
    auto step = [&](int u, bool refill) {
        const int g = gb[u];
        const int lim = limb[u];
        const unsigned o = ob[u];
        us2 C[NP];
        census_costs<DPL>(clb[u], cb[u], lim, __any(lim < DPL - 1) != 0, C);   // consume the slot, then refill it
        if (refill) {
            advance();
            ob[u] = off;
            limb[u] = x - lim_bias;
            fetch(cb[u], clb[u], gb[u]);
        }
        const int dg = g > g_prev ? g - g_prev : g_prev - g;
        CellVec<DPL> packed;
        min_prev = agg_step<DPL, PAD, LPP>(C, Lp, min_prev, lut_s[dg], p1v, padmask, first_lane, last_lane, packed);
        g_prev = g;
        if (store_ok) store_cells<DPL>(plane + o, packed);
    };
    // hot loop: all PF steps and all PF refills are in range, no per-step conditions
    int k0 = 1;
    for (; k0 + 2 * PF - 1 <= nsteps; k0 += PF) {
#pragma unroll
        for (int u = 0; u < PF; ++u) step(u, true);
    }
    // tail: at most 2*PF-1 steps
    for (; k0 <= nsteps; k0 += PF) {
#pragma unroll
        for (int u = 0; u < PF; ++u)
            if (k0 + u <= nsteps) step(u, k0 + u + PF <= nsteps);
    }
}

// The anomalous line of a diagonal direction: walked with the reference's full state
// machine incl. the out-of-image end (Q6); its L_r go to the extras rows (the pixels it visits are
// also visited by regular lines), and it zeroes the cells no line visits (W >= H: the track it
// should have taken).  One wave per diagonal direction; all four DPP rows compute the same line,
// row 0 stores.
template <int DPL, bool PAD, int LPP>
static __device__ __forceinline__ void agg_anomalous(const AggArgs& a, const AggFrame& fr, const unsigned short* lut_s,
                                                     int dir)
{
    constexpr int NP = DPL / 2;
    constexpr int NW = (DPL + 3) / 4;
    const int lane = threadIdx.x;
    const int dx = a.dx[dir], dy = a.dy[dir];
    const int W = a.W, H = a.H, Dp = a.Dp;
    const bool fwd = (dx == 1 && dy == 1) || (dx == -1 && dy == 1);
    const int s = fwd ? 1 : -1;
    const int diag_step = s * (W + ((dx == dy) ? 1 : -1));                 // ref :311-322
    const int col_step = (dx == dy) ? s : -s;
    const int nsteps = H - 1;
    const long long npx = (long long)W * H;
    const int line = a.anom_line[dir];
    const int slot = dir - 4;
    const bool store_ok = lane < LPP;
    const int sub = lane & (LPP - 1);
    const bool first_lane = (sub == 0), last_lane = (sub == LPP - 1);
    const unsigned lane_off = (unsigned)(sub * DPL);
    uint8_t* const plane = fr.planes + (size_t)dir * a.plane_bytes;
    uint8_t* const extras = fr.extras + (size_t)slot * H * Dp + lane_off;

    us2 padmask[NP];
#pragma unroll
    for (int j = 0; j < NP; ++j) {
        const unsigned lo = ((int)lane_off + 2 * j >= a.D) ? 0x00FFu : 0u;
        const unsigned hi = ((int)lane_off + 2 * j + 1 >= a.D) ? 0x00FF0000u : 0u;
        padmask[j] = as_p(lo | hi);
    }
    auto ghost_zero = [&](int k) {
        if (store_ok && a.ghost_zero) {
            int gc = line + dx * k;
            if (gc >= W) gc -= W;
            if (gc < 0) gc += W;
            const int gr = fwd ? k : H - 1 - k;
            CellVec<DPL> z;
#pragma unroll
            for (int i = 0; i < NW; ++i) z.w[i] = 0;
            store_cells<DPL>(plane + ((size_t)gr * W + gc) * Dp + lane_off, z);
        }
    };

    long long p = (fwd ? 0 : (long long)(H - 1) * W) + line;
    int row = fwd ? 0 : H - 1, col = line;
    us2 Lp[NP];
    unsigned min_prev;
    int g_prev;
    const int back = a.dmin + (int)lane_off + DPL - 1;
    const int lim_bias = a.dmin + (int)lane_off;
    {
        CensusVec<DPL> cv;
        load_census<DPL>(fr.census_r + (p - back), cv);
        g_prev = fr.img[p];
        census_costs<DPL>(fr.census_l[p], cv, (int)(p % W) - lim_bias, true, Lp);
        if (PAD) {
#pragma unroll
            for (int j = 0; j < NP; ++j) Lp[j] = as_p(as_u(Lp[j]) | as_u(padmask[j]));
        }
        us2 m = Lp[0];
#pragma unroll
        for (int j = 1; j < NP; ++j) m = pk_min(m, Lp[j]);
        min_prev = row_allmin<LPP>(min(as_u(m) & 0xFFFFu, as_u(m) >> 16));
        if (store_ok) {
            CellVec<DPL> o;
            pack_cells<DPL>(Lp, o);
            store_cells<DPL>(extras, o);
        }
        ghost_zero(0);
    }
    const us2 p1v = splat((unsigned)a.p1);
    bool dead = false;
    for (int k = 1; k <= nsteps; ++k) {
        if (!dead) {
            const bool not_last = fwd ? (row < H - 1) : (row > 0);
            if (col == W - 1 && not_last)      { p = (long long)(row + s) * W;           col = 0; }       // ref :297-303
            else if (col == 0 && not_last)     { p = (long long)(row + s) * W + (W - 1); col = W - 1; }   // ref :304-310
            else                               { p += diag_step; }
            row = (row + s) & 0xFFFF;
            col = (col + col_step) & 0xFFFF;
            if (p < 0 || p >= npx) dead = true;                            // Q6: the line ends
        }
        if (!dead) {                                                       // uniform (all rows walk the same line)
            CellVec<DPL> packed;
            CensusVec<DPL> cv;
            us2 C[NP];
            load_census<DPL>(fr.census_r + (p - back), cv);
            const int g = fr.img[p];
            census_costs<DPL>(fr.census_l[p], cv, (int)(p % W) - lim_bias, true, C);
            const int dg = g > g_prev ? g - g_prev : g_prev - g;
            min_prev = agg_step<DPL, PAD, LPP>(C, Lp, min_prev, lut_s[dg], p1v, padmask, first_lane, last_lane, packed);
            g_prev = g;
            if (store_ok) store_cells<DPL>(extras + (size_t)k * Dp, packed);
        }
        ghost_zero(k);
    }
}

template <int DPL, bool PAD, int LPP, int HL>
__global__ __launch_bounds__(64) void sgm_aggregate_k(const AggArgs a)
{
    __shared__ unsigned short lut_s[256];
    const int lane = threadIdx.x;
#pragma unroll
    for (int i = 0; i < 4; ++i) lut_s[lane * 4 + i] = a.lut[lane * 4 + i];
    __syncthreads();

    // Batch: consecutive blocks are the same line group of consecutive frames, so every frame's long
    // horizontal lines are dispatched first.  Per frame, blocks [block_begin[d], block_begin[d+1]) are the
    // regular lines of direction d; the last blocks (one per diagonal direction) are the anomalous lines.
    const int frame = blockIdx.x % a.B;
    const int b = blockIdx.x / a.B;
    AggFrame fr;
    fr.img = a.img + (size_t)frame * a.W * a.H;
    fr.census_l = a.census_l + (size_t)frame * a.W * a.H;
    fr.census_r = a.census_r + (size_t)frame * a.W * a.H;
    fr.planes = a.planes + (size_t)frame * 8 * a.plane_bytes;
    fr.extras = a.extras + (size_t)frame * 4 * a.H * a.Dp;
    if (b >= a.block_begin[8]) {
        agg_anomalous<DPL, PAD, LPP>(a, fr, lut_s, 4 + (b - a.block_begin[8]));
        return;
    }
    int dir = 0;
    while (dir + 1 < a.ndirs && b >= a.block_begin[dir + 1]) ++dir;
    const int grp = b - a.block_begin[dir];
    // the horizontal lines are the longest serial chains of the launch (W-1 dependent steps): their waves
    // get issue priority over the shorter vertical/diagonal ones sharing the SIMD, also across frames in flight
    if (a.dy[dir] == 0) __builtin_amdgcn_s_setprio(3);
    // HL (single-frame mode): the horizontal lines are the critical path of the launch (W-1 serial steps), so
    // they get 32 lanes per pixel -- fewer disparities per lane, the shortest step -- while the vertical and
    // diagonal lines keep the lane count that costs the fewest instructions per cell
    if (a.dy[dir] == 0) {
        if constexpr (HL != 0) agg_regular<DPL * LPP / HL, PAD, HL, AGG_H>(a, fr, lut_s, dir, grp);
        else               agg_regular<DPL, PAD, LPP, AGG_H>(a, fr, lut_s, dir, grp);
    }
    else if (a.dx[dir] == 0) agg_regular<DPL, PAD, LPP, AGG_V>(a, fr, lut_s, dir, grp);
    else                     agg_regular<DPL, PAD, LPP, AGG_D>(a, fr, lut_s, dir, grp);
}

// ============================================================================================
// S = [S +] sum over directions of L_r (+ the second visits of the anomalous lines)
// ============================================================================================

struct WtaState {
    unsigned m1, m2;   // smallest cost (lowest d wins ties, ref :390) and smallest among the others (ref :413-419)
    int d1;            // index (d - dmin) of m1, -1 if nothing beat 65535
    unsigned c1, c2;   // cost_local[best-1], cost_local[best+1] (ref :432-435)
    unsigned pv;       // cost of the previous index
    bool want_next;
};

static __device__ __forceinline__ void wta_feed(WtaState& s, unsigned v, int di)
{
    if (s.want_next) { s.c2 = v; s.want_next = false; }
    if (v < s.m1) {
        s.m2 = s.m1; s.m1 = v; s.d1 = di; s.c1 = s.pv; s.want_next = true; s.c2 = 0xFFFFu;
    } else if (v < s.m2) {
        s.m2 = v;
    }
    s.pv = v;
}

static __device__ __forceinline__ float wta_finish(const WtaState& s, int D, int dmin, int check_unique,
                                                   float one_minus_ratio)
{
    const float inf = __builtin_inff();
    if (s.d1 < 0) return inf;                             // no candidate at all (see oracle/sgm_oracle.c sgmo_wta)
    if (check_unique) {                                   // ref :412-426 (Q10)
        const unsigned margin = (unsigned)(unsigned short)(int)((float)s.m1 * one_minus_ratio);
        if ((int)s.m2 - (int)s.m1 <= (int)margin) return inf;
    }
    if (s.d1 == 0 || s.d1 == D - 1) return inf;          // ref :428
    const int c1 = (int)(short)s.c1, c2 = (int)(short)s.c2;       // (int16_t) casts, 65535 -> -1 (Q11b)
    int denom = (int)(short)(c1 + c2 - 2 * (int)s.m1);
    if (denom < 1) denom = 1;
    return (float)(s.d1 + dmin) + (float)(c1 - c2) / ((float)denom * 2.0f);     // ref :440
}

// OR over the 16 lanes of a DPP row, result in every lane
static __device__ __forceinline__ unsigned row_allor(unsigned v)
{
    v |= dpp_perm<DPP_QUAD_XOR1>(v);
    v |= dpp_perm<DPP_QUAD_XOR2>(v);
    v |= dpp_perm<DPP_ROW_HALF_MIRROR>(v);
    v |= dpp_perm<DPP_ROW_MIRROR>(v);
    return v;
}

// S = [S +] sum of the L_r planes (+ second visits of the anomalous lines), and -- while the 16 lanes of a
// pixel still hold its S vector in registers -- the LEFT-view winner-take-all (ref :374-443 with
// inverse == 0).  16 lanes per pixel, DPL disparities per lane; the kernel is HBM-bound (it streams the 8
// planes once), so the WTA arithmetic rides along for free.
//   key = S << 16 | d: the row-wide minimum key is the first minimum the reference's strict '>' finds.
template <int DPL>
__global__ __launch_bounds__(256) void sgm_sum_wta_k(const uint8_t* __restrict__ planes, size_t plane_bytes, int ndirs,
                                                     const uint8_t* __restrict__ extras,
                                                     const sgmd_row_extra* __restrict__ row_extras,
                                                     const int* __restrict__ row_extra_count, int row_cap, int accumulate,
                                                     uint16_t* __restrict__ S, float* __restrict__ disp_l, int W, int H, int D,
                                                     int Dp, int dmin, int check_unique, float one_minus_ratio, int row0)
{
    const int sub = threadIdx.x & 15;
    const int xr = blockIdx.x * 16 + (threadIdx.x >> 4);
    const bool inside = xr < W;
    const int x = inside ? xr : W - 1;                                  // keep the DPP rows converged; stores are masked
    const int row = row0 + blockIdx.y;                                  // row0: first row of this GPU's row tile
    const size_t off = ((size_t)row * W + x) * Dp + sub * DPL;
    planes += (size_t)blockIdx.z * 8 * plane_bytes;                     // batch: z = frame
    extras += (size_t)blockIdx.z * 4 * H * Dp;
    S += (size_t)blockIdx.z * W * H * Dp;
    disp_l += (size_t)blockIdx.z * W * H;

    unsigned acc[DPL];
    if (accumulate) {                                    // Q14: S was not reset since the last frame
#pragma unroll
        for (int i = 0; i < DPL; ++i) acc[i] = S[off + i];
    } else {
#pragma unroll
        for (int i = 0; i < DPL; ++i) acc[i] = 0;
    }
    auto add_cells = [&](const uint8_t* p, bool nt) {
        CellVec<DPL> v;
        if (nt) load_cells_nt<DPL>(p, v); else load_cells<DPL>(p, v);
#pragma unroll
        for (int i = 0; i < DPL; ++i) acc[i] += (v.w[i >> 2] >> (8 * (i & 3))) & 0xFF;
    };
    for (int d = 0; d < ndirs; ++d) add_cells(planes + (size_t)d * plane_bytes + off, true);
    if (ndirs > 4) {
        const int n = row_extra_count[row];
        for (int j = 0; j < n; ++j) {
            const sgmd_row_extra e = row_extras[row * row_cap + j];
            if ((e.col_slot & 0xFFFF) == x)
                add_cells(extras + ((size_t)(e.col_slot >> 16) * H + e.step) * Dp + sub * DPL, false);
        }
    }
    if (inside) {
        unsigned short* dst = S + off;
#pragma unroll
        for (int i = 0; i < DPL; i += 2)
            *reinterpret_cast<unsigned*>(dst + i) = (acc[i] & 0xFFFFu) | (acc[i + 1] << 16);
    }

    // ---- left-view WTA over the 16 lanes of the pixel ----
    unsigned key[DPL];
    unsigned kmin = 0xFFFFFFFFu;
#pragma unroll
    for (int i = 0; i < DPL; ++i) {
        const int idx = sub * DPL + i;
        key[i] = (idx < D) ? (((acc[i] & 0xFFFFu) << 16) | (unsigned)idx) : 0xFFFFFFFFu;
        kmin = min(kmin, key[i]);
    }
    const unsigned kbest = row_allmin<16>(kmin);
    unsigned k2 = 0xFFFFFFFFu;
#pragma unroll
    for (int i = 0; i < DPL; ++i) k2 = min(k2, key[i] == kbest ? 0xFFFFFFFFu : key[i]);
    const unsigned ksecond = row_allmin<16>(k2);
    const int dbest = (int)(kbest & 0xFFFFu);
    unsigned nb = 0;                                     // S[best-1] | S[best+1] << 16 (ref :432-435)
#pragma unroll
    for (int i = 0; i < DPL; ++i) {
        const int idx = sub * DPL + i;
        if (idx == dbest - 1) nb |= acc[i] & 0xFFFFu;
        if (idx == dbest + 1) nb |= acc[i] << 16;
    }
    nb = row_allor(nb);
    if (inside && sub == 0) {
        WtaState st;
        st.m1 = kbest >> 16;
        st.m2 = ksecond >> 16;                           // 0xFFFF if there is no other disparity, as ref :381
        st.d1 = (kbest == 0xFFFFFFFFu) ? -1 : dbest;
        st.c1 = nb & 0xFFFFu;
        st.c2 = nb >> 16;
        st.pv = 0; st.want_next = false;
        disp_l[(size_t)row * W + x] = wta_finish(st, D, dmin, check_unique, one_minus_ratio);
    }
}

// ============================================================================================
// Fused cost sum + BOTH winner-take-all passes (Dp <= 256): one workgroup walks one image row, 16 columns per
// iteration (16 lanes per pixel as in sgm_sum_wta_k).  The S vectors of the last Dp+32 columns stay in an LDS
// ring, so the right view -- cost of right pixel xr at disparity d is S[y][xr+d][d-dmin] (ref :397-408), a
// diagonal through Dp consecutive columns -- is evaluated from LDS as soon as its last column has been summed,
// with the same 16-lane key-min reduction as the left view.  S itself is then needed by nobody: it is written
// only on request (stage read-back; the host materialises it lazily for a Match without Reset, Q14).  Per
// frame that removes the S write (119 MB at KITTI size) and the S read of sgm_wta_right_k (143 MB) of 1.37 GB.
//   ring: u16 [R][LD], R = Dp + 32 columns, LD = Dp + 2 (odd dword stride); entries of columns >= W and of
//   padding disparities hold 65535 = the reference's "off the image" cost (ref :407).
// ============================================================================================
template <int DPL, int STAGE>
static __device__ __forceinline__ void sumlr_prefetch(CellVec<DPL> (&pre)[2][8], const uint8_t* planes, size_t plane_bytes,
                                                      int ndirs, size_t off)
{
    // always 8 unconditional loads (see SLOW below): with four paths the upper four re-read planes 0..3 and are
    // masked out when they are added
#pragma unroll
    for (int d = 0; d < 8; ++d)
        load_cells_nt<DPL>(planes + (size_t)(d < ndirs ? d : d - 4) * plane_bytes + off, pre[STAGE][d]);
}

// SLOW = the variant that may read (accumulate) or write (store_S) S.  The common one has no conditional global
// access inside the loop at all: hipcc merges s_waitcnt counts over all control-flow paths, and a load behind a
// condition or in a loop of unknown trip count makes it drain the two-iterations-deep plane prefetch (vmcnt(0))
// every iteration -- which is why columns past the row end re-read the last column instead of being skipped, why
// the right-view-only iterations after the last column are a loop of their own, and why the anomalous-line
// visits of the row are staged in LDS up front.
#define SUMLR_MAX_EXTRA 8
template <int DPL, bool SLOW, int THREADS>
__global__ __launch_bounds__(THREADS) void sgm_sum_wta_lr_k(const uint8_t* __restrict__ planes, size_t plane_bytes, int ndirs,
                                                        const uint8_t* __restrict__ extras,
                                                        const sgmd_row_extra* __restrict__ row_extras,
                                                        const int* __restrict__ row_extra_count, int row_cap,
                                                        int accumulate, int store_S, int do_right,
                                                        uint16_t* __restrict__ S, float* __restrict__ disp_l,
                                                        float* __restrict__ disp_r, int W, int H, int D, int dmin,
                                                        int check_unique, float one_minus_ratio, int row0, int seg_len)
{
    constexpr int Dp = 16 * DPL;
    constexpr int LD = Dp + 2;
    constexpr int COLS = THREADS / 16;                                   // columns per iteration
    constexpr int R = Dp + 2 * COLS;
    static_assert(R % COLS == 0, "a ring slot must always belong to the same px");
    __shared__ unsigned short ring[R * LD];
    __shared__ unsigned ex_val[SUMLR_MAX_EXTRA * (Dp / 4)];
    __shared__ int ex_col[SUMLR_MAX_EXTRA];

    const int sub = threadIdx.x & 15;
    const int px = threadIdx.x >> 4;
    const int row = row0 + blockIdx.x;
    planes += (size_t)blockIdx.y * 8 * plane_bytes;                     // batch: y = frame
    extras += (size_t)blockIdx.y * 4 * H * Dp;
    S += (size_t)blockIdx.y * W * H * Dp;
    disp_l += (size_t)blockIdx.y * W * H;
    disp_r += (size_t)blockIdx.y * W * H;
    const size_t row_cells = (size_t)row * W * Dp;
    const int n_extra = (ndirs > 4) ? min(row_extra_count[row], SUMLR_MAX_EXTRA) : 0;    // host: row_cap <= SUMLR_MAX_EXTRA
    for (int t = threadIdx.x; t < n_extra * (Dp / 4); t += THREADS) {
        const int j = t / (Dp / 4), w = t % (Dp / 4);
        const sgmd_row_extra e = row_extras[row * row_cap + j];
        if (w == 0) ex_col[j] = e.col_slot & 0xFFFF;
        ex_val[t] = *reinterpret_cast<const unsigned*>(extras + ((size_t)(e.col_slot >> 16) * H + e.step) * Dp + w * 4);
    }
    __syncthreads();

    // A row may be cut into segments (blockIdx.z) so that a single frame still fills the GPU: the segment owns the
    // left- and right-view pixels [xa, xb) and sums the columns they need, [xa, xb + dmin + D - 1) -- the overlap with
    // the next segment is summed twice (never with SLOW: the host then uses one segment).
    // Main iterations cover real columns; the right view of the last pixels may need (virtual) columns >= W.
    const int xa = blockIdx.z * seg_len;
    const int xb = min(W, xa + seg_len);
    const int x_last = do_right ? xb - 1 + dmin + D - 1 : xb - 1;       // last column any pixel of the segment needs
    const int n_main = (min(x_last + 1, W) - xa + COLS - 1) / COLS;
    const int n_iter = (x_last - xa) / COLS + 1;
    const unsigned upper_mask = (ndirs > 4) ? 0xFFu : 0u;               // planes 4..7 count only with eight paths

    auto cell_off = [&](int x) { return row_cells + (size_t)min(x, W - 1) * Dp + sub * DPL; };
    CellVec<DPL> pre[2][8];
    sumlr_prefetch<DPL, 0>(pre, planes, plane_bytes, ndirs, cell_off(xa + px));
    sumlr_prefetch<DPL, 1>(pre, planes, plane_bytes, ndirs, cell_off(xa + COLS + px));

    int slot = px;                                                       // ring slot of this thread's column: x mod R

    // right-view WTA of the pixel whose last column (disparity D-1) is column x, after the ring holds column x
    auto right_view = [&](int x) {
        __syncthreads();                                                 // the new columns are in the ring
        const int xr = x - dmin - (D - 1);
        int base = slot + R - (D - 1);                                   // ring slot of column xr + dmin = x - (D-1)
        if (base >= R) base -= R;
        unsigned key[DPL], val[DPL];
        unsigned kmin = 0xFFFFFFFFu;
#pragma unroll
        for (int i = 0; i < DPL; ++i) {
            const int k = sub * DPL + i;
            int sl = base + k;
            if (sl >= R) sl -= R;
            val[i] = ring[sl * LD + k];                                  // padding disparities hold 65535
        }
#pragma unroll
        for (int i = 0; i < DPL; ++i) {
            const int k = sub * DPL + i;
            key[i] = (k < D) ? ((val[i] << 16) | (unsigned)k) : 0xFFFFFFFFu;
            kmin = min(kmin, key[i]);
        }
        const unsigned kbest = row_allmin<16>(kmin);
        // runner-up: keys are distinct (they carry d), so key - kbest - 1 (mod 2^32) sends the best to the top
        // and keeps the order of all others
        const unsigned nbest = ~kbest;
        unsigned k2 = 0xFFFFFFFFu;
#pragma unroll
        for (int i = 0; i < DPL; ++i) k2 = min(k2, key[i] + nbest);
        const unsigned ksecond = row_allmin<16>(k2) + kbest + 1;
        const int dbest = (int)(kbest & 0xFFFFu);
        if (xr >= xa && xr < xb && sub == 0) {
            // S[best-1], S[best+1] straight from the ring (a best at either end of the range is invalid anyway,
            // ref :428: clamp the index, the value is not used)
            const int km = max(dbest - 1, 0), kp = min(dbest + 1, Dp - 1);
            int sm = base + km, sp = base + kp;
            if (sm >= R) sm -= R;
            if (sp >= R) sp -= R;
            WtaState st;
            st.m1 = kbest >> 16;
            st.m2 = ksecond >> 16;
            st.d1 = ((kbest >> 16) == 0xFFFFu) ? -1 : dbest;             // nothing beat 65535 (ref :381, strict '>')
            st.c1 = ring[sm * LD + km];
            st.c2 = ring[sp * LD + kp];
            st.pv = 0; st.want_next = false;
            disp_r[(size_t)row * W + xr] = wta_finish(st, D, dmin, check_unique, one_minus_ratio);
        }
    };
    auto next_slot = [&]() {
        slot += COLS;
        if (slot >= R) slot -= R;
    };

    auto main_body = [&](int it, auto stage_tag) {
        constexpr int STAGE = decltype(stage_tag)::value;
        const int x = xa + it * COLS + px;
        const bool inside = x < W;
        const bool mine = x < xb;                                        // left-view output (and S) of this segment
        const size_t off = cell_off(x);
        unsigned acc[DPL];
#pragma unroll
        for (int i = 0; i < DPL; ++i) acc[i] = 0;
        if (SLOW) {
            if (accumulate) {                                            // Q14: S was not reset since the last frame
#pragma unroll
                for (int i = 0; i < DPL; ++i) acc[i] = S[off + i];
            }
        }
#pragma unroll
        for (int d = 0; d < 8; ++d) {
            const unsigned m = (d < 4) ? 0xFFu : upper_mask;
#pragma unroll
            for (int i = 0; i < DPL; ++i) acc[i] += (pre[STAGE][d].w[i >> 2] >> (8 * (i & 3))) & m;
        }
        sumlr_prefetch<DPL, STAGE>(pre, planes, plane_bytes, ndirs, cell_off(x + 2 * COLS));     // columns of iteration it + 2
        for (int j = 0; j < n_extra; ++j) {
            if (ex_col[j] == x) {                                        // second visit of an anomalous line (LDS)
#pragma unroll
                for (int i = 0; i < DPL; ++i) {
                    const int b = sub * DPL + i;
                    acc[i] += (ex_val[j * (Dp / 4) + (b >> 2)] >> (8 * (b & 3))) & 0xFF;
                }
            }
        }
        if (SLOW) {
            if (store_S && mine) {
                unsigned short* dst = S + off;
#pragma unroll
                for (int i = 0; i < DPL; i += 2)
                    *reinterpret_cast<unsigned*>(dst + i) = (acc[i] & 0xFFFFu) | (acc[i + 1] << 16);
            }
        }
        // ---- this column's S vector into the ring (65535 outside the image / the disparity range) ----
        // (also without a right view: the left view fetches S[best +- 1] from here.  A column's slot is only ever
        // written by the wave that owns px = slot mod 16 -- R is a multiple of 16 -- so that read-back needs no barrier)
        {
            unsigned* dst = reinterpret_cast<unsigned*>(&ring[slot * LD + sub * DPL]);
#pragma unroll
            for (int i = 0; i < DPL; i += 2) {
                const int idx = sub * DPL + i;
                const unsigned lo = (inside && idx < D) ? (acc[i] & 0xFFFFu) : 0xFFFFu;
                const unsigned hi = (inside && idx + 1 < D) ? (acc[i + 1] & 0xFFFFu) : 0xFFFFu;
                dst[i >> 1] = lo | (hi << 16);
            }
        }
        // ---- left-view WTA over the 16 lanes of the pixel (as in sgm_sum_wta_k) ----
        unsigned key[DPL];
        unsigned kmin = 0xFFFFFFFFu;
#pragma unroll
        for (int i = 0; i < DPL; ++i) {
            const int idx = sub * DPL + i;
            key[i] = (idx < D) ? (((acc[i] & 0xFFFFu) << 16) | (unsigned)idx) : 0xFFFFFFFFu;
            kmin = min(kmin, key[i]);
        }
        const unsigned kbest = row_allmin<16>(kmin);
        const unsigned nbest = ~kbest;                                   // runner-up as in right_view()
        unsigned k2 = 0xFFFFFFFFu;
#pragma unroll
        for (int i = 0; i < DPL; ++i) k2 = min(k2, key[i] + nbest);
        const unsigned ksecond = row_allmin<16>(k2) + kbest + 1;
        const int dbest = (int)(kbest & 0xFFFFu);
        asm volatile("" ::: "memory");                                   // the wave's ring writes above stay above
        if (mine && sub == 0) {
            const int km = max(dbest - 1, 0), kp = min(dbest + 1, Dp - 1);
            WtaState st;
            st.m1 = kbest >> 16;
            st.m2 = ksecond >> 16;
            st.d1 = (kbest == 0xFFFFFFFFu) ? -1 : dbest;
            st.c1 = ring[slot * LD + km];                                // S[best-1], S[best+1] (unused when best is at an end)
            st.c2 = ring[slot * LD + kp];
            st.pv = 0; st.want_next = false;
            disp_l[(size_t)row * W + x] = wta_finish(st, D, dmin, check_unique, one_minus_ratio);
        }
        if (do_right) right_view(x);
        next_slot();
    };

    int it = 0;
    for (; it + 1 < n_main; it += 2) {
        main_body(it, std::integral_constant<int, 0>{});
        main_body(it + 1, std::integral_constant<int, 1>{});
    }
    if (it < n_main) {
        main_body(it, std::integral_constant<int, 0>{});
        ++it;
    }
    // ---- columns past the image: only the right view is still working (no global loads) ----
    for (; it < n_iter; ++it) {
        unsigned* dst = reinterpret_cast<unsigned*>(&ring[slot * LD + sub * DPL]);
#pragma unroll
        for (int i = 0; i < DPL; i += 2) dst[i >> 1] = 0xFFFFFFFFu;
        right_view(xa + it * COLS + px);
        next_slot();
    }
}

// ============================================================================================
// right-view winner-take-all  (ref :374-443 with inverse == 1): cost of right pixel x at disparity d is
// S[y][x+d][d], 65535 where x+d is off the image (ref :397-408).
//
// One lane = one right-view pixel; a workgroup handles WTA_T consecutive pixels of a row and walks the
// disparity range in chunks of WTA_DC, staging the S columns x0+dmin+dc .. +T+DC-1 through LDS so the
// diagonal gather reads conflict-free LDS (row stride 33 dwords) instead of strided HBM.  LDS reads are
// issued 8 at a time so their latency overlaps the compare chain.
// ============================================================================================

#define WTA_T 256
#define WTA_DC 64
#define WTA_LD (WTA_DC + 2)       // u16 row stride (33 dwords: odd, conflict-free lane stride)

__global__ __launch_bounds__(WTA_T) void sgm_wta_right_k(const uint16_t* __restrict__ S, float* __restrict__ disp_r, int W,
                                                         int H, int D, int Dp, int dmin, int check_unique,
                                                         float one_minus_ratio, int row0)
{
    __shared__ unsigned short tr[(WTA_T + WTA_DC) * WTA_LD];
    const int row = row0 + blockIdx.y;
    const int x0 = blockIdx.x * WTA_T;
    const int i = threadIdx.x;
    const int x = x0 + i;
    const size_t frame_px = (size_t)blockIdx.z * W * H;                // batch: z = frame
    const uint16_t* Srow = S + (frame_px + (size_t)row * W) * Dp;
    disp_r += frame_px;

    WtaState sr;
    sr.m1 = sr.m2 = 0xFFFFu; sr.d1 = -1; sr.c1 = sr.c2 = 0xFFFFu; sr.pv = 0xFFFFu; sr.want_next = false;

    for (int dc = 0; dc < D; dc += WTA_DC) {
        __syncthreads();                                  // previous chunk fully consumed
        // columns x0+dmin+dc .. +T+DC-2, disparities dc..dc+DC-1 in 16-byte pieces; off-image columns and
        // disparities >= D read 65535 (ref :407; feeding 65535 never changes the state of a valid result)
        for (int t = i; t < (WTA_T + WTA_DC) * (WTA_DC / 8); t += WTA_T) {
            const int px = t / (WTA_DC / 8), piece = t % (WTA_DC / 8);
            const int xx = x0 + dmin + dc + px;
            const int d0 = dc + piece * 8;
            uint4 v = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu);
            if (xx < W && d0 < Dp) {
                v = *reinterpret_cast<const uint4*>(Srow + (size_t)xx * Dp + d0);
                if (d0 + 8 > D) {                         // partially / fully padded piece
                    unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                    for (int k = 0; k < 8; ++k)
                        if (d0 + k >= D) w[k >> 1] |= (k & 1) ? 0xFFFF0000u : 0x0000FFFFu;
                    v = make_uint4(w[0], w[1], w[2], w[3]);
                }
            }
            unsigned* dst = reinterpret_cast<unsigned*>(&tr[px * WTA_LD + piece * 8]);
            dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; dst[3] = v.w;
        }
        __syncthreads();
        for (int e0 = 0; e0 < WTA_DC; e0 += 8) {
            unsigned v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = tr[(i + e0 + k) * WTA_LD + e0 + k];
#pragma unroll
            for (int k = 0; k < 8; ++k) wta_feed(sr, v[k], dc + e0 + k);
        }
    }
    if (x < W) disp_r[(size_t)row * W + x] = wta_finish(sr, D, dmin, check_unique, one_minus_ratio);
}

// ============================================================================================
// left-right consistency  (ref :445-470)
// ============================================================================================

__global__ __launch_bounds__(256) void sgm_lrcheck_k(float* __restrict__ dl, const float* __restrict__ dr, int W, int H,
                                                     float thres, int row0)
{
    const int x = blockIdx.x * 256 + threadIdx.x;
    const int y = row0 + blockIdx.y;
    if (x >= W) return;
    const float inf = __builtin_inff();
    dl += (size_t)blockIdx.z * W * H;                                   // batch: z = frame
    dr += (size_t)blockIdx.z * W * H;
    const size_t idx = (size_t)y * W + x;
    const float d = dl[idx];
    if (d == inf) return;
    const int xr = (int)((double)((float)x - d) + 0.5);          // ref :454: float subtract, double add, truncate (Q12)
    if (xr >= 0 && xr < W) {
        const float r = dr[(size_t)y * W + xr];
        if (r == inf) return;                                    // left kept
        if (fabs((double)(d - r)) > (double)thres) dl[idx] = inf;
    } else {
        dl[idx] = inf;
    }
}

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
#define MED_SKEW 3
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
                                                        int Tq)
{
    const int l = threadIdx.x, tq = blockIdx.x, g = blockIdx.y;
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

            float o1 = 0.f, o2 = 0.f;                                // own outputs of the last two steps
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
                    // out(y-1, x+1): produced by the lane above two steps ago (its o2); lane 0 takes the top row
                    const float b = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(tv[j]), __float_as_int(o2),
                                                                               0x138 /* wave_shr:1 */, 0xF, 0xF, false));
                    const float lo = fminf(fminf(T0, T1), b), hi = fmaxf(fmaxf(T0, T1), b);
                    const float mid = __builtin_amdgcn_fmed3f(T0, T1, b);
                    const float s3 = fmaxf(fmaxf(fminf(e3, lo), fminf(e2, mid)), fmaxf(fminf(e1, hi), e0));
                    const float s4 = fmaxf(fmaxf(fminf(e4, lo), fminf(e3, mid)), fmaxf(fminf(e2, hi), e1));
                    const float outv = __builtin_amdgcn_fmed3f(s3, o1, s4);   // border columns: s3 == s4 == original
                    res[j] = outv;
                    T0 = T1; T1 = b;
                    o2 = o1; o1 = outv;
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
// host-callable launchers
// ============================================================================================

extern "C" size_t sgmd_census_slack(const sgmd_geom* g)
{
    // lowest census-right index read is p - (dmin + Dp - 1) with p >= 0; round up to 256 B
    return (((size_t)g->dmin + g->Dp + 8) * sizeof(uint32_t) + 255) & ~(size_t)255;
}

template <int DPL, int LPP, int HL>
static bool launch_aggregate_hl(const AggArgs& a, int blocks, bool pad, hipStream_t st)
{
    constexpr int per = (HL != 0) ? DPL * LPP / HL : DPL;                 // disparities per lane on the horizontal lines
    constexpr bool ok = (HL == 0) || ((DPL * LPP) % HL == 0 && (per == 2 || per == 4 || per == 8 || per == 16) && HL != LPP);
    if constexpr (ok) {
        if (pad) hipLaunchKernelGGL((sgm_aggregate_k<DPL, true, LPP, HL>), dim3(blocks), dim3(64), 0, st, a);
        else     hipLaunchKernelGGL((sgm_aggregate_k<DPL, false, LPP, HL>), dim3(blocks), dim3(64), 0, st, a);
        return true;
    }
    return false;
}
template <int DPL, int LPP>
static void launch_aggregate(const AggArgs& a, int blocks, bool pad, int hl, hipStream_t st)
{
    if (hl == 64 && launch_aggregate_hl<DPL, LPP, 64>(a, blocks, pad, st)) return;
    if (hl == 32 && launch_aggregate_hl<DPL, LPP, 32>(a, blocks, pad, st)) return;
    launch_aggregate_hl<DPL, LPP, 0>(a, blocks, pad, st);
}

template <int DPL, int THREADS>
static void launch_sum_wta_lr(dim3 grid, hipStream_t st, const void* planes, size_t plane_bytes, int ndirs, const void* extras,
                              const void* row_extras, const void* row_extra_count, int row_cap, int accumulate, int store_S,
                              int do_right, void* S, void* disp_l, void* disp_r, const sgmd_geom* g, int check_unique,
                              float one_minus_ratio, int seg_len)
{
#define SUMLR_CALL(SLOW)                                                                                              \
    hipLaunchKernelGGL((sgm_sum_wta_lr_k<DPL, SLOW, THREADS>), grid, dim3(THREADS), 0, st, (const uint8_t*)planes, plane_bytes, ndirs, \
                       (const uint8_t*)extras, (const sgmd_row_extra*)row_extras, (const int*)row_extra_count, row_cap,  \
                       accumulate, store_S, do_right, (uint16_t*)S, (float*)disp_l, (float*)disp_r, g->W, g->H, g->D,    \
                       g->dmin, check_unique, one_minus_ratio, g->row_begin, seg_len)
    if (accumulate || store_S) SUMLR_CALL(true);
    else SUMLR_CALL(false);
#undef SUMLR_CALL
}

template <int DPL>
static void launch_sum_wta(dim3 grid, hipStream_t st, const void* planes, size_t plane_bytes, int ndirs, const void* extras,
                           const void* row_extras, const void* row_extra_count, int row_cap, int accumulate, void* S,
                           void* disp_l, const sgmd_geom* g, int check_unique, float one_minus_ratio)
{
    hipLaunchKernelGGL((sgm_sum_wta_k<DPL>), grid, dim3(256), 0, st, (const uint8_t*)planes, plane_bytes, ndirs,
                       (const uint8_t*)extras, (const sgmd_row_extra*)row_extras, (const int*)row_extra_count, row_cap,
                       accumulate, (uint16_t*)S, (float*)disp_l, g->W, g->H, g->D, g->Dp, g->dmin, check_unique,
                       one_minus_ratio, g->row_begin);
}

extern "C" {

int sgmd_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int sgmd_device_is_gfx950(int ordinal)
{
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, ordinal) != hipSuccess) return -1;
    return strncmp(prop.gcnArchName, "gfx950", 6) == 0 ? 1 : 0;
}

int sgmd_stream_create(int ord, void** stream)
{
    HIP_TRY(hipSetDevice(ord));
    hipStream_t s;
    HIP_TRY(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    *stream = (void*)s;
    return 0;
}
int sgmd_stream_destroy(int ord, void* stream)
{
    HIP_TRY(hipSetDevice(ord));
    HIP_TRY(hipStreamDestroy((hipStream_t)stream));
    return 0;
}
int sgmd_stream_sync(int ord, void* stream)
{
    HIP_TRY(hipSetDevice(ord));
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    return 0;
}
int sgmd_alloc(int ord, void** dptr, size_t bytes)
{
    HIP_TRY(hipSetDevice(ord));
    HIP_TRY(hipMalloc(dptr, bytes ? bytes : 16));
    return 0;
}
int sgmd_free(int ord, void* dptr)
{
    if (!dptr) return 0;
    HIP_TRY(hipSetDevice(ord));
    HIP_TRY(hipFree(dptr));
    return 0;
}
int sgmd_alloc_pinned(int ord, void** hptr, size_t bytes)
{
    HIP_TRY(hipSetDevice(ord));
    HIP_TRY(hipHostMalloc(hptr, bytes ? bytes : 16, hipHostMallocDefault));
    return 0;
}
int sgmd_free_pinned(int ord, void* hptr)
{
    if (!hptr) return 0;
    HIP_TRY(hipSetDevice(ord));
    HIP_TRY(hipHostFree(hptr));
    return 0;
}
int sgmd_h2d_async(int ord, void* stream, void* dst, const void* src, size_t bytes)
{
    HIP_TRY(hipSetDevice(ord));
    HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, (hipStream_t)stream));
    return 0;
}
int sgmd_d2h_async(int ord, void* stream, void* dst, const void* src, size_t bytes)
{
    HIP_TRY(hipSetDevice(ord));
    HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, (hipStream_t)stream));
    return 0;
}
int sgmd_d2d_async(int ord, void* stream, void* dst, const void* src, size_t bytes)
{
    HIP_TRY(hipSetDevice(ord));
    HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return 0;
}
int sgmd_memset_async(int ord, void* stream, void* dst, int value, size_t bytes)
{
    HIP_TRY(hipSetDevice(ord));
    HIP_TRY(hipMemsetAsync(dst, value, bytes, (hipStream_t)stream));
    return 0;
}

struct sgmd_timer { int n; hipEvent_t* ev; };

int sgmd_timer_create(int ord, void** timer, int max_marks)
{
    HIP_TRY(hipSetDevice(ord));
    sgmd_timer* t = new sgmd_timer;
    t->n = max_marks;
    t->ev = new hipEvent_t[max_marks];
    for (int i = 0; i < max_marks; ++i) HIP_TRY(hipEventCreate(&t->ev[i]));
    *timer = t;
    return 0;
}
void sgmd_timer_destroy(int ord, void* timer)
{
    if (!timer) return;
    (void)hipSetDevice(ord);
    sgmd_timer* t = (sgmd_timer*)timer;
    for (int i = 0; i < t->n; ++i) (void)hipEventDestroy(t->ev[i]);
    delete[] t->ev;
    delete t;
}
int sgmd_timer_mark(int ord, void* timer, void* stream, int index)
{
    sgmd_timer* t = (sgmd_timer*)timer;
    if (!t || index < 0 || index >= t->n) return -1;
    HIP_TRY(hipSetDevice(ord));
    HIP_TRY(hipEventRecord(t->ev[index], (hipStream_t)stream));
    return 0;
}
int sgmd_timer_elapsed(int ord, void* timer, int from, int to, float* ms)
{
    sgmd_timer* t = (sgmd_timer*)timer;
    if (!t) return -1;
    HIP_TRY(hipSetDevice(ord));
    HIP_TRY(hipEventElapsedTime(ms, t->ev[from], t->ev[to]));
    return 0;
}

int sgmd_census(int ord, void* stream, const sgmd_geom* g, const void* left, const void* right, void* cl, void* cr)
{
    HIP_TRY(hipSetDevice(ord));
    dim3 grid((g->W + 63) / 64, (g->H + 3) / 4, 2 * g->B);
    hipLaunchKernelGGL(sgm_census_k, grid, dim3(256), 0, (hipStream_t)stream, (const uint8_t*)left,
                       (const uint8_t*)right, (uint32_t*)cl, (uint32_t*)cr, g->W, g->H);
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

int sgmd_aggregate(int ord, void* stream, const sgmd_geom* g, const sgmd_paths* paths, const void* img_left,
                   const void* census_l, const void* census_r, const void* lut, void* planes, size_t plane_bytes,
                   void* extras)
{
    HIP_TRY(hipSetDevice(ord));
    AggArgs a;
    a.img = (const uint8_t*)img_left;
    a.census_l = (const uint32_t*)census_l;
    a.census_r = (const uint32_t*)census_r;
    a.dmin = g->dmin;
    a.lut = (const uint16_t*)lut;
    a.planes = (uint8_t*)planes;
    a.plane_bytes = plane_bytes;
    a.extras = (uint8_t*)extras;
    a.W = g->W; a.H = g->H; a.D = g->D; a.Dp = g->Dp;
    a.B = g->B;
    a.row_begin = g->row_begin; a.row_end = g->row_end;
    a.run_anom = (paths->ndirs > 4 && paths->run_anom) ? 1 : 0;
    a.p1 = paths->p1;
    a.ndirs = paths->ndirs;
    a.ghost_zero = paths->ghost_zero;
    int blocks = 0;
    for (int d = 0; d < 8; ++d) {
        a.dx[d] = paths->dx[d]; a.dy[d] = paths->dy[d]; a.anom_line[d] = paths->anom_line[d];
        a.block_begin[d] = blocks;
        if (d < paths->ndirs && ((paths->dir_mask >> d) & 1)) {
            const int nlines = (paths->dy[d] == 0) ? g->row_end - g->row_begin : g->W;
            const int lines_per_wave = 64 / ((paths->dy[d] == 0 && g->HL) ? g->HL : g->LPP);
            blocks += (nlines + lines_per_wave - 1) / lines_per_wave;
        }
    }
    a.block_begin[8] = blocks;
    if (a.run_anom) blocks += 4;                       // one extra wave per diagonal direction: its anomalous line
    if (blocks == 0) return 0;
    blocks *= g->B;                                    // every frame of the batch in the same launch
    const bool pad = (g->D != g->Dp);
    hipStream_t st = (hipStream_t)stream;
    // (DPL, LPP): disparities per lane x lanes per pixel = Dp.  16 lanes per pixel (4 lines per wave) gives
    // the shortest serial step; 8 lanes per pixel (8 lines per wave) spends ~40 % fewer VALU instructions
    // per cell and is what a batch of frames (VALU-bound) uses.
    const int key = g->LPP * 100 + g->DPL;
    switch (key) {
    case 1602: launch_aggregate<2, 16>(a, blocks, pad, g->HL, st); break;
    case 1604: launch_aggregate<4, 16>(a, blocks, pad, g->HL, st); break;
    case 1608: launch_aggregate<8, 16>(a, blocks, pad, g->HL, st); break;
    case 1612: launch_aggregate<12, 16>(a, blocks, pad, g->HL, st); break;
    case 1616: launch_aggregate<16, 16>(a, blocks, pad, g->HL, st); break;
    case 1632: launch_aggregate<32, 16>(a, blocks, pad, g->HL, st); break;
    case 804:  launch_aggregate<4, 8>(a, blocks, pad, g->HL, st); break;
    case 808:  launch_aggregate<8, 8>(a, blocks, pad, g->HL, st); break;
    case 816:  launch_aggregate<16, 8>(a, blocks, pad, g->HL, st); break;
    default:
        fprintf(stderr, "sgm_mi355x: unsupported lanes-per-pixel/DPL combination %d/%d\n", g->LPP, g->DPL);
        return -1;
    }
    HIP_TRY(hipGetLastError());
    return 0;
}

int sgmd_sum_wta(int ord, void* stream, const sgmd_geom* g, int ndirs, const void* planes, size_t plane_bytes,
                 const void* extras, const void* row_extras, const void* row_extra_count, int row_cap, int accumulate,
                 void* S, int check_unique, float one_minus_ratio, void* disp_l)
{
    HIP_TRY(hipSetDevice(ord));
    const dim3 grid((g->W + 15) / 16, g->row_end - g->row_begin, g->B);
    hipStream_t st = (hipStream_t)stream;
#define SUM_ARGS grid, st, planes, plane_bytes, ndirs, extras, row_extras, row_extra_count, row_cap, accumulate, S, disp_l, g, check_unique, one_minus_ratio
    switch (g->Dp / 16) {                                // 16 lanes per pixel here, whatever the aggregation used
    case 2:  launch_sum_wta<2>(SUM_ARGS); break;
    case 4:  launch_sum_wta<4>(SUM_ARGS); break;
    case 8:  launch_sum_wta<8>(SUM_ARGS); break;
    case 12: launch_sum_wta<12>(SUM_ARGS); break;
    case 16: launch_sum_wta<16>(SUM_ARGS); break;
    case 32: launch_sum_wta<32>(SUM_ARGS); break;
    default:
        fprintf(stderr, "sgm_mi355x: unsupported Dp %d\n", g->Dp);
        return -1;
    }
#undef SUM_ARGS
    HIP_TRY(hipGetLastError());
    return 0;
}

int sgmd_sum_wta_lr_supported(const sgmd_geom* g, int row_cap)
{
    return (g->Dp == 32 || g->Dp == 64 || g->Dp == 128 || g->Dp == 192 || g->Dp == 256) && row_cap <= SUMLR_MAX_EXTRA;
}

int sgmd_sum_wta_lr(int ord, void* stream, const sgmd_geom* g, int ndirs, const void* planes, size_t plane_bytes,
                    const void* extras, const void* row_extras, const void* row_extra_count, int row_cap, int accumulate,
                    int store_S, int do_right, void* S, int check_unique, float one_minus_ratio, void* disp_l, void* disp_r)
{
    HIP_TRY(hipSetDevice(ord));
    // segments per row: enough workgroups for ~4 per CU when a launch has few rows (one frame), but never segments
    // shorter than 2 Dp columns (each re-sums dmin + D - 1 columns of its right neighbour), and one segment whenever S
    // is read or written (the overlap would be accumulated twice)
    const int rows = (g->row_end - g->row_begin) * g->B;
    int segs = 1;
    if (!accumulate && !store_S) {
        const char* e = getenv("SGM_SUM_SEGMENTS");
        segs = (e && *e) ? atoi(e) : (1024 + rows - 1) / rows;
        if (segs > 4) segs = 4;
        while (segs > 1 && g->W / segs < 2 * g->Dp) --segs;
        if (segs < 1) segs = 1;
    }
    const int seg_len = (((g->W + segs - 1) / segs) + 15) / 16 * 16;
    const dim3 grid(g->row_end - g->row_begin, g->B, (g->W + seg_len - 1) / seg_len);
    hipStream_t st = (hipStream_t)stream;
#define SUMLR_ARGS grid, st, planes, plane_bytes, ndirs, extras, row_extras, row_extra_count, row_cap, accumulate, store_S, do_right, S, disp_l, disp_r, g, check_unique, one_minus_ratio, seg_len
    switch (g->Dp / 16) {
    case 2: launch_sum_wta_lr<2, 256>(SUMLR_ARGS); break;
    case 4: launch_sum_wta_lr<4, 256>(SUMLR_ARGS); break;
    case 8: launch_sum_wta_lr<8, 256>(SUMLR_ARGS); break;   // (512 threads = 32 columns per iteration measured the same)
    // larger ranges: the ring takes most of the CU's 160 KB of LDS, one workgroup per CU
    case 12: launch_sum_wta_lr<12, 512>(SUMLR_ARGS); break;  // Dp 192: ring 256 x 194 u16 =  97 KB, 8 waves
    case 16: launch_sum_wta_lr<16, 256>(SUMLR_ARGS); break;  // Dp 256: ring 288 x 258 u16 = 145 KB, 4 waves
    default:
        fprintf(stderr, "sgm_mi355x: fused sum/WTA needs Dp <= 256 (got %d)\n", g->Dp);
        return -1;
    }
#undef SUMLR_ARGS
    HIP_TRY(hipGetLastError());
    return 0;
}

int sgmd_wta_right(int ord, void* stream, const sgmd_geom* g, const void* S, int check_unique, float one_minus_ratio,
                   void* disp_r)
{
    HIP_TRY(hipSetDevice(ord));
    dim3 grid((g->W + WTA_T - 1) / WTA_T, g->row_end - g->row_begin, g->B);
    hipLaunchKernelGGL(sgm_wta_right_k, grid, dim3(WTA_T), 0, (hipStream_t)stream, (const uint16_t*)S, (float*)disp_r,
                       g->W, g->H, g->D, g->Dp, g->dmin, check_unique, one_minus_ratio, g->row_begin);
    HIP_TRY(hipGetLastError());
    return 0;
}

int sgmd_lrcheck(int ord, void* stream, const sgmd_geom* g, void* disp_l, const void* disp_r, float thres)
{
    HIP_TRY(hipSetDevice(ord));
    dim3 grid((g->W + 255) / 256, g->row_end - g->row_begin, g->B);
    hipLaunchKernelGGL(sgm_lrcheck_k, grid, dim3(256), 0, (hipStream_t)stream, (float*)disp_l, (const float*)disp_r,
                       g->W, g->H, thres, g->row_begin);
    HIP_TRY(hipGetLastError());
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

size_t sgmd_median_scratch_bytes(const sgmd_geom* g)
{
    const int groups = (g->H - 2 + 63) / 64;
    if (groups <= 0) return 16;
    return (size_t)g->B * groups * med_tq(g->W) * (MED_NE + 1) * 64 * sizeof(float4);   // inputs (MED_NE planes) + results
}

int sgmd_median(int ord, void* stream, const sgmd_geom* g, void* disp, void* scratch)
{
    HIP_TRY(hipSetDevice(ord));
    const int rows = g->H - 2;
    if (rows <= 0 || g->W <= 2) return 0;                            // no interior pixel: the filter is a no-op
    const int groups = (rows + 63) / 64;
    const int Tq = med_tq(g->W);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(sgm_median_prep_k, dim3(Tq, groups, g->B), dim3(64), 0, st, (const float*)disp, (float4*)scratch, g->W,
                       g->H, Tq);
    const int waves = groups < MED_WAVES ? groups : MED_WAVES;
    float4* const results = (float4*)scratch + (size_t)g->B * groups * Tq * MED_NE * 64;
    hipLaunchKernelGGL(sgm_median_serial_k, dim3(g->B), dim3(64 * waves), 0, st, (const float*)disp, (float4*)scratch, results,
                       g->W, g->H, Tq);
    hipLaunchKernelGGL(sgm_median_unskew_k, dim3((g->W - 2 + 255) / 256, g->H - 2, g->B), dim3(256), 0, st,
                       (const float*)results, (float*)disp, g->W, g->H, Tq);
    HIP_TRY(hipGetLastError());
    return 0;
}

}  // extern "C"

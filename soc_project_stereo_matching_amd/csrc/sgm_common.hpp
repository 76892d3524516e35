// sgm_common.hpp -- shared by the HIP translation units of libsgm_mi355x.so (see sgm_device.h for the C interface).
// Hand-written gfx950 (CDNA4, wave64) kernels of the SGM hot path, one translation unit per stage:
//   sgm_census.hip  sgm_aggregate.hip  sgm_sum_wta.hip  sgm_post.hip (speckle, median)  sgm_runtime.hip  No MFMA: the path is integer min-plus and
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

#pragma once
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


// ---- one pixel's cells (DPL bytes per lane) ----
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

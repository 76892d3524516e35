#pragma once
#include "sgm_common.hpp"

#ifndef SGM_AGG_PF
#define SGM_AGG_PF 2      // steps of census words / grey values a wave keeps in flight (8- and 16-lane lines); 3 measured no faster (NOTES.md 9)
#endif
#ifndef SGM_ANOM_PF
#define SGM_ANOM_PF 4     // ... and for the anomalous diagonal lines (a kernel of their own: registers are no concern there)
#endif
#ifndef SGM_AGG_HBURST
#define SGM_AGG_HBURST 1  // horizontal lines: L_r of this many consecutive steps are stored back to back (a burst of HBURST x 128 B per line); 1 = store every step
#endif
#ifndef SGM_AGG_PF_HL
#define SGM_AGG_PF_HL 4   // the same for the 32- / 64-lane horizontal lines of a single frame (the launch's critical chain)
#endif

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
    const uint8_t* cost;        // VOL kernels only: the materialised cost volume u8 [B][H][W][Dp] (census pointers unused)
    int dmin;
    const uint16_t* lut;        // (uint16) max(P1, P2 / (|dg| + 1)), 256 entries (ref :335)
    uint8_t* planes;
    size_t plane_bytes;
    uint8_t* extras;
    int W, H, D, Dp;
    int row_begin, row_end;     // rows of the frame this launch covers (a row tile of a multi-GPU run; [0,H) normally)
    int run_anom;               // 1: also run the four anomalous diagonal lines (whole frame)
    int anom_inline;            // 1: ... as the FIRST four blocks of every frame of the regular lines' launch (sgm_aggregate_k); 0: sgm_aggregate_anom_k
    int B;                      // frames per launch; frame f uses img/census + f*W*H, planes + f*8*plane_bytes, extras + f*4*H*Dp
    int p1;
    int ndirs;
    int dx[8], dy[8];
    int anom_line[8];
    int block_begin[9];
    int ghost_zero;
    int strips;                 // XCD-aware numbering of the workgroups (see sgm_aggregate_k): XCDs per frame, 1 = plain numbering
    // the fused last sweep (sgm_upsum.hip) computes the upward directions itself except the POST-wrap cells of the diagonal lines that
    // wrap around the image edge: for those directions this launch walks only lines [line_lo, line_lo + line_n) and stores only
    // what lies behind the wrap (post_wrap_mask: bit d).  Everything else: line_lo = 0, line_n = all lines, mask 0.
    int line_lo[8], line_n[8];
    int post_wrap_mask;
};

// per-frame base pointers of a batched launch (kept apart from the kernel-argument struct so that struct
// stays in the scalar kernarg segment)
struct AggFrame {
    const uint8_t* img;
    const uint32_t* census_l;
    const uint32_t* census_r;
    const uint8_t* cost;
    uint8_t* planes;
    uint8_t* extras;
};

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

// VOL kernels: the lane's DPL cost bytes (loaded from the volume into cv.r[0..]) -> packed u16 pairs
template <int DPL>
static __device__ __forceinline__ void volume_costs(const CensusVec<DPL>& cv, us2 (&C)[DPL / 2])
{
#pragma unroll
    for (int j = 0; j < DPL / 2; ++j) {
        const unsigned w = cv.r[j >> 1];
        C[j] = as_p(__builtin_amdgcn_perm(w, w, (j & 1) ? 0x0c030c02u : 0x0c010c00u));
    }
}
template <int DPL>
static __device__ __forceinline__ void load_volume(const uint8_t* p, CensusVec<DPL>& cv)
{
    CellVec<DPL> c;
    load_cells<DPL>(p, c);
#pragma unroll
    for (int i = 0; i < (DPL + 3) / 4; ++i) cv.r[i] = c.w[i];
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

// ---- the same step for non-negative penalties (P1 >= 0; every sane configuration), rearranged ----
//   L(d) = C(d) + min( min(Lp(d), Lp(d-1)+P1, Lp(d+1)+P1) - min_prev,  P2' )        (ref :329-343)
// Every candidate is >= min_prev and < 2^15 then (Lp <= 255, sentinels 255, 0 <= P1, P2' <= 32767), so min_prev can
// leave the minimum, the bracket is a small non-negative number in both 16-bit halves, and v_bcnt_u32_b32's
// accumulate operand adds the Hamming cost onto it for free (no carry can cross the halves): one instruction per pair
// fewer, three fewer per step for the P2 term, and less of the step waits for min_prev.  hipcc does not form the
// accumulating popcount on its own, hence the two inline-asm helpers.  Disparities left of column 0 cost 127
// (ref :170-171): patched in afterwards on the (wave-uniform, rare) border steps.
static __device__ __forceinline__ unsigned bcnt_acc(unsigned x, unsigned acc)
{
    unsigned r;
    asm("v_bcnt_u32_b32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(acc));
    return r;
}
static __device__ __forceinline__ unsigned shl16_add(unsigned a, unsigned b)
{
    unsigned r;
    asm("v_lshl_add_u32 %0, %1, 16, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// FAST (NN == 2): two more things the host guarantees for ordinary penalties (sgm_aggregate.hip: 0 <= P1 <= 31488 and
// max(P1, P2_init) <= 223):
//  * every operand of the neighbour minimum is a 16-bit pattern below 0x7C00 (L <= 255, L + P1 <= 31743), i.e. a positive finite
//    binary16 -- and those order exactly like their bit patterns, so gfx950's packed three-way v_pk_minimum3_f16 IS the unsigned
//    minimum of (Lp(d-1)+P1, Lp(d+1)+P1, Lp(d)): one instruction instead of two v_pk_min_u16 (no value is ever interpreted as a
//    number: denormal patterns pass through unchanged, tools/ubench/pk_min3_probe.hip checks all 31744^2 pairs); the in-lane tree
//    of the row minimum halves the same way;
//  * away from the left border C <= 32 and the bracket <= P2' <= 223, so C + bracket < 256 and the uint8 truncation (ref :343,
//    Q7) is the identity there: the AND only runs on the (wave-uniform, rare) border steps, where C = 127 can wrap.
static __device__ __forceinline__ us2 pk_min3_f16(us2 a, us2 b, us2 c)
{
    unsigned r;
    asm("v_pk_minimum3_f16 %0, %1, %2, %3" : "=v"(r) : "v"(as_u(a)), "v"(as_u(b)), "v"(as_u(c)));
    return as_p(r);
}
template <int N, bool FAST>
static __device__ __forceinline__ us2 pk_min_tree(const us2 (&v)[N])
{
    if constexpr (!FAST || N < 3) {
        us2 m = v[0];
#pragma unroll
        for (int j = 1; j < N; ++j) m = pk_min(m, v[j]);
        return m;
    } else {
        us2 m = pk_min3_f16(v[0], v[1], v[2]);
        int j = 3;
#pragma unroll
        for (; j + 1 < N; j += 2) m = pk_min3_f16(m, v[j], v[j + 1]);
        if (j < N) m = pk_min(m, v[j]);
        return m;
    }
}
// min over the lanes of a pixel of both halves of m, returned in BOTH halves of every lane (the packed operand the next step
// subtracts): one VOP3P min with swapped halves, then the DPP all-reduce on the replicated word (u32 order = u16 order then)
template <int LPP>
static __device__ __forceinline__ unsigned row_allmin_pk(us2 m)
{
    unsigned r;
    asm("v_pk_min_u16 %0, %1, %1 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(r) : "v"(as_u(m)));
    return row_allmin<LPP>(r);
}
// `mp` = min over d of the previous pixel's L in both halves (in and out); `sent` = two registers the caller initialises to
// 0x00FF00FF and carries from step to step: the DPP shifts write into them, so the lanes without a source lane keep the 255
// sentinels (ref :260-263) without a move per step.
template <int DPL, bool PAD, int LPP, bool FAST = false>
static __device__ __forceinline__ unsigned agg_step_nn(unsigned cl, const CensusVec<DPL>& cv, int lim, bool border,
                                                       us2 (&Lp)[DPL / 2], unsigned mp, unsigned pen16, us2 p1v,
                                                       const us2 (&padmask)[DPL / 2], bool first_lane, bool last_lane,
                                                       unsigned (&sent)[2], CellVec<DPL>& packed_out)
{
    constexpr int NP = DPL / 2;
    const us2 p2v = as_p(pen16);                                    // both halves hold the penalty (32-bit table)
    const unsigned p1u = as_u(p1v);
    unsigned from_left, from_right;
    if (LPP >= 32) {
        from_left = sent[0] = dpp_mov<0x138 /* wave_shr:1 */>(sent[0], as_u(Lp[NP - 1]));
        from_right = sent[1] = dpp_mov<0x130 /* wave_shl:1 */>(sent[1], as_u(Lp[0]));
    } else {
        from_left = sent[0] = dpp_mov<DPP_ROW_SHR1>(sent[0], as_u(Lp[NP - 1]));
        from_right = sent[1] = dpp_mov<DPP_ROW_SHL1>(sent[1], as_u(Lp[0]));
    }
    if (LPP != 16) {
        from_left = first_lane ? 0x00FF00FFu : from_left;
        from_right = last_lane ? 0x00FF00FFu : from_right;
    }
    unsigned w[NP], md[NP];
#pragma unroll
    for (int j = 0; j < NP; ++j) {
        const unsigned below = (j == 0) ? from_left : as_u(Lp[j - 1]);
        const unsigned above = (j == NP - 1) ? from_right : as_u(Lp[j + 1]);
        // plain 32-bit add / subtract instead of v_pk_add_u16 / v_pk_sub_u16 (full rate against half rate, tools/ubench):
        // no carry or borrow crosses the halves -- Lp <= 255, 0 <= P1 <= 32767, and every candidate is >= min_prev
        const us2 dm1 = as_p(__builtin_amdgcn_alignbit(as_u(Lp[j]), below, 16) + p1u);
        const us2 dp1 = as_p(__builtin_amdgcn_alignbit(above, as_u(Lp[j]), 16) + p1u);
        us2 m;
        if constexpr (FAST) m = pk_min3_f16(dm1, dp1, Lp[j]);
        else m = pk_min(pk_min(dm1, dp1), Lp[j]);
        md[j] = as_u(pk_min(as_p(as_u(m) - mp), p2v));
        const unsigned lo = bcnt_acc(cl ^ cv.r[DPL - 1 - 2 * j], md[j]);              // C(2j) + md, high half = md's
        w[j] = shl16_add((unsigned)__popc(cl ^ cv.r[DPL - 2 - 2 * j]), lo);           // + C(2j+1) << 16
    }
    if (border) {                                                                     // wave-uniform and rare: a real branch
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            const unsigned msk = (2 * j > lim ? 0xFFFFu : 0u) | (2 * j + 1 > lim ? 0xFFFF0000u : 0u);
            w[j] = (w[j] & ~msk) | ((md[j] + 0x007F007Fu) & msk);
            if constexpr (FAST) w[j] &= 0x00FF00FFu;                                  // only C = 127 can pass 255 (see above)
        }
    }
    us2 Ln[NP];
#pragma unroll
    for (int j = 0; j < NP; ++j) {
        unsigned r = w[j];
        if constexpr (!FAST) r &= 0x00FF00FFu;                                        // uint8 truncation (Q7)
        if (PAD) r |= as_u(padmask[j]);
        Ln[j] = as_p(r);
    }
    const us2 m = pk_min_tree<NP, FAST>(Ln);
#pragma unroll
    for (int j = 0; j < NP; ++j) Lp[j] = Ln[j];
    pack_cells<DPL>(Ln, packed_out);
    return row_allmin_pk<LPP>(m);
}

enum { AGG_H = 0, AGG_V = 1, AGG_D = 2 };

// Regular lines of one direction kind.  All addressing is 32-bit offsets from the volume bases
// (the host guarantees W*H*Dp < 2^32); the walk is the reference's (ref :281-323, 359-367) with the
// row test dropped (a regular line is never in the last row before its final step) and the two
// edge tests turned into selects.
// WIDE (diagonal lines only): W > H.  A regular line of such an image wraps at most once, exactly where its true column
// leaves the image, so the walk is the closed form of SURVEY.md Q5 -- pixel (k, (i + dx k) mod W) -- and costs 7
// instructions per step instead of the ~17 of the reference's two-tracker state machine (which W <= H still needs:
// there the off-by-one tracker makes lines wrap early).
// VOL: the matching cost comes from a materialised volume (wide census windows) instead of the census images.
// PWO (diagonal lines of W > H only): store only the cells BEHIND the line's wrap around the image edge -- what the fused last sweep
// (sgm_upsum.hip) cannot compute itself; a separate instantiation, so the ordinary lines pay nothing for it.
template <int DPL, bool PAD, int LPP, int KIND, int NN, bool WIDE = false, bool VOL = false, bool PWO = false>
static __device__ __forceinline__ void agg_regular(const AggArgs& a, const AggFrame& fr, const unsigned short* lut_s,
                                                   const unsigned* lut32_s, int dir, int grp)
{
    constexpr int NP = DPL / 2;
    constexpr int PF = (LPP >= 32) ? SGM_AGG_PF_HL : SGM_AGG_PF;                              // prefetch depth (steps): 2 keeps the 8-lines-per-wave kernel at 8 waves/SIMD; the latency-critical 32-lane lines look further ahead
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
    const int nlines = (KIND == AGG_H) ? rows : a.line_lo[dir] + a.line_n[dir];   // ref :238 (W lines; a subset for the fused last sweep)
    constexpr bool pwo = PWO;
    bool wrapped = false;
    const int nsteps = (KIND == AGG_H) ? W - 1 : (import_state ? rows : rows - 1);   // ref :281
    if (KIND == AGG_D && W < 2) return;                                    // the only line is the anomalous one

    constexpr int LPW = 64 / LPP;                                          // path lines per wave
    const int sub = lane & (LPP - 1);
    const bool first_lane = (sub == 0), last_lane = (sub == LPP - 1);
    int line = ((KIND == AGG_H) ? 0 : a.line_lo[dir]) + grp * LPW + lane / LPP;
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
    // WIDE: column the line wraps from / to, and the pixel / byte steps of an ordinary and of a wrapping move
    const unsigned wrap_at = (col_step > 0) ? (unsigned)(W - 1) : 0u, wrap_to = (col_step > 0) ? 0u : (unsigned)(W - 1);
    const unsigned pstep_n = (unsigned)(dstep_p + col_step), pstep_w = (unsigned)(dstep_p - col_step * (W - 1));
    const unsigned ostep_n = pstep_n * (unsigned)Dp, ostep_w = pstep_w * (unsigned)Dp;
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
        } else if (WIDE) {
            const bool wrap = (pcol == wrap_at);
            if constexpr (PWO) wrapped = wrapped || wrap;
            pcol = wrap ? wrap_to : pcol + (unsigned)col_step;
            p += wrap ? pstep_w : pstep_n;
            off += wrap ? ostep_w : ostep_n;
            x = (int)pcol;
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
    // census words through 32-bit byte offsets from uniform bases (scalar base + VGPR offset addressing; a 64-bit address per
    // load costs four VALU instructions per step): the right image's base is moved dmin + Dp words down into the slack in front
    // of the allocation, so the lane's first word, p - back, is never negative
    const unsigned cbias = (unsigned)(a.dmin + Dp - back);
    const char* const crb = reinterpret_cast<const char*>(fr.census_r - (a.dmin + Dp));
    const char* const clb_base = reinterpret_cast<const char*>(fr.census_l);
    auto fetch = [&](CensusVec<DPL>& cv, unsigned& cl, uint8_t& g) {
        if constexpr (VOL) {
            load_volume<DPL>(fr.cost + off, cv);
            cl = 0;
        } else {
            load_census<DPL>(reinterpret_cast<const uint32_t*>(crb + (size_t)((p + cbias) << 2)), cv);
            cl = *reinterpret_cast<const uint32_t*>(clb_base + (size_t)(p << 2));
        }
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
        uint8_t g0;
        fetch(cv, cl, g0);
        g_prev = g0;
        const int lim = x - lim_bias;
        if constexpr (VOL) volume_costs<DPL>(cv, Lp);
        else census_costs<DPL>(cl, cv, lim, true, Lp);
        if (PAD) {
#pragma unroll
            for (int j = 0; j < NP; ++j) Lp[j] = as_p(as_u(Lp[j]) | as_u(padmask[j]));
        }
        us2 m = Lp[0];
#pragma unroll
        for (int j = 1; j < NP; ++j) m = pk_min(m, Lp[j]);
        min_prev = row_allmin<LPP>(min(as_u(m) & 0xFFFFu, as_u(m) >> 16));
        if (store_ok && !pwo) {
            CellVec<DPL> o;
            pack_cells<DPL>(Lp, o);
            store_cells<DPL>(plane + off, o);
        }
    }

    // ---- prefetch ring: census-right words, census-left word, grey value, offset, in-image limit ----
    // The ring's grey values stay BYTES and are widened where a step uses them.  As 32-bit values LLVM zero-extends them at the end
    // of every pass of the hot loop (one v_and per slot on the just-loaded registers, sunk to the loop latch), so the wave drained
    // ALL its outstanding loads and plane stores (s_waitcnt vmcnt(0)) every PF steps whatever the prefetch depth.
    CensusVec<DPL> cb[PF];
    unsigned clb[PF];
    uint8_t gb[PF];
    int limb[PF];
    unsigned ob[PF];
    bool wb[PF];                                                           // the slot's pixel lies behind the line's wrap
#pragma unroll
    for (int u = 0; u < PF; ++u) {
        gb[u] = 0; ob[u] = off; clb[u] = 0; limb[u] = 0; wb[u] = false;
#pragma unroll
        for (int i = 0; i < DPL; ++i) cb[u].r[i] = 0;
        if (1 + u <= nsteps) {
            advance();
            ob[u] = off;
            if constexpr (PWO) wb[u] = wrapped;
            limb[u] = x - lim_bias;
            fetch(cb[u], clb[u], gb[u]);
        }
    }
    const us2 p1v = splat((unsigned)a.p1);
    unsigned sent[2] = {0x00FF00FFu, 0x00FF00FFu};                         // agg_step_nn's carried sentinel registers
    if constexpr (NN) min_prev |= min_prev << 16;                          // ... and its packed minimum

    // one step on ring slot u; `refill` = also fetch step k + PF into the slot; `defer` != nullptr: hand the cells and their offset
    // back instead of storing them (the horizontal lines' bursts)
    auto step = [&](int u, bool refill, CellVec<DPL>* defer = nullptr, unsigned* defer_o = nullptr) {
        const int g = (int)gb[u];
        const int lim = limb[u];
        const unsigned o = ob[u];
        const bool st_ok = store_ok && (!pwo || wb[u]);
        const unsigned dg = __builtin_amdgcn_sad_u8((unsigned)g, (unsigned)g_prev, 0u);   // |g - g_prev| (grey values: one byte)
        CellVec<DPL> packed;
        if constexpr (NN) {
            min_prev = agg_step_nn<DPL, PAD, LPP, NN == 2>(clb[u], cb[u], lim, __any(lim < DPL - 1) != 0, Lp, min_prev, lut32_s[dg], p1v,
                                                  padmask, first_lane, last_lane, sent, packed);
            if (refill) {                                                      // the slot's census words are consumed now
                advance();
                ob[u] = off;
                if constexpr (PWO) wb[u] = wrapped;
                limb[u] = x - lim_bias;
                fetch(cb[u], clb[u], gb[u]);
            }
        } else {
            us2 C[NP];
            if constexpr (VOL) volume_costs<DPL>(cb[u], C);
            else census_costs<DPL>(clb[u], cb[u], lim, __any(lim < DPL - 1) != 0, C);   // consume the slot, then refill it
            if (refill) {
                advance();
                ob[u] = off;
                if constexpr (PWO) wb[u] = wrapped;
                limb[u] = x - lim_bias;
                fetch(cb[u], clb[u], gb[u]);
            }
            min_prev = agg_step<DPL, PAD, LPP>(C, Lp, min_prev, lut_s[dg], p1v, padmask, first_lane, last_lane, packed);
        }
        g_prev = g;
        if (defer) { *defer = packed; *defer_o = o; }
        else if (st_ok) store_cells<DPL>(plane + o, packed);
    };
    // Everything the prologue has in flight lands before the hot loop (once per line).  The waits inside the loop are placed for
    // the state merged over BOTH ways into the loop head; with the prologue's loads still pending in whatever order the scheduler
    // left them, that merge is "wait for everything" -- s_waitcnt vmcnt(0) at the top of EVERY pass, plane stores included.
    __builtin_amdgcn_s_waitcnt(0x0F70);                                     // vmcnt(0)
    // hot loop: all PF steps and all PF refills are in range, no per-step conditions
    int k0 = 1;
    if constexpr (KIND == AGG_H && NN != 0 && SGM_AGG_HBURST > 1 && SGM_AGG_HBURST % PF == 0) {
        // horizontal lines: a wave's lines are rows apart, so every step writes one 128-byte line per row; stored step by step those
        // lines reach HBM microseconds apart, interleaved with thousands of other rows.  HBURST steps' cells go out back to back instead.
        constexpr int HB = SGM_AGG_HBURST;
        for (; k0 + HB + PF - 1 <= nsteps; k0 += HB) {
            CellVec<DPL> pk[HB];
            unsigned po[HB];
#pragma unroll
            for (int j = 0; j < HB; ++j) step(j % PF, true, &pk[j], &po[j]);
            if (store_ok) {
#pragma unroll
                for (int j = 0; j < HB; ++j) store_cells<DPL>(plane + po[j], pk[j]);
            }
        }
    }
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
template <int DPL, bool PAD, int LPP, int NN, bool VOL = false>
static __device__ __forceinline__ void agg_anomalous(const AggArgs& a, const AggFrame& fr, const unsigned short* lut_s,
                                                     const unsigned* lut32_s, int dir)
{
    static_assert(!(NN && VOL), "the volume-fed lines run the generic step");
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
            if (gr >= a.row_begin && gr < a.row_end) {             // a row tile's planes only have storage for its own rows
                CellVec<DPL> z;
#pragma unroll
                for (int i = 0; i < NW; ++i) z.w[i] = 0;
                store_cells<DPL>(plane + ((size_t)gr * W + gc) * Dp + lane_off, z);
            }
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
        g_prev = fr.img[p];
        if constexpr (VOL) {
            load_volume<DPL>(fr.cost + (size_t)p * Dp + lane_off, cv);
            volume_costs<DPL>(cv, Lp);
        } else {
            load_census<DPL>(fr.census_r + (p - back), cv);
            census_costs<DPL>(fr.census_l[p], cv, (int)(p % W) - lim_bias, true, Lp);
        }
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
    // The walk (ref :281-323: positions only) does not depend on the path costs, so it runs PF steps ahead of them and the loads
    // of a step are in flight while the steps before it compute -- as on the regular lines.  Without that every step waited for its
    // own image byte and census words (374 dependent round trips at KITTI size: 0.3 ms -- the longest wave of a single frame's
    // aggregation, and the tail of a batch's).
    constexpr int PF = (DPL >= 12) ? 2 : SGM_ANOM_PF;                      // wide lanes: the ring must not outgrow the regular lines' registers
    int pc = line;                                                         // the TRUE column of p (the walk's own `col` runs one off, Q5)
    auto advance = [&]() {
        const bool not_last = fwd ? (row < H - 1) : (row > 0);
        if (col == W - 1 && not_last)      { p = (long long)(row + s) * W;           col = 0;     pc = 0; }       // ref :297-303
        else if (col == 0 && not_last)     { p = (long long)(row + s) * W + (W - 1); col = W - 1; pc = W - 1; }   // ref :304-310
        else {
            p += diag_step;
            pc += col_step;                                                // diag_step = +-(W +- 1): the column moves by +-1 ...
            if (pc >= W) pc -= W;                                          // ... and runs over the row end into the neighbouring row
            if (pc < 0) pc += W;
        }
        row = (row + s) & 0xFFFF;
        col = (col + col_step) & 0xFFFF;
    };
    // Q6: the line may leave the image before its H - 1 steps are done and ends there.  Where, is found by walking it once without
    // the costs (a few scalar instructions per step), so that the loops below have no end-of-line test -- and no load behind one:
    // hipcc merges s_waitcnt counts over all control-flow paths, a conditional load makes it wait for (nearly) everything in
    // flight at every step, ring or no ring.
    int n_live = nsteps;
    {
        const long long p0 = p;
        const int row0 = row, col0 = col;
        for (int k = 1; k <= nsteps; ++k) {
            advance();
            if (p < 0 || p >= npx) { n_live = k - 1; break; }
        }
        p = p0; row = row0; col = col0; pc = line;
    }
    CensusVec<DPL> cb[PF];
    unsigned clb[PF];
    uint8_t gb[PF];                                                        // bytes, widened at use (see agg_regular)
    int limb[PF];
    // The position is the same in every lane, so the image byte and the census-left word would be SCALAR loads -- which return out
    // of order and can only be waited for all at once (lgkmcnt(0)): every step would wait for the loads the step before it has
    // just issued (measured: 0.48 us per step).  An opaque zero in a vector register makes them vector loads.
    unsigned vzero;
    asm("v_mov_b32 %0, 0" : "=v"(vzero));
    auto fetch = [&](int u) {
        const long long pv = p + vzero;
        gb[u] = fr.img[pv];
        if constexpr (VOL) {
            load_volume<DPL>(fr.cost + (size_t)p * Dp + lane_off, cb[u]);
            clb[u] = 0;
        } else {
            load_census<DPL>(fr.census_r + (p - back), cb[u]);
            clb[u] = fr.census_l[pv];
        }
        limb[u] = pc - lim_bias;
    };
#pragma unroll
    for (int u = 0; u < PF; ++u) {
        gb[u] = 0; clb[u] = 0; limb[u] = 0;
#pragma unroll
        for (int i = 0; i < DPL; ++i) cb[u].r[i] = 0;
        if (1 + u <= n_live) {
            advance();
            fetch(u);
        }
    }
    unsigned sent[2] = {0x00FF00FFu, 0x00FF00FFu};                         // agg_step_nn's carried sentinel registers
    if constexpr (NN) min_prev |= min_prev << 16;                          // ... and its packed minimum
    auto step = [&](int u, int k, bool refill) {
        const int g = (int)gb[u];
        const unsigned dg = __builtin_amdgcn_sad_u8((unsigned)g, (unsigned)g_prev, 0u);
        CellVec<DPL> packed;
        if constexpr (NN) {
            min_prev = agg_step_nn<DPL, PAD, LPP, false>(clb[u], cb[u], limb[u], __any(limb[u] < DPL - 1) != 0, Lp, min_prev, lut32_s[dg],
                                                         p1v, padmask, first_lane, last_lane, sent, packed);
            if (refill) {                                                  // the slot's census words are consumed now
                advance();
                fetch(u);
            }
        } else {
            us2 C[NP];
            if constexpr (VOL) volume_costs<DPL>(cb[u], C);
            else census_costs<DPL>(clb[u], cb[u], limb[u], true, C);       // consume the slot, then refill it
            if (refill) {
                advance();
                fetch(u);
            }
            min_prev = agg_step<DPL, PAD, LPP>(C, Lp, min_prev, lut_s[dg], p1v, padmask, first_lane, last_lane, packed);
        }
        g_prev = g;
        if (store_ok) store_cells<DPL>(extras + (size_t)k * Dp, packed);
        ghost_zero(k);
    };
    __builtin_amdgcn_s_waitcnt(0x0F70);                                     // vmcnt(0): the prologue's loads, once (see agg_regular)
    int k0 = 1;
    for (; k0 + 2 * PF - 1 <= n_live; k0 += PF) {
#pragma unroll
        for (int u = 0; u < PF; ++u) step(u, k0 + u, true);
    }
    for (; k0 <= n_live; k0 += PF) {
#pragma unroll
        for (int u = 0; u < PF; ++u)
            if (k0 + u <= n_live) step(u, k0 + u, k0 + u + PF <= n_live);
    }
    for (int k = n_live + 1; k <= nsteps; ++k) ghost_zero(k);               // the cells the line should have visited (W >= H)
}

// Lane layout of an anomalous line for a cell of DP disparities: with the non-negative-P1 step the whole wave works on the one
// line where that leaves an even number >= 2 of disparities per lane (the shortest step, as on the horizontal lines of a single
// frame); the generic step keeps 16 lanes per pixel.
template <int DP> struct anom_layout   { static constexpr int lpp = (DP % 128 == 0) ? 64 : (DP == 64 ? 32 : 16), dpl = DP / lpp; };
template <int DP> struct anom_layout16 { static constexpr int lpp = 16, dpl = DP / 16; };

template <bool VOL>
static __device__ __forceinline__ AggFrame agg_frame(const AggArgs& a, int frame)
{
    AggFrame fr;
    fr.img = a.img + (size_t)frame * a.W * a.H;
    fr.census_l = a.census_l + (size_t)frame * a.W * a.H;
    fr.census_r = a.census_r + (size_t)frame * a.W * a.H;
    fr.cost = VOL ? a.cost + (size_t)frame * a.W * a.H * a.Dp : nullptr;
    fr.planes = a.planes + (size_t)frame * 8 * a.plane_bytes;
    fr.extras = a.extras + (size_t)frame * 4 * a.H * a.Dp;
    return fr;
}

// The four anomalous lines of every frame of the launch: one wave each (block = frame + B * (direction - 4)).  Its own kernel so
// that its deep prefetch ring does not set the register count -- the occupancy -- of the regular lines' kernel, and its own launch
// so that the host can run it BESIDE that kernel on a second stream (sgm_host.c launch_aggregation): inside the big launch these
// four waves, one dependent memory round trip per step, were the last to finish (single frame: 0.34 ms against 0.21 for the
// horizontal chains).  Lane layout of its own, whatever the regular lines of the batch use (extras rows and planes are plain [Dp]
// cells): with non-negative P1 the whole wave works on the one line -- up to 64 lanes per pixel, the shortest step, as on the
// horizontal lines of a single frame --; negative P1 and the volume-fed variant keep the generic step on 16 lanes.
template <int DPL, bool PAD, int LPP, int NN, bool VOL>
__global__ __launch_bounds__(64) void sgm_aggregate_anom_k(const AggArgs a)
{
    __shared__ unsigned short lut_s[256];
    __shared__ unsigned lut32_s[256];
    const int lane = threadIdx.x;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const unsigned pen = a.lut[lane * 4 + i];
        lut_s[lane * 4 + i] = (unsigned short)pen;
        lut32_s[lane * 4 + i] = pen * 0x00010001u;
    }
    __syncthreads();
    const AggFrame fr = agg_frame<VOL>(a, blockIdx.x % a.B);
    agg_anomalous<DPL, PAD, LPP, NN, VOL>(a, fr, lut_s, lut32_s, 4 + blockIdx.x / a.B);
}
// NN: 0 = the generic step (any penalties; serves negative P1, for which the host keeps to 16 lanes per pixel everywhere,
// sgm_host.c, so only those combinations are instantiated), 1 = the step for non-negative P1 (agg_step_nn), 2 = the same with the
// FAST shortcuts for ordinary penalties (sgm_aggregate_fast.hip)
// Diagnostics, compiled in only with -DSGM_CLOCK_PROBE (make CLOCK_PROBE=1; tools/agg_clock.py builds such a variant): the shader
// clock the launch runs at -- s_memtime (shader-clock ticks) against s_memrealtime (100 MHz) over the lifetime of one long-running
// wave (block 0 = horizontal lines of frame 0: W-1 steps); read back with sgmd_debug_clock(), which covers the kernels of
// sgm_aggregate.hip (the symbol is per translation unit).  The default build has no diagnostic loads or stores in the kernel.
#ifdef SGM_CLOCK_PROBE
static __device__ unsigned long long g_sgm_agg_clock[2];
#endif

template <int DPL, bool PAD, int LPP, int HL, int NN, bool VOL = false>
__global__ __launch_bounds__(64) void sgm_aggregate_k(const AggArgs a)
{
#ifdef SGM_CLOCK_PROBE
    const unsigned long long clk0 = __builtin_amdgcn_s_memtime(), rt0 = __builtin_amdgcn_s_memrealtime();
    struct ClockProbe {
        unsigned long long c0, r0;
        __device__ ~ClockProbe() {
            if (blockIdx.x == 0 && threadIdx.x == 0) {
                g_sgm_agg_clock[0] = __builtin_amdgcn_s_memtime() - c0;
                g_sgm_agg_clock[1] = __builtin_amdgcn_s_memrealtime() - r0;
            }
        }
    } probe{clk0, rt0};
#endif
    __shared__ unsigned short lut_s[256];
    __shared__ unsigned lut32_s[256];                        // the same penalties in both halves of a dword (packed u16 operand)
    const int lane = threadIdx.x;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const unsigned pen = a.lut[lane * 4 + i];
        lut_s[lane * 4 + i] = (unsigned short)pen;
        lut32_s[lane * 4 + i] = pen * 0x00010001u;
    }
    __syncthreads();

    // Batch: consecutive blocks are the same line group of consecutive frames, so every frame's long
    // horizontal lines are dispatched first.  Per frame, blocks [block_begin[d], block_begin[d+1]) are the
    // regular lines of direction d (the four anomalous diagonal lines run in a kernel of their own, sgm_aggregate_anom_k).
    //
    // Workgroups are dealt round-robin over the 8 XCDs, each with an L2 of its own.  With 8 frames per launch frame f's blocks all
    // land on XCD f and its census images stay in that L2.  With fewer frames (the large shapes run 2 per launch, a single frame 1)
    // the plain numbering scatters neighbouring line groups -- whose census-right windows overlap almost completely -- over
    // 8 / B XCDs, and every one of those L2s fetches the same rows (2880x1988 D=256: 5.3 GB of census reads per frame against
    // 46 MB of census data).  a.strips = 8 / B > 1: the frame keeps its 8 / B XCDs, and each of them takes one CONTIGUOUS strip of
    // every direction's line groups, in the order horizontal lines first.
    // The four anomalous lines of a frame (a.anom_inline) come FIRST in both numberings: one wave each, H - 1 steps -- started
    // behind everything else they would be the tail of the launch.  b >= block_begin[8] marks them below.
    int frame, b;
    const int n_anom = a.anom_inline ? 4 : 0;
    if (a.strips > 1) {
        const int xcd = blockIdx.x & 7;
        frame = xcd % a.B;
        const int sub = xcd / a.B;
        int q = blockIdx.x >> 3;
        b = -1;
        for (int dd = 0; dd <= 8; ++dd) {
            const int d = dd == 0 ? 8 : dd - 1;                            // 8 = the anomalous lines
            const int begin = a.block_begin[d];
            const int n = (d < 8 ? a.block_begin[d + 1] : begin + n_anom) - begin;
            const int lo = sub * n / a.strips, share = (sub + 1) * n / a.strips - lo;
            if (q < share) { b = begin + lo + q; break; }
            q -= share;
        }
        if (b < 0) return;                                                 // the grid is padded to 8 x the longest strip
    } else {
        frame = blockIdx.x % a.B;
        b = blockIdx.x / a.B;
        b = b < n_anom ? a.block_begin[8] + b : b - n_anom;
    }
    const AggFrame fr = agg_frame<VOL>(a, frame);
    if constexpr (NN != 0 && !VOL) {
        // lane layout of its own (anom_layout: the whole wave on the one line where the cell is wide enough): fewer registers
        // than the regular lines need, so the branch does not cost them occupancy
        if (b >= a.block_begin[8]) {
            using AL = anom_layout<DPL * LPP>;
            agg_anomalous<AL::dpl, PAD, AL::lpp, 1, false>(a, fr, lut_s, lut32_s, 4 + (b - a.block_begin[8]));
            return;
        }
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
        if constexpr (HL != 0) agg_regular<DPL * LPP / HL, PAD, HL, AGG_H, NN, false, VOL>(a, fr, lut_s, lut32_s, dir, grp);
        else               agg_regular<DPL, PAD, LPP, AGG_H, NN, false, VOL>(a, fr, lut_s, lut32_s, dir, grp);
    }
    else if (a.dx[dir] == 0) agg_regular<DPL, PAD, LPP, AGG_V, NN, false, VOL>(a, fr, lut_s, lut32_s, dir, grp);
    else if (a.W > a.H) {
        if constexpr (NN != 0 && !VOL) {                                      // the fused last sweep exists for the non-negative-P1 census path only
            if ((a.post_wrap_mask >> dir) & 1) {
                agg_regular<DPL, PAD, LPP, AGG_D, NN, true, VOL, true>(a, fr, lut_s, lut32_s, dir, grp);
                return;
            }
        }
        agg_regular<DPL, PAD, LPP, AGG_D, NN, true, VOL>(a, fr, lut_s, lut32_s, dir, grp);
    }
    else                     agg_regular<DPL, PAD, LPP, AGG_D, NN, false, VOL>(a, fr, lut_s, lut32_s, dir, grp);
}

template <int DPL, int LPP, int HL, int NN>
static bool launch_aggregate_hl(const AggArgs& a, int blocks, bool pad, hipStream_t st)
{
    constexpr int per = (HL != 0) ? DPL * LPP / HL : DPL;                 // disparities per lane on the horizontal lines
    constexpr bool ok = (HL == 0) || ((DPL * LPP) % HL == 0 && (per == 2 || per == 4 || per == 8 || per == 16) && HL != LPP);
    if constexpr (ok) {
        if (pad) hipLaunchKernelGGL((sgm_aggregate_k<DPL, true, LPP, HL, NN>), dim3(blocks), dim3(64), 0, st, a);
        else     hipLaunchKernelGGL((sgm_aggregate_k<DPL, false, LPP, HL, NN>), dim3(blocks), dim3(64), 0, st, a);
        return true;
    }
    return false;
}
template <int DPL, int LPP, int NN>
static void launch_aggregate(const AggArgs& a, int blocks, bool pad, int hl, hipStream_t st)
{
    if constexpr (NN) {
        if (hl == 64 && launch_aggregate_hl<DPL, LPP, 64, NN>(a, blocks, pad, st)) return;
        if (hl == 32 && launch_aggregate_hl<DPL, LPP, 32, NN>(a, blocks, pad, st)) return;
        if (hl == 16 && launch_aggregate_hl<DPL, LPP, 16, NN>(a, blocks, pad, st)) return;
    }
    launch_aggregate_hl<DPL, LPP, 0, NN>(a, blocks, pad, st);
}

// (DPL, LPP): disparities per lane x lanes per pixel = Dp.  16 lanes per pixel (4 lines per wave) gives the shortest
// serial step; 8 lanes per pixel (8 lines per wave) spends ~40 % fewer VALU instructions per cell and is what a
// batch of frames (VALU-bound) uses.  Returns false for a combination that is not instantiated.
template <int NN>
static bool launch_aggregate_key(int lpp, int dpl, const AggArgs& a, int blocks, bool pad, int hl, hipStream_t st)
{
    switch (lpp * 100 + dpl) {
    case 1602: launch_aggregate<2, 16, NN>(a, blocks, pad, hl, st); return true;
    case 1604: launch_aggregate<4, 16, NN>(a, blocks, pad, hl, st); return true;
    case 1608: launch_aggregate<8, 16, NN>(a, blocks, pad, hl, st); return true;
    case 1612: launch_aggregate<12, 16, NN>(a, blocks, pad, hl, st); return true;
    case 1616: launch_aggregate<16, 16, NN>(a, blocks, pad, hl, st); return true;
    case 1632: launch_aggregate<32, 16, NN>(a, blocks, pad, hl, st); return true;
    default: break;
    }
    if constexpr (NN) {
        switch (lpp * 100 + dpl) {
        case 804: launch_aggregate<4, 8, NN>(a, blocks, pad, hl, st); return true;
        case 808: launch_aggregate<8, 8, NN>(a, blocks, pad, hl, st); return true;
        case 816: launch_aggregate<16, 8, NN>(a, blocks, pad, hl, st); return true;
        default: break;
        }
    }
    return false;
}


#include "sgm_common.hpp"


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
// a * b + c for a, b < 2^24: v_mad_u32_u24 (half rate); hipcc turns the plain 32-bit form into v_mad_u64_u32 / v_mul_lo_u32
static __device__ __forceinline__ unsigned umad24(unsigned a, unsigned b, unsigned c) { return __umul24(a, b) + c; }

template <int DPL, int STAGE>
static __device__ __forceinline__ void sumlr_prefetch(CellVec<DPL> (&pre)[2][8], const uint8_t* const (&pb)[8], unsigned off)
{
    // always 8 unconditional loads (see SLOW below): with four paths the upper four re-read planes 0..3 and are masked out
    // when they are added.  pb[] = the eight plane bases, computed once per workgroup (wave-uniform: SGPR pairs), so a load
    // is base + zero-extended 32-bit lane offset with no 64-bit vector arithmetic
#pragma unroll
    for (int d = 0; d < 8; ++d) load_cells_nt<DPL>(pb[d] + off, pre[STAGE][d]);
}

// SLOW = the variant that may read (accumulate) or write (store_S) S.  The common one has no conditional global
// access inside the loop at all: hipcc merges s_waitcnt counts over all control-flow paths, and a load behind a
// condition or in a loop of unknown trip count makes it drain the two-iterations-deep plane prefetch (vmcnt(0))
// every iteration -- which is why columns past the row end re-read the last column instead of being skipped, why
// the right-view-only iterations after the last column are a loop of their own, and why the anomalous-line
// visits of the row are staged in LDS up front.
// TIGHT: the ring holds Dp + COLS columns instead of Dp + 2 COLS and the iteration ends with a second barrier (no wave may
// write the next iteration's columns while another still reads this one's oldest): for Dp = 256 that is what lets the ring
// (288 x 258 u16 = 145 KB either way) serve 32 columns = 8 waves per iteration instead of 16 columns = 4 waves -- the kernel
// is alone on its CU there, so its waves are all the latency hiding it has.
#define SUMLR_MAX_EXTRA 8
template <int DPL, bool SLOW, int THREADS, bool TIGHT = false>
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
#ifdef SGM_SUM_PRIO
    __builtin_amdgcn_s_setprio(SGM_SUM_PRIO);
#endif
    constexpr int R = Dp + (TIGHT ? 1 : 2) * COLS;
    static_assert(R % COLS == 0, "a ring slot must always belong to the same px");
    // slots R .. R + MIR - 1 mirror slots 0 .. MIR - 1 (written together): the DPL consecutive slots a lane gathers its piece of
    // the right view's diagonal from then never wrap, so the gather is one base address per lane and immediate offsets
    constexpr int MIR = DPL;
    __shared__ unsigned short ring[(R + MIR) * LD];
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
    const unsigned row_cells = (unsigned)row * (unsigned)W * Dp;      // 32-bit cell offsets: the host guarantees W*H*Dp < 2^32
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
    // planes 4..7 count only with eight paths (with four they re-read planes 0..3, see sumlr_prefetch): byte-extraction
    // mask / selector that give zeros then
    const unsigned upper_lo = (ndirs > 4) ? 0x00FF00FFu : 0u;
    const unsigned upper_sel = (ndirs > 4) ? 0x0c030c01u : 0x0c0c0c0cu;
    unsigned padpair[DPL / 2];                                          // 65535 in the halves of padding disparities
#pragma unroll
    for (int m = 0; m < DPL / 2; ++m)
        padpair[m] = ((sub * DPL + 2 * m >= D) ? 0xFFFFu : 0u) | ((sub * DPL + 2 * m + 1 >= D) ? 0xFFFF0000u : 0u);

    auto cell_off = [&](int x) { return row_cells + (unsigned)min(x, W - 1) * Dp + (unsigned)(sub * DPL); };
    const uint8_t* pb[8];
#pragma unroll
    for (int d = 0; d < 8; ++d) pb[d] = planes + (size_t)(d < ndirs ? d : d - 4) * plane_bytes;
    CellVec<DPL> pre[2][8];
    sumlr_prefetch<DPL, 0>(pre, pb, cell_off(xa + px));
    sumlr_prefetch<DPL, 1>(pre, pb, cell_off(xa + COLS + px));

    int slot = px;                                                       // ring slot of this thread's column: x mod R

    // Both views end in ONE wta_finish per iteration (uniqueness test, float divide of the sub-pixel term: ~35
    // instructions a wave pays in full even for a single active lane): lane 0 of a pixel finishes the left view,
    // lane 1 the right view.  kl/ks = best and runner-up key of the left view of column x (ignored unless `left`).
    auto finish_views = [&](int x, bool left, unsigned kl, unsigned ks) {
        unsigned kbest_r = 0, ksecond_r = 0;
        int base = 0;
        const int xr = x - dmin - (D - 1);
        if (do_right) {
        __syncthreads();                                                 // the new columns are in the ring
        base = slot + R - (D - 1);                                       // ring slot of column xr + dmin = x - (D-1)
        if (base >= R) base -= R;
        unsigned key[DPL], val[DPL];
        unsigned kmin = 0xFFFFFFFFu;
        int first = base + sub * DPL;                                    // slot of this lane's first disparity; the others follow (mirror)
        if (first >= R) first -= R;
        const unsigned short* const diag = &ring[umad24((unsigned)first, (unsigned)LD, (unsigned)(sub * DPL))];
#pragma unroll
        for (int i = 0; i < DPL; ++i) val[i] = diag[i * (LD + 1)];      // padding disparities and columns past the image hold 65535
        if (D == Dp) {                                                   // wave-uniform: no padding disparities, every slot read was written
#pragma unroll
            for (int i = 0; i < DPL; ++i) {
                key[i] = (val[i] << 16) | (unsigned)(sub * DPL + i);
                kmin = min(kmin, key[i]);
            }
        } else {                                                         // padding disparities reach into slots ahead of the newest column
#pragma unroll
            for (int i = 0; i < DPL; ++i) {
                const int k = sub * DPL + i;
                key[i] = (k < D) ? ((val[i] << 16) | (unsigned)k) : 0xFFFFFFFFu;
                kmin = min(kmin, key[i]);
            }
        }
        const unsigned kbest = row_allmin<16>(kmin);
        // runner-up: keys are distinct (they carry d), so key - kbest - 1 (mod 2^32) sends the best to the top
        // and keeps the order of all others
        const unsigned nbest = ~kbest;
        unsigned k2 = 0xFFFFFFFFu;
#pragma unroll
        for (int i = 0; i < DPL; ++i) k2 = min(k2, key[i] + nbest);
        kbest_r = kbest;
        ksecond_r = row_allmin<16>(k2) + kbest + 1;
        }
        const bool is_r = (sub == 1);
        const bool active = is_r ? (do_right && xr >= xa && xr < xb) : (sub == 0 && left);
        if (active) {
            const unsigned kb = is_r ? kbest_r : kl, k2nd = is_r ? ksecond_r : ks;
            const int dbest = (int)(kb & 0xFFFFu);
            // S[best-1], S[best+1] straight from the ring: the column's own slot for the left view, the diagonal for
            // the right one (a best at either end of the range is invalid anyway, ref :428: clamp, value unused)
            const int km = max(dbest - 1, 0), kp = min(dbest + 1, Dp - 1);
            int sm = base + km, sp = base + kp;
            if (sm >= R) sm -= R;
            if (sp >= R) sp -= R;
            if (!is_r) sm = sp = slot;
            WtaState st;
            st.m1 = kb >> 16;
            st.m2 = k2nd >> 16;
            st.d1 = (is_r && (kb >> 16) == 0xFFFFu) ? -1 : dbest;        // right view: nothing beat 65535 (ref :381, strict '>')
            st.c1 = ring[umad24((unsigned)sm, (unsigned)LD, (unsigned)km)];
            st.c2 = ring[umad24((unsigned)sp, (unsigned)LD, (unsigned)kp)];
            st.pv = 0; st.want_next = false;
            float* const out = is_r ? disp_r + xr : disp_l + x;
            out[(size_t)row * W] = wta_finish(st, D, dmin, check_unique, one_minus_ratio);
        }
        if (TIGHT && do_right) __syncthreads();                          // every diagonal of this iteration has been read
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
        const unsigned off = cell_off(x);
        // S of this lane's DPL disparities as packed u16 pairs: aL[k] = (S(4k), S(4k+2)), aH[k] = (S(4k+1), S(4k+3)) --
        // a plane dword gives both with one AND and one v_perm, and eight planes add up in 16-bit halves without
        // carries (8 x 255 + a few anomalous visits < 2^16; the accumulating variant adds with v_pk_add_u16, where
        // the reference's uint16 sums may wrap)
        constexpr int NW = (DPL + 3) / 4;
        constexpr int NPAIR = DPL / 2;
        unsigned aL[NW], aH[NW];
#pragma unroll
        for (int k = 0; k < NW; ++k) aL[k] = aH[k] = 0;
        auto add_packed = [&](unsigned& dst, unsigned v) {
            if (SLOW) dst = as_u(as_p(dst) + as_p(v));                   // per-half wrap (Q14 sums are uint16)
            else dst += v;
        };
        auto add_bytes = [&](int k, unsigned w, unsigned mask_lo, unsigned sel_hi) {
            add_packed(aL[k], w & mask_lo);                              // bytes 0, 2
            add_packed(aH[k], __builtin_amdgcn_perm(0u, w, sel_hi));     // bytes 1, 3
        };
        if (SLOW) {
            if (accumulate) {                                            // Q14: S was not reset since the last frame
                const unsigned* sp = reinterpret_cast<const unsigned*>(S + off);
                if constexpr (DPL == 2) {
                    const unsigned v = sp[0];
                    aL[0] = v & 0xFFFFu; aH[0] = v >> 16;
                } else {
#pragma unroll
                    for (int k = 0; k < NW; ++k) {
                        const unsigned e = sp[2 * k], o = sp[2 * k + 1]; // (S(4k), S(4k+1)), (S(4k+2), S(4k+3))
                        aL[k] = __builtin_amdgcn_perm(o, e, 0x05040100u);
                        aH[k] = __builtin_amdgcn_perm(o, e, 0x07060302u);
                    }
                }
            }
        }
#pragma unroll
        for (int d = 0; d < 8; ++d) {
            const unsigned mask_lo = (d < 4) ? 0x00FF00FFu : upper_lo;
            const unsigned sel_hi = (d < 4) ? 0x0c030c01u : upper_sel;
#pragma unroll
            for (int k = 0; k < NW; ++k) add_bytes(k, pre[STAGE][d].w[k], mask_lo, sel_hi);
        }
        sumlr_prefetch<DPL, STAGE>(pre, pb, cell_off(x + 2 * COLS));     // columns of iteration it + 2
        for (int j = 0; j < n_extra; ++j) {
            if (ex_col[j] == x) {                                        // second visit of an anomalous line (LDS)
                if constexpr (DPL == 2) {
                    const unsigned w = (ex_val[j * (Dp / 4) + (sub >> 1)] >> (16 * (sub & 1))) & 0xFFFFu;
                    add_bytes(0, w, 0x00FF00FFu, 0x0c030c01u);
                } else {
#pragma unroll
                    for (int k = 0; k < NW; ++k) add_bytes(k, ex_val[j * (Dp / 4) + sub * (DPL / 4) + k], 0x00FF00FFu, 0x0c030c01u);
                }
            }
        }
        // back to disparity order: pr[m] = (S(2m), S(2m+1))
        unsigned pr[NPAIR];
#pragma unroll
        for (int m = 0; m < NPAIR; ++m)
            pr[m] = __builtin_amdgcn_perm(aH[m >> 1], aL[m >> 1], (m & 1) ? 0x07060302u : 0x05040100u);
        if (SLOW) {
            if (store_S && mine) {
                unsigned* dst = reinterpret_cast<unsigned*>(S + off);
#pragma unroll
                for (int m = 0; m < NPAIR; ++m) dst[m] = pr[m];
            }
        }
        // ---- this column's S vector into the ring (65535 outside the image / the disparity range) ----
        // (also without a right view: the left view fetches S[best +- 1] from here.  A column's slot is only ever
        // written by the wave that owns px = slot mod 16 -- R is a multiple of 16 -- so that read-back needs no barrier)
        {
            unsigned* dst = reinterpret_cast<unsigned*>(&ring[umad24((unsigned)slot, (unsigned)LD, (unsigned)(sub * DPL))]);
#pragma unroll
            for (int m = 0; m < NPAIR; ++m) {
                pr[m] = inside ? (pr[m] | padpair[m]) : 0xFFFFFFFFu;
                dst[m] = pr[m];
            }
            if (slot < MIR) {                                            // ... and its mirror behind the ring
#pragma unroll
                for (int m = 0; m < NPAIR; ++m) dst[(R * LD) / 2 + m] = pr[m];
            }
        }
        // ---- left-view WTA over the 16 lanes of the pixel (as in sgm_sum_wta_k); padding disparities carry 65535,
        //      so their keys lose against every real one ----
        unsigned key[DPL];
        unsigned kmin = 0xFFFFFFFFu;
#pragma unroll
        for (int m = 0; m < NPAIR; ++m) {
            const unsigned idx = (unsigned)(sub * DPL + 2 * m);
            key[2 * m] = (pr[m] << 16) | idx;
            key[2 * m + 1] = (pr[m] & 0xFFFF0000u) | (idx + 1);
            kmin = min(kmin, min(key[2 * m], key[2 * m + 1]));
        }
        const unsigned kbest = row_allmin<16>(kmin);
        const unsigned nbest = ~kbest;                                   // runner-up as in right_view()
        unsigned k2 = 0xFFFFFFFFu;
#pragma unroll
        for (int i = 0; i < DPL; ++i) k2 = min(k2, key[i] + nbest);
        const unsigned ksecond = row_allmin<16>(k2) + kbest + 1;
        asm volatile("" ::: "memory");                                   // the wave's ring writes above stay above
        finish_views(x, mine, kbest, ksecond);
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
        unsigned* dst = reinterpret_cast<unsigned*>(&ring[umad24((unsigned)slot, (unsigned)LD, (unsigned)(sub * DPL))]);
#pragma unroll
        for (int i = 0; i < DPL; i += 2) dst[i >> 1] = 0xFFFFFFFFu;
        if (slot < MIR) {
#pragma unroll
            for (int i = 0; i < DPL; i += 2) dst[(R * LD) / 2 + (i >> 1)] = 0xFFFFFFFFu;
        }
        finish_views(xa + it * COLS + px, false, 0u, 0u);
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

// Extension: the mirror image of sgm_lrcheck_k with the right view as the reference view (out-of-place: the left map
// is read at other columns of the row)
__global__ __launch_bounds__(256) void sgm_lrcheck_right_k(const float* __restrict__ dr, const float* __restrict__ dl,
                                                           float* __restrict__ out, int W, int H, float thres, int do_check, int row0)
{
    const int x = blockIdx.x * 256 + threadIdx.x;
    const int y = row0 + blockIdx.y;
    if (x >= W) return;
    const float inf = __builtin_inff();
    const size_t fo = (size_t)blockIdx.z * W * H;
    const size_t idx = fo + (size_t)y * W + x;
    float d = dr[idx];
    if (do_check && d != inf) {
        const int xl = (int)((double)((float)x + d) + 0.5);
        if (xl >= 0 && xl < W) {
            const float l = dl[fo + (size_t)y * W + xl];
            if (l != inf && fabs((double)(d - l)) > (double)thres) d = inf;
        } else {
            d = inf;
        }
    }
    out[idx] = d;
}

template <int DPL, int THREADS, bool TIGHT = false>
static void launch_sum_wta_lr(dim3 grid, hipStream_t st, const void* planes, size_t plane_bytes, int ndirs, const void* extras,
                              const void* row_extras, const void* row_extra_count, int row_cap, int accumulate, int store_S,
                              int do_right, void* S, void* disp_l, void* disp_r, const sgmd_geom* g, int check_unique,
                              float one_minus_ratio, int seg_len)
{
#define SUMLR_CALL(SLOW)                                                                                              \
    hipLaunchKernelGGL((sgm_sum_wta_lr_k<DPL, SLOW, THREADS, TIGHT>), grid, dim3(THREADS), 0, st, (const uint8_t*)planes, plane_bytes, ndirs, \
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
    static const int tight = getenv("SGM_SUM_TIGHT") ? atoi(getenv("SGM_SUM_TIGHT")) : 1;   // Dp 192 / 256: see TIGHT above
    int segs = 1;
    if (!accumulate && !store_S) {
        const char* e = getenv("SGM_SUM_SEGMENTS");
        segs = (e && *e) ? atoi(e) : (1024 + rows - 1) / rows;
        if (segs > 4) segs = 4;
        while (segs > 1 && g->W / segs < 2 * g->Dp) --segs;
        if (segs < 1) segs = 1;
        if (!(e && *e)) {
            // ... and where some segment count lets ALL workgroups of the launch be resident at once, the largest such count: a
            // second, partly filled round of workgroups costs more than the longer segments (KITTI, one frame: 375 rows x 3
            // segments = 1125 workgroups on 768 slots took 0.123 ms, x 2 = 750 workgroups 0.103 ms).  Slots: 256 CUs x what the
            // ring (LDS) and the wave count of a workgroup allow.
            const int threads = g->Dp == 192 ? (tight ? 1024 : 512) : g->Dp == 256 ? (tight ? 512 : 256) : 256;
            const int ring_cols = g->Dp + (((g->Dp == 192 || g->Dp == 256) && tight) ? 1 : 2) * (threads / 16) + g->Dp / 16;
            const size_t lds = (size_t)ring_cols * (size_t)(g->Dp + 2) * 2 + SUMLR_MAX_EXTRA * (size_t)g->Dp + 64;
            const size_t by_lds = ((size_t)160 << 10) / lds, by_waves = 32 / (size_t)(threads / 64);
            const long slots = 256L * (long)(by_lds < by_waves ? by_lds : by_waves);
            for (int sg = 4; sg >= 1; --sg)
                if ((long)rows * sg <= slots && (sg == 1 || g->W / sg >= 2 * g->Dp)) { segs = sg; break; }
        }
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
    case 12:                                                 // Dp 192: ring 256 x 194 u16 = 97 KB either way
        if (tight) launch_sum_wta_lr<12, 1024, true>(SUMLR_ARGS);    //   64 columns = 16 waves per iteration
        else       launch_sum_wta_lr<12, 512>(SUMLR_ARGS);           //   32 columns =  8 waves
        break;
    case 16:                                                 // Dp 256: ring 288 x 258 u16 = 145 KB either way
        if (tight) launch_sum_wta_lr<16, 512, true>(SUMLR_ARGS);     //   32 columns = 8 waves per iteration
        else       launch_sum_wta_lr<16, 256>(SUMLR_ARGS);           //   16 columns = 4 waves
        break;
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

int sgmd_lrcheck_right(int ord, void* stream, const sgmd_geom* g, const void* disp_r, const void* disp_l, float thres,
                       int do_check, void* out)
{
    HIP_TRY(hipSetDevice(ord));
    dim3 grid((g->W + 255) / 256, g->row_end - g->row_begin, g->B);
    hipLaunchKernelGGL(sgm_lrcheck_right_k, grid, dim3(256), 0, (hipStream_t)stream, (const float*)disp_r, (const float*)disp_l,
                       (float*)out, g->W, g->H, thres, do_check, g->row_begin);
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

}  // extern "C"

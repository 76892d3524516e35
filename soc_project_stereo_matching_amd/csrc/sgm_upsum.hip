#include "sgm_aggregate_impl.hpp"

// ============================================================================================
// The LAST vertical sweep fused with the cost sum and both winner-take-all passes.
//
// The reference runs its eight directions one after the other and read-modify-writes S (ref :213-220, :345).  The
// per-direction planes of sgm_aggregate_k removed the RMW but still put all eight directions through HBM once (written) and
// once more (read by the sum kernel): 16 B per cell.  The three directions that walk UP the image (ref :216 (0,-1), :218
// (-1,-1), :219 (1,-1)) are the last anybody needs before S is complete, so they are computed HERE, row by row from the
// bottom, in the same lanes that sum the other five planes, and never reach HBM:
//
//   * a workgroup = R consecutive image rows (R "teams" of 4 waves; a team walks its row left to right, 16 columns per
//     iteration, exactly like sgm_sum_wta_lr_k: S of the last Dp+16 columns in an LDS ring for the right view);
//   * the row above needs L_r(y+1, x-1 .. x+1) of the three directions: inside a workgroup through an LDS exchange ring (team
//     t runs one iteration and one column behind team t-1, so one barrier per step orders producer and consumer); between
//     workgroups through a small global hand-over row (sc1 stores, drained by a counted s_waitcnt, one progress word per
//     workgroup; the consumer polls it with sc1 loads and prefetches one iteration ahead) -- 3 B per cell of every R-th row;
//   * a regular diagonal line wraps around the image edge carrying its state.  Its POST-wrap part (the
//     upper-left triangle x < H-1-y for (1,-1), the upper-right one x >= W-(H-1-y) for (-1,-1)) cannot be had from the row below in
//     time -- its predecessor is the far end of that row -- so the aggregation launch still walks the H-1 wrapping lines of
//     those two directions and stores their post-wrap cells (15 % of a plane each); this kernel reads them there (elsewhere the
//     load is pointed at one hot line) and computes everything else itself.  The cells no regular line visits (the track of
//     the anomalous line, Q5) contribute 0; the anomalous lines' own visits come from `extras` as in the sum kernel.
//
// HBM per cell: 5.3 B written + 5.3 B read (+ the hand-over rows) instead of 8 + 8.  Only for what the timed configurations
// are: W > H, eight paths, non-negative P1, Dp = 128 or 64, batches, no S read-back (everything else keeps the separate
// kernels; sgm_host.c re-creates the three planes on demand for a Match without Reset, Q14).
// ============================================================================================

struct UpArgs {
    const uint8_t* img;
    const uint32_t* census_l;
    const uint32_t* census_r;
    const uint8_t* planes;
    size_t plane_bytes;
    const uint8_t* extras;
    const sgmd_row_extra* row_extras;
    const int* row_extra_count;
    int row_cap;
    const uint16_t* lut;
    float* disp_l;
    float* disp_r;
    uint8_t* xbuf;              // hand-over rows between workgroups: [B][2][3][W][Dp] bytes
    unsigned* progress;         // [B][ngroups]: (generation << 13) + iterations the workgroup's top team has published
    unsigned* ticket;           // [B], zero at launch: workgroups of a frame take their row group in arrival order
    int* status;                // pinned host word: set to 1 when a poll gave up (the maps are then wrong)
    unsigned long long* trace;  // diagnostics (SGM_UPSUM_TRACE): [B][ngroups][12]: 100 MHz timestamps (ticket drawn, first hand-over seen, last step done), xcc/cu id,
                                // then shader-clock sums over the group's steps of wave 0: work before barrier A, wait at A, work before B, wait at B (bottom team), the same four for the top team
    unsigned gen;
    int W, H, D, dmin, B, p1, ngroups;
    int check_unique;
    float one_minus_ratio;
    int do_right;
};

#define UPSUM_MAX_EXTRA 8
#define UPSUM_XC 36            // columns of an exchange ring (34 are live at any time)
#define UPSUM_POLL_LIMIT (1u << 22)

static __device__ __forceinline__ unsigned up_umad24(unsigned a, unsigned b, unsigned c) { return __umul24(a, b) + c; }

// One step of one direction with the matching cost already packed (shared by the three directions of a pixel): the
// non-negative-P1 step of agg_step_nn -- L(d) = C(d) + min(min(Lp(d), Lp(d-1)+P1, Lp(d+1)+P1) - min_prev, P2') (ref :329-343) --
// Cp = (C(2j), C(2j+1)) pairs, 127 where x - d is left of the image (census_costs).  Returns the new row minimum in both halves.
template <int DPL, int LPP, bool PAD, bool FAST>
static __device__ __forceinline__ unsigned up_step(const us2 (&Cp)[DPL / 2], bool border, const us2 (&Lp)[DPL / 2], unsigned mp, unsigned pen32,
                                                   unsigned p1u, const us2 (&padmask)[DPL / 2], unsigned (&sent)[2], bool first_lane, bool last_lane,
                                                   us2 (&Ln)[DPL / 2])
{
    constexpr int NP = DPL / 2;
    const us2 p2v = as_p(pen32);
    unsigned from_left = sent[0] = dpp_mov<DPP_ROW_SHR1>(sent[0], as_u(Lp[NP - 1]));
    unsigned from_right = sent[1] = dpp_mov<DPP_ROW_SHL1>(sent[1], as_u(Lp[0]));
    if (LPP != 16) {                                    // two pixels share a DPP row: cut the shift at the pixel boundary (255 sentinels, ref :260-263)
        from_left = first_lane ? 0x00FF00FFu : from_left;
        from_right = last_lane ? 0x00FF00FFu : from_right;
    }
#pragma unroll
    for (int j = 0; j < NP; ++j) {
        const unsigned below = (j == 0) ? from_left : as_u(Lp[j - 1]);
        const unsigned above = (j == NP - 1) ? from_right : as_u(Lp[j + 1]);
        const us2 dm1 = as_p(__builtin_amdgcn_alignbit(as_u(Lp[j]), below, 16) + p1u);
        const us2 dp1 = as_p(__builtin_amdgcn_alignbit(above, as_u(Lp[j]), 16) + p1u);
        us2 m;
        if constexpr (FAST) m = pk_min3_f16(dm1, dp1, Lp[j]);
        else m = pk_min(pk_min(dm1, dp1), Lp[j]);
        unsigned w = as_u(pk_min(as_p(as_u(m) - mp), p2v)) + as_u(Cp[j]);
        if constexpr (FAST) { if (border) w &= 0x00FF00FFu; }      // only C = 127 can pass 255 (agg_step_nn)
        else w &= 0x00FF00FFu;                                      // uint8 truncation (ref :343, Q7)
        if (PAD) w |= as_u(padmask[j]);
        Ln[j] = as_p(w);
    }
    return row_allmin_pk<LPP>(pk_min_tree<NP, FAST>(Ln));
}

template <int DPL>
static __device__ __forceinline__ void unpack_cells(const CellVec<DPL>& c, us2 (&L)[DPL / 2])
{
#pragma unroll
    for (int j = 0; j < DPL / 2; ++j) {
        const unsigned w = c.w[j >> 1];
        L[j] = as_p(__builtin_amdgcn_perm(w, w, (j & 1) ? 0x0c030c02u : 0x0c010c00u));   // bytes -> u16 pairs
    }
}

template <int DPL>
static __device__ __forceinline__ void load_cells_sc1(const uint8_t* p, CellVec<DPL>& v)
{
    static_assert(DPL == 16 || DPL == 8 || DPL == 4, "hand-over cells are 16, 8 or 4 bytes per lane");
    if constexpr (DPL == 16) {
        const unsigned long long t0 = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long t1 = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p) + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        v.w[0] = (unsigned)t0; v.w[1] = (unsigned)(t0 >> 32); v.w[2] = (unsigned)t1; v.w[3] = (unsigned)(t1 >> 32);
    } else if constexpr (DPL == 8) {
        const unsigned long long t = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        v.w[0] = (unsigned)t; v.w[1] = (unsigned)(t >> 32);
    } else {
        v.w[0] = __hip_atomic_load(reinterpret_cast<const unsigned*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
template <int DPL>
static __device__ __forceinline__ void store_cells_sc1(uint8_t* p, const CellVec<DPL>& v)
{
    if constexpr (DPL == 16) {
        __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), (unsigned long long)v.w[0] | ((unsigned long long)v.w[1] << 32), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(reinterpret_cast<unsigned long long*>(p) + 1, (unsigned long long)v.w[2] | ((unsigned long long)v.w[3] << 32), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
    } else if constexpr (DPL == 8)
        __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), (unsigned long long)v.w[0] | ((unsigned long long)v.w[1] << 32), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
    else
        __hip_atomic_store(reinterpret_cast<unsigned*>(p), v.w[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// planes summed whole by this kernel (reference order ref :213-220: 0 (1,0), 1 (-1,0), 2 (0,1), 4 (1,1), 7 (-1,1)); 3 = (0,-1),
// 5 = (-1,-1) and 6 = (1,-1) are the sweep computed here (5 and 6 are read in their post-wrap triangles only)
// LPP lanes per pixel x DPL disparities per lane = the padded range.  16 lanes is sgm_sum_wta_lr_k's layout; 8 lanes (16 disparities per
// lane, a team = 2 waves) halves the wave instructions per pixel of everything that is per lane group -- the three direction steps
// above all -- at the price of fewer waves per CU (the S rings of three rows fill its LDS either way).
template <int DPL, int LPP, int R, bool PAD, bool FAST>
__global__ __launch_bounds__(R * 16 * LPP + 64) void sgm_upsum_k(const UpArgs a)
{
    constexpr int Dp = LPP * DPL;
    constexpr int LD = Dp + 2;
    constexpr int COLS = 16;
    constexpr int TEAM = COLS * LPP;                                     // threads of a team (one image row)
    constexpr int RC = Dp + COLS;                                        // TIGHT ring of sgm_sum_wta_lr_k: second barrier per step
    constexpr int MIR = DPL;
    constexpr int NP = DPL / 2;
    constexpr int NW = (DPL + 3) / 4;
    constexpr int XP = (R > 1) ? R - 1 : 1;
    static_assert(RC % COLS == 0, "a ring slot must always belong to the same px");
    __shared__ unsigned short ring[R][(RC + MIR) * LD];
    constexpr int XCN = (R > 1) ? UPSUM_XC : 1;                          // one row per workgroup: no exchange through LDS
    __shared__ unsigned xch[XP][3][XCN][Dp / 4];                         // L_r bytes of the row below, by column mod 48
    __shared__ unsigned xmin[XP][3][XCN];                                // ... and their minimum over d (both halves)
    __shared__ unsigned ex_val[R][UPSUM_MAX_EXTRA * (Dp / 4)];
    __shared__ int ex_col[R][UPSUM_MAX_EXTRA];
    __shared__ unsigned lut32_s[256];
    __shared__ unsigned group_s;

    const int W = a.W, H = a.H, D = a.D, dmin = a.dmin;
    const int frame = blockIdx.x % a.B;
    for (int q = threadIdx.x; q < 256; q += R * TEAM + 64) lut32_s[q] = (unsigned)a.lut[q] * 0x00010001u;
    // A frame's row groups are taken in arrival order by the few workgroups the launch has per frame (a.ngroups of them would mostly
    // sit waiting for the rows below while holding a CU's LDS): whoever finishes a group takes the next one.  A group only ever
    // waits for the group before it, whose ticket was drawn earlier -- by a workgroup that is running or done.
    // The last wave of the workgroup computes nothing: it polls the progress word of the group below and publishes this group's.
    // Vector-memory results return in issue order, so a poll in a computing wave would drain that wave's plane prefetch every step.
    const bool helper = threadIdx.x >= R * TEAM;
    unsigned* const prog = a.progress + (size_t)frame * a.ngroups;
    const unsigned gen_base = a.gen << 13;
    const int x_last = a.do_right ? W - 1 + dmin + D - 1 : W - 1;        // last column any pixel of a row needs
    const int n_steps = (x_last + R - 1) / 16 + 1 + (R - 1);             // team t runs iteration step - t (clamped work outside its range)
    int k_prev = -1;
    for (;;) {
    __syncthreads();                                                     // the previous group's ring / extras / ticket word are done with, its stores drained
    if (helper && k_prev >= 0)                                           // ... so all of its top team's iterations are out
        __hip_atomic_store(&prog[k_prev], gen_base + (unsigned)(n_steps - R + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (threadIdx.x == 0) group_s = atomicAdd(&a.ticket[frame], 1u);
    __syncthreads();
    const int k = (int)group_s;                                          // row group, counted from the bottom of the image
    if (k >= a.ngroups) break;
    k_prev = k;
    unsigned long long* const tr = a.trace ? a.trace + ((size_t)frame * a.ngroups + k) * 12 : nullptr;
    if (helper) {
        if (tr && threadIdx.x == R * TEAM) {
            tr[0] = __builtin_amdgcn_s_memrealtime();
            unsigned xcc, hw;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
            tr[3] = ((unsigned long long)(xcc & 0xF) << 32) | hw;
        }
        // the group below must have published every column < W up to `col` (its top team's windows are shifted R - 1 columns)
        bool gave_up = false;
        auto wait_for = [&](int col) {
            if (gave_up || k == 0) return;                               // one timed-out poll: finish the launch without waiting again
            const unsigned want = gen_base + (unsigned)((min(col, W - 1) + R - 1) / 16 + 1);
            unsigned polls = 0;
            while ((int)(__hip_atomic_load(&prog[k - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - want) < 0) {
                __builtin_amdgcn_s_sleep(1);
                if (++polls > UPSUM_POLL_LIMIT) {                        // never in a healthy launch; do not hang the GPU
                    if (a.status) *a.status = 1;
                    gave_up = true;
                    break;
                }
            }
        };
        wait_for(16 + 16);                                               // iterations 0 and 1 of the bottom team read columns -1 .. 32
        if (tr && threadIdx.x == R * TEAM) tr[1] = __builtin_amdgcn_s_memrealtime();
        __syncthreads();
        for (int step = 0; step < n_steps; ++step) {
            wait_for(16 * (step + 2) + 16);                              // what the bottom team asks for at the START of the next step: iteration step + 2
            __syncthreads();                                             // A
            __syncthreads();                                             // B: every wave of the top team has seen its stores of this step complete
            if (step - R + 2 >= 1)
                __hip_atomic_store(&prog[k], gen_base + (unsigned)(step - R + 2), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (tr && threadIdx.x == R * TEAM) tr[2] = __builtin_amdgcn_s_memrealtime();
        continue;
    }
    const int t = threadIdx.x / TEAM;                                    // team: 0 = the group's bottom row
    const int tid = threadIdx.x % TEAM;
    const int sub = tid & (LPP - 1), px = tid / LPP;
    const bool first_lane = sub == 0, last_lane = sub == LPP - 1;
    const int y = H - 1 - (k * R + t);
    const bool row_ok = y >= 0;
    const int yc = row_ok ? y : 0;
    const bool start = (y == H - 1);                                     // first pixel of every line of the sweep: L = C (ref :266-275)

    const uint8_t* const img = a.img + (size_t)frame * W * H;
    const char* const clb = reinterpret_cast<const char*>(a.census_l + (size_t)frame * W * H);
    const uint8_t* const planes = a.planes + (size_t)frame * 8 * a.plane_bytes;
    const uint8_t* const extras = a.extras + (size_t)frame * 4 * H * Dp;
    float* const disp_l = a.disp_l + (size_t)frame * W * H + (size_t)yc * W;
    float* const disp_r = a.disp_r + (size_t)frame * W * H + (size_t)yc * W;
    uint8_t* const xb_out = a.xbuf + ((size_t)frame * 2 + (size_t)(k & 1)) * 3 * (size_t)W * Dp;
    const uint8_t* const xb_in = a.xbuf + ((size_t)frame * 2 + (size_t)((k + 1) & 1)) * 3 * (size_t)W * Dp;

    // ---- second visits of the anomalous lines that land on this team's row (as in sgm_sum_wta_lr_k) ----
    const int n_extra = row_ok ? min(a.row_extra_count[yc], UPSUM_MAX_EXTRA) : 0;
    for (int q = tid; q < n_extra * (Dp / 4); q += TEAM) {
        const int j = q / (Dp / 4), w = q % (Dp / 4);
        const sgmd_row_extra e = a.row_extras[yc * a.row_cap + j];
        if (w == 0) ex_col[t][j] = e.col_slot & 0xFFFF;
        ex_val[t][q] = *reinterpret_cast<const unsigned*>(extras + ((size_t)(e.col_slot >> 16) * H + e.step) * Dp + w * 4);
    }

    // ---- per-lane constants ----
    const unsigned lane_off = (unsigned)(sub * DPL);
    us2 padmask[NP];
    unsigned padpair[NP];                                                // 65535 in the ring for padding disparities
#pragma unroll
    for (int j = 0; j < NP; ++j) {
        const unsigned lo = ((int)lane_off + 2 * j >= D) ? 0x00FFu : 0u;
        const unsigned hi = ((int)lane_off + 2 * j + 1 >= D) ? 0x00FF0000u : 0u;
        padmask[j] = as_p(lo | hi);
        padpair[j] = (lo ? 0xFFFFu : 0u) | (hi ? 0xFFFF0000u : 0u);
    }
    const unsigned p1u = ((unsigned)a.p1 & 0xFFFFu) * 0x00010001u;
    const int back = dmin + (int)lane_off + DPL - 1;                     // census-right words of this lane start at pixel p - back
    const int lim_bias = dmin + (int)lane_off;
    const unsigned cbias = (unsigned)(dmin + Dp - back);
    const char* const crb = reinterpret_cast<const char*>(a.census_r + (size_t)frame * W * H - (dmin + Dp));
    const unsigned row_pix = (unsigned)yc * (unsigned)W;
    const unsigned below_pix = (unsigned)min(yc + 1, H - 1) * (unsigned)W;   // the row the sweep comes from
    const unsigned row_cells = row_pix * Dp + lane_off;
    const int tri_l = H - 1 - y;                                         // (1,-1): post-wrap cells x < tri_l; x == tri_l: nobody's cell
    const int tri_r = W - (H - 1 - y);                                   // (-1,-1): post-wrap cells x >= tri_r; x == tri_r - 1: nobody's
    const uint8_t* pb[5];
    pb[0] = planes; pb[1] = planes + a.plane_bytes; pb[2] = planes + 2 * a.plane_bytes; pb[3] = planes + 4 * a.plane_bytes;
    pb[4] = planes + 7 * a.plane_bytes;
    const uint8_t* const pd_ul = planes + 5 * a.plane_bytes;             // (-1,-1)
    const uint8_t* const pd_ur = planes + 6 * a.plane_bytes;             // (1,-1)


    // column of this thread in iteration i: 16 i - t + px (team t is shifted t columns to the left of the bottom team, so the
    // cells x-1 .. x+1 of the row below are always at least one iteration old)
    auto col_of = [&](int i) { return COLS * i - t + px; };
    auto cell_off = [&](int x) { return row_cells + (unsigned)min(max(x, 0), W - 1) * Dp; };

    // ---- prefetch state ----
    CellVec<DPL> pre[2][5], pre_ul[2], pre_ur[2];                        // planes of the iteration two steps ahead
    CensusVec<DPL> cvb[2];                                               // census / grey of the iterations one and two steps ahead (as the planes)
    unsigned clbuf[2];
    uint8_t gb_here[2], gb_up[2], gb_ul[2], gb_ur[2];
    CellVec<DPL> hinb[2][3];                                             // bottom team of a group above the first: L_r of the group below, one step ahead
    auto fetch_planes = [&](int stage, int x) {
        const unsigned off = cell_off(x);
#pragma unroll
        for (int d = 0; d < 5; ++d) load_cells_nt<DPL>(pb[d] + off, pre[stage][d]);
        const int xc = min(max(x, 0), W - 1);
        // the diagonal planes hold data only in their post-wrap triangles: elsewhere every lane reads the row's first cell (one hot line)
        load_cells<DPL>(pd_ul + ((xc >= tri_r) ? off : row_cells), pre_ul[stage]);
        load_cells<DPL>(pd_ur + ((xc < tri_l) ? off : row_cells), pre_ur[stage]);
    };
    auto fetch_row = [&](int stage, int x) {
        const int xc = min(max(x, 0), W - 1);
        const unsigned p = row_pix + (unsigned)xc;
        load_census<DPL>(reinterpret_cast<const uint32_t*>(crb + (size_t)((p + cbias) << 2)), cvb[stage]);
        clbuf[stage] = *reinterpret_cast<const uint32_t*>(clb + (size_t)(p << 2));
        gb_here[stage] = img[p];
        gb_up[stage] = img[below_pix + (unsigned)xc];
        gb_ul[stage] = img[below_pix + (unsigned)min(xc + 1, W - 1)];
        gb_ur[stage] = img[below_pix + (unsigned)max(xc - 1, 0)];
    };
    auto fetch_handover = [&](int stage, int x) {                                   // L_r(y+1, x), (y+1, x+1), (y+1, x-1) of directions 3, 5, 6
        const size_t dirb = (size_t)W * Dp;
        const unsigned c0 = (unsigned)min(max(x, 0), W - 1), c1 = (unsigned)min(max(x + 1, 0), W - 1), c2 = (unsigned)min(max(x - 1, 0), W - 1);
        load_cells_sc1<DPL>(xb_in + (size_t)c0 * Dp + lane_off, hinb[stage][0]);
        load_cells_sc1<DPL>(xb_in + dirb + (size_t)c1 * Dp + lane_off, hinb[stage][1]);
        load_cells_sc1<DPL>(xb_in + 2 * dirb + (size_t)c2 * Dp + lane_off, hinb[stage][2]);
    };
    const bool from_global = (t == 0 && k > 0);

    __syncthreads();                                                     // ex_val / ex_col are in place, the helper has seen iteration 0's columns
    // issue order = the order of use: a load's data waits for every older load (vmcnt counts in issue order)
    fetch_planes(0, col_of(-t));
    fetch_row(0, col_of(-t));
    if (from_global) fetch_handover(0, col_of(-t));
    fetch_planes(1, col_of(1 - t));
    fetch_row(1, col_of(1 - t));
    if (!from_global) {
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int dd = 0; dd < 3; ++dd)
#pragma unroll
                for (int w = 0; w < NW; ++w) hinb[q][dd].w[w] = 0;
    }

    unsigned long long ph[4] = {0, 0, 0, 0};                             // diagnostics: shader clocks per phase, summed over the steps
    const bool timed = tr != nullptr && tid == 0 && (t == 0 || t == R - 1);
    int slot = px;                                                       // ring slot of this thread's column (advances 16 per step)
    unsigned sent[2] = {0x00FF00FFu, 0x00FF00FFu};

    auto body = [&](int step, auto stage_tag) {
        constexpr int STAGE = decltype(stage_tag)::value;
        const int i = step - t;                                          // this team's iteration (outside [0, n) the work is clamped and masked)
        const int x = col_of(i);
        const int xc = min(max(x, 0), W - 1);
        const bool inside = row_ok && x >= 0 && x < W;
        const unsigned long long c0_ = tr ? __builtin_amdgcn_s_memtime() : 0;
        // the helper saw the group below publish iteration i + 1's columns before the previous step ended: its cells are asked for now and
        // used a whole step later
        if (from_global) fetch_handover(STAGE ^ 1, col_of(i + 1));
        // ---- S = five planes + the sweep (+ anomalous visits): packed u16 pairs as in sgm_sum_wta_lr_k ----
        unsigned aL[NW], aH[NW];
#pragma unroll
        for (int w = 0; w < NW; ++w) aL[w] = aH[w] = 0;
        auto add_bytes = [&](int w, unsigned v) {
            aL[w] += v & 0x00FF00FFu;                                        // bytes 0, 2
            aH[w] += __builtin_amdgcn_perm(0u, v, 0x0c030c01u);              // bytes 1, 3
        };
#pragma unroll
        for (int d = 0; d < 5; ++d)
#pragma unroll
            for (int w = 0; w < NW; ++w) add_bytes(w, pre[STAGE][d].w[w]);
        const bool in_ul = xc >= tri_r, in_ur = xc < tri_l;                  // post-wrap cells: the aggregation launch computed them
        const unsigned m_pl_ul = in_ul ? 0xFFFFFFFFu : 0u, m_pl_ur = in_ur ? 0xFFFFFFFFu : 0u;
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            add_bytes(w, pre_ul[STAGE].w[w] & m_pl_ul);
            add_bytes(w, pre_ur[STAGE].w[w] & m_pl_ur);
        }
        for (int j = 0; j < n_extra; ++j) {
            if (ex_col[t][j] == x) {
#pragma unroll
                for (int w = 0; w < NW; ++w) add_bytes(w, ex_val[t][j * (Dp / 4) + sub * NW + w]);
            }
        }
        // ---- the three directions of the sweep ----
        us2 C[NP];
        const int lim = xc - lim_bias;
        const bool border = __any(lim < DPL - 1) != 0;
        census_costs<DPL>(clbuf[STAGE], cvb[STAGE], lim, border, C);
        us2 Lup[NP], Lul[NP], Lur[NP];
        unsigned m_up, m_ul, m_ur;
        if (start) {
#pragma unroll
            for (int j = 0; j < NP; ++j) Lup[j] = Lul[j] = Lur[j] = PAD ? as_p(as_u(C[j]) | as_u(padmask[j])) : C[j];
            m_up = m_ul = m_ur = row_allmin_pk<LPP>(pk_min_tree<NP, FAST>(Lup));
        } else {
            us2 Pup[NP], Pul[NP], Pur[NP];
            unsigned q_up, q_ul, q_ur;
            if (t == 0) {
                unpack_cells<DPL>(hinb[STAGE][0], Pup);
                unpack_cells<DPL>(hinb[STAGE][1], Pul);
                unpack_cells<DPL>(hinb[STAGE][2], Pur);
                q_up = row_allmin_pk<LPP>(pk_min_tree<NP, FAST>(Pup));
                q_ul = row_allmin_pk<LPP>(pk_min_tree<NP, FAST>(Pul));
                q_ur = row_allmin_pk<LPP>(pk_min_tree<NP, FAST>(Pur));
            } else {
                const int tp = (R > 1) ? t - 1 : 0;
                const unsigned c0 = (unsigned)(x + 2 * UPSUM_XC) % XCN, c1 = (unsigned)(x + 1 + 2 * UPSUM_XC) % XCN,
                               c2 = (unsigned)(x - 1 + 2 * UPSUM_XC) % XCN;
                CellVec<DPL> e0, e1, e2;
#pragma unroll
                for (int w = 0; w < NW; ++w) {
                    e0.w[w] = xch[tp][0][c0][sub * NW + w];
                    e1.w[w] = xch[tp][1][c1][sub * NW + w];
                    e2.w[w] = xch[tp][2][c2][sub * NW + w];
                }
                unpack_cells<DPL>(e0, Pup);
                unpack_cells<DPL>(e1, Pul);
                unpack_cells<DPL>(e2, Pur);
                q_up = xmin[tp][0][c0];
                q_ul = xmin[tp][1][c1];
                q_ur = xmin[tp][2][c2];
            }
            const unsigned g = gb_here[STAGE];
            const unsigned pen_up = lut32_s[__builtin_amdgcn_sad_u8(g, (unsigned)gb_up[STAGE], 0u)];      // ref :335, |g - g_prev| of the visited pixels
            const unsigned pen_ul = lut32_s[__builtin_amdgcn_sad_u8(g, (unsigned)gb_ul[STAGE], 0u)];
            const unsigned pen_ur = lut32_s[__builtin_amdgcn_sad_u8(g, (unsigned)gb_ur[STAGE], 0u)];
            m_up = up_step<DPL, LPP, PAD, FAST>(C, border, Pup, q_up, pen_up, p1u, padmask, sent, first_lane, last_lane, Lup);
            m_ul = up_step<DPL, LPP, PAD, FAST>(C, border, Pul, q_ul, pen_ul, p1u, padmask, sent, first_lane, last_lane, Lul);
            m_ur = up_step<DPL, LPP, PAD, FAST>(C, border, Pur, q_ur, pen_ur, p1u, padmask, sent, first_lane, last_lane, Lur);
        }
        // ---- hand the new L_r to the row above: LDS inside the workgroup, the global hand-over row from the top team ----
        {
            CellVec<DPL> o0, o1, o2;
            pack_cells<DPL>(Lup, o0);
            pack_cells<DPL>(Lul, o1);
            pack_cells<DPL>(Lur, o2);
            if (t < R - 1) {
                const unsigned c = (unsigned)(x + 2 * UPSUM_XC) % XCN;
#pragma unroll
                for (int w = 0; w < NW; ++w) {
                    xch[t < XP ? t : 0][0][c][sub * NW + w] = o0.w[w];
                    xch[t < XP ? t : 0][1][c][sub * NW + w] = o1.w[w];
                    xch[t < XP ? t : 0][2][c][sub * NW + w] = o2.w[w];
                }
                if (sub == 0) {
                    xmin[t < XP ? t : 0][0][c] = m_up;
                    xmin[t < XP ? t : 0][1][c] = m_ul;
                    xmin[t < XP ? t : 0][2][c] = m_ur;
                }
            } else if (inside) {
                const size_t dirb = (size_t)W * Dp;
                uint8_t* const o = xb_out + (size_t)xc * Dp + lane_off;
                store_cells_sc1<DPL>(o, o0);
                store_cells_sc1<DPL>(o + dirb, o1);
                store_cells_sc1<DPL>(o + 2 * dirb, o2);
            }
        }
        // own cells of the two diagonals: everything that is neither post-wrap nor the anomalous line's empty track
        const unsigned m_own_ul = (in_ul || xc == tri_r - 1) ? 0u : 0xFFFFFFFFu;
        const unsigned m_own_ur = (in_ur || xc == tri_l) ? 0u : 0xFFFFFFFFu;
        unsigned pr[NP];
#pragma unroll
        for (int m = 0; m < NP; ++m) {
            unsigned v = __builtin_amdgcn_perm(aH[m >> 1], aL[m >> 1], (m & 1) ? 0x07060302u : 0x05040100u);
            unsigned lu = as_u(Lup[m]), l1 = as_u(Lul[m]) & m_own_ul, l2 = as_u(Lur[m]) & m_own_ur;
            if (PAD) { lu &= ~as_u(padmask[m]); l1 &= ~as_u(padmask[m]); l2 &= ~as_u(padmask[m]); }   // padding cells hold 255 in L, nothing in S
            v += lu + l1 + l2;
            pr[m] = inside ? (v | padpair[m]) : 0xFFFFFFFFu;
        }
        // ---- this column's S vector into the ring ----
        {
            unsigned* dst = reinterpret_cast<unsigned*>(&ring[t][up_umad24((unsigned)slot, (unsigned)LD, (unsigned)(sub * DPL))]);
#pragma unroll
            for (int m = 0; m < NP; ++m) dst[m] = pr[m];
            if (slot < MIR) {
#pragma unroll
                for (int m = 0; m < NP; ++m) dst[(RC * LD) / 2 + m] = pr[m];
            }
        }
        // ---- left-view WTA over the 16 lanes of the pixel ----
        unsigned key[DPL];
        unsigned kmin = 0xFFFFFFFFu;
#pragma unroll
        for (int m = 0; m < NP; ++m) {
            const unsigned idx = (unsigned)(sub * DPL + 2 * m);
            key[2 * m] = (pr[m] << 16) | idx;
            key[2 * m + 1] = (pr[m] & 0xFFFF0000u) | (idx + 1);
            kmin = min(kmin, min(key[2 * m], key[2 * m + 1]));
        }
        const unsigned kbest_l = row_allmin<LPP>(kmin);
        unsigned k2 = 0xFFFFFFFFu;
        {
            const unsigned nbest = ~kbest_l;
#pragma unroll
            for (int q = 0; q < DPL; ++q) k2 = min(k2, key[q] + nbest);
        }
        const unsigned ksecond_l = row_allmin<LPP>(k2) + kbest_l + 1;

        const unsigned long long c1_ = tr ? __builtin_amdgcn_s_memtime() : 0;
        __syncthreads();                                                 // A: ring columns, exchange cells and the helper's poll are done
        const unsigned long long c2_ = tr ? __builtin_amdgcn_s_memtime() : 0;
        // ---- prefetch for the iteration two steps ahead (this stage's registers are free again): census / grey, then the planes ----
        fetch_row(STAGE, col_of(i + 2));
        fetch_planes(STAGE, col_of(i + 2));

        // ---- right view from the ring, then ONE wta_finish for both views (lane 0: left, lane 1: right) ----
        unsigned kbest_r = 0, ksecond_r = 0;
        int base = 0;
        const int xr = x - dmin - (D - 1);
        if (a.do_right) {
            base = slot + RC - (D - 1);
            if (base >= RC) base -= RC;
            unsigned val[DPL];
            int first = base + sub * DPL;
            if (first >= RC) first -= RC;
            const unsigned short* const diag = &ring[t][up_umad24((unsigned)first, (unsigned)LD, (unsigned)(sub * DPL))];
#pragma unroll
            for (int q = 0; q < DPL; ++q) val[q] = diag[q * (LD + 1)];
            unsigned km = 0xFFFFFFFFu;
#pragma unroll
            for (int q = 0; q < DPL; ++q) {
                const int kk = sub * DPL + q;
                key[q] = (!PAD || kk < D) ? ((val[q] << 16) | (unsigned)kk) : 0xFFFFFFFFu;
                km = min(km, key[q]);
            }
            const unsigned kb = row_allmin<LPP>(km);
            const unsigned nb = ~kb;
            unsigned k3 = 0xFFFFFFFFu;
#pragma unroll
            for (int q = 0; q < DPL; ++q) k3 = min(k3, key[q] + nb);
            kbest_r = kb;
            ksecond_r = row_allmin<LPP>(k3) + kb + 1;
        }
        {
            const bool is_r = (sub == 1);
            const bool active = is_r ? (a.do_right && row_ok && xr >= 0 && xr < W) : (sub == 0 && inside);
            if (active) {
                const unsigned kb = is_r ? kbest_r : kbest_l, k2nd = is_r ? ksecond_r : ksecond_l;
                const int dbest = (int)(kb & 0xFFFFu);
                const int km = max(dbest - 1, 0), kp = min(dbest + 1, Dp - 1);
                int sm = base + km, sp = base + kp;
                if (sm >= RC) sm -= RC;
                if (sp >= RC) sp -= RC;
                if (!is_r) sm = sp = slot;
                WtaState st;
                st.m1 = kb >> 16;
                st.m2 = k2nd >> 16;
                st.d1 = (is_r && (kb >> 16) == 0xFFFFu) ? -1 : dbest;
                st.c1 = ring[t][up_umad24((unsigned)sm, (unsigned)LD, (unsigned)km)];
                st.c2 = ring[t][up_umad24((unsigned)sp, (unsigned)LD, (unsigned)kp)];
                st.pv = 0; st.want_next = false;
                float* const out = is_r ? disp_r + xr : disp_l + x;
                *out = wta_finish(st, D, dmin, a.check_unique, a.one_minus_ratio);
            }
        }
        // The top team's hand-over stores were issued before barrier A; behind them this wave has issued at least 14 loads (fetch_row:
        // >= 7, fetch_planes: 7).  Vector-memory operations complete in issue order, so once at most 12 are outstanding the stores
        // are in L2 -- without draining the prefetch -- and the helper may publish this iteration behind barrier B.
        if (t == R - 1) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        const unsigned long long c3_ = tr ? __builtin_amdgcn_s_memtime() : 0;
        __syncthreads();                                                 // B: every diagonal of this step has been read (TIGHT ring)
        if (timed) {
            const unsigned long long c4_ = __builtin_amdgcn_s_memtime();
            ph[0] += c1_ - c0_; ph[1] += c2_ - c1_; ph[2] += c3_ - c2_; ph[3] += c4_ - c3_;
        }
        slot += COLS;
        if (slot >= RC) slot -= RC;
    };

    int step = 0;
    for (; step + 1 < n_steps; step += 2) {
        body(step, std::integral_constant<int, 0>{});
        body(step + 1, std::integral_constant<int, 1>{});
    }
    if (step < n_steps) body(step, std::integral_constant<int, 0>{});
    if (timed) {
        const int o = (t == 0) ? 4 : 8;
        if (!(R == 1 && o == 8)) { tr[o] = ph[0]; tr[o + 1] = ph[1]; tr[o + 2] = ph[2]; tr[o + 3] = ph[3]; }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                     // this wave's hand-over stores are out before the group is called done
    }
}

template <int DPL, int LPP, int R>
static void launch_upsum(const UpArgs& a, bool pad, bool fast, int wgs_per_frame, hipStream_t st)
{
    const dim3 grid((unsigned)(wgs_per_frame * a.B)), block(R * 16 * LPP + 64);
    if (pad) {
        if (fast) hipLaunchKernelGGL((sgm_upsum_k<DPL, LPP, R, true, true>), grid, block, 0, st, a);
        else      hipLaunchKernelGGL((sgm_upsum_k<DPL, LPP, R, true, false>), grid, block, 0, st, a);
    } else {
        if (fast) hipLaunchKernelGGL((sgm_upsum_k<DPL, LPP, R, false, true>), grid, block, 0, st, a);
        else      hipLaunchKernelGGL((sgm_upsum_k<DPL, LPP, R, false, false>), grid, block, 0, st, a);
    }
}

extern "C" {

int sgmd_upsum_rows(const sgmd_geom* g)                                 // most rows per workgroup, 0 = this shape keeps the separate kernels
{
    if (g->Dp != 128 || g->W <= g->H || g->W < 64 || g->H < 4 || g->row_begin != 0 || g->row_end != g->H) return 0;
    return 3;
}

size_t sgmd_upsum_scratch_bytes(const sgmd_geom* g)                      // hand-over rows + progress words + tickets (for any rows per workgroup)
{
    if (!sgmd_upsum_rows(g)) return 0;
    const size_t groups = (size_t)g->H;
    return (size_t)g->B * 2 * 3 * g->W * g->Dp + ((size_t)g->B * groups + (size_t)g->B + 64) * sizeof(unsigned);
}

int sgmd_upsum(int ord, void* stream, const sgmd_geom* g, const sgmd_paths* paths, const void* img_left, const void* census_l,
               const void* census_r, const void* lut, const void* planes, size_t plane_bytes, const void* extras,
               const void* row_extras, const void* row_extra_count, int row_cap, int do_right, int check_unique, float one_minus_ratio,
               void* scratch, unsigned generation, void* status, int rows, int wgs_per_frame, void* disp_l, void* disp_r)
{
    HIP_TRY(hipSetDevice(ord));
    const int R = rows;
    if (R < 1 || R > sgmd_upsum_rows(g) || paths->ndirs != 8 || paths->p1 < 0 || row_cap > UPSUM_MAX_EXTRA) {
        fprintf(stderr, "sgm_mi355x: the fused last sweep does not cover this configuration\n");
        return -1;
    }
    hipStream_t st = (hipStream_t)stream;
    UpArgs a;
    a.img = (const uint8_t*)img_left;
    a.census_l = (const uint32_t*)census_l;
    a.census_r = (const uint32_t*)census_r;
    a.planes = (const uint8_t*)planes;
    a.plane_bytes = plane_bytes;
    a.extras = (const uint8_t*)extras;
    a.row_extras = (const sgmd_row_extra*)row_extras;
    a.row_extra_count = (const int*)row_extra_count;
    a.row_cap = row_cap;
    a.lut = (const uint16_t*)lut;
    a.disp_l = (float*)disp_l;
    a.disp_r = (float*)disp_r;
    a.ngroups = (g->H + R - 1) / R;
    a.xbuf = (uint8_t*)scratch;
    a.progress = (unsigned*)((char*)scratch + (size_t)g->B * 2 * 3 * g->W * g->Dp);
    a.ticket = a.progress + (size_t)g->B * a.ngroups;
    a.status = (int*)status;
    a.gen = generation & 0x7FFFFu;
    a.W = g->W; a.H = g->H; a.D = g->D; a.dmin = g->dmin; a.B = g->B; a.p1 = paths->p1;
    a.check_unique = check_unique;
    a.one_minus_ratio = one_minus_ratio;
    a.do_right = do_right;
    HIP_TRY(hipMemsetAsync(a.ticket, 0, (size_t)g->B * sizeof(unsigned), st));
    // diagnostics: SGM_UPSUM_TRACE=file -> per-group timestamps of THIS launch are written there (the launch is waited for)
    static const char* trace_path = getenv("SGM_UPSUM_TRACE");
    static unsigned long long* d_trace = nullptr;
    static size_t trace_cap = 0;
    const size_t trace_n = (size_t)g->B * a.ngroups * 12;
    a.trace = nullptr;
    if (trace_path && *trace_path) {
        if (trace_n > trace_cap) {
            if (d_trace) (void)hipFree(d_trace);
            HIP_TRY(hipMalloc((void**)&d_trace, trace_n * 8));
            trace_cap = trace_n;
        }
        HIP_TRY(hipMemsetAsync(d_trace, 0, trace_n * 8, st));
        a.trace = d_trace;
    }
    const bool pad = g->D != g->Dp;
    const bool fast = paths->allow_fast && paths->p1 <= 31488 && paths->pen_max <= 223;
    // workgroups per frame: what the chain of row groups can keep busy at once (steps of a group / steps between the starts of two
    // groups), not one per group -- the others would hold LDS waiting
    int wgs = wgs_per_frame > 0 ? wgs_per_frame : (R == 1 ? 48 : 16);
    if (wgs > a.ngroups) wgs = a.ngroups;
    // 8 lanes per pixel x 16 disparities per lane (the 16-lane layout of sgm_sum_wta_lr_k measured the same per step with twice the
    // instructions and spilled registers at three rows per workgroup: profiles/r04_fused_last_sweep.txt)
    switch (R) {
    case 1: launch_upsum<16, 8, 1>(a, pad, fast, wgs, st); break;
    case 2: launch_upsum<16, 8, 2>(a, pad, fast, wgs, st); break;
    default: launch_upsum<16, 8, 3>(a, pad, fast, wgs, st); break;
    }
    HIP_TRY(hipGetLastError());
    if (a.trace) {
        unsigned long long* h = (unsigned long long*)malloc(trace_n * 8);
        HIP_TRY(hipStreamSynchronize(st));
        HIP_TRY(hipMemcpy(h, d_trace, trace_n * 8, hipMemcpyDeviceToHost));
        if (FILE* f = fopen(trace_path, "wb")) {
            const int hdr[4] = {g->B, a.ngroups, R, 0};
            fwrite(hdr, sizeof hdr, 1, f);
            fwrite(h, 8, trace_n, f);
            fclose(f);
        }
        free(h);
    }
    return 0;
}

}  // extern "C"

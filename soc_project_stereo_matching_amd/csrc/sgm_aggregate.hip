#include "sgm_aggregate_impl.hpp"

// kernels for negative P1 live in sgm_aggregate_generic.hip (their own translation unit: parallel build)
bool sgmd_aggregate_launch_generic(int lpp, int dpl, const AggArgs* a, int blocks, int pad, hipStream_t st);
bool sgmd_aggregate_launch_volume(int dpl, const AggArgs* a, int blocks, int pad, hipStream_t st);
bool sgmd_aggregate_launch_anom(int dp, const AggArgs* a, int pad, int from_volume, hipStream_t st);
// ... and those with the shortcuts for ordinary penalties in sgm_aggregate_fast.hip
bool sgmd_aggregate_launch_fast(int lpp, int dpl, const AggArgs* a, int blocks, int pad, int hl, hipStream_t st);

static int aggregate_any(int ord, void* stream, const sgmd_geom* g, const sgmd_paths* paths, const void* img_left,
                         const void* census_l, const void* census_r, const void* cost, const void* lut, void* planes,
                         size_t plane_bytes, void* extras)
{
    HIP_TRY(hipSetDevice(ord));
    AggArgs a;
    a.img = (const uint8_t*)img_left;
    a.census_l = (const uint32_t*)census_l;
    a.census_r = (const uint32_t*)census_r;
    a.cost = (const uint8_t*)cost;
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
    // SGM_DIR_MASK (diagnostics only: wrong results): run a subset of the directions, to time them apart; bit 8 = the anomalous lines
    static const int debug_mask = getenv("SGM_DIR_MASK") ? (int)strtol(getenv("SGM_DIR_MASK"), nullptr, 0) : 0x1FF;
    int blocks = 0;
    a.post_wrap_mask = 0;
    // the fused last sweep (sgmd_upsum) computes the upward directions itself, except the cells of the diagonal lines BEHIND their
    // wrap around the image edge (W > H: line i of (1,-1) visits (H-1-k, (i+k) mod W) and wraps iff i >= W-H+1; line i of (-1,-1)
    // visits (H-1-k, (i-k) mod W) and wraps iff i <= H-2)
    const bool up_fused = paths->up_fused && g->W > g->H && g->row_begin == 0 && g->row_end == g->H;
    for (int d = 0; d < 8; ++d) {
        a.dx[d] = paths->dx[d]; a.dy[d] = paths->dy[d]; a.anom_line[d] = paths->anom_line[d];
        a.block_begin[d] = blocks;
        a.line_lo[d] = 0;
        a.line_n[d] = g->W;
        bool run = d < paths->ndirs && ((paths->dir_mask & debug_mask) >> d) & 1;
        if (up_fused && paths->dy[d] == -1) {
            if (paths->dx[d] == 0) run = false;
            else {
                a.line_lo[d] = paths->dx[d] > 0 ? g->W - g->H + 1 : 0;
                a.line_n[d] = g->H - 1;
                a.post_wrap_mask |= 1 << d;
            }
        }
        if (run) {
            const int nlines = (paths->dy[d] == 0) ? g->row_end - g->row_begin : a.line_n[d];
            const int lines_per_wave = 64 / ((paths->dy[d] == 0 && g->HL) ? g->HL : g->LPP);
            blocks += (nlines + lines_per_wave - 1) / lines_per_wave;
        }
    }
    a.block_begin[8] = blocks;
    if (!(debug_mask & 0x100)) a.run_anom = 0;
    const bool pad = (g->D != g->Dp);
    hipStream_t st = (hipStream_t)stream;
    a.strips = 1;
    // The anomalous line of every diagonal direction (one wave each): with the kernels of the non-negative-P1 step the FIRST four
    // blocks of every frame of the regular lines' launch; otherwise (negative P1, cost volume: the generic step, whose registers
    // would set the occupancy of the whole kernel) or where no regular line is asked for, a small launch of its own in front.
    const bool nn_kernels = !cost && a.p1 >= 0;
    a.anom_inline = (a.run_anom && nn_kernels && blocks > 0) ? 1 : 0;
    if (a.run_anom && !a.anom_inline) {
        if (!sgmd_aggregate_launch_anom(g->Dp, &a, pad ? 1 : 0, cost ? 1 : 0, st)) {
            fprintf(stderr, "sgm_mi355x: no anomalous-line kernel for a cell of %d disparities\n", g->Dp);
            return -1;
        }
        HIP_TRY(hipGetLastError());
    }
    if (blocks == 0) return 0;
    if (a.anom_inline) blocks += 4;
    // XCD-aware numbering for 2 or 4 frames per launch (sgm_aggregate_k): each of a frame's 8 / B XCDs takes a contiguous strip of
    // every direction; the grid is 8 x the longest per-XCD list.  Measured (tools/fetch_probe.sh): census reads of the launch
    // 4.07 -> 1.06 GB per frame at 2880x1988 D=256, 0.34 -> 0.12 GB at 1762x800 D=192.  Not for a single frame: its launch is
    // bound by the horizontal chains, not by traffic, and runs 7 % longer with the strips (0.340 -> 0.364 ms at KITTI size).
    // SGM_XCD_STRIPS = 0: never; 2: also for one frame per launch (the counter passes of tools/profile_counters.py run one
    // frame per launch and use it to see the traffic of the two-frame launches bench.py times).
    static const int use_strips = getenv("SGM_XCD_STRIPS") ? atoi(getenv("SGM_XCD_STRIPS")) : 1;
    a.strips = (use_strips && g->B < 8 && 8 % g->B == 0 && (g->B > 1 || use_strips > 1)) ? 8 / g->B : 1;
    if (a.strips > 1) {
        int longest = 0;
        for (int sub = 0; sub < a.strips; ++sub) {
            int t = 0;
            for (int d = 0; d <= 8; ++d) {
                const int n = (d < 8 ? a.block_begin[d + 1] - a.block_begin[d] : (a.anom_inline ? 4 : 0));
                t += (sub + 1) * n / a.strips - sub * n / a.strips;
            }
            if (t > longest) longest = t;
        }
        blocks = 8 * longest;
    } else
        blocks *= g->B;                                // every frame of the batch in the same launch
    bool launched;
    if (cost)           launched = sgmd_aggregate_launch_volume(g->DPL, &a, blocks, pad ? 1 : 0, st);
    else if (a.p1 >= 0) {
        // ordinary penalties (agg_step_nn's FAST conditions) unless the host asks for the plain non-negative-P1 step
        const bool fast = paths->allow_fast && a.p1 <= 31488 && paths->pen_max <= 223;
        launched = fast ? sgmd_aggregate_launch_fast(g->LPP, g->DPL, &a, blocks, pad ? 1 : 0, g->HL, st)
                        : launch_aggregate_key<1>(g->LPP, g->DPL, a, blocks, pad, g->HL, st);
    }
    else                launched = sgmd_aggregate_launch_generic(g->LPP, g->DPL, &a, blocks, pad ? 1 : 0, st);
    if (!launched) {
        fprintf(stderr, "sgm_mi355x: unsupported lanes-per-pixel/DPL combination %d/%d\n", g->LPP, g->DPL);
        return -1;
    }
    HIP_TRY(hipGetLastError());
    return 0;
}

extern "C" {

int sgmd_aggregate(int ord, void* stream, const sgmd_geom* g, const sgmd_paths* paths, const void* img_left,
                   const void* census_l, const void* census_r, const void* lut, void* planes, size_t plane_bytes,
                   void* extras)
{
    return aggregate_any(ord, stream, g, paths, img_left, census_l, census_r, nullptr, lut, planes, plane_bytes, extras);
}

/* diagnostics (builds with -DSGM_CLOCK_PROBE only; -1 otherwise): shader-clock ticks and 100 MHz ticks over the lifetime of
 * block 0 of the last aggregation launch of this translation unit's kernels */
int sgmd_debug_clock(int ord, unsigned long long out[2])
{
#ifdef SGM_CLOCK_PROBE
    HIP_TRY(hipSetDevice(ord));
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_sgm_agg_clock), 2 * sizeof(unsigned long long)));
    return 0;
#else
    (void)ord; out[0] = out[1] = 0;
    return -1;
#endif
}

int sgmd_aggregate_volume(int ord, void* stream, const sgmd_geom* g, const sgmd_paths* paths, const void* img_left,
                          const void* cost, const void* lut, void* planes, size_t plane_bytes, void* extras)
{
    if (g->LPP != 16 || g->HL != 0) {
        fprintf(stderr, "sgm_mi355x: the volume-fed aggregation needs 16 lanes per pixel on every line\n");
        return -1;
    }
    return aggregate_any(ord, stream, g, paths, img_left, nullptr, nullptr, cost, lut, planes, plane_bytes, extras);
}

}  // extern "C"

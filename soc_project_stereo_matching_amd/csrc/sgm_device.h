/*
 * sgm_device.h -- the thin C interface between the C host (sgm_host.c) and the HIP translation
 * units (sgm_*.hip).  Plain C types only; device memory is passed as void*.
 * Every function returns 0 on success and a non-zero HIP error code otherwise (the message is
 * printed to stderr by the HIP side), except where noted.
 */
#ifndef SGM_DEVICE_H
#define SGM_DEVICE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- device, stream, memory ---- */
int   sgmd_device_count(void);                       /* <= 0: no usable device */
int   sgmd_device_is_gfx950(int ordinal);            /* 1 yes, 0 no, <0 error */
int   sgmd_stream_create(int ordinal, void** stream);
/* a stream whose kernels run only on compute units [first, first + count) of EVERY XCD (hipExtStreamCreateWithCUMask; the mask's
 * bit i is CU i / xcds of XCD i % xcds).  count <= 0: an ordinary stream on all CUs */
int   sgmd_stream_create_cus(int ordinal, void** stream, int first_per_xcd, int count_per_xcd);
int   sgmd_device_cus(int ordinal, int* cus_per_xcd, int* xcds);     /* 32 x 8 on MI355X */
int   sgmd_stream_create_prio(int ordinal, void** stream, int priority);   /* 0 normal, < 0 higher, > 0 lower (clamped) */
int   sgmd_stream_destroy(int ordinal, void* stream);
int   sgmd_stream_sync(int ordinal, void* stream);
/* events without timing, for ordering one stream behind another */
int   sgmd_event_create(int ordinal, void** event);
void  sgmd_event_destroy(int ordinal, void* event);
int   sgmd_event_record(int ordinal, void* event, void* stream);
int   sgmd_stream_wait_event(int ordinal, void* stream, void* event);
int   sgmd_event_sync(int ordinal, void* event);                     /* the host waits */
int   sgmd_set_device(int ordinal);                                  /* the calling thread's current device (for libraries such as RCCL) */
int   sgmd_mem_info(int ordinal, size_t* free_bytes, size_t* total_bytes);
int   sgmd_alloc(int ordinal, void** dptr, size_t bytes);
int   sgmd_free(int ordinal, void* dptr);
int   sgmd_alloc_pinned(int ordinal, void** hptr, size_t bytes);
int   sgmd_free_pinned(int ordinal, void* hptr);
int   sgmd_host_is_pinned(int ordinal, const void* hptr, size_t bytes);   /* 1: [hptr, hptr+bytes) is page-locked HIP host memory, else 0 */
int   sgmd_h2d_async(int ordinal, void* stream, void* dst, const void* src, size_t bytes);
int   sgmd_d2h_async(int ordinal, void* stream, void* dst, const void* src, size_t bytes);
int   sgmd_d2d_async(int ordinal, void* stream, void* dst, const void* src, size_t bytes);
/* `rows` pieces of `width` bytes each, `src_pitch` / `dst_pitch` bytes apart (device to device; the row gather of a batch packs the same
 * rows of every map into one message this way) */
int   sgmd_d2d_2d_async(int ordinal, void* stream, void* dst, size_t dst_pitch, const void* src, size_t src_pitch, size_t width, size_t rows);
/* One image row of up to 8 direction planes of every frame of the batch <-> a packed buffer [frame][k][row_bytes] (the row-tile
 * hand-over): plane d = dirs[k] of frame f starts at planes + (f * 8 + d) * plane_bytes; to_buf != 0 gathers, else scatters. */
int   sgmd_plane_rows_copy(int ordinal, void* stream, void* planes, size_t plane_bytes, size_t row_offset, size_t row_bytes,
                           const int* dirs, int ndirs, int frames, void* buf, int to_buf);
int   sgmd_memset_async(int ordinal, void* stream, void* dst, int value, size_t bytes);

/* ---- per-stage timing with HIP events on the stream ---- */
int   sgmd_timer_create(int ordinal, void** timer, int max_marks);
void  sgmd_timer_destroy(int ordinal, void* timer);
int   sgmd_timer_mark(int ordinal, void* timer, void* stream, int index);           /* records event #index */
int   sgmd_timer_elapsed(int ordinal, void* timer, int from, int to, float* ms);    /* after a sync */

/* ---- geometry shared by all launchers ---- */
typedef struct {
    int W, H;          /* image */
    int D;             /* max_disparity - min_disparity */
    int Dp;            /* padded cell stride of the volumes: 16 * DPL >= D */
    int DPL;           /* disparities per lane in the aggregation kernel (2,4,8,12,16,32) */
    int LPP;           /* lanes per pixel in the aggregation kernel: 16 (4 lines per wave) or 8 (8 lines); Dp = LPP*DPL */
    int HL;            /* lanes per pixel of the horizontal lines: 0 = LPP, or 32 / 64 (2 / 1 lines per wave; needs
                          Dp/HL in {2,4,8,16}): the launcher falls back to LPP where the combination does not exist */
    int dmin;          /* min_disparity */
    int B;             /* frames per launch (batch): every buffer is [B][...] frame-major, every kernel
                          processes all B frames */
    int row_begin, row_end;  /* the rows this GPU computes in aggregate / sum_wta / wta_right / lrcheck: [0,H) normally,
                          a row tile when a frame is spread over several GPUs (buffers stay frame-sized) */
} sgmd_geom;

/* one anomalous-line visit that lands on a pixel of row `row` (built by the host) */
typedef struct {
    int32_t col_slot;  /* column | (diagonal slot 0..3) << 16 */
    int32_t step;      /* index k into the slot's extras rows */
} sgmd_row_extra;

typedef struct {
    int ndirs;                 /* 4 or 8, reference order SemiGlobalMatching.c:213-220 */
    int dx[8], dy[8];
    int anom_line[8];          /* line index whose first step trips the wrong edge test, or -1 */
    int ghost_zero;            /* 1: W >= H, the anomalous wave zeroes the cells no line visits */
    int p1;
    int pen_max;               /* largest entry of the adaptive-P2 table = max(P1, P2_init, 0): picks the aggregation step family */
    int allow_fast;            /* 0: never the FAST step family (sgm_aggregate_fast.hip), whatever the penalties (tests run both) */
    int dir_mask;              /* bit d: run direction d in this launch (0xFF normally; a tile sweep runs a subset) */
    int run_anom;              /* 1: run the anomalous diagonal lines in this launch (whole frame, never tiled) */
    int up_fused;              /* 1: the upward directions belong to the fused last sweep (sgmd_upsum): this launch skips (0,-1) and walks
                                  of (-1,-1) / (1,-1) only the H-1 lines that wrap around the image edge, storing their post-wrap cells */
} sgmd_paths;

/* ---- stage launchers (all asynchronous on `stream`) ---- */

/* 5x5 census of both images; border = 0.  SemiGlobalMatching.c:134-159 */
/* need: NULL, or one byte per block of the sgmd_census_blocks grid (row-major): 0 = leave the block's words as they are (a
 * row-tile instance only needs its own rows and the pixels the four anomalous lines read) */
/* keep_border != 0: the words the reference never writes (the 2-pixel border; everything when W <= 5 or H <= 5, .c:136,140-141)
 * are left as they are instead of being written as 0 -- the reference's static buffers keep what an earlier frame of another
 * shape left at the same linear index */
int sgmd_census(int ord, void* stream, const sgmd_geom* g, const void* left, const void* right,
                void* census_l, void* census_r, const void* need, int keep_border);
void sgmd_census_blocks(const sgmd_geom* g, int* blocks_x, int* blocks_y);     /* blocks of 64 x 16 pixels */

/* Hamming matching cost volume, u8 [H][W][Dp].  SemiGlobalMatching.c:161-196 */
int sgmd_cost(int ord, void* stream, const sgmd_geom* g, const void* census_l, const void* census_r, void* cost);

/* Extension: census over an odd cw x ch window of at most 64 pixels into u64 words, and the Hamming cost
 * volume of those words (u8 [H][W][Dp], as sgmd_cost); sgmd_aggregate then takes the volume instead of the census images */
int sgmd_census_window(int ord, void* stream, const sgmd_geom* g, int cw, int ch, const void* left, const void* right,
                       void* census64_l, void* census64_r);
int sgmd_cost64(int ord, void* stream, const sgmd_geom* g, const void* census64_l, const void* census64_r, void* cost);

/* All directions of the path aggregation in ONE launch.  SemiGlobalMatching.c:198-372.
 * The matching cost (SemiGlobalMatching.c:161-196) is recomputed from the census images inside the
 * kernel; census_r must be preceded by at least sgmd_census_slack(g) readable bytes (disparities that
 * reach left of column 0 read there and are masked to 127).
 * planes: u8 [ndirs][H][W][Dp] per-direction path costs L_r; extras: u8 [4][H][Dp] path costs of
 * the four anomalous diagonal lines (step-major); lut: u16[256] = (uint16)max(P1, P2/(a+1)). */
size_t sgmd_census_slack(const sgmd_geom* g);
int sgmd_aggregate(int ord, void* stream, const sgmd_geom* g, const sgmd_paths* paths, const void* img_left,
                   const void* census_l, const void* census_r, const void* lut, void* planes, size_t plane_bytes,
                   void* extras);
/* the same aggregation fed from a materialised cost volume (sgmd_cost / sgmd_cost64) instead of recomputing the cost:
 * what the wide census windows use (generic step, 16 lanes per pixel) */
int sgmd_aggregate_volume(int ord, void* stream, const sgmd_geom* g, const sgmd_paths* paths, const void* img_left,
                          const void* cost, const void* lut, void* planes, size_t plane_bytes, void* extras);

/* S = (accumulate ? S : 0) + sum of planes + anomalous-line visits (u16 [H][W][Dp]) and, fused, the LEFT-view
 * winner-take-all with uniqueness and sub-pixel (SemiGlobalMatching.c:374-443, inverse == 0) -> disp_l */
int sgmd_sum_wta(int ord, void* stream, const sgmd_geom* g, int ndirs, const void* planes, size_t plane_bytes,
                 const void* extras, const void* row_extras, const void* row_extra_count, int row_cap, int accumulate,
                 void* S, int check_unique, float one_minus_ratio, void* disp_l);

/* The same sum and BOTH winner-take-all passes in one kernel (Dp <= 256: sgmd_sum_wta_lr_supported): a workgroup
 * walks an image row and keeps the last Dp+32 columns of S in LDS for the right view's diagonal gather, so S is
 * written only if store_S (and read only if accumulate); do_right = 0 skips the right view (no LR check). */
int sgmd_sum_wta_lr_supported(const sgmd_geom* g, int row_cap);   /* row_cap: most anomalous-line visits in one row */
int sgmd_sum_wta_lr(int ord, void* stream, const sgmd_geom* g, int ndirs, const void* planes, size_t plane_bytes,
                    const void* extras, const void* row_extras, const void* row_extra_count, int row_cap, int accumulate,
                    int store_S, int do_right, void* S, int check_unique, float one_minus_ratio, void* disp_l, void* disp_r);

/* The LAST vertical sweep -- directions (0,-1), (-1,-1), (1,-1), SemiGlobalMatching.c:216,218,219 -- fused with the cost sum and both
 * winner-take-all passes (sgm_upsum.hip): reads the five other planes and the post-wrap cells of the two diagonal ones (an
 * aggregation launch with paths->up_fused), writes the two disparity maps; S and the three planes never exist.
 * sgmd_upsum_rows: the most image rows a workgroup may take (sgmd_upsum's rows_per_workgroup: 1 .. that), 0 where the shape keeps the separate kernels (needs W > H, Dp = 128, whole frames).
 * scratch: sgmd_upsum_scratch_bytes(g) bytes, zero when allocated; generation: a different number for every launch on that scratch;
 * status: NULL or a page-locked int set to 1 if a workgroup gave up waiting for the rows below (the maps are then wrong);
 * workgroups_per_frame: 0 = the launcher's choice (a frame's row groups are drawn from a ticket by that many persistent workgroups). */
int sgmd_upsum_rows(const sgmd_geom* g);
size_t sgmd_upsum_scratch_bytes(const sgmd_geom* g);
int sgmd_upsum(int ord, void* stream, const sgmd_geom* g, const sgmd_paths* paths, const void* img_left, const void* census_l,
               const void* census_r, const void* lut, const void* planes, size_t plane_bytes, const void* extras,
               const void* row_extras, const void* row_extra_count, int row_cap, int do_right, int check_unique, float one_minus_ratio,
               void* scratch, unsigned generation, void* status, int rows_per_workgroup, int workgroups_per_frame, void* disp_l, void* disp_r);

/* right-view winner-take-all on S[y][x+d][d] (SemiGlobalMatching.c:395-408) -> disp_r; needed by the LR check only */
int sgmd_wta_right(int ord, void* stream, const sgmd_geom* g, const void* S, int check_unique, float one_minus_ratio,
                   void* disp_r);

/* SemiGlobalMatching.c:445-470 */
int sgmd_lrcheck(int ord, void* stream, const sgmd_geom* g, void* disp_l, const void* disp_r, float thres);

/* Extension: the right view as the reference view.  out = disp_r where the mirrored check against disp_l passes (all of
 * disp_r if !do_check), +INF elsewhere; out must not alias either input */
int sgmd_lrcheck_right(int ord, void* stream, const sgmd_geom* g, const void* disp_r, const void* disp_l, float thres,
                       int do_check, void* out);

/* connected components (|delta| <= diff, 8-neighbourhood) smaller than min_area -> +INF.
 * labels/sizes/totals: int32 [H][W] scratch each.  SemiGlobalMatching.c:585-642 */
int sgmd_speckle(int ord, void* stream, const sgmd_geom* g, void* disp, float diff, unsigned min_area,
                 void* labels, void* sizes, void* totals);

/* in-place raster-order 3x3 median (the reference calls MedianFilter with in == out, .c:120).
 * scratch: sgmd_median_scratch_bytes(g) bytes for the pre-sorted neighbourhoods. */
/* status: NULL, or an int in page-locked host memory (sgmd_alloc_pinned) that the chained kernel of tall frames sets to 1 when a
 * band gave up waiting for the rows of the band above (bounded polls): the map is then wrong and the host fails the match */
size_t sgmd_median_scratch_bytes(const sgmd_geom* g);
int sgmd_median(int ord, void* stream, const sgmd_geom* g, void* disp, void* scratch, void* status);

/* SURVEY.md 8f-3 on device buffers: depth[mm] = float32(fx * baseline) / (disparity + doffs), NaN where the denominator is not
 * finite or zero; and (blocking) the sums behind RMSE / bad-pixel rate over the pixels finite in both depth images */
int sgmd_depth(int ord, void* stream, const void* disp, size_t n, float fx, float baseline, float doffs, void* depth);
int sgmd_score(int ord, void* stream, const void* ground_truth, const void* test, size_t n, float abs_thresh, double* sum_sq,
               unsigned long long* n_valid, unsigned long long* n_bad);

/* grey = (weight_r r + 150 g + 29 b) >> 8 of three consecutive n-byte planes B, G, R (the test platform's frame
 * format, server.py:105-131; the firmware's conversion, stereo_matching.c:18-25) */
int sgmd_gray_planes(int ord, void* stream, const void* bgr, size_t n, int weight_r, void* gray);

#ifdef __cplusplus
}
#endif
#endif

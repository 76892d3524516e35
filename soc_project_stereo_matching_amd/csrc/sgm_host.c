/*
 * sgm_host.c -- the C host of libsgm_mi355x.so.
 *
 * Implements the reference's library boundary (SGM_Initialize / SGM_Reset / SGM_Match,
 * /root/reference/SemiGlobalMatching/SemiGlobalMatching/SemiGlobalMatching.h:78-80) plus the
 * extensions of include/sgm_mi355x.h on top of the HIP stage launchers of sgm_device.h.
 * This file owns everything that is not a kernel: option validation, buffer sizing, the
 * adaptive-P2 table, the path-geometry tables for the anomalous diagonal lines and the order
 * of the stages (the body of SGM_Match, SemiGlobalMatching.c:77-122).
 *
 * There is no CPU fallback: without a usable gfx950 device every entry point fails loudly.
 */
#include "../../include/sgm_mi355x.h"
#include "sgm_device.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define SGM_VERSION_STRING "sgm_mi355x 0.3 (gfx950, hand-written HIP)"
#define CENSUS_FRONT_SLACK ((size_t)(65535 + SGM_MAX_DISPARITY_RANGE + 8 + 63) / 64 * 64 * 4)   /* >= sgmd_census_slack() for any options */

#define TIMING_RING 64
enum { T_CENSUS, T_COST, T_AGGREGATE, T_SUM, T_WTA, T_LRCHECK, T_SPECKLE, T_MEDIAN, T_COUNT };
/* event marks of a match: 0 .. T_COUNT bracket the stages in order; one more, M_SUM_BEGIN, sits right in front of the cost sum
 * BEHIND its waits for other streams' events, so that "sum" is the kernel's time and not the wait for the previous post pass */
#define M_SUM_BEGIN (T_COUNT + 1)
#define MARKS_PER_MATCH (T_COUNT + 2)
static const char* const k_stage_names[T_COUNT] = {"census", "cost", "aggregate", "sum", "wta", "lrcheck", "speckle", "median"};

/* reference direction order, SemiGlobalMatching.c:213-220 */
static const int k_dir_dx[8] = {1, -1, 0, 0, 1, -1, 1, -1};
static const int k_dir_dy[8] = {0, 0, 1, -1, 1, -1, -1, 1};

struct sgm_instance {
    int device;
    void* stream;                /* census and aggregation; and every other stage unless it has a stream of its own (sgm_set_stage_cus) */
    void* sum_stream;            /* cost sum + both WTAs, behind ev_agg (the aggregation is done) */
    void* post_stream;           /* LR check, speckle removal and median, behind ev_sum (cost sum + WTAs done); ev_post = post pass done */
    void *ev_agg, *ev_sum, *ev_post;
    int overlap_post;
    bool sum_pending;            /* a cost sum is (possibly) still reading the planes on sum_stream */
    bool post_pending;           /* a post pass is (possibly) still running on post_stream */
    void* tail_stream;           /* the stream the last match's final kernel was queued on (NULL: stream) */
    int cu_first[3], cu_count[3];/* CUs per XCD of the main / sum / post stream (count 0: all CUs) */
    int up_rows;                 /* > 0: the last vertical sweep (directions (0,-1), (-1,-1), (1,-1)) runs fused with the cost sum and both WTAs
                                    (sgmd_upsum) wherever a match allows it: image rows per workgroup of that kernel */
    void* d_up_scratch;          /* its hand-over rows, progress words and tickets */
    size_t cap_up_scratch;
    void* d_left_keep;           /* copy of the last fused match's left image(s): what re-creating the three planes needs (Q14, stage read-back) */
    size_t cap_left_keep;
    unsigned up_gen;             /* launch counter of the fused kernel (its progress words carry it) */
    int last_up_rows;            /* rows per workgroup of the fused sweep in the LAST match, 0 if it ran the separate kernels */
    bool planes_partial;         /* the planes of the last frame lack the upward directions: materialize_S re-creates them first */
    int env_upsum, env_upsum_rows, env_upsum_wgs;   /* SGM_UPSUM, SGM_UPSUM_ROWS, SGM_UPSUM_WGS */
    int env_lanes, env_hl, env_agg_fast, env_fused;   /* SGM_LANES_PER_PIXEL, SGM_HL, SGM_AGG_FAST, SGM_FUSED_WTA as read at sgm_create
                                    (-1: not set) -- tuning / test knobs, not looked up again on the per-frame sgm_reset path */
    bool stage_prio[3];          /* that stream was made by sgm_set_stage_priority (an all-CU request must replace it, not keep it) */
    int* h_status;               /* page-locked word the chained median kernel sets when a band gave up waiting (sgmd_median) */
    void* timer;
    int timing;
    int keep_stages;
    int honor_num_paths;
    int census_w, census_h;      /* census window (sgm_set_census_window); 0 = the reference's 5x5 */
    int reference_view;          /* 0 = left (reference), 1 = right (sgm_set_reference_view) */
    int reference_statics;       /* the default instance behind SGM_Initialize / SGM_Reset / SGM_Match: its census buffers behave like
                                    the reference's static arrays (SemiGlobalMatching.h:67-68) -- zero at first, never cleared, the words
                                    census_transform_5x5 does not write (.c:136,140-141) keep what an earlier frame of another shape left
                                    at the same linear index.  Explicit instances (an extension) write those words as 0 */
    int batch;                   /* frames per match call (>= 1); takes effect at the next initialize */
    int read_frame;              /* which frame of the batch sgm_read_stage returns */
    int tile_begin, tile_end;    /* row tile this instance computes (sgm_set_rows); tile_end == 0: the whole frame */
    const void* tile_left;       /* left image of the frame a tile sequence is working on */

    bool initialized;
    bool s_is_zero;              /* aggregated-cost volume logically zero (set by Initialize/Reset, Q14) */
    int fused_wta;               /* Dp <= 256: cost sum and both WTA passes in one kernel, S not written */
    bool s_pending;              /* the planes hold a frame whose sum has not been put into d_S (fused kernel, S not
                                    stored): done lazily when somebody needs S -- a Match without Reset, a stage read */
    bool s_pending_accumulate;   /* ... and that sum adds to d_S (true) or replaces it */
    SGMOption opt;
    sgmd_geom g;
    sgmd_paths paths;
    int need_plane_memset;       /* W < H: diagonal planes are cleared before aggregation */
    int row_cap;
    float last_ms[T_COUNT];
    bool have_ms;
    /* timing history: TIMING_RING event sets, one per match since the last collection */
    int ring_next, ring_pending;          /* next set to record into; sets recorded and not yet read */
    double sum_ms[T_COUNT], min_ms[T_COUNT];
    long n_timed;

    /* device buffers (capacity tracked so a Reset with the same shape allocates nothing) */
    size_t cap_px, cap_planes, cap_S, cap_cost, cap_extras, cap_median;
    int plane_row_lo, plane_rows;        /* image rows a direction plane has storage for: [plane_row_lo, plane_row_lo + plane_rows) --
                                            the whole frame normally, the tile's rows + one hand-over row either side in row-tile mode */
    int cap_H, cap_row_cap;
    int tab_W, tab_H, tab_ndirs, tab_p1, tab_p2;   /* what the uploaded tables were built for */
    void *d_left, *d_right, *d_census_l, *d_census_r, *d_census_r_alloc, *d_cost, *d_planes, *d_planes_alloc, *d_extras, *d_S;
                                 /* d_planes = d_planes_alloc - plane_row_lo rows: kernels address cells by their frame
                                    position; d_cost and d_S exist only once somebody needs them (stage read-back, Q14, D > 256) */
    void *d_disp, *d_disp_r, *d_labels, *d_sizes, *d_lut, *d_row_extras, *d_row_count;
    void *d_snap_wta, *d_snap_lr, *d_snap_speckle, *d_totals, *d_median_scratch;
    void *d_census64_l, *d_census64_r;   /* u64 census words of the wide windows (allocated on first use) */
    size_t cap_census64;
    void* d_census_need;                 /* row tiles: which 64 x 16 blocks of the census this instance reads (sgmd_census) */
    size_t cap_census_need;
    int need_key[7];                     /* W, H, rows, dmin, Dp, ndirs the map was built for */
    void *d_bgr, *d_depth, *h_bgr;       /* a test-platform frame's six colour planes, its depth map, pinned staging (first use) */
    size_t cap_bgr;
    size_t plane_bytes;
    /* pinned staging for the host-pointer entry point */
    void *h_left, *h_right, *h_disp;
    /* sgm_match_async: a match whose result has been queued on the stream and not yet handed to the caller */
    bool async_pending;
    float* async_out;            /* caller's buffer the staged result still has to be copied to (NULL: it was pinned, the
                                    device wrote it directly) */
    size_t async_bytes;
    /* a staged result (pageable caller buffer) comes back in RESULT_CHUNKS pieces, an event behind each: sgm_match_wait copies piece i
     * to the caller while piece i + 1 is still on the bus (a 1242x375 map: 1.86 MB, ~40 us of DMA + ~90 us of memcpy in sequence otherwise) */
    void* ev_chunk[4];
    int async_chunks;
};
#define UPSUM_DEFAULT 0       /* the fused last sweep is opt-in (SGM_UPSUM=1) until it beats the separate kernels in the timed pipeline */
#define RESULT_CHUNKS 4
#define RESULT_CHUNK_MIN ((size_t)256 << 10)      /* smaller results are not worth the events */
#define RESULT_CHUNK_SPLIT ((size_t)4 << 20)      /* below this: two pieces */

#define FAIL(...)                                  \
    do {                                           \
        fprintf(stderr, "sgm_mi355x: " __VA_ARGS__); \
        fputc('\n', stderr);                       \
        return false;                              \
    } while (0)

/* wait for everything the instance has queued (its stream and, with sgm_set_overlap_post, the post-pass stream) */
static int sync_streams(sgm_instance* s)
{
    int rc = sgmd_stream_sync(s->device, s->stream);
    if (s->sum_stream && sgmd_stream_sync(s->device, s->sum_stream) != 0) rc = -1;
    if (s->post_stream && sgmd_stream_sync(s->device, s->post_stream) != 0) rc = -1;
    if (rc == 0) { s->post_pending = s->sum_pending = false; s->tail_stream = NULL; }
    if (rc == 0 && s->h_status && *s->h_status) {
        *s->h_status = 0;
        fprintf(stderr, "sgm_mi355x: the median of a tall frame gave up waiting for the band above; the result is not valid\n");
        rc = -1;
    }
    return rc;
}

/* ------------------------------------------------------------------ path geometry (host) */

/* The reference's pointer walk (SemiGlobalMatching.c:243-255, 281-323, 359-367)
 * for one line.  Writes visited linear pixel indices, returns their count; a step that leaves
 * the image ends the line (the reference's undefined behaviour, defined away: SURVEY.md Q6). */
static int walk_line(int W, int H, int dx, int dy, int line, int32_t* pix)
{
    const int fwd = (dx == 1 && dy == 0) || (dx == 0 && dy == 1) || (dx == 1 && dy == 1) || (dx == -1 && dy == 1);
    const int s = fwd ? 1 : -1;
    const long long npx = (long long)W * H;
    long long p = (dy == 0) ? (long long)line * W + (fwd ? 0 : W - 1) : (fwd ? 0 : (long long)(H - 1) * W) + line;
    const int steps = (dy == 0 ? W : H) - 1;
    unsigned row = (unsigned)(fwd ? 0 : H - 1) & 0xFFFFu, col = (unsigned)line & 0xFFFFu;
    int n = 0;
    pix[n++] = (int32_t)p;
    for (int j = 0; j < steps; ++j) {
        if (dy == 0) p += s;
        else if (dx == 0) p += (long long)s * W;
        else {
            const int not_last = fwd ? ((int)row < H - 1) : (row > 0);
            if ((int)col == W - 1 && not_last) { p = ((long long)row + s) * W; col = 0; }
            else if (col == 0 && not_last) { p = ((long long)row + s) * W + (W - 1); col = (unsigned)(W - 1); }
            else p += (long long)s * (W + (dx == dy ? 1 : -1));
        }
        if (p < 0 || p >= npx) break;
        pix[n++] = (int32_t)p;
        row = (row + (unsigned)s) & 0xFFFFu;
        col = (col + (unsigned)((dx == dy || dx == 0 || dy == 0) ? s : -s)) & 0xFFFFu;
    }
    return n;
}

/* exported for the host-logic tests (not part of the public header) */
int sgm_host_walk_line(int W, int H, int dx, int dy, int line, int32_t* pix) { return walk_line(W, H, dx, dy, line, pix); }

/* The line of a diagonal direction whose very first step trips the wrong edge test
 * (SemiGlobalMatching.c:297,304 do not look at dx): it starts in column 0 while
 * moving right, or in column W-1 while moving left. */
static int anomalous_line(int W, int dx) { return dx > 0 ? 0 : W - 1; }
int sgm_host_anomalous_line(int W, int dx) { return anomalous_line(W, dx); }

/* (uint16) max(P1, P2 / (a + 1)), a = |grey difference| (SemiGlobalMatching.c:335) */
static void build_p2_table(int p1, int p2_init, uint16_t* lut)
{
    for (int a = 0; a < 256; ++a) {
        int pen = p2_init / (a + 1);
        if (p1 > pen) pen = p1;
        lut[a] = (uint16_t)pen;
    }
}
void sgm_host_p2_table(int p1, int p2_init, uint16_t* lut) { build_p2_table(p1, p2_init, lut); }

static int pick_dpl(int D)
{
    if (D <= 32) return 2;
    if (D <= 64) return 4;
    if (D <= 128) return 8;
    if (D <= 192) return 12;
    if (D <= 256) return 16;
    return 32;
}

/* ------------------------------------------------------------------ instance management */

static bool device_usable(int device)
{
    const int n = sgmd_device_count();
    if (n <= 0) FAIL("no HIP device available (this library has no CPU fallback)");
    if (device < 0 || device >= n) FAIL("device %d out of range (%d visible)", device, n);
    if (sgmd_device_is_gfx950(device) != 1) FAIL("device %d is not gfx950 (MI355X); kernels are built for gfx950 only", device);
    return true;
}

static int env_int(const char* name)
{
    const char* e = getenv(name);
    return (e && *e) ? atoi(e) : -1;
}

sgm_instance* sgm_create(int device)
{
    if (!device_usable(device)) return NULL;
    sgm_instance* s = (sgm_instance*)calloc(1, sizeof *s);
    if (!s) return NULL;
    s->device = device;
    s->batch = 1;
    if (sgmd_stream_create(device, &s->stream) != 0) { free(s); return NULL; }
    void* st = NULL;
    if (sgmd_alloc_pinned(device, &st, 64) != 0) { sgmd_stream_destroy(device, s->stream); free(s); return NULL; }
    s->h_status = (int*)st;
    *s->h_status = 0;
    s->env_lanes = env_int("SGM_LANES_PER_PIXEL");
    s->env_hl = env_int("SGM_HL");
    s->env_agg_fast = env_int("SGM_AGG_FAST");
    s->env_fused = env_int("SGM_FUSED_WTA");
    s->env_upsum = env_int("SGM_UPSUM");
    s->env_upsum_rows = env_int("SGM_UPSUM_ROWS");
    s->env_upsum_wgs = env_int("SGM_UPSUM_WGS");
    return s;
}

static void free_device_buffers(sgm_instance* s)
{
    s->d_census_r = NULL;                                        /* points into d_census_r_alloc */
    s->d_planes = NULL;                                          /* points into (or in front of) d_planes_alloc */
    void** all[] = {&s->d_left, &s->d_right, &s->d_census_l, &s->d_census_r_alloc, &s->d_cost, &s->d_planes_alloc, &s->d_extras,
                    &s->d_S, &s->d_disp, &s->d_disp_r, &s->d_labels, &s->d_sizes, &s->d_lut, &s->d_row_extras,
                    &s->d_row_count, &s->d_snap_wta, &s->d_snap_lr, &s->d_snap_speckle, &s->d_totals,
                    &s->d_median_scratch, &s->d_census64_l, &s->d_census64_r, &s->d_bgr, &s->d_depth, &s->d_census_need,
                    &s->d_up_scratch, &s->d_left_keep};
    for (size_t i = 0; i < sizeof all / sizeof all[0]; ++i) {
        sgmd_free(s->device, *all[i]);
        *all[i] = NULL;
    }
    sgmd_free_pinned(s->device, s->h_left);
    sgmd_free_pinned(s->device, s->h_right);
    sgmd_free_pinned(s->device, s->h_disp);
    sgmd_free_pinned(s->device, s->h_bgr);
    s->h_left = s->h_right = s->h_disp = s->h_bgr = NULL;
    s->cap_px = s->cap_planes = s->cap_S = s->cap_cost = s->cap_extras = s->cap_median = s->cap_census64 = s->cap_bgr = 0;
    s->cap_census_need = 0;
    s->cap_up_scratch = s->cap_left_keep = 0;
    s->need_key[0] = 0;
    s->cap_H = s->cap_row_cap = 0;
    s->tab_W = s->tab_H = 0;
}

void sgm_destroy(sgm_instance* s)
{
    if (!s) return;
    sgm_match_wait(s);
    sync_streams(s);
    free_device_buffers(s);
    sgmd_timer_destroy(s->device, s->timer);
    sgmd_event_destroy(s->device, s->ev_agg);
    sgmd_event_destroy(s->device, s->ev_sum);
    sgmd_event_destroy(s->device, s->ev_post);
    for (int i = 0; i < 4; ++i)
        if (s->ev_chunk[i]) sgmd_event_destroy(s->device, s->ev_chunk[i]);
    if (s->sum_stream) sgmd_stream_destroy(s->device, s->sum_stream);
    if (s->post_stream) sgmd_stream_destroy(s->device, s->post_stream);
    sgmd_stream_destroy(s->device, s->stream);
    sgmd_free_pinned(s->device, s->h_status);
    free(s);
}

void sgm_set_honor_num_paths(sgm_instance* s, int honor) { if (s) s->honor_num_paths = honor; }

/* the three ordering events exist as soon as any stage has a stream of its own; all or none */
static bool ensure_stage_events(sgm_instance* s)
{
    void** ev[3] = {&s->ev_agg, &s->ev_sum, &s->ev_post};
    for (int i = 0; i < 3; ++i)
        if (!*ev[i] && sgmd_event_create(s->device, ev[i]) != 0) { *ev[i] = NULL; return false; }
    return true;
}

/* Stage groups on streams of their own, optionally on their own compute units (include/sgm_mi355x.h).  The instance is idle
 * while its streams change. */
bool sgm_set_stage_cus(sgm_instance* s, int which, int first_per_xcd, int count_per_xcd)
{
    if (!s || which < SGM_STAGE_MAIN || which > SGM_STAGE_POST) return false;
    if (!sgm_match_wait(s) || sync_streams(s) != 0) return false;
    void** slot = which == SGM_STAGE_MAIN ? &s->stream : (which == SGM_STAGE_SUM ? &s->sum_stream : &s->post_stream);
    if (count_per_xcd < 0) {                                     /* back to the default */
        if (which == SGM_STAGE_MAIN) return sgm_set_stage_cus(s, which, 0, 0);
        if (*slot) sgmd_stream_destroy(s->device, *slot);
        *slot = NULL;
        if (which == SGM_STAGE_POST) s->overlap_post = 0;
        s->cu_first[which] = s->cu_count[which] = 0;
        s->stage_prio[which] = false;
        return true;
    }
    if (*slot && !s->stage_prio[which] && s->cu_first[which] == first_per_xcd && s->cu_count[which] == count_per_xcd) {
        if (which == SGM_STAGE_POST) s->overlap_post = 1;
        return true;
    }
    void* fresh = NULL;
    if (!ensure_stage_events(s) || sgmd_stream_create_cus(s->device, &fresh, first_per_xcd, count_per_xcd) != 0) return false;
    if (*slot) sgmd_stream_destroy(s->device, *slot);
    *slot = fresh;
    s->cu_first[which] = first_per_xcd;
    s->cu_count[which] = count_per_xcd;
    s->stage_prio[which] = false;
    if (which == SGM_STAGE_POST) s->overlap_post = 1;
    return true;
}

/* the group's own stream, on all CUs, with a dispatch priority (include/sgm_mi355x.h) */
bool sgm_set_stage_priority(sgm_instance* s, int which, int priority)
{
    if (!s || which < SGM_STAGE_MAIN || which > SGM_STAGE_POST) return false;
    if (!sgm_match_wait(s) || sync_streams(s) != 0) return false;
    void** slot = which == SGM_STAGE_MAIN ? &s->stream : (which == SGM_STAGE_SUM ? &s->sum_stream : &s->post_stream);
    void* fresh = NULL;
    if (!ensure_stage_events(s) || sgmd_stream_create_prio(s->device, &fresh, priority) != 0) return false;
    if (*slot) sgmd_stream_destroy(s->device, *slot);
    *slot = fresh;
    s->cu_first[which] = s->cu_count[which] = 0;
    s->stage_prio[which] = true;
    if (which == SGM_STAGE_POST) s->overlap_post = 1;
    return true;
}

bool sgm_set_overlap_post(sgm_instance* s, int enable)
{
    if (!s) return false;
    if (enable) return s->post_stream ? (s->overlap_post = 1, true) : sgm_set_stage_cus(s, SGM_STAGE_POST, 0, 0);
    if (s->overlap_post) sync_streams(s);                        /* the next match is ordered on sgm_stream alone again */
    s->overlap_post = 0;
    return true;
}

bool sgm_set_census_window(sgm_instance* s, int width, int height)
{
    if (!s || width < 1 || height < 1 || !(width & 1) || !(height & 1) || width * height > 64) return false;
    if (width == 5 && height == 5) width = height = 0;           /* the reference's window: the fused fast path */
    if (width != s->census_w || height != s->census_h) s->initialized = false;   /* takes effect at the next initialize */
    s->census_w = width; s->census_h = height;
    return true;
}

void sgm_set_reference_view(sgm_instance* s, int right) { if (s) s->reference_view = right ? 1 : 0; }
void sgm_keep_stages(sgm_instance* s, int enable) { if (s) s->keep_stages = enable; }

bool sgm_set_batch(sgm_instance* s, int frames)
{
    if (!s || frames < 1 || frames > 1024) return false;
    if (frames != s->batch) s->initialized = false;              /* buffers are sized at the next initialize */
    s->batch = frames;
    return true;
}

void sgm_select_frame(sgm_instance* s, int frame)
{
    if (s && frame >= 0 && frame < s->batch) s->read_frame = frame;
}
void* sgm_stream(sgm_instance* s) { return s ? s->stream : NULL; }
int sgm_fused_sweep_rows(const sgm_instance* s) { return s ? s->last_up_rows : 0; }

void sgm_enable_timing(sgm_instance* s, int enable)
{
    if (!s) return;
    if (enable && !s->timer && sgmd_timer_create(s->device, &s->timer, TIMING_RING * MARKS_PER_MATCH) != 0) return;
    s->timing = enable;
    /* (re-)enabling starts a new statistics window: drop what was recorded before */
    sync_streams(s);
    s->ring_next = s->ring_pending = 0;
    s->n_timed = 0;
    for (int i = 0; i < T_COUNT; ++i) { s->sum_ms[i] = 0; s->min_ms[i] = 1e30; }
}

int sgm_mean_timing(sgm_instance* s, const char** names, float* mean_ms, float* min_ms, int max_entries, long* matches)
{
    if (matches) *matches = s ? s->n_timed : 0;
    if (!s || s->n_timed == 0) return 0;
    int n = 0;
    for (int i = 0; i < T_COUNT && n < max_entries; ++i, ++n) {
        if (names) names[n] = k_stage_names[i];
        if (mean_ms) mean_ms[n] = (float)(s->sum_ms[i] / (double)s->n_timed);
        if (min_ms) min_ms[n] = (float)s->min_ms[i];
    }
    return n;
}

int sgm_last_timing(sgm_instance* s, const char** names, float* ms, int max_entries)
{
    if (!s || !s->have_ms) return 0;
    int n = 0;
    for (int i = 0; i < T_COUNT && n < max_entries; ++i, ++n) {
        if (names) names[n] = k_stage_names[i];
        if (ms) ms[n] = s->last_ms[i];
    }
    return n;
}

/* Row tiles: the census words this instance reads are those of its own rows (horizontal lines, the sweeps, the cost sum's
 * recomputation) and, on every row of the frame, the pixel an anomalous line visits with the Dp words to its left.  Everything
 * else of the replicated images is skipped: sgmd_census takes one byte per 64 x 16 block.  (The aggregation's masked and
 * prefetched reads beyond that see whatever the buffer holds, as they always did at row starts.) */
static bool upload_census_need(sgm_instance* s)
{
    const int W = s->g.W, H = s->g.H;
    const int key[7] = {W, H, s->g.row_begin, s->g.row_end, s->g.dmin, s->g.Dp, s->paths.ndirs};
    if (s->d_census_need && memcmp(key, s->need_key, sizeof key) == 0) return true;
    int bx, by;
    sgmd_census_blocks(&s->g, &bx, &by);
    const size_t n = (size_t)bx * by;
    uint8_t* need = (uint8_t*)calloc(n, 1);
    int32_t* pix = (int32_t*)malloc(sizeof(int32_t) * (size_t)(W > H ? W : H));
    if (!need || !pix) { free(need); free(pix); FAIL("out of host memory"); }
    const int bw = 64, bh = 16;                               /* sgmd_census_blocks */
    for (int r = s->g.row_begin / bh; r <= (s->g.row_end - 1) / bh; ++r) memset(need + (size_t)r * bx, 1, (size_t)bx);
    if (s->paths.ndirs > 4) {
        const int back = s->g.dmin + s->g.Dp + 64;            /* words left of the pixel: range + the widest vector load */
        for (int d = 4; d < 8; ++d) {
            const int cnt = walk_line(W, H, k_dir_dx[d], k_dir_dy[d], s->paths.anom_line[d], pix);
            for (int k = 0; k < cnt; ++k) {
                const int y = pix[k] / W, x = pix[k] % W;
                const int c0 = (x - back < 0 ? 0 : x - back) / bw, c1 = x / bw;
                memset(need + (size_t)(y / bh) * bx + c0, 1, (size_t)(c1 - c0 + 1));
                /* a window that starts left of column 0 continues at the end of the row above (masked, but keep it defined) */
                if (x - back < 0 && y > 0) need[(size_t)((y - 1) / bh) * bx + bx - 1] = 1;
            }
        }
    }
    free(pix);
    bool ok = true;
    if (n > s->cap_census_need || !s->d_census_need) {
        sync_streams(s);
        sgmd_free(s->device, s->d_census_need);
        s->d_census_need = NULL;
        ok = sgmd_alloc(s->device, &s->d_census_need, n) == 0;
        s->cap_census_need = ok ? n : 0;
    }
    ok = ok && sgmd_h2d_async(s->device, s->stream, s->d_census_need, need, n) == 0 && sync_streams(s) == 0;
    free(need);
    if (!ok) FAIL("uploading the census block map failed");
    memcpy(s->need_key, key, sizeof key);
    return true;
}

/* Build the per-row table of anomalous-line visits and upload it together with the P2 table. */
static bool upload_tables(sgm_instance* s)
{
    const int W = s->g.W, H = s->g.H;
    uint16_t lut[256];
    build_p2_table(s->opt.p1, s->opt.p2_init, lut);

    int32_t* pix = (int32_t*)malloc(sizeof(int32_t) * (size_t)(W > H ? W : H));
    int* count = (int*)calloc((size_t)H, sizeof(int));
    int32_t* visits = (int32_t*)malloc(sizeof(int32_t) * 4 * (size_t)H);      /* [slot][k] pixel or -1 */
    if (!pix || !count || !visits) { free(pix); free(count); free(visits); FAIL("out of host memory"); }
    for (int i = 0; i < 4 * H; ++i) visits[i] = -1;
    if (s->paths.ndirs > 4) {
        for (int slot = 0; slot < 4; ++slot) {
            const int d = 4 + slot;
            const int n = walk_line(W, H, k_dir_dx[d], k_dir_dy[d], s->paths.anom_line[d], pix);
            for (int k = 0; k < n; ++k) {
                visits[slot * H + k] = pix[k];
                count[pix[k] / W]++;
            }
        }
    }
    int cap = 1;
    for (int r = 0; r < H; ++r) if (count[r] > cap) cap = count[r];
    sgmd_row_extra* table = (sgmd_row_extra*)calloc((size_t)H * cap, sizeof *table);
    if (!table) { free(pix); free(count); free(visits); FAIL("out of host memory"); }
    memset(count, 0, sizeof(int) * (size_t)H);
    for (int slot = 0; slot < 4; ++slot)
        for (int k = 0; k < H; ++k) {
            const int32_t p = visits[slot * H + k];
            if (p < 0) continue;
            const int r = p / W, c = p % W;
            table[(size_t)r * cap + count[r]].col_slot = c | (slot << 16);
            table[(size_t)r * cap + count[r]].step = k;
            count[r]++;
        }

    bool ok = true;
    if (cap > s->cap_row_cap || H > s->cap_H || !s->d_row_extras) {
        sgmd_free(s->device, s->d_row_extras);
        sgmd_free(s->device, s->d_row_count);
        s->d_row_extras = s->d_row_count = NULL;
        ok = sgmd_alloc(s->device, &s->d_row_extras, sizeof *table * (size_t)H * cap) == 0 &&
             sgmd_alloc(s->device, &s->d_row_count, sizeof(int) * (size_t)H) == 0;
        s->cap_row_cap = cap;
        s->cap_H = H;
    }
    s->row_cap = cap;
    /* plain blocking-safe uploads: the tables live on the host stack/heap only until the sync below */
    ok = ok && sgmd_h2d_async(s->device, s->stream, s->d_row_extras, table, sizeof *table * (size_t)H * cap) == 0 &&
         sgmd_h2d_async(s->device, s->stream, s->d_row_count, count, sizeof(int) * (size_t)H) == 0 &&
         sgmd_h2d_async(s->device, s->stream, s->d_lut, lut, sizeof lut) == 0 &&
         sync_streams(s) == 0;
    free(table); free(visits); free(count); free(pix);
    if (!ok) FAIL("uploading path tables failed");
    return true;
}

/* Per-pixel buffers (sized for the whole batch) and the per-direction planes.  A plane only has storage for the rows
 * the instance works on: every row of the frame normally; in row-tile mode the tile's rows plus the row either side that
 * a neighbouring GPU's hand-over lands in -- 1/N of the frame, which is what lets N + 2 frames be in flight per GPU. */
static bool ensure_buffers(sgm_instance* s)
{
    const int dev = s->device;
    const size_t px = (size_t)s->g.B * s->g.W * s->g.H;          /* all frames of the batch, frame-major */
    int rc = 0;
    if (px > s->cap_px || !s->d_disp) {
        sync_streams(s);
        /* the census words of earlier frames outlive a re-allocation (reference_statics): set the old buffers aside */
        void *old_l = s->d_census_l, *old_r_alloc = s->d_census_r_alloc;
        const size_t old_px = s->cap_px;
        s->d_census_l = s->d_census_r_alloc = NULL;
        free_device_buffers(s);
        rc |= sgmd_alloc(dev, &s->d_left, px);
        rc |= sgmd_alloc(dev, &s->d_right, px);
        rc |= sgmd_alloc(dev, &s->d_census_l, px * 4);
        /* the aggregation kernel reads census-right up to dmin + Dp - 1 words left of a row start (masked to 127
         * afterwards); give the buffer that much readable slack in front, sized for the largest options */
        rc |= sgmd_alloc(dev, &s->d_census_r_alloc, CENSUS_FRONT_SLACK + px * 4);
        if (rc == 0) s->d_census_r = (char*)s->d_census_r_alloc + CENSUS_FRONT_SLACK;
        /* zero like the reference's statics; then the words earlier frames left, at their linear indices */
        if (rc == 0) rc |= sgmd_memset_async(dev, s->stream, s->d_census_l, 0, px * 4);
        if (rc == 0) rc |= sgmd_memset_async(dev, s->stream, s->d_census_r_alloc, 0, CENSUS_FRONT_SLACK + px * 4);
        if (rc == 0 && s->reference_statics && old_l && old_r_alloc && old_px) {
            rc |= sgmd_d2d_async(dev, s->stream, s->d_census_l, old_l, old_px * 4);
            rc |= sgmd_d2d_async(dev, s->stream, s->d_census_r, (char*)old_r_alloc + CENSUS_FRONT_SLACK, old_px * 4);
        }
        if (rc == 0) rc |= sgmd_stream_sync(dev, s->stream);
        sgmd_free(dev, old_l);
        sgmd_free(dev, old_r_alloc);
        rc |= sgmd_alloc(dev, &s->d_disp, px * 4);
        rc |= sgmd_alloc(dev, &s->d_disp_r, px * 4);
        rc |= sgmd_alloc(dev, &s->d_labels, px * 4);
        rc |= sgmd_alloc(dev, &s->d_sizes, px * 4);
        rc |= sgmd_alloc(dev, &s->d_totals, px * 4);
        rc |= sgmd_alloc(dev, &s->d_lut, 512);
        rc |= sgmd_alloc(dev, &s->d_snap_wta, px * 4);
        rc |= sgmd_alloc(dev, &s->d_snap_lr, px * 4);
        rc |= sgmd_alloc(dev, &s->d_snap_speckle, px * 4);
        rc |= sgmd_alloc_pinned(dev, &s->h_left, px);
        rc |= sgmd_alloc_pinned(dev, &s->h_right, px);
        rc |= sgmd_alloc_pinned(dev, &s->h_disp, px * 4);
        if (rc != 0) { free_device_buffers(s); FAIL("device allocation failed for %dx%dx%d", s->g.W, s->g.H, s->g.D); }
        s->cap_px = px;
    }
    const bool tiled = s->tile_end != 0;
    s->plane_row_lo = tiled && s->g.row_begin > 0 ? s->g.row_begin - 1 : 0;
    const int row_hi = tiled && s->g.row_end < s->g.H ? s->g.row_end + 1 : s->g.H;
    s->plane_rows = row_hi - s->plane_row_lo;
    s->plane_bytes = (size_t)s->plane_rows * s->g.W * s->g.Dp;
    const size_t need = (size_t)s->g.B * 8 * s->plane_bytes;
    if (need > s->cap_planes || !s->d_planes_alloc) {
        sync_streams(s);
        sgmd_free(dev, s->d_planes_alloc);
        s->d_planes_alloc = NULL;
        s->cap_planes = 0;
        if (sgmd_alloc(dev, &s->d_planes_alloc, need) != 0)
            FAIL("device allocation failed for the path-cost planes of %dx%dx%d (%zu bytes)", s->g.W, s->g.H, s->g.D, need);
        s->cap_planes = need;
    }
    s->d_planes = (void*)((uintptr_t)s->d_planes_alloc - (uintptr_t)s->plane_row_lo * s->g.W * s->g.Dp);
    return true;
}

/* S (u16 per cell) and the cost volume (u8 per cell) are frame-sized and rarely needed: the fused kernels neither read
 * nor write them.  They are allocated (S zero-filled) the first time something does: a Match without Reset (Q14), the
 * separate sum / right-view kernels (D > 256, SGM_FUSED_WTA=0), sgm_keep_stages, a stage read-back. */
static int ensure_S(sgm_instance* s)
{
    const size_t need = (size_t)s->g.B * s->g.W * s->g.H * s->g.Dp * 2;
    if (s->d_S && need <= s->cap_S) return 0;
    sync_streams(s);
    sgmd_free(s->device, s->d_S);
    s->d_S = NULL;
    s->cap_S = 0;
    int rc = sgmd_alloc(s->device, &s->d_S, need);
    if (rc == 0) rc = sgmd_memset_async(s->device, s->stream, s->d_S, 0, need);
    if (rc == 0 && s->sum_stream) rc = sgmd_stream_sync(s->device, s->stream);   /* the cost sum may run on another stream */
    if (rc == 0) s->cap_S = need;
    else fprintf(stderr, "sgm_mi355x: device allocation failed for the aggregated-cost volume (%zu bytes)\n", need);
    return rc;
}

static int ensure_cost(sgm_instance* s)
{
    const size_t need = (size_t)s->g.B * s->g.W * s->g.H * s->g.Dp;
    if (s->d_cost && need <= s->cap_cost) return 0;
    sync_streams(s);
    sgmd_free(s->device, s->d_cost);
    s->d_cost = NULL;
    s->cap_cost = 0;
    const int rc = sgmd_alloc(s->device, &s->d_cost, need);
    if (rc == 0) s->cap_cost = need;
    else fprintf(stderr, "sgm_mi355x: device allocation failed for the cost volume (%zu bytes)\n", need);
    return rc;
}

bool sgm_initialize(sgm_instance* s, uint16_t width, uint16_t height, const SGMOption* option)
{
    if (!s || !option) return false;
    s->initialized = false;
    if (s->async_pending && !sgm_match_wait(s)) return false;    /* its buffers may be re-sized below */
    s->opt = *option;                                            /* SemiGlobalMatching.c:41 */
    if (width == 0 || height == 0) return false;                 /* .c:43 */
    if (option->max_disparity <= option->min_disparity) return false;   /* .c:46 */
    const int D = (uint16_t)(option->max_disparity - option->min_disparity);    /* .c:49 */
    if (D > SGM_MAX_DISPARITY_RANGE) FAIL("disparity range %d exceeds SGM_MAX_DISPARITY_RANGE=%d", D, SGM_MAX_DISPARITY_RANGE);
    if ((long long)width * height > 0x7FFFFFFFLL) FAIL("image too large (width*height must fit in 31 bits)");

    s->g.W = width; s->g.H = height; s->g.D = D;
    s->g.DPL = pick_dpl(D);
    s->g.LPP = 16;
    s->g.Dp = 16 * s->g.DPL;
    /* a batch of frames is VALU-bound: 8 lanes per pixel (twice the disparities per lane, same Dp) spends the
     * fewest instructions per cell; a single frame keeps 16 lanes per pixel for the shorter serial step (measured
     * at KITTI size, one frame: 16 lanes + 32-lane horizontals 0.35 ms, 8 lanes + 32-lane horizontals 0.39 ms, 8 lanes 0.59 ms) */
    {
        const int want = s->env_lanes >= 0 ? s->env_lanes : (s->batch >= 2 ? 8 : 16);      /* SGM_LANES_PER_PIXEL */
        /* negative P1 (defined by the reference's C arithmetic, covered by the parity tests, used by nobody) runs the
         * generic aggregation step, which only exists for 16 lanes per pixel */
        /* the wide census windows feed the aggregation from a cost volume: generic step, 16 lanes per pixel */
        if (want == 8 && option->p1 >= 0 && !s->census_w && s->g.DPL >= 2 && s->g.DPL <= 8 && s->g.DPL != 6) { s->g.LPP = 8; s->g.DPL *= 2; }
    }
    /* one frame per launch: the horizontal lines (W-1 serial steps) are the longest chains of the launch -> spread each
     * pixel of those over 32 lanes (2 lines per wave).  64 lanes (SGM_HL=64, one line per wave) measures the same at
     * KITTI size (0.351 vs 0.349 ms): with one frame the launch is then bound by total VALU issue at ~2 waves per SIMD.
     * Batches with 8 lanes per pixel: 16 lanes on the horizontal lines (4 lines per wave: a third fewer instructions per step of
     * the launch's longest chains for 4 % more instructions in total) -- KITTI 3850 -> 4010 fps through host pointers, aggregation
     * 1.19 -> 1.16 ms per 8 frames alone; 32 and 64 lanes cost more than they shorten (round 2) */
    {
        const int ok64 = (s->g.Dp % 64 == 0) && (s->g.Dp / 64 == 2 || s->g.Dp / 64 == 4 || s->g.Dp / 64 == 8);
        const int ok32 = (s->g.Dp % 32 == 0) && (s->g.Dp / 32 == 2 || s->g.Dp / 32 == 4 || s->g.Dp / 32 == 8 || s->g.Dp / 32 == 16);
        int want = s->env_hl >= 0 ? s->env_hl : ((s->batch == 1 || s->g.LPP == 16) ? 32 : 16);   /* SGM_HL; */   /* D > 128 in batches (16 lanes elsewhere): 32, +2 % at 2880x1988 D=256 */
        const int ok16 = s->g.LPP == 8 && (s->g.Dp / 16 == 2 || s->g.Dp / 16 == 4 || s->g.Dp / 16 == 8 || s->g.Dp / 16 == 16);
        if (want == 64 && !ok64) want = 32;
        if (want == 32 && !ok32) want = 0;
        if (want == 16 && !ok16) want = 0;
        if (want == s->g.LPP || option->p1 < 0 || s->census_w) want = 0;
        s->g.HL = want;
    }
    s->g.dmin = option->min_disparity;
    s->g.B = s->batch;
    s->g.row_begin = 0; s->g.row_end = height;
    if (s->tile_end != 0) {
        if (s->tile_begin < 0 || s->tile_begin >= s->tile_end || s->tile_end > height)
            FAIL("row tile [%d,%d) does not fit a frame of %d rows", s->tile_begin, s->tile_end, height);
        s->g.row_begin = s->tile_begin; s->g.row_end = s->tile_end;
    }
    s->tile_left = NULL;
    if (s->read_frame >= s->batch) s->read_frame = 0;
    if ((unsigned long long)width * height * (unsigned)s->g.Dp >= 0xFFFFFFFFull)
        FAIL("cost volume too large: width*height*%d must stay below 2^32 cells (32-bit offsets in the kernels)", s->g.Dp);

    s->paths.ndirs = (s->honor_num_paths && option->num_paths == 4) ? 4 : 8;    /* Q1 */
    s->paths.p1 = option->p1;
    {
        uint16_t lut[256];
        build_p2_table(option->p1, option->p2_init, lut);
        s->paths.pen_max = 0;
        for (int a = 0; a < 256; ++a) if (lut[a] > s->paths.pen_max) s->paths.pen_max = lut[a];
        s->paths.allow_fast = s->env_agg_fast >= 0 ? s->env_agg_fast != 0 : 1;   /* SGM_AGG_FAST=0: keep the plain non-negative-P1 step (parity tests run both) */
    }
    for (int d = 0; d < 8; ++d) {
        s->paths.dx[d] = k_dir_dx[d];
        s->paths.dy[d] = k_dir_dy[d];
        s->paths.anom_line[d] = (d >= 4) ? anomalous_line(width, k_dir_dx[d]) : -1;
    }
    s->paths.ghost_zero = (width >= height);
    s->paths.dir_mask = 0xFF;
    s->paths.run_anom = 1;
    s->need_plane_memset = !s->paths.ghost_zero;

    if (!ensure_buffers(s)) return false;
    /* extras: 4 anomalous lines x H steps x Dp bytes */
    const size_t extras_bytes = (size_t)s->g.B * 4 * height * s->g.Dp;
    if (extras_bytes > s->cap_extras || !s->d_extras) {
        sync_streams(s);
        sgmd_free(s->device, s->d_extras);
        s->d_extras = NULL;
        if (sgmd_alloc(s->device, &s->d_extras, extras_bytes) != 0) FAIL("device allocation failed (extras)");
        s->cap_extras = extras_bytes;
    }
    /* median scratch depends on W and H separately (64-row groups x time slots) */
    const size_t median_bytes = sgmd_median_scratch_bytes(&s->g);
    if (median_bytes > s->cap_median || !s->d_median_scratch) {
        sync_streams(s);
        sgmd_free(s->device, s->d_median_scratch);
        s->d_median_scratch = NULL;
        if (sgmd_alloc(s->device, &s->d_median_scratch, median_bytes) != 0) FAIL("device allocation failed (median scratch)");
        /* the granule rows between the bands of a tall frame carry a generation tag: start from "never written" */
        if (sgmd_memset_async(s->device, s->stream, s->d_median_scratch, 0, median_bytes) != 0) FAIL("clearing the median scratch failed");
        s->cap_median = median_bytes;
    }
    /* a Reset with unchanged shape and penalties (the per-frame case, Q14) re-uploads nothing */
    if (s->tab_W != width || s->tab_H != height || s->tab_ndirs != s->paths.ndirs || s->tab_p1 != option->p1 ||
        s->tab_p2 != option->p2_init) {
        if (!upload_tables(s)) return false;
        s->tab_W = width; s->tab_H = height; s->tab_ndirs = s->paths.ndirs;
        s->tab_p1 = option->p1; s->tab_p2 = option->p2_init;
    }

    if (s->tile_end != 0 && !s->census_w && !upload_census_need(s)) return false;

    s->s_is_zero = true;                                         /* .c:57: memset of cost_aggr, done lazily */
    s->s_pending = false;
    {
        /* one workgroup per image row segment (the launcher cuts rows into up to 4 segments when a launch has few rows);
         * at KITTI size: a batch of 8 frames 0.093 ms per frame against 0.115 + 0.043 for the two separate kernels, a
         * single frame 0.144 against 0.163 */
        const int want = s->env_fused >= 0 ? s->env_fused != 0 : 1;        /* SGM_FUSED_WTA */
        s->fused_wta = sgmd_sum_wta_lr_supported(&s->g, s->row_cap) && want;
    }
    /* the fused last sweep: batches of whole frames with eight paths and non-negative P1 on the census path (sgm_upsum.hip has
     * the shapes: W > H, Dp = 128).  SGM_UPSUM=0 / 1 forces it off / on (also for one frame per launch, where its row-to-row chain
     * costs latency) */
    s->up_rows = 0;
    s->planes_partial = false;
    if (s->fused_wta && s->tile_end == 0 && !s->census_w && s->paths.ndirs == 8 && option->p1 >= 0 && s->row_cap <= 8 &&
        (s->env_upsum >= 0 ? s->env_upsum != 0 : UPSUM_DEFAULT && s->batch >= 2))
        s->up_rows = sgmd_upsum_rows(&s->g);
    if (s->up_rows > 0 && s->env_upsum_rows >= 1 && s->env_upsum_rows < s->up_rows) s->up_rows = s->env_upsum_rows;
    s->have_ms = false;
    s->initialized = true;
    return true;
}

bool sgm_reset(sgm_instance* s, uint16_t width, uint16_t height, const SGMOption* option)
{
    if (!s) return false;
    s->initialized = false;                                      /* .c:130 */
    return sgm_initialize(s, width, height, option);
}

static void mark_on(sgm_instance* s, void* stream, int idx)
{
    if (s->timing && s->timer) sgmd_timer_mark(s->device, s->timer, stream, s->ring_next * MARKS_PER_MATCH + idx);
}
static void mark(sgm_instance* s, int idx) { mark_on(s, s->stream, idx); }

static int sweep_mask(const sgm_instance* s, int forward);
static int launch_aggregation(sgm_instance* s, const sgmd_paths* paths, const void* d_left);

/* d_S <- [d_S +] sum of the planes of the last frame, if the fused kernel skipped that store */
static int materialize_S(sgm_instance* s)
{
    if (!s->s_pending) return 0;
    if (ensure_S(s) != 0) return -1;
    /* d_S may still be in use by a cost sum on its own stream; the scratch map below is the speckle pass's label map: a post
     * pass still running on its own stream comes first */
    if (s->sum_pending && sgmd_stream_wait_event(s->device, s->stream, s->ev_sum) != 0) return -1;
    if (s->post_pending && sgmd_stream_wait_event(s->device, s->stream, s->ev_post) != 0) return -1;
    if (s->planes_partial) {
        /* the last match ran the fused sweep: the three upward planes do not exist.  Walk those directions now (the census images
         * and the kept copy of the left image are still that frame's; the anomalous lines and their cells were done then) */
        sgmd_paths p = s->paths;
        p.dir_mask = sweep_mask(s, 0);
        p.run_anom = 0;
        p.up_fused = 0;
        if (launch_aggregation(s, &p, s->d_left_keep) != 0) return -1;
        s->planes_partial = false;
    }
    /* the left-view WTA this kernel also produces goes to a dead scratch map (speckle labels) */
    const int rc = sgmd_sum_wta(s->device, s->stream, &s->g, s->paths.ndirs, s->d_planes, s->plane_bytes, s->d_extras,
                                s->d_row_extras, s->d_row_count, s->row_cap, s->s_pending_accumulate ? 1 : 0, s->d_S, 0, 0.0f,
                                s->d_labels);
    if (rc == 0) s->s_pending = false;                           /* a failed launch leaves the sum pending */
    return rc;
}

/* .c:94 (sum over the directions), .c:99 and .c:105 (both ComputeDisparity calls).  The Q14 bookkeeping (s_is_zero,
 * s_pending) changes only when every launch of the stage was accepted. */
static int sum_and_wta(sgm_instance* s, void* st, void* d_out, bool with_marks)
{
    const SGMOption* o = &s->opt;
    const int accumulate = s->s_is_zero ? 0 : 1;                 /* Q14 */
    const int uniq = o->is_check_unique ? 1 : 0;
    const float keep = 1 - o->uniqueness_ratio;
    int rc;
    if ((!s->fused_wta || accumulate || s->keep_stages) && ensure_S(s) != 0) return -1;
    if (s->fused_wta) {
        const int store = s->keep_stages ? 1 : 0;
        rc = sgmd_sum_wta_lr(s->device, st, &s->g, s->paths.ndirs, s->d_planes, s->plane_bytes, s->d_extras,
                             s->d_row_extras, s->d_row_count, s->row_cap, accumulate, store, (o->is_check_lr || s->reference_view) ? 1 : 0, s->d_S,
                             uniq, keep, d_out, s->d_disp_r);
        if (rc != 0) return rc;
        s->s_pending = !store;
        s->s_pending_accumulate = accumulate != 0;
        if (with_marks) mark_on(s, st, 4);
    } else {
        rc = sgmd_sum_wta(s->device, st, &s->g, s->paths.ndirs, s->d_planes, s->plane_bytes, s->d_extras,
                          s->d_row_extras, s->d_row_count, s->row_cap, accumulate, s->d_S, uniq, keep, d_out);
        if (rc != 0) return rc;
        /* d_S now holds this frame's sum whatever happens next */
        s->s_pending = false;
        s->s_is_zero = false;
        if (with_marks) mark_on(s, st, 4);
        if (o->is_check_lr || s->reference_view) rc = sgmd_wta_right(s->device, st, &s->g, s->d_S, uniq, keep, s->d_disp_r);
        if (rc != 0) return rc;
    }
    s->s_is_zero = false;
    return 0;
}

/* .c:82-83 (+ .c:89 for the wide census windows, whose cost is materialised): census of both images */
static int prepare_costs(sgm_instance* s, const void* d_left, const void* d_right)
{
    if (!s->census_w) {
        const bool tiled = s->tile_end != 0 && !s->keep_stages;          /* stage read-back wants the whole census */
        if (tiled && getenv("SGM_DEBUG_POISON_CENSUS")) {                /* tests: a read of a skipped block must not go unnoticed */
            const size_t bytes = (size_t)s->g.B * s->g.W * s->g.H * 4;
            if (sgmd_memset_async(s->device, s->stream, s->d_census_l, 0xA5, bytes) != 0 ||
                sgmd_memset_async(s->device, s->stream, s->d_census_r, 0x5A, bytes) != 0) return -1;
        }
        /* the reference's own boundary, one whole frame per match: the unwritten census words stay as they are (Q3) */
        const int keep_border = s->reference_statics && s->g.B == 1 && s->tile_end == 0;
        return sgmd_census(s->device, s->stream, &s->g, d_left, d_right, s->d_census_l, s->d_census_r, tiled ? s->d_census_need : NULL,
                           keep_border);
    }
    const size_t need = (size_t)s->g.B * s->g.W * s->g.H * 8;
    if (need > s->cap_census64 || !s->d_census64_l) {
        sync_streams(s);
        sgmd_free(s->device, s->d_census64_l);
        sgmd_free(s->device, s->d_census64_r);
        s->d_census64_l = s->d_census64_r = NULL;
        s->cap_census64 = 0;
        if (sgmd_alloc(s->device, &s->d_census64_l, need) != 0 || sgmd_alloc(s->device, &s->d_census64_r, need) != 0) return -1;
        s->cap_census64 = need;
    }
    int rc = ensure_cost(s);
    if (rc == 0) rc = sgmd_census_window(s->device, s->stream, &s->g, s->census_w, s->census_h, d_left, d_right, s->d_census64_l,
                                         s->d_census64_r);
    if (rc == 0) rc = sgmd_cost64(s->device, s->stream, &s->g, s->d_census64_l, s->d_census64_r, s->d_cost);
    return rc;
}

/* .c:94: the path aggregation, from the census images or (wide windows) from the cost volume */
static int launch_aggregation(sgm_instance* s, const sgmd_paths* paths, const void* d_left)
{
    if (s->census_w)
        return sgmd_aggregate_volume(s->device, s->stream, &s->g, paths, d_left, s->d_cost, s->d_lut, s->d_planes, s->plane_bytes,
                                     s->d_extras);
    return sgmd_aggregate(s->device, s->stream, &s->g, paths, d_left, s->d_census_l, s->d_census_r, s->d_lut, s->d_planes,
                          s->plane_bytes, s->d_extras);
}

/* .c:109 LRCheck on the left map -- or, with the right view as the reference view (extension), the mirrored check on the
 * right map, whose result replaces the left map in d_out */
static int lr_stage(sgm_instance* s, void* st, void* d_out)
{
    const SGMOption* o = &s->opt;
    if (!s->reference_view) return o->is_check_lr ? sgmd_lrcheck(s->device, st, &s->g, d_out, s->d_disp_r, o->lrcheck_thres) : 0;
    /* the rows this instance computes: all rows of all frames of the batch, or its row tile of each of them */
    int rc = sgmd_lrcheck_right(s->device, st, &s->g, s->d_disp_r, d_out, o->lrcheck_thres, o->is_check_lr ? 1 : 0, s->d_labels);
    const size_t frame = (size_t)s->g.W * s->g.H * sizeof(float), first = (size_t)s->g.row_begin * s->g.W * sizeof(float);
    const size_t rows = (size_t)(s->g.row_end - s->g.row_begin) * s->g.W * sizeof(float);
    if (rows == frame)                                        /* d_labels: scratch until the speckle pass */
        return rc ? rc : sgmd_d2d_async(s->device, st, d_out, s->d_labels, frame * s->g.B);
    for (int f = 0; rc == 0 && f < s->g.B; ++f)
        rc = sgmd_d2d_async(s->device, st, (char*)d_out + f * frame + first, (char*)s->d_labels + f * frame + first, rows);
    return rc;
}

/* The body of SGM_Match (SemiGlobalMatching.c:80-122) on device buffers.  The first launch that is refused ends the
 * match: nothing further is queued and false is returned with the instance in a consistent state -- d_S (or the
 * pending planes) still describe exactly the matches that completed, so a later Match without Reset (Q14) accumulates
 * onto the right thing. */
#define LAUNCH(expr) do { if ((expr) != 0) goto failed; } while (0)
/* scratch of the fused last sweep (zero when allocated: its progress words start below every generation) and the kept left image */
static int ensure_upsum(sgm_instance* s)
{
    const size_t need = sgmd_upsum_scratch_bytes(&s->g), px = (size_t)s->g.B * s->g.W * s->g.H;
    if (need > s->cap_up_scratch || !s->d_up_scratch) {
        sync_streams(s);
        sgmd_free(s->device, s->d_up_scratch);
        s->d_up_scratch = NULL; s->cap_up_scratch = 0;
        if (sgmd_alloc(s->device, &s->d_up_scratch, need) != 0) return -1;
        if (sgmd_memset_async(s->device, s->stream, s->d_up_scratch, 0, need) != 0) return -1;
        s->cap_up_scratch = need;
    }
    if (px > s->cap_left_keep || !s->d_left_keep) {
        sync_streams(s);
        sgmd_free(s->device, s->d_left_keep);
        s->d_left_keep = NULL; s->cap_left_keep = 0;
        if (sgmd_alloc(s->device, &s->d_left_keep, px) != 0) return -1;
        s->cap_left_keep = px;
    }
    return 0;
}

static bool run_pipeline(sgm_instance* s, const void* d_left, const void* d_right, void* d_out)
{
    const int dev = s->device;
    void* st = s->stream;
    const sgmd_geom* g = &s->g;
    const SGMOption* o = &s->opt;
    const size_t px_bytes = (size_t)g->B * g->W * g->H * sizeof(float);

    /* stage groups on streams of their own (sgm_set_stage_cus / sgm_set_overlap_post; never in row-tile mode) */
    const bool own_sum = s->sum_stream && s->tile_end == 0;
    const bool overlap = s->overlap_post && s->post_stream && s->tile_end == 0;
    void *sts = st, *st2 = st;
    /* the aggregation rewrites the planes the previous match's cost sum may still be reading on its own stream */
    if (s->sum_pending) LAUNCH(sgmd_stream_wait_event(dev, st, s->ev_sum));
    if (!s->s_is_zero) LAUNCH(materialize_S(s));             /* Match without Reset: S of the previous frame is needed now */
    mark(s, 0);
    LAUNCH(prepare_costs(s, d_left, d_right));                                                      /* .c:82-83 */
    mark(s, 1);
    /* .c:89: the cost volume is recomputed inside the aggregation kernel; it is only materialised when a
     * test wants to read it back (stage 2) */
    if (s->keep_stages && !s->census_w) {
        LAUNCH(ensure_cost(s));
        LAUNCH(sgmd_cost(dev, st, g, s->d_census_l, s->d_census_r, s->d_cost));
    }
    mark(s, 2);
    if (s->need_plane_memset && s->paths.ndirs > 4)
        for (int f = 0; f < g->B; ++f)
            LAUNCH(sgmd_memset_async(dev, st, (char*)s->d_planes_alloc + ((size_t)f * 8 + 4) * s->plane_bytes, 0, 4 * s->plane_bytes));
    /* the last vertical sweep fused with the cost sum (sgmd_upsum): whenever this match neither adds to an earlier S (Q14) nor has
     * to leave S behind for a test */
    const bool use_up = s->up_rows > 0 && !s->keep_stages && s->s_is_zero && s->fused_wta;
    s->last_up_rows = use_up ? s->up_rows : 0;
    if (use_up) {
        LAUNCH(ensure_upsum(s));
        LAUNCH(sgmd_d2d_async(dev, st, s->d_left_keep, d_left, (size_t)g->B * g->W * g->H));
        sgmd_paths p = s->paths;
        p.up_fused = 1;
        LAUNCH(launch_aggregation(s, &p, d_left));
    } else
        LAUNCH(launch_aggregation(s, &s->paths, d_left));                                           /* .c:94 */
    mark(s, 3);
    if (own_sum) {
        LAUNCH(sgmd_event_record(dev, s->ev_agg, st));
        LAUNCH(sgmd_stream_wait_event(dev, s->sum_stream, s->ev_agg));
        sts = st2 = s->sum_stream;
    }
    /* the cost sum writes d_out and the right-view map, which the previous match's post pass may still be reading */
    if (s->post_pending) LAUNCH(sgmd_stream_wait_event(dev, sts, s->ev_post));
    mark_on(s, sts, M_SUM_BEGIN);
    if (use_up) {
        LAUNCH(sgmd_upsum(dev, sts, g, &s->paths, s->d_left_keep, s->d_census_l, s->d_census_r, s->d_lut, s->d_planes, s->plane_bytes, s->d_extras,
                          s->d_row_extras, s->d_row_count, s->row_cap, (o->is_check_lr || s->reference_view) ? 1 : 0, o->is_check_unique ? 1 : 0,
                          1 - o->uniqueness_ratio, s->d_up_scratch, ++s->up_gen, s->h_status, s->up_rows, s->env_upsum_wgs > 0 ? s->env_upsum_wgs : 0, d_out, s->d_disp_r));
        s->planes_partial = true;                                /* S of this frame = five planes + what materialize_S re-creates */
        s->s_pending = true;
        s->s_pending_accumulate = false;
        s->s_is_zero = false;
        mark_on(s, sts, 4);
    } else
        LAUNCH(sum_and_wta(s, sts, d_out, true));                                                   /* .c:94 sum, .c:99, .c:105 */
    if (s->keep_stages) LAUNCH(sgmd_d2d_async(dev, sts, s->d_snap_wta, d_out, px_bytes));
    mark_on(s, sts, 5);
    /* the post pass (latency-bound kernels that fill a fraction of the GPU) on its own stream, so that the stream(s) before it
     * can start the next match's census, aggregation and cost sum beside it */
    if (overlap || own_sum) LAUNCH(sgmd_event_record(dev, s->ev_sum, sts));
    if (own_sum) s->sum_pending = true;
    if (overlap) {
        LAUNCH(sgmd_stream_wait_event(dev, s->post_stream, s->ev_sum));
        st2 = s->post_stream;
        s->post_pending = true;                                  /* from here on the post stream has work of this match */
    }
    LAUNCH(lr_stage(s, st2, d_out));                                                                /* .c:109 */
    if (s->keep_stages) LAUNCH(sgmd_d2d_async(dev, st2, s->d_snap_lr, d_out, px_bytes));
    mark_on(s, st2, 6);
    if (o->is_remove_speckles)                                                                      /* .c:115 */
        LAUNCH(sgmd_speckle(dev, st2, g, d_out, 1.0f, o->min_speckle_area, s->d_labels, s->d_sizes, s->d_totals));
    if (s->keep_stages) LAUNCH(sgmd_d2d_async(dev, st2, s->d_snap_speckle, d_out, px_bytes));
    mark_on(s, st2, 7);
    LAUNCH(sgmd_median(dev, st2, g, d_out, s->d_median_scratch, s->h_status));                                   /* .c:120 */
    mark_on(s, st2, 8);
    if (overlap) LAUNCH(sgmd_event_record(dev, s->ev_post, st2));
    else if (own_sum) LAUNCH(sgmd_event_record(dev, s->ev_sum, st2));   /* the post pass ran on the sum stream: "sum done" = all of it */
    s->tail_stream = st2;
    if (s->timing && s->timer) {
        s->ring_next = (s->ring_next + 1) % TIMING_RING;
        if (s->ring_pending < TIMING_RING) ++s->ring_pending;      /* older sets are overwritten */
    }
    return true;
failed:
    /* this match's timing set is incomplete: it is recorded over by the next match (ring_next did not advance).  Work may sit
     * on the sum / post stream without the event a later match would wait for: drain everything, so that nothing of the
     * abandoned match is still running when its buffers are reused */
    if (s->sum_stream || s->post_stream) sync_streams(s);
    s->tail_stream = st;
    FAIL("a kernel launch failed; the match was abandoned");
}

/* ------------------------------------------------------------------ row tiles (one frame over several GPUs)
 *
 * The instance computes rows [row_begin,row_end) of aggregation, cost sum, both WTAs and the LR check.  The
 * vertical and diagonal paths cross tile borders, so a tile's sweep in one vertical sense starts from the path
 * costs of the row just outside the tile -- the neighbouring GPU's last row of that sweep -- which the caller
 * moves between GPUs (sgm_tile_export_boundary -> RCCL send/recv or a peer copy -> sgm_tile_import_boundary).
 * Census and the four anomalous diagonal lines are computed on the whole frame by every GPU (the images are
 * replicated: 2 x W*H bytes; that work is ~1 % of a frame).  Speckle removal and the median are whole-frame
 * passes over the gathered W*H disparity map (sgm_tile_post). */

static int sweep_mask(const sgm_instance* s, int forward)
{
    int m = 0;
    for (int d = 0; d < s->paths.ndirs; ++d)
        if (s->paths.dy[d] == (forward ? 1 : -1)) m |= 1 << d;
    return m;
}

bool sgm_set_rows(sgm_instance* s, int row_begin, int row_end)
{
    if (!s || row_begin < 0 || row_end < 0 || (row_end != 0 && row_begin >= row_end)) return false;
    s->tile_begin = row_begin;
    s->tile_end = row_end;
    s->initialized = false;      /* takes effect at the next sgm_initialize / sgm_reset */
    return true;
}

static bool tile_aggregate(sgm_instance* s, int dir_mask, int run_anom)
{
    sgmd_paths p = s->paths;
    p.dir_mask = dir_mask;
    p.run_anom = run_anom;
    return launch_aggregation(s, &p, s->tile_left) == 0;
}

bool sgm_tile_begin(sgm_instance* s, const uint8_t* d_left, const uint8_t* d_right)
{
    if (!s || !s->initialized || !d_left || !d_right) return false;
    int rc = s->s_is_zero ? 0 : materialize_S(s);
    if (rc == 0) rc = prepare_costs(s, d_left, d_right);
    if (s->need_plane_memset && s->paths.ndirs > 4)
        for (int f = 0; rc == 0 && f < s->g.B; ++f)
            rc = sgmd_memset_async(s->device, s->stream, (char*)s->d_planes_alloc + ((size_t)f * 8 + 4) * s->plane_bytes, 0, 4 * s->plane_bytes);
    if (rc != 0) FAIL("a kernel launch failed");
    s->tile_left = d_left;
    int hmask = 0;
    for (int d = 0; d < s->paths.ndirs; ++d)
        if (s->paths.dy[d] == 0) hmask |= 1 << d;
    return tile_aggregate(s, hmask, 1);
}

size_t sgm_tile_boundary_bytes(const sgm_instance* s)
{
    if (!s || !s->initialized) return 0;
    return (size_t)s->g.B * (s->paths.ndirs > 4 ? 3 : 1) * s->g.W * s->g.Dp;     /* [frame of the batch][direction of the sweep][W][Dp] */
}

/* rows of the planes a sweep hands over: `inside` = the tile's last row in walking order, else the row just past
 * the tile's first row against the walking order (where the neighbour's hand-over lands) */
static bool boundary_copy(sgm_instance* s, int forward, void* d_buf, bool do_export)
{
    if (!s || !s->initialized || !d_buf) return false;
    const int row = do_export ? (forward ? s->g.row_end - 1 : s->g.row_begin)
                              : (forward ? s->g.row_begin - 1 : s->g.row_end);
    if (row < 0 || row >= s->g.H) return false;               /* the tile touches the frame edge: nothing to import */
    const size_t row_bytes = (size_t)s->g.W * s->g.Dp;
    const int mask = sweep_mask(s, forward);
    int dirs[8], n = 0;
    for (int d = 0; d < s->paths.ndirs; ++d)
        if ((mask >> d) & 1) dirs[n++] = d;
    /* one launch for all frames and directions: [frame][direction of the sweep][W][Dp] in the buffer */
    return sgmd_plane_rows_copy(s->device, s->stream, s->d_planes, s->plane_bytes, (size_t)row * row_bytes, row_bytes, dirs, n, s->g.B,
                                d_buf, do_export ? 1 : 0) == 0;
}

bool sgm_tile_export_boundary(sgm_instance* s, int forward, void* d_buf) { return boundary_copy(s, forward, d_buf, true); }
bool sgm_tile_import_boundary(sgm_instance* s, int forward, const void* d_buf)
{
    return boundary_copy(s, forward, (void*)d_buf, false);
}

bool sgm_tile_sweep(sgm_instance* s, int forward)
{
    if (!s || !s->initialized || !s->tile_left) return false;
    return tile_aggregate(s, sweep_mask(s, forward), 0);
}

bool sgm_tile_finish(sgm_instance* s, float* d_disp_left)
{
    if (!s || !s->initialized || !s->tile_left || !d_disp_left) return false;
    int rc = sum_and_wta(s, s->stream, d_disp_left, false);
    if (rc == 0) rc = lr_stage(s, s->stream, d_disp_left);
    s->tile_left = NULL;
    if (rc != 0) FAIL("a kernel launch failed");
    return true;
}

bool sgm_tile_post(sgm_instance* s, float* d_disp_left)
{
    if (!s || !s->initialized || !d_disp_left) return false;
    int rc = 0;
    if (s->opt.is_remove_speckles)
        rc = sgmd_speckle(s->device, s->stream, &s->g, d_disp_left, 1.0f, s->opt.min_speckle_area, s->d_labels, s->d_sizes,
                          s->d_totals);
    if (rc == 0) rc = sgmd_median(s->device, s->stream, &s->g, d_disp_left, s->d_median_scratch, s->h_status);
    if (rc != 0) FAIL("a kernel launch failed");
    return true;
}

/* the stream the instance's last match finishes on (the post-pass or cost-sum stream when those stages have their own) */
static void* result_stream(sgm_instance* s) { return s->tail_stream ? s->tail_stream : s->stream; }

/* the event later matches wait for before they reuse what the tail of the last match works on */
static void* result_event(sgm_instance* s)
{
    void* st = result_stream(s);
    if (st == s->stream) return NULL;                            /* stream order does it */
    return st == s->post_stream ? s->ev_post : s->ev_sum;
}

/* more work on the result of the last match, queued behind it on the stream it finished on: re-record that stream's "done" event
 * so that the next match's waits cover it */
static int rerecord_result_event(sgm_instance* s)
{
    void* ev = result_event(s);
    return ev ? sgmd_event_record(s->device, ev, result_stream(s)) : 0;
}

/* make `stream` wait for the last match's result */
static int wait_for_result(sgm_instance* s, void* stream)
{
    void* ev = result_event(s);
    if (!ev || stream == result_stream(s)) return 0;
    return sgmd_stream_wait_event(s->device, stream, ev);
}

/* D2H of a result behind the match that produced it.  No event is recorded behind the copy: every entry point that could
 * overwrite the copy's source (the instance's own d_disp / d_depth) starts with sgm_match_wait, i.e. after the copy has
 * completed -- and an event record queued behind a D2H costs the pipelined host-pointer path 3 % (measured: 3830 -> 3700 fps) */
static bool queue_result_copy(sgm_instance* s, void* host_dst, const void* d_src, size_t bytes)
{
    return sgmd_d2h_async(s->device, result_stream(s), host_dst, d_src, bytes) == 0;
}

static void collect_timing(sgm_instance* s)
{
    if (!(s->timing && s->timer)) return;
    /* every match recorded since the last collection (the stream is idle here), oldest first */
    while (s->ring_pending > 0) {
        const int set = ((s->ring_next - s->ring_pending) % TIMING_RING + TIMING_RING) % TIMING_RING;
        const int base = set * MARKS_PER_MATCH;
        --s->ring_pending;
        float ms[T_COUNT];
        bool ok = true;
        for (int i = 0; i < T_COUNT && ok; ++i)
            ok = sgmd_timer_elapsed(s->device, s->timer, base + (i == T_SUM ? M_SUM_BEGIN : i), base + i + 1, &ms[i]) == 0;
        if (!ok) continue;
        for (int i = 0; i < T_COUNT; ++i) {
            s->last_ms[i] = ms[i];
            s->sum_ms[i] += ms[i];
            if (ms[i] < s->min_ms[i]) s->min_ms[i] = ms[i];
        }
        ++s->n_timed;
        s->have_ms = true;
    }
}

bool sgm_match_device(sgm_instance* s, const uint8_t* d_left, const uint8_t* d_right, float* d_disp_left)
{
    if (!s || !s->initialized) return false;                     /* .c:70 */
    if (!d_left || !d_right) return false;                       /* .c:73 */
    if (!d_disp_left) return false;
    if (s->tile_end != 0) FAIL("the instance is in row-tile mode (sgm_set_rows): use the sgm_tile_* sequence");
    return run_pipeline(s, d_left, d_right, d_disp_left);
}

bool sgm_synchronize(sgm_instance* s)
{
    if (!s) return false;
    if (sync_streams(s) != 0) return false;
    collect_timing(s);
    return true;
}

/* Hands the result of a queued sgm_match_async to its caller: waits for the stream, copies the staged disparity map
 * to the caller's buffer (nothing to copy when that buffer was pinned: the device wrote it). */
bool sgm_match_wait(sgm_instance* s)
{
    if (!s) return false;
    if (!s->async_pending) return true;
    s->async_pending = false;
    size_t done = 0;
    const int chunks = s->async_chunks;
    s->async_chunks = 1;                                         /* whatever happens below, the next match starts clean */
    if (s->async_out && chunks > 1) {
        /* the pieces as they arrive; a map that turns out invalid (sgm_synchronize below) has been handed over in part, as a
         * failed SGM_Match leaves its output undefined */
        const size_t piece = s->async_bytes / (size_t)chunks / 4 * 4;
        for (int i = 0; i + 1 < chunks; ++i) {
            if (sgmd_event_sync(s->device, s->ev_chunk[i]) != 0) break;
            memcpy((char*)s->async_out + done, (const char*)s->h_disp + done, piece);
            done += piece;
        }
    }
    if (!sgm_synchronize(s)) return false;
    if (s->async_out) memcpy((char*)s->async_out + done, (const char*)s->h_disp + done, s->async_bytes - done);   /* .c:122 */
    s->async_out = NULL;
    s->async_chunks = 1;
    return true;
}

/* The host-pointer match without the final wait: stages the images (not at all when the caller's buffers are pinned,
 * sgm_host_alloc), queues H2D, the pipeline and D2H on the instance's stream and returns.  With a few instances
 * round-robined by the caller, the copies of one overlap the kernels of the others (separate DMA engines). */
bool sgm_match_async(sgm_instance* s, const uint8_t* img_left, const uint8_t* img_right, float* disp_left)
{
    if (!s || !s->initialized) return false;                     /* .c:70 */
    if (!img_left || !img_right) return false;                   /* .c:73 */
    if (!disp_left) return false;
    if (s->tile_end != 0) FAIL("the instance is in row-tile mode (sgm_set_rows): use the sgm_tile_* sequence");
    if (!sgm_match_wait(s)) return false;                        /* the staging buffers are free again */
    const size_t px = (size_t)s->g.B * s->g.W * s->g.H;           /* batch > 1: B consecutive frames */
    const void *src_l = img_left, *src_r = img_right;
    /* the left image is on the bus while the right one is staged */
    if (!sgmd_host_is_pinned(s->device, img_left, px)) { memcpy(s->h_left, img_left, px); src_l = s->h_left; }
    bool ok = sgmd_h2d_async(s->device, s->stream, s->d_left, src_l, px) == 0;
    if (!sgmd_host_is_pinned(s->device, img_right, px)) { memcpy(s->h_right, img_right, px); src_r = s->h_right; }
    const bool out_pinned = sgmd_host_is_pinned(s->device, disp_left, px * sizeof(float)) != 0;
    const size_t bytes = px * sizeof(float);
    ok = ok && sgmd_h2d_async(s->device, s->stream, s->d_right, src_r, px) == 0 &&
         run_pipeline(s, s->d_left, s->d_right, s->d_disp);
    int chunks = 1;
    if (ok && !out_pinned && bytes >= RESULT_CHUNK_MIN) {        /* a single frame: 0.92 -> 0.88 ms per blocking call; batches of 8 through
                                                                   four pipelined instances on pageable buffers: 3500 -> 3640 fps */
        /* two pieces for a map of a few MB (an event and a wait per piece: 1.86 MB in 2 / 4 / 8 pieces = 0.708 / 0.733 / 0.79 ms per
         * blocking KITTI frame), RESULT_CHUNKS for more (a batch of 8 such maps: 3520 / 3575 / 3590 fps pipelined) */
        chunks = bytes < RESULT_CHUNK_SPLIT ? 2 : RESULT_CHUNKS;
        for (int i = 0; ok && i < chunks - 1; ++i)
            if (!s->ev_chunk[i]) ok = sgmd_event_create(s->device, &s->ev_chunk[i]) == 0;
    }
    if (ok && chunks > 1) {
        const size_t piece = bytes / (size_t)chunks / 4 * 4;
        size_t off = 0;
        for (int i = 0; ok && i < chunks; ++i) {
            const size_t n = i + 1 < chunks ? piece : bytes - off;
            ok = queue_result_copy(s, (char*)s->h_disp + off, (const char*)s->d_disp + off, n) &&
                 (i + 1 == chunks || sgmd_event_record(s->device, s->ev_chunk[i], result_stream(s)) == 0);
            off += n;
        }
    } else if (ok)
        ok = queue_result_copy(s, out_pinned ? (void*)disp_left : s->h_disp, s->d_disp, bytes);
    if (!ok) {
        sync_streams(s);                  /* queued copies may still read the caller's / staging buffers */
        return false;
    }
    s->async_pending = true;
    s->async_out = out_pinned ? NULL : disp_left;
    s->async_bytes = bytes;
    s->async_chunks = chunks;
    return true;
}

bool sgm_match(sgm_instance* s, const uint8_t* img_left, const uint8_t* img_right, float* disp_left)
{
    return sgm_match_async(s, img_left, img_right, disp_left) && sgm_match_wait(s);
}

void* sgm_host_alloc(sgm_instance* s, size_t bytes)
{
    void* p = NULL;
    if (!s || sgmd_alloc_pinned(s->device, &p, bytes) != 0) return NULL;
    return p;
}
void sgm_host_free(sgm_instance* s, void* p) { if (s && p) sgmd_free_pinned(s->device, p); }

/* ------------------------------------------------------------------ test-platform arithmetic on device buffers (8f-3) */

bool sgm_disparity_to_depth(sgm_instance* s, const float* d_disparity, size_t count, float fx, float baseline, float doffs,
                            float* d_depth)
{
    if (!s || !d_disparity || !d_depth) return false;
    /* the map may be the result of a match whose last stages run on a stream of their own (sgm_set_overlap_post / _stage_cus) */
    if (wait_for_result(s, s->stream) != 0) return false;
    return sgmd_depth(s->device, s->stream, d_disparity, count, fx, baseline, doffs, d_depth) == 0;
}

/* ------------------------------------------------------------------ a test-platform frame end to end (8f-2) */

bool sgm_gray_from_planes(sgm_instance* s, const uint8_t* d_bgr, size_t count, int weight_r, uint8_t* d_gray)
{
    if (!s || !d_bgr || !d_gray) return false;
    if (weight_r != 76 && weight_r != 77) FAIL("grey weight of red must be 76 (stereo_matching.c:18-25) or 77 (stb_image.h:1746-1749)");
    return sgmd_gray_planes(s->device, s->stream, d_bgr, count, weight_r, d_gray) == 0;
}

static int ensure_planes_io(sgm_instance* s)
{
    const size_t px = (size_t)s->g.B * s->g.W * s->g.H;
    if (px <= s->cap_bgr && s->d_bgr) return 0;
    sync_streams(s);
    sgmd_free(s->device, s->d_bgr);
    sgmd_free(s->device, s->d_depth);
    sgmd_free_pinned(s->device, s->h_bgr);
    s->d_bgr = s->d_depth = s->h_bgr = NULL;
    s->cap_bgr = 0;
    if (sgmd_alloc(s->device, &s->d_bgr, 6 * px) != 0 || sgmd_alloc(s->device, &s->d_depth, px * sizeof(float)) != 0 ||
        sgmd_alloc_pinned(s->device, &s->h_bgr, 6 * px) != 0)
        return -1;
    s->cap_bgr = px;
    return 0;
}

bool sgm_match_planes_async(sgm_instance* s, const uint8_t* planes, float fx, float baseline, float doffs, float* depth)
{
    if (!s || !s->initialized) return false;
    if (!planes || !depth) return false;
    if (s->tile_end != 0) FAIL("the instance is in row-tile mode (sgm_set_rows): use the sgm_tile_* sequence");
    if (!sgm_match_wait(s)) return false;                        /* the staging buffers are free again */
    if (ensure_planes_io(s) != 0) FAIL("device allocation failed for the colour planes of %dx%d", s->g.W, s->g.H);
    const int dev = s->device;
    const size_t fpx = (size_t)s->g.W * s->g.H, px = (size_t)s->g.B * fpx;
    const void* src = planes;
    if (!sgmd_host_is_pinned(dev, planes, 6 * px)) { memcpy(s->h_bgr, planes, 6 * px); src = s->h_bgr; }
    const bool out_pinned = sgmd_host_is_pinned(dev, depth, px * sizeof(float)) != 0;
    bool ok = sgmd_h2d_async(dev, s->stream, s->d_bgr, src, 6 * px) == 0;
    for (int f = 0; ok && f < s->g.B; ++f) {                    /* frame f: left B G R, right B G R (server.py:105-131) */
        const char* fr = (const char*)s->d_bgr + (size_t)f * 6 * fpx;
        ok = sgmd_gray_planes(dev, s->stream, fr, fpx, 76, (char*)s->d_left + f * fpx) == 0 &&
             sgmd_gray_planes(dev, s->stream, fr + 3 * fpx, fpx, 76, (char*)s->d_right + f * fpx) == 0;
    }
    ok = ok && run_pipeline(s, s->d_left, s->d_right, s->d_disp);
    void* st = result_stream(s);
    /* the depth conversion reads the map the next match's cost sum rewrites: "result done" moves behind it (not behind the copy) */
    ok = ok && sgmd_depth(dev, st, s->d_disp, px, fx, baseline, doffs, s->d_depth) == 0 && rerecord_result_event(s) == 0 &&
         queue_result_copy(s, out_pinned ? (void*)depth : s->h_disp, s->d_depth, px * sizeof(float));
    if (!ok) {
        sync_streams(s);
        return false;
    }
    s->async_pending = true;
    s->async_out = out_pinned ? NULL : depth;
    s->async_bytes = px * sizeof(float);
    s->async_chunks = 1;                                         /* one copy: no chunk events of an earlier match to wait for */
    return true;
}

bool sgm_match_planes(sgm_instance* s, const uint8_t* planes, float fx, float baseline, float doffs, float* depth)
{
    return sgm_match_planes_async(s, planes, fx, baseline, doffs, depth) && sgm_match_wait(s);
}

bool sgm_compare_depth(sgm_instance* s, const float* d_ground_truth, const float* d_test, size_t count, float abs_thresh,
                       double* rmse, double* bad_pixel_rate, uint64_t* n_valid)
{
    if (!s || !d_ground_truth || !d_test) return false;
    double sumsq = 0.0;
    unsigned long long n = 0, bad = 0;
    if (wait_for_result(s, s->stream) != 0) return false;
    if (sgmd_score(s->device, s->stream, d_ground_truth, d_test, count, abs_thresh, &sumsq, &n, &bad) != 0) return false;
    if (n_valid) *n_valid = n;
    /* depth_image.py:306-308: (nan, nan, 0) when no pixel is finite in both images */
    if (rmse) *rmse = n ? sqrt(sumsq / (double)n) : NAN;
    if (bad_pixel_rate) *bad_pixel_rate = n ? (double)bad / (double)n : NAN;
    return true;
}

/* ------------------------------------------------------------------ stage read-back */

static size_t compact_volume(const sgm_instance* s, const void* padded, size_t elem, void* out)
{
    const size_t px = (size_t)s->g.W * s->g.H;
    const char* src = (const char*)padded;
    char* dst = (char*)out;
    for (size_t p = 0; p < px; ++p)
        memcpy(dst + p * s->g.D * elem, src + p * s->g.Dp * elem, (size_t)s->g.D * elem);
    return px * s->g.D * elem;
}

size_t sgm_read_stage(sgm_instance* s, int which, void* host_out, size_t capacity)
{
    if (!s || !s->initialized || !host_out) return 0;
    const size_t px = (size_t)s->g.W * s->g.H;                    /* one frame */
    const size_t f = (size_t)s->read_frame;
    const char* src = NULL;
    size_t elem = 0;
    bool volume = false;
    int row_a = 0, row_b = s->g.H;                                /* rows the device holds of a volume stage */
    if ((which == 4 || which == 6 || which == 7 || (which == 2 && !s->census_w)) && !s->keep_stages) return 0;
    if (which == 2 && !s->d_cost) return 0;
    if (which == 3 && (ensure_S(s) != 0 || materialize_S(s) != 0)) return 0;
    switch (which) {
    case 0: src = s->census_w ? (const char*)s->d_census64_l + f * px * 8 : (const char*)s->d_census_l + f * px * 4; elem = s->census_w ? 8 : 4; break;
    case 1: src = s->census_w ? (const char*)s->d_census64_r + f * px * 8 : (const char*)s->d_census_r + f * px * 4; elem = s->census_w ? 8 : 4; break;
    case 2: src = (const char*)s->d_cost + f * px * s->g.Dp; elem = 1; volume = true; break;
    case 3: src = (const char*)s->d_S + f * px * s->g.Dp * 2; elem = 2; volume = true; break;
    case 4: src = (const char*)s->d_snap_wta + f * px * 4; elem = 4; break;
    case 5: src = (const char*)s->d_disp_r + f * px * 4; elem = 4; break;
    case 6: src = (const char*)s->d_snap_lr + f * px * 4; elem = 4; break;
    case 7: src = (const char*)s->d_snap_speckle + f * px * 4; elem = 4; break;
    case 8: src = (const char*)s->d_disp + f * px * 4; elem = 4; break;
    default:
        if (which >= 10 && which < 10 + s->paths.ndirs) {
            /* frame-addressed base of the plane; only rows [plane_row_lo, plane_row_lo + plane_rows) have storage */
            src = (const char*)s->d_planes + (f * 8 + (size_t)(which - 10)) * s->plane_bytes;
            elem = 1; volume = true;
            row_a = s->plane_row_lo; row_b = s->plane_row_lo + s->plane_rows;
        }
    }
    if (!src) return 0;
    const size_t need = volume ? px * s->g.D * elem : px * elem;
    if (capacity < need) return 0;
    if (sync_streams(s) != 0) return 0;
    if (!volume) {
        if (sgmd_d2h_async(s->device, s->stream, host_out, src, need) != 0) return 0;
        if (sync_streams(s) != 0) return 0;
        return need;
    }
    const size_t row_bytes = (size_t)s->g.W * s->g.Dp * elem;
    void* tmp = calloc(px * s->g.Dp, elem);                       /* rows without storage read 0 */
    if (!tmp) return 0;
    size_t got = 0;
    if (sgmd_d2h_async(s->device, s->stream, (char*)tmp + (size_t)row_a * row_bytes, src + (size_t)row_a * row_bytes,
                       (size_t)(row_b - row_a) * row_bytes) == 0 &&
        sync_streams(s) == 0)
        got = compact_volume(s, tmp, elem, host_out);
    free(tmp);
    return got;
}

/* ------------------------------------------------------------------ the reference boundary */

static sgm_instance* g_default;
static int g_default_device = -1;
static int g_default_honor;

static int default_device(void)
{
    if (g_default_device >= 0) return g_default_device;
    const char* e = getenv("SGM_DEVICE");
    return (e && *e) ? atoi(e) : 0;
}

bool SGM_SetDevice(int device_ordinal)
{
    if (device_ordinal < 0) return false;
    if (g_default && g_default->device != device_ordinal) {
        sgm_destroy(g_default);
        g_default = NULL;
    }
    g_default_device = device_ordinal;
    return true;
}

static int g_default_census_w, g_default_census_h, g_default_view;

bool SGM_SetCensusWindow(int width, int height)
{
    if (width < 1 || height < 1 || !(width & 1) || !(height & 1) || width * height > 64) return false;
    g_default_census_w = width; g_default_census_h = height;
    return g_default ? sgm_set_census_window(g_default, width, height) : true;
}

void SGM_SetReferenceView(int right)
{
    g_default_view = right ? 1 : 0;
    if (g_default) sgm_set_reference_view(g_default, right);
}

void SGM_SetHonorNumPaths(int honor)
{
    g_default_honor = honor;
    if (g_default) g_default->honor_num_paths = honor;
}

bool SGM_Initialize(uint16_t width, uint16_t height, const SGMOption* option)
{
    if (!option) return false;
    /* argument errors are reported exactly like the reference, before any device is touched */
    if (width == 0 || height == 0) { if (g_default) g_default->initialized = false; return false; }
    if (option->max_disparity <= option->min_disparity) { if (g_default) g_default->initialized = false; return false; }
    if (!g_default) {
        g_default = sgm_create(default_device());
        if (!g_default) return false;
        g_default->honor_num_paths = g_default_honor;
        g_default->reference_statics = 1;
        if (g_default_census_w) sgm_set_census_window(g_default, g_default_census_w, g_default_census_h);
        sgm_set_reference_view(g_default, g_default_view);
    }
    return sgm_initialize(g_default, width, height, option);
}

bool SGM_Reset(uint16_t width, uint16_t height, const SGMOption* option)
{
    if (g_default) g_default->initialized = false;               /* SemiGlobalMatching.c:130 */
    return SGM_Initialize(width, height, option);
}

bool SGM_Match(const uint8_t* img_left, const uint8_t* img_right, float* disp_left)
{
    if (!g_default) return false;
    return sgm_match(g_default, img_left, img_right, disp_left);
}

/* north_star's one-call form: SGM_Reset + SGM_Match (SemiGlobalMatching.c:128-132, 68-125; main.c:72,83) */
bool sgm_compute(const uint8_t* img_left, const uint8_t* img_right, uint16_t width, uint16_t height, const SGMOption* option,
                 float* disp_left)
{
    return SGM_Reset(width, height, option) && SGM_Match(img_left, img_right, disp_left);
}

bool SGM_MatchDevice(const uint8_t* d_left, const uint8_t* d_right, float* d_disp_left)
{
    if (!g_default) return false;
    return sgm_match_device(g_default, d_left, d_right, d_disp_left);
}

bool SGM_Synchronize(void) { return g_default ? sgm_synchronize(g_default) : false; }

void SGM_Shutdown(void)
{
    sgm_destroy(g_default);
    g_default = NULL;
}

size_t SGM_ReadStage(int which, void* host_out, size_t capacity)
{
    return g_default ? sgm_read_stage(g_default, which, host_out, capacity) : 0;
}

void SGM_KeepStages(int enable)
{
    if (g_default) g_default->keep_stages = enable;
}

const char* SGM_Version(void) { return SGM_VERSION_STRING; }

/* ------------------------------------------------------------------ synthetic input */

static uint32_t lcg(uint32_t* s) { *s = *s * 1664525u + 1013904223u; return *s; }

void SGM_SynthPair(int W, int H, int D, uint32_t seed, uint8_t* left, uint8_t* right)
{
    const size_t px = (size_t)W * H;
    uint8_t* noise = (uint8_t*)malloc(px);
    if (!noise) return;
    uint32_t st = seed;
    for (size_t i = 0; i < px; ++i) noise[i] = (uint8_t)(lcg(&st) >> 24);
    for (int y = 0; y < H; ++y) {
        const int yb = (y + 1 < H) ? y + 1 : H - 1;
        for (int x = 0; x < W; ++x) {
            const int xb = (x + 1 < W) ? x + 1 : W - 1;
            const unsigned sum = noise[(size_t)y * W + x] + noise[(size_t)y * W + xb] + noise[(size_t)yb * W + x] +
                                 noise[(size_t)yb * W + xb];
            left[(size_t)y * W + x] = (uint8_t)((sum + 2) >> 2);
        }
    }
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            const int shift = D / 16 + ((5 * D / 8) * y) / H + 4 * ((x >> 6) & 1);
            int v = (x + shift < W) ? left[(size_t)y * W + x + shift] : (int)(lcg(&st) >> 24);
            v += (int)((lcg(&st) >> 24) & 3u) - 1;
            right[(size_t)y * W + x] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
        }
    free(noise);
}

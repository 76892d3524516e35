/*
 * stub_device.c -- TEST INFRASTRUCTURE ONLY (tests/test_host_error_paths.py).
 *
 * A stand-in for the HIP side of csrc/sgm_device.h so that the product's C host (csrc/sgm_host.c) can be driven on a
 * machine without a GPU: "device" memory is malloc, copies are memcpy, kernel launchers compute nothing -- they append
 * their name (and the arguments the tests look at) to a log and can be told to refuse the n-th call.  This checks
 * the ORDER of the stages and the error paths of the host, never results.  It is linked with sgm_host.c into a
 * test-only library under a temporary directory; nothing of it is part of libsgm_mi355x.so.
 */
#include "sgm_device.h"

#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define LOG_MAX 4096
static char g_log[LOG_MAX][40];
static int g_log_arg[LOG_MAX];
static int g_n;
static char g_fail_name[40];
static int g_fail_countdown = -1;

void stub_clear(void) { g_n = 0; g_fail_countdown = -1; g_fail_name[0] = 0; }
int stub_log_size(void) { return g_n; }
const char* stub_log_name(int i) { return (i >= 0 && i < g_n) ? g_log[i] : ""; }
int stub_log_arg(int i) { return (i >= 0 && i < g_n) ? g_log_arg[i] : -1; }
/* the nth (0-based) call of launcher `name` from now on returns an error */
void stub_fail_at(const char* name, int nth) { snprintf(g_fail_name, sizeof g_fail_name, "%s", name); g_fail_countdown = nth; }

static pthread_mutex_t g_mu = PTHREAD_MUTEX_INITIALIZER;     /* the tile-pipeline tests run ranks as threads */
static int note_locked(const char* name, int arg);
static int note(const char* name, int arg)
{
    pthread_mutex_lock(&g_mu);
    const int rc = note_locked(name, arg);
    pthread_mutex_unlock(&g_mu);
    return rc;
}
static int note_locked(const char* name, int arg)
{
    if (g_n < LOG_MAX) { snprintf(g_log[g_n], sizeof g_log[g_n], "%s", name); g_log_arg[g_n] = arg; ++g_n; }
    if (g_fail_countdown >= 0 && strcmp(name, g_fail_name) == 0 && g_fail_countdown-- == 0) {
        g_fail_countdown = -1;
        return 719;                                  /* some HIP error code */
    }
    return 0;
}

int sgmd_device_count(void) { return 1; }
int sgmd_device_is_gfx950(int o) { (void)o; return 1; }
int sgmd_stream_create(int o, void** st) { (void)o; *st = malloc(8); return 0; }
int sgmd_stream_create_cus(int o, void** st, int first, int count) { (void)o; *st = malloc(8); return note("stream_cus", first * 100 + count); }
int sgmd_stream_create_prio(int o, void** st, int prio) { (void)o; *st = malloc(8); return note("stream_prio", prio); }
int sgmd_device_cus(int o, int* per, int* xcds) { (void)o; *per = 32; *xcds = 8; return 0; }
int sgmd_stream_destroy(int o, void* st) { (void)o; free(st); return 0; }
int sgmd_stream_sync(int o, void* st) { (void)o; (void)st; return note("sync", 0); }
int sgmd_event_create(int o, void** e) { (void)o; *e = malloc(8); return 0; }
void sgmd_event_destroy(int o, void* e) { (void)o; free(e); }
int sgmd_event_record(int o, void* e, void* st) { (void)o; (void)e; (void)st; return note("event_record", 0); }
int sgmd_stream_wait_event(int o, void* st, void* e) { (void)o; (void)st; (void)e; return note("wait_event", 0); }
int sgmd_event_sync(int o, void* e) { (void)o; (void)e; return note("event_sync", 0); }
int sgmd_set_device(int o) { (void)o; return 0; }
int sgmd_mem_info(int o, size_t* f, size_t* t) { (void)o; *f = (size_t)200 << 30; *t = (size_t)288 << 30; return 0; }
int sgmd_alloc(int o, void** p, size_t n)      /* logged with its size in KiB; nothing the tests do reads the bytes of a volume */
{ (void)o; *p = calloc(1, n > (1u << 20) ? (1u << 20) : (n ? n : 16)); note("alloc", (int)(n >> 10)); return *p ? 0 : 2; }
int sgmd_free(int o, void* p) { (void)o; free(p); return 0; }
int sgmd_alloc_pinned(int o, void** p, size_t n) { (void)o; *p = calloc(1, n ? n : 16); return *p ? 0 : 2; }
int sgmd_free_pinned(int o, void* p) { (void)o; free(p); return 0; }
int sgmd_host_is_pinned(int o, const void* p, size_t n) { (void)o; (void)p; (void)n; return 0; }
int sgmd_h2d_async(int o, void* st, void* d, const void* s, size_t n) { (void)o; (void)st; if (n <= (1u << 20)) memcpy(d, s, n); return note("h2d", (int)n); }
int sgmd_d2h_async(int o, void* st, void* d, const void* s, size_t n) { (void)o; (void)st; if (n <= (1u << 20)) memcpy(d, s, n); return note("d2h", (int)n); }
int sgmd_plane_rows_copy(int o, void* st, void* planes, size_t pb, size_t ro, size_t rb, const int* dirs, int nd, int frames, void* buf, int to_buf)
{
    (void)o; (void)st;
    /* real copies while the whole plane allocation is below the allocator's cap: the sanitizer build checks the offsets */
    if (pb * 8 * (size_t)frames <= (1u << 20))
        for (int f = 0; f < frames; ++f)
            for (int k = 0; k < nd; ++k) {
                char* cell = (char*)planes + ((size_t)f * 8 + (size_t)dirs[k]) * pb + ro;
                char* slot = (char*)buf + ((size_t)f * nd + k) * rb;
                if (to_buf) memcpy(slot, cell, rb); else memcpy(cell, slot, rb);
            }
    return note(to_buf ? "rows_out" : "rows_in", nd * frames);
}
int sgmd_d2d_async(int o, void* st, void* d, const void* s, size_t n) { (void)o; (void)st; if (n <= (1u << 20)) memmove(d, s, n); return note("d2d", (int)n); }
int sgmd_d2d_2d_async(int o, void* st, void* d, size_t dp, const void* s, size_t sp, size_t w, size_t rows)
{
    (void)o; (void)st;
    if ((rows ? (rows - 1) * (dp > sp ? dp : sp) : 0) + w <= (1u << 20))
        for (size_t r = 0; r < rows; ++r) memmove((char*)d + r * dp, (const char*)s + r * sp, w);
    return note("d2d_2d", (int)(w * rows));
}
int sgmd_memset_async(int o, void* st, void* d, int v, size_t n) { (void)o; (void)st; if (n <= (1u << 20)) memset(d, v, n); return note("memset", (int)n); }

int sgmd_timer_create(int o, void** t, int n) { (void)o; (void)n; *t = malloc(8); return 0; }
void sgmd_timer_destroy(int o, void* t) { (void)o; free(t); }
int sgmd_timer_mark(int o, void* t, void* st, int i) { (void)o; (void)t; (void)st; (void)i; return 0; }
int sgmd_timer_elapsed(int o, void* t, int a, int b, float* ms) { (void)o; (void)t; (void)a; (void)b; *ms = 0.f; return 0; }

/* ---- toy compute (tests/test_tiles_c.py): with stub_toy_compute(1) the aggregation, the cost sum and the post pass compute a
 * small recurrence with the data dependencies of SGM's vertical / diagonal paths on byte 0 of every cell, so that the row-tile
 * pipeline (csrc/sgm_tiles.c) can be checked END TO END on the CPU -- hand-over routing, slot reuse, row gather, batches -- against
 * the same recurrence on the whole frame.  Only for frames whose planes fit the allocator's cap. ---- */
static int g_toy;
void stub_toy_compute(int on) { g_toy = on; }
static unsigned char* toy_cell(void* planes, size_t pb, const sgmd_geom* g, int f, int d, int y, int x)
{ return (unsigned char*)planes + ((size_t)f * 8 + (size_t)d) * pb + ((size_t)y * g->W + x) * g->Dp; }
static void toy_aggregate(const sgmd_geom* g, const sgmd_paths* p, const unsigned char* img, void* planes, size_t pb)
{
    for (int f = 0; f < g->B; ++f)
        for (int d = 0; d < p->ndirs; ++d) {
            if (!((p->dir_mask >> d) & 1)) continue;
            const int dy = p->dy[d], dx = p->dx[d];
            if (dy == 0) {                                           /* row-local */
                for (int y = g->row_begin; y < g->row_end; ++y)
                    for (int x = 0; x < g->W; ++x) *toy_cell(planes, pb, g, f, d, y, x) = (unsigned char)(3 * d + img[((size_t)f * g->H + y) * g->W + x]);
                continue;
            }
            for (int k = 0; k < g->row_end - g->row_begin; ++k) {   /* along y, reading the row before (the imported one at a tile edge) */
                const int y = dy > 0 ? g->row_begin + k : g->row_end - 1 - k, py = y - dy;
                for (int x = 0; x < g->W; ++x) {
                    const int px = ((x - dx) % g->W + g->W) % g->W;     /* wraps like the reference's diagonals */
                    const unsigned prev = (py >= 0 && py < g->H) ? *toy_cell(planes, pb, g, f, d, py, px) : 0u;
                    *toy_cell(planes, pb, g, f, d, y, x) = (unsigned char)(prev * 5u + img[((size_t)f * g->H + y) * g->W + x] + (unsigned)d);
                }
            }
        }
}
static void toy_sum(const sgmd_geom* g, int nd, const void* planes, size_t pb, void* disp)
{
    for (int f = 0; f < g->B; ++f)
        for (int y = g->row_begin; y < g->row_end; ++y)
            for (int x = 0; x < g->W; ++x) {
                unsigned s = 0;
                for (int d = 0; d < nd; ++d) s += *toy_cell((void*)planes, pb, g, f, d, y, x);
                ((float*)disp)[((size_t)f * g->H + y) * g->W + x] = (float)s;
            }
}

int sgmd_census(int o, void* st, const sgmd_geom* g, const void* l, const void* r, void* cl, void* cr, const void* need, int keep)
{ (void)o; (void)st; (void)l; (void)r; (void)cl; (void)cr; (void)need; return note("census", g->B | (keep << 16)); }
void sgmd_census_blocks(const sgmd_geom* g, int* bx, int* by) { *bx = (g->W + 63) / 64; *by = (g->H + 15) / 16; }
int sgmd_cost(int o, void* st, const sgmd_geom* g, const void* cl, const void* cr, void* c)
{ (void)o; (void)st; (void)g; (void)cl; (void)cr; (void)c; return note("cost", 0); }
int sgmd_census_window(int o, void* st, const sgmd_geom* g, int cw, int ch, const void* l, const void* r, void* cl, void* cr)
{ (void)o; (void)st; (void)g; (void)l; (void)r; (void)cl; (void)cr; return note("census_window", cw * 100 + ch); }
int sgmd_cost64(int o, void* st, const sgmd_geom* g, const void* cl, const void* cr, void* c)
{ (void)o; (void)st; (void)g; (void)cl; (void)cr; (void)c; return note("cost64", 0); }
int sgmd_aggregate_volume(int o, void* st, const sgmd_geom* g, const sgmd_paths* p, const void* img, const void* cost, const void* lut,
                          void* planes, size_t pb, void* ex)
{ (void)o; (void)st; (void)img; (void)cost; (void)lut; (void)planes; (void)pb; (void)ex; return note("aggregate_volume", p->dir_mask | (g->LPP << 8)); }
int sgmd_lrcheck_right(int o, void* st, const sgmd_geom* g, const void* dr, const void* dl, float th, int chk, void* out)
{ (void)o; (void)st; (void)g; (void)dr; (void)dl; (void)th; (void)out; return note("lrcheck_right", chk); }
size_t sgmd_census_slack(const sgmd_geom* g) { return ((size_t)g->dmin + g->Dp + 8) * 4; }
int sgmd_aggregate(int o, void* st, const sgmd_geom* g, const sgmd_paths* p, const void* img, const void* cl, const void* cr,
                   const void* lut, void* planes, size_t pb, void* ex)
{ (void)o; (void)st; (void)cl; (void)cr; (void)lut; (void)ex; if (g_toy) toy_aggregate(g, p, (const unsigned char*)img, planes, pb); return note("aggregate", p->dir_mask | (p->up_fused ? 0x100 : 0)); }
int sgmd_sum_wta(int o, void* st, const sgmd_geom* g, int nd, const void* pl, size_t pb, const void* ex, const void* re,
                 const void* rc, int cap, int accumulate, void* S, int cu, float omr, void* dl)
{ (void)o; (void)st; (void)g; (void)nd; (void)pl; (void)pb; (void)ex; (void)re; (void)rc; (void)cap; (void)S; (void)cu; (void)omr; (void)dl;
  return note("sum_wta", accumulate); }
int sgmd_sum_wta_lr_supported(const sgmd_geom* g, int cap) { (void)cap; return g->Dp <= 256; }
int sgmd_sum_wta_lr(int o, void* st, const sgmd_geom* g, int nd, const void* pl, size_t pb, const void* ex, const void* re,
                    const void* rc, int cap, int accumulate, int store_S, int do_right, void* S, int cu, float omr, void* dl, void* dr)
{ (void)o; (void)st; (void)ex; (void)re; (void)rc; (void)cap; (void)do_right; (void)S; (void)cu; (void)omr; (void)dr;
  if (g_toy) toy_sum(g, nd, pl, pb, dl);
  return note("sum_wta_lr", accumulate | (store_S << 1)); }
/* the fused last sweep: logged with the rows per workgroup; the aggregation launch in front of it is logged by sgmd_aggregate */
int sgmd_upsum_rows(const sgmd_geom* g) { return (g->Dp == 128 && g->W > g->H && g->row_begin == 0 && g->row_end == g->H) ? 3 : 0; }
size_t sgmd_upsum_scratch_bytes(const sgmd_geom* g) { return sgmd_upsum_rows(g) ? (size_t)g->B * 6 * g->W * g->Dp + 4096 : 0; }
int sgmd_upsum(int o, void* st, const sgmd_geom* g, const sgmd_paths* p, const void* img, const void* cl, const void* cr, const void* lut,
               const void* pl, size_t pb, const void* ex, const void* re, const void* rc, int cap, int do_right, int cu, float omr,
               void* scratch, unsigned gen, void* status, int rows, int wgs, void* dl, void* dr)
{ (void)o; (void)st; (void)g; (void)p; (void)img; (void)cl; (void)cr; (void)lut; (void)pl; (void)pb; (void)ex; (void)re; (void)rc; (void)cap;
  (void)do_right; (void)cu; (void)omr; (void)scratch; (void)gen; (void)status; (void)wgs; (void)dl; (void)dr;
  return note("upsum", rows); }
int sgmd_wta_right(int o, void* st, const sgmd_geom* g, const void* S, int cu, float omr, void* dr)
{ (void)o; (void)st; (void)g; (void)S; (void)cu; (void)omr; (void)dr; return note("wta_right", 0); }
int sgmd_lrcheck(int o, void* st, const sgmd_geom* g, void* dl, const void* dr, float th)
{ (void)o; (void)st; (void)g; (void)dl; (void)dr; (void)th; return note("lrcheck", 0); }
int sgmd_speckle(int o, void* st, const sgmd_geom* g, void* d, float diff, unsigned area, void* a, void* b, void* c)
{ (void)o; (void)st; (void)diff; (void)a; (void)b; (void)c;
  if (g_toy) for (size_t i = 0; i < (size_t)g->B * g->W * g->H; ++i) ((float*)d)[i] += 1000.0f;       /* whole-frame pass, once per frame */
  return note("speckle", (int)area); }
size_t sgmd_median_scratch_bytes(const sgmd_geom* g) { return (size_t)g->W * g->H * 4; }
static int g_median_stall;
void stub_median_stall(int on) { g_median_stall = on; }      /* the next median launches report a band that gave up waiting */
int sgmd_median(int o, void* st, const sgmd_geom* g, void* d, void* s, void* status)
{ (void)o; (void)st; (void)g; (void)d; (void)s; if (g_median_stall && status) *(int*)status = 1; return note("median", 0); }
int sgmd_depth(int o, void* st, const void* d, size_t n, float fx, float b, float doffs, void* out)
{ (void)o; (void)st; (void)d; (void)fx; (void)b; (void)doffs; (void)out; return note("depth", (int)n); }
int sgmd_gray_planes(int o, void* st, const void* bgr, size_t n, int wr, void* gray)
{ (void)o; (void)st; (void)bgr; (void)wr; (void)gray; return note("gray", (int)n); }
int sgmd_score(int o, void* st, const void* g, const void* t, size_t n, float th, double* s, unsigned long long* nv, unsigned long long* nb)
{ (void)o; (void)st; (void)g; (void)t; (void)th; *s = 0; *nv = 0; *nb = 0; return note("score", (int)n); }

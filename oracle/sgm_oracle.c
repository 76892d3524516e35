/*
 * sgm_oracle.c -- TEST INFRASTRUCTURE ONLY (see sgm_oracle.h).
 *
 * Plain single-threaded C restatement of the reference pipeline with run-time sizes.
 * "ref:" comments give the line of
 * /root/reference/SemiGlobalMatching/SemiGlobalMatching/SemiGlobalMatching.c
 * whose behaviour the code below reproduces.  Integer widths and the order of the
 * float operations follow the reference exactly because they are part of "bit-exact"
 * (SURVEY.md 8a, quirks Q1-Q16).
 */
#include "sgm_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ census */

void sgmo_census5x5(const uint8_t* img, int W, int H, uint32_t* census)
{
    memset(census, 0, (size_t)W * H * sizeof(uint32_t));      /* Q3: border stays 0 (a fresh, zero-initialised buffer) */
    sgmo_census5x5_interior(img, W, H, census);
}

/* Exactly the stores of ref :134-159: interior pixels only.  The 2-pixel border -- and, for W <= 5 or H <= 5, every pixel
 * (ref :136) -- is left as it is: in the reference the buffer is a static array (SemiGlobalMatching.h:67-68) that is never
 * cleared, so those words keep what an earlier frame, possibly of another shape, left at the same linear index. */
void sgmo_census5x5_interior(const uint8_t* img, int W, int H, uint32_t* census)
{
    if (W <= 5 || H <= 5) return;                             /* ref :136 */
    for (int y = 2; y < H - 2; ++y) {
        for (int x = 2; x < W - 2; ++x) {
            const uint8_t centre = img[(size_t)y * W + x];
            uint32_t bits = 0;
            /* raster order, first comparison ends up in bit 24 (ref :146-154) */
            for (int k = 0; k < 25; ++k) {
                const int yy = y + k / 5 - 2, xx = x + k % 5 - 2;
                bits = (bits << 1) | (uint32_t)(img[(size_t)yy * W + xx] < centre);
            }
            census[(size_t)y * W + x] = bits;
        }
    }
}

/* Extension (not in the reference, which only has the 5x5 window): the same transform for any odd window of at most 64
 * pixels -- raster order, one bit per window pixel incl. the centre (always 0), first comparison in the highest bit,
 * border of cw/2 columns and ch/2 rows stays 0, nothing at all for W <= cw or H <= ch.  For 5x5 it equals
 * sgmo_census5x5.  Pinned by nothing but this restatement ("parity unpinned by the reference"). */
void sgmo_census_window(const uint8_t* img, int W, int H, int cw, int ch, uint64_t* census)
{
    memset(census, 0, (size_t)W * H * sizeof(uint64_t));
    if (W <= cw || H <= ch) return;
    const int rx = cw / 2, ry = ch / 2;
    for (int y = ry; y < H - ry; ++y)
        for (int x = rx; x < W - rx; ++x) {
            const uint8_t centre = img[(size_t)y * W + x];
            uint64_t bits = 0;
            for (int r = -ry; r <= ry; ++r)
                for (int c = -rx; c <= rx; ++c)
                    bits = (bits << 1) | (uint64_t)(img[(size_t)(y + r) * W + (x + c)] < centre);
            census[(size_t)y * W + x] = bits;
        }
}

/* -------------------------------------------------------------------- cost */

static inline uint8_t bitcount32(uint32_t v)
{
    uint8_t n = 0;
    for (; v; v &= v - 1) ++n;                                /* ref :185-196 */
    return n;
}

void sgmo_cost(const uint32_t* cl, const uint32_t* cr, int W, int H,
               int dmin, int dmax, uint8_t* cost)
{
    const int D = dmax - dmin;
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            uint8_t* out = cost + ((size_t)y * W + x) * D;
            const uint32_t a = cl[(size_t)y * W + x];
            for (int d = dmin; d < dmax; ++d) {
                const int xr = x - d;
                /* off-image right pixel -> UINT8_MAX/2 (ref :170-171) */
                out[d - dmin] = (xr < 0 || xr >= W) ? (uint8_t)127
                                                    : bitcount32(a ^ cr[(size_t)y * W + xr]);
            }
        }
}

/* Hamming cost of 64-bit census words (extension, see sgmo_census_window); off-image = 127 as in ref :170-171 */
void sgmo_cost64(const uint64_t* cl, const uint64_t* cr, int W, int H, int dmin, int dmax, uint8_t* cost)
{
    const int D = dmax - dmin;
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            uint8_t* out = cost + ((size_t)y * W + x) * D;
            const uint64_t a = cl[(size_t)y * W + x];
            for (int d = dmin; d < dmax; ++d) {
                const int xr = x - d;
                if (xr < 0 || xr >= W) { out[d - dmin] = 127; continue; }
                uint64_t v = a ^ cr[(size_t)y * W + xr];
                uint8_t n = 0;
                for (; v; v &= v - 1) ++n;
                out[d - dmin] = n;
            }
        }
}

/* ----------------------------------------------------------- path geometry */

static int dir_is_forward(int dx, int dy)
{
    /* ref :232 */
    return (dx == 1 && dy == 0) || (dx == 0 && dy == 1) || (dx == 1 && dy == 1) || (dx == -1 && dy == 1);
}

int sgmo_path_lines(int W, int H, int dx, int dy)
{
    (void)dx;
    return dy == 0 ? H : W;                                   /* ref :238 */
}

int sgmo_path_walk(int W, int H, int dx, int dy, int line, int32_t* pix)
{
    const int fwd = dir_is_forward(dx, dy);
    const int s = fwd ? 1 : -1;
    const int64_t npx = (int64_t)W * H;
    int64_t p;
    int steps;
    if (dy == 0) {                                            /* ref :243-248 */
        p = (int64_t)line * W + (fwd ? 0 : W - 1);
        steps = W - 1;
    } else {                                                  /* ref :250-255 */
        p = (fwd ? 0 : (int64_t)(H - 1) * W) + line;
        steps = H - 1;
    }
    /* the two uint16_t trackers of ref :278-279 */
    uint16_t row = (uint16_t)(fwd ? 0 : H - 1);
    uint16_t col = (uint16_t)line;
    int n = 0;
    pix[n++] = (int32_t)p;
    for (int j = 0; j < steps; ++j) {
        if (dy == 0) {
            p += s;                                           /* ref :283-288 */
        } else if (dx == 0) {
            p += (int64_t)s * W;                              /* ref :289-294 */
        } else {
            const int not_last_row = fwd ? (row < H - 1) : (row > 0);
            if (col == W - 1 && not_last_row) {               /* ref :297-303 */
                p = (int64_t)(row + s) * W;
                col = 0;
            } else if (col == 0 && not_last_row) {            /* ref :304-310 */
                p = (int64_t)(row + s) * W + (W - 1);
                col = (uint16_t)(W - 1);
            } else if (dx == dy) {                            /* ref :311-316 */
                p += (int64_t)s * (W + 1);
            } else {                                          /* ref :317-322 */
                p += (int64_t)s * (W - 1);
            }
        }
        if (p < 0 || p >= npx) break;                         /* Q6: out-of-image step dropped */
        pix[n++] = (int32_t)p;
        row = (uint16_t)(row + s);                            /* ref :359 */
        col = (uint16_t)((dx != dy && dx != 0 && dy != 0) ? col - s : col + s); /* ref :360-367 */
    }
    return n;
}

/* ------------------------------------------------------------- aggregation */

static uint64_t g_wraps, g_dropped;     /* diagnostics of the last sgmo_aggregate_* calls */

void sgmo_aggregate_dir(const uint8_t* img, const uint8_t* cost, int W, int H, int D,
                        int p1, int p2_init, int dx, int dy,
                        uint16_t* S, uint8_t* L_last, uint8_t* visits)
{
    const int lines = sgmo_path_lines(W, H, dx, dy);
    const int full = (dy == 0 ? W : H);
    int32_t* pix = (int32_t*)malloc(sizeof(int32_t) * (size_t)(W > H ? W : H));
    /* previous-pixel path costs with the two 255 sentinels at d=-1 and d=D (ref :260-263, Q8) */
    uint8_t* prev = (uint8_t*)malloc((size_t)D + 2);
    uint8_t* cur = (uint8_t*)malloc((size_t)D);

    for (int line = 0; line < lines; ++line) {
        const int n = sgmo_path_walk(W, H, dx, dy, line, pix);
        g_dropped += (uint64_t)(n < full);     /* lines ended by the out-of-image guard */

        /* first pixel: L = C (ref :266-275) */
        size_t cell = (size_t)pix[0] * D;
        memset(prev, 0xFF, (size_t)D + 2);
        uint8_t min_prev = 0xFF;
        for (int d = 0; d < D; ++d) {
            const uint8_t c = cost[cell + d];
            S[cell + d] = (uint16_t)(S[cell + d] + c);
            prev[d + 1] = c;
            if (c < min_prev) min_prev = c;
            if (L_last) L_last[cell + d] = c;
        }
        if (visits) visits[pix[0]]++;
        uint8_t g_prev = img[pix[0]];

        for (int k = 1; k < n; ++k) {
            cell = (size_t)pix[k] * D;
            const uint8_t g = img[pix[k]];
            /* adaptive P2 (ref :335, Q9): C integer division, truncated to uint16 with the sum */
            int pen = p2_init / (abs((int)g - (int)g_prev) + 1);
            if (p1 > pen) pen = p1;
            const uint16_t l4 = (uint16_t)((int)min_prev + pen);
            uint8_t min_cur = 0xFF;
            for (int d = 0; d < D; ++d) {
                const uint16_t l1 = prev[d + 1];
                const uint16_t l2 = (uint16_t)((int)prev[d] + p1);
                const uint16_t l3 = (uint16_t)((int)prev[d + 2] + p1);
                uint16_t m = l1;
                if (l2 < m) m = l2;
                if (l3 < m) m = l3;
                if (l4 < m) m = l4;
                const int wide = (int)cost[cell + d] + (int)m - (int)min_prev;
                if (wide < 0 || wide > 255) ++g_wraps;
                const uint8_t l = (uint8_t)wide;              /* Q7: mod 256 */
                cur[d] = l;
                S[cell + d] = (uint16_t)(S[cell + d] + l);    /* ref :345 */
                if (l < min_cur) min_cur = l;
                if (L_last) L_last[cell + d] = l;
            }
            memcpy(prev + 1, cur, (size_t)D);
            min_prev = min_cur;
            g_prev = g;
            if (visits) visits[pix[k]]++;
        }
    }
    free(cur);
    free(prev);
    free(pix);
}

static const int8_t k_dirs[8][2] = {                          /* ref :213-220 */
    {1, 0}, {-1, 0}, {0, 1}, {0, -1}, {1, 1}, {-1, -1}, {1, -1}, {-1, 1}};

void sgmo_aggregate_all(const uint8_t* img, const uint8_t* cost, int W, int H, int D,
                        int p1, int p2_init, int n_dirs, uint16_t* S)
{
    for (int i = 0; i < n_dirs; ++i)
        sgmo_aggregate_dir(img, cost, W, H, D, p1, p2_init, k_dirs[i][0], k_dirs[i][1], S, NULL, NULL);
}

/* --------------------------------------------------------------------- WTA */

void sgmo_wta(const uint16_t* S, int W, int H, int dmin, int dmax,
              bool check_unique, float uniqueness_ratio, int right_view, float* disp)
{
    const int D = dmax - dmin;
    uint16_t* local = (uint16_t*)malloc(sizeof(uint16_t) * (size_t)D);
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            float* out = disp + (size_t)y * W + x;
            uint16_t best_cost = 0xFFFF;
            uint16_t best_d = 0;                              /* ref :382 (stays 0 if nothing beats 65535) */
            for (int d = dmin; d < dmax; ++d) {
                uint16_t c;
                if (!right_view) {
                    c = S[((size_t)y * W + x) * D + (d - dmin)];
                } else {
                    const int xl = x + d;                     /* ref :397-408 */
                    c = (xl >= 0 && xl < W) ? S[((size_t)y * W + xl) * D + (d - dmin)] : (uint16_t)0xFFFF;
                    if (!(xl >= 0 && xl < W)) { local[d - dmin] = c; continue; }
                }
                local[d - dmin] = c;
                if (best_cost > c) { best_cost = c; best_d = (uint16_t)d; }   /* strict: lowest d wins */
            }
            if (check_unique) {                               /* ref :412-426, Q10 */
                uint16_t second = 0xFFFF;
                for (int d = dmin; d < dmax; ++d)
                    if (d != best_d && local[d - dmin] < second) second = local[d - dmin];
                const uint16_t margin = (uint16_t)((float)best_cost * (1 - uniqueness_ratio));
                if ((int)second - (int)best_cost <= (int)margin) { *out = INFINITY; continue; }
            }
            if (best_d == dmin || best_d == dmax - 1) { *out = INFINITY; continue; }  /* ref :428 */
            const int i1 = (int)best_d - 1 - dmin, i2 = (int)best_d + 1 - dmin;
            if (i1 < 0 || i2 >= D) { *out = INFINITY; continue; }   /* only reachable where the reference is UB
                                                                        (right view, dmin>0, no candidate) */
            const int16_t c1 = (int16_t)local[i1];            /* 65535 -> -1 (Q11b) */
            const int16_t c2 = (int16_t)local[i2];
            int16_t denom = (int16_t)(c1 + c2 - 2 * (int)best_cost);
            if (denom < 1) denom = 1;
            *out = (float)best_d + (float)(c1 - c2) / ((float)denom * 2.0f);   /* ref :440 */
        }
    free(local);
}

/* ---------------------------------------------------------------- LR check */

void sgmo_lrcheck(float* dl, const float* dr, int W, int H, float thres)
{
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            float* p = dl + (size_t)y * W + x;
            const float d = *p;
            if (d == INFINITY) continue;
            /* ref :454 -- float subtract, double add, truncation toward zero (Q12) */
            const int32_t xr = (int32_t)((double)((float)x - d) + 0.5);
            if (xr >= 0 && xr < W) {
                const float r = dr[(size_t)y * W + xr];
                if (r == INFINITY) continue;                  /* left kept */
                if (fabs(d - r) > thres) *p = INFINITY;
            } else {
                *p = INFINITY;
            }
        }
}

/* Extension: the RIGHT view as the reference view.  The mirror image of ref :445-470: right pixel x with disparity d
 * corresponds to left column x + d (same float subtract->add, double rounding, truncation); invalid if that column is
 * off the image or the left map disagrees by more than thres; kept if the left pixel is invalid. */
void sgmo_lrcheck_right(float* dr, const float* dl, int W, int H, float thres)
{
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            float* p = dr + (size_t)y * W + x;
            const float d = *p;
            if (d == INFINITY) continue;
            const int32_t xl = (int32_t)((double)((float)x + d) + 0.5);
            if (xl >= 0 && xl < W) {
                const float l = dl[(size_t)y * W + xl];
                if (l == INFINITY) continue;
                if (fabs(d - l) > thres) *p = INFINITY;
            } else {
                *p = INFINITY;
            }
        }
}

/* ---------------------------------------------------------------- speckles */

void sgmo_remove_speckles(float* disp, int W, int H, float diff_insame, unsigned min_area)
{
    const size_t n = (size_t)W * H;
    uint8_t* seen = (uint8_t*)calloc(n, 1);
    uint32_t* queue = (uint32_t*)malloc(n * sizeof(uint32_t));
    for (size_t start = 0; start < n; ++start) {
        if (seen[start] || disp[start] == INFINITY) continue;
        size_t head = 0, tail = 0;
        queue[tail++] = (uint32_t)start;
        seen[start] = 1;
        while (head < tail) {                                 /* breadth-first flood, ref :607-631 */
            const uint32_t q = queue[head++];
            const int qy = (int)(q / (uint32_t)W), qx = (int)(q % (uint32_t)W);
            const float base = disp[q];
            for (int oy = -1; oy <= 1; ++oy)
                for (int ox = -1; ox <= 1; ++ox) {
                    if (!oy && !ox) continue;
                    const int yy = qy + oy, xx = qx + ox;
                    if (yy < 0 || yy >= H || xx < 0 || xx >= W) continue;
                    const size_t t = (size_t)yy * W + xx;
                    if (!seen[t] && disp[t] != INFINITY && fabs(disp[t] - base) <= diff_insame) {
                        queue[tail++] = (uint32_t)t;
                        seen[t] = 1;
                    }
                }
        }
        if (tail < min_area)                                  /* ref :633 */
            for (size_t i = 0; i < tail; ++i) disp[queue[i]] = INFINITY;
    }
    free(queue);
    free(seen);
}

/* ------------------------------------------------------------------ median */

static float fifth_smallest_of_9(float v[9])
{
    /* ref :496-523 returns the 5th smallest; +INF orders last, equal values are identical */
    for (int i = 1; i < 9; ++i) {
        const float t = v[i];
        int j = i - 1;
        while (j >= 0 && v[j] > t) { v[j + 1] = v[j]; --j; }
        v[j + 1] = t;
    }
    return v[4];
}

void sgmo_median3_inplace(float* d, int W, int H)
{
    /* in == out (ref :120): already-filtered values of row y-1 and of (y,x-1) feed pixel (y,x) (Q13) */
    for (int y = 1; y < H - 1; ++y)
        for (int x = 1; x < W - 1; ++x) {
            float w[9];
            int k = 0;
            for (int oy = -1; oy <= 1; ++oy)
                for (int ox = -1; ox <= 1; ++ox) w[k++] = d[(size_t)(y + oy) * W + (x + ox)];
            d[(size_t)y * W + x] = fifth_smallest_of_9(w);
        }
}

/* -------------------------------------------------------- main.c normalise */

void sgmo_normalize_u8(const float* disp, int W, int H, uint8_t* out)
{
    float lo = (float)W, hi = -(float)W;                      /* main.c:92 */
    const size_t n = (size_t)W * H;
    for (size_t i = 0; i < n; ++i)
        if (disp[i] != INFINITY) {
            if (disp[i] < lo) lo = disp[i];
            if (disp[i] > hi) hi = disp[i];
        }
    const float range = (hi - lo) != 0.0f ? (hi - lo) : 1.0f;
    for (size_t i = 0; i < n; ++i) {
        if (disp[i] == INFINITY) { out[i] = 0; continue; }
        float v = (disp[i] - lo) / range * 255.0f;            /* main.c:111 */
        if (v < 0) v = 0;
        if (v > 255) v = 255;
        out[i] = (uint8_t)v;
    }
}

/* ---------------------------------------------------------- whole pipeline */

struct sgmo_ctx {
    sgmo_option opt;
    int W, H, D;
    int honor_num_paths;
    int census_w, census_h;   /* 0 = the reference's 5x5 */
    int reference_view;       /* 0 = left (reference), 1 = right (extension) */
    uint64_t *census64_l, *census64_r;
    bool ready;
    /* ref .h:67-68: the census buffers are statics that outlive SGM_Reset; modelled as two buffers that only ever grow
     * (zero-extended, contents kept at their linear indices) and that sgmo_match writes in the interior only */
    uint32_t *census_l, *census_r;
    size_t census_px;
    uint8_t* cost;
    uint16_t* aggr;
    float* stage_f[5];      /* dispL after WTA, dispR, after LR, after speckle, final */
    uint64_t counters[2];
};

sgmo_ctx* sgmo_create(void) { return (sgmo_ctx*)calloc(1, sizeof(sgmo_ctx)); }

static void release_buffers(sgmo_ctx* c)
{
    free(c->cost); free(c->aggr);
    free(c->census64_l); free(c->census64_r);
    c->census64_l = c->census64_r = NULL;
    for (int i = 0; i < 5; ++i) free(c->stage_f[i]);
    c->cost = NULL; c->aggr = NULL;
    memset(c->stage_f, 0, sizeof c->stage_f);
}

static bool grow_census(sgmo_ctx* c, size_t px)
{
    if (px <= c->census_px) return true;
    uint32_t* l = (uint32_t*)realloc(c->census_l, px * sizeof(uint32_t));
    if (l) c->census_l = l;
    uint32_t* r = (uint32_t*)realloc(c->census_r, px * sizeof(uint32_t));
    if (r) c->census_r = r;
    if (!l || !r) return false;
    memset(c->census_l + c->census_px, 0, (px - c->census_px) * sizeof(uint32_t));
    memset(c->census_r + c->census_px, 0, (px - c->census_px) * sizeof(uint32_t));
    c->census_px = px;
    return true;
}

/* the reference's statics as a new process finds them (what oracle/ref_harness_tail.c's ref_clear_census does to the reference) */
void sgmo_clear_census(sgmo_ctx* c)
{
    if (c->census_l) memset(c->census_l, 0, c->census_px * sizeof(uint32_t));
    if (c->census_r) memset(c->census_r, 0, c->census_px * sizeof(uint32_t));
}

void sgmo_destroy(sgmo_ctx* c)
{
    if (!c) return;
    release_buffers(c);
    free(c->census_l); free(c->census_r);
    free(c);
}

void sgmo_set_honor_num_paths(sgmo_ctx* c, int honor) { c->honor_num_paths = honor; }

/* extensions (see sgmo_census_window / sgmo_lrcheck_right); both take effect at the next sgmo_match */
bool sgmo_set_census_window(sgmo_ctx* c, int cw, int ch)
{
    if (cw < 1 || ch < 1 || !(cw & 1) || !(ch & 1) || cw * ch > 64) return false;
    if (cw == 5 && ch == 5) cw = ch = 0;
    c->census_w = cw; c->census_h = ch;
    return true;
}
void sgmo_set_reference_view(sgmo_ctx* c, int right) { c->reference_view = right ? 1 : 0; }

bool sgmo_initialize(sgmo_ctx* c, uint16_t width, uint16_t height, const sgmo_option* opt)
{
    c->ready = false;
    c->opt = *opt;                                            /* ref :41, before validation */
    if (width == 0 || height == 0) return false;              /* ref :43 */
    if (opt->max_disparity <= opt->min_disparity) return false;   /* ref :46 */
    const int D = (uint16_t)(opt->max_disparity - opt->min_disparity);
    if (c->W != width || c->H != height || c->D != D || !c->aggr) {
        release_buffers(c);
        const size_t px = (size_t)width * height;
        c->cost = (uint8_t*)malloc(px * D);
        c->aggr = (uint16_t*)malloc(px * D * sizeof(uint16_t));
        for (int i = 0; i < 5; ++i) c->stage_f[i] = (float*)calloc(px, sizeof(float));
    }
    c->W = width; c->H = height; c->D = D;
    if (!grow_census(c, (size_t)width * height)) return false;
    memset(c->aggr, 0, (size_t)width * height * D * sizeof(uint16_t));   /* ref :57, Q14 */
    c->ready = c->census_l && c->census_r && c->cost && c->aggr && c->stage_f[4];
    return c->ready;
}

bool sgmo_reset(sgmo_ctx* c, uint16_t width, uint16_t height, const sgmo_option* opt)
{
    return sgmo_initialize(c, width, height, opt);            /* ref :128-132 */
}

bool sgmo_match(sgmo_ctx* c, const uint8_t* left, const uint8_t* right, float* out)
{
    if (!c->ready) return false;                              /* ref :70 */
    if (!left || !right) return false;                        /* ref :73 */
    const int W = c->W, H = c->H, D = c->D;
    const sgmo_option* o = &c->opt;
    const size_t px = (size_t)W * H;

    if (c->census_w == 0) {
        sgmo_census5x5_interior(left, W, H, c->census_l);     /* ref :82-83 on the never-cleared statics (Q3) */
        sgmo_census5x5_interior(right, W, H, c->census_r);
        sgmo_cost(c->census_l, c->census_r, W, H, o->min_disparity, o->max_disparity, c->cost);
    } else {
        if (!c->census64_l) {
            c->census64_l = (uint64_t*)malloc(px * sizeof(uint64_t));
            c->census64_r = (uint64_t*)malloc(px * sizeof(uint64_t));
            if (!c->census64_l || !c->census64_r) return false;
        }
        sgmo_census_window(left, W, H, c->census_w, c->census_h, c->census64_l);
        sgmo_census_window(right, W, H, c->census_w, c->census_h, c->census64_r);
        sgmo_cost64(c->census64_l, c->census64_r, W, H, o->min_disparity, o->max_disparity, c->cost);
    }

    g_wraps = g_dropped = 0;
    const int n_dirs = (c->honor_num_paths && o->num_paths == 4) ? 4 : 8;     /* Q1 */
    sgmo_aggregate_all(left, c->cost, W, H, D, o->p1, o->p2_init, n_dirs, c->aggr);
    c->counters[0] = g_dropped;
    c->counters[1] = g_wraps;

    float* cur = c->stage_f[0];
    sgmo_wta(c->aggr, W, H, o->min_disparity, o->max_disparity, o->is_check_unique, o->uniqueness_ratio, 0, cur);
    memcpy(c->stage_f[2], cur, px * sizeof(float));
    cur = c->stage_f[2];
    if (o->is_check_lr || c->reference_view) {
        sgmo_wta(c->aggr, W, H, o->min_disparity, o->max_disparity, o->is_check_unique, o->uniqueness_ratio, 1,
                 c->stage_f[1]);
        if (c->reference_view) {                              /* extension: the right view's map, checked against the left */
            memcpy(cur, c->stage_f[1], px * sizeof(float));
            if (o->is_check_lr) sgmo_lrcheck_right(cur, c->stage_f[0], W, H, o->lrcheck_thres);
        } else {
            sgmo_lrcheck(cur, c->stage_f[1], W, H, o->lrcheck_thres);
        }
    }
    memcpy(c->stage_f[3], cur, px * sizeof(float));
    cur = c->stage_f[3];
    if (o->is_remove_speckles) sgmo_remove_speckles(cur, W, H, 1.0f, o->min_speckle_area);
    memcpy(c->stage_f[4], cur, px * sizeof(float));
    sgmo_median3_inplace(c->stage_f[4], W, H);
    memcpy(out, c->stage_f[4], px * sizeof(float));
    return true;
}

const void* sgmo_stage(const sgmo_ctx* c, int which, size_t* bytes)
{
    const size_t px = (size_t)c->W * c->H;
    size_t n = 0;
    const void* p = NULL;
    switch (which) {
    case 0: p = c->census_w ? (const void*)c->census64_l : (const void*)c->census_l; n = px * (c->census_w ? 8 : 4); break;
    case 1: p = c->census_w ? (const void*)c->census64_r : (const void*)c->census_r; n = px * (c->census_w ? 8 : 4); break;
    case 2: p = c->cost; n = px * c->D; break;
    case 3: p = c->aggr; n = px * c->D * 2; break;
    default:
        if (which >= 4 && which <= 8) { p = c->stage_f[which - 4]; n = px * 4; }
    }
    if (bytes) *bytes = n;
    return p;
}

void sgmo_counters(const sgmo_ctx* c, uint64_t out[2]) { out[0] = c->counters[0]; out[1] = c->counters[1]; }

/* --------------------------------------------------------- synthetic input */

static inline uint32_t lcg_next(uint32_t* s) { *s = *s * 1664525u + 1013904223u; return *s; }

void sgmo_synth_pair(int W, int H, int D, uint32_t seed, uint8_t* left, uint8_t* right)
{
    const size_t px = (size_t)W * H;
    uint8_t* noise = (uint8_t*)malloc(px);
    uint32_t s = seed;
    for (size_t i = 0; i < px; ++i) noise[i] = (uint8_t)(lcg_next(&s) >> 24);
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            const int y1 = y + 1 < H ? y + 1 : H - 1, x1 = x + 1 < W ? x + 1 : W - 1;
            left[(size_t)y * W + x] = (uint8_t)((noise[(size_t)y * W + x] + noise[(size_t)y * W + x1] +
                                                 noise[(size_t)y1 * W + x] + noise[(size_t)y1 * W + x1] + 2) >> 2);
        }
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            const int delta = D / 16 + ((5 * D / 8) * y) / H + 4 * ((x >> 6) & 1);
            int v = (x + delta < W) ? left[(size_t)y * W + x + delta] : (int)(lcg_next(&s) >> 24);
            v += (int)((lcg_next(&s) >> 24) & 3u) - 1;
            right[(size_t)y * W + x] = (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v);
        }
    free(noise);
}

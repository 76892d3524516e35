/*
 * sgm_tile_sched.c -- the step schedule of the row-tile pipeline (include/sgm_tiles.h, layer 1).  No device code: which
 * operation a rank performs in which step, as calls on an engine vtable.  The device pipeline (sgm_tiles.c) and tiling.py's
 * TilePipeline (Python engines, the CPU tests over gloo) both run exactly this function.
 *
 * Why the schedule looks like this: six of the eight path directions of the reference
 * (SemiGlobalMatching.c:213-220; the recurrence of .c:229-372) are recurrences along y, so tile k cannot start its downward
 * directions before tile k-1 has handed over the path costs of its last row, and the upward ones flow the other way.  With
 * frames in flight the hand-overs form a systolic pipeline over the ranks.
 */
#include "../../include/sgm_tiles.h"

bool sgm_tile_rows(int height, int world, int rank, int* r0, int* r1)
{
    if (world < 1 || height < world || rank < 0 || rank >= world) return false;
    const int base = height / world, extra = height % world;
    const int begin = rank * base + (rank < extra ? rank : extra);
    if (r0) *r0 = begin;
    if (r1) *r1 = begin + base + (rank < extra ? 1 : 0);
    return true;
}

size_t sgm_tile_slot_bytes(int r0, int r1, uint16_t width, uint16_t height, const SGMOption* option, int batch)
{
    if (!option || batch < 1 || r0 < 0 || r1 <= r0 || r1 > height) return 0;
    const int D = (uint16_t)(option->max_disparity - option->min_disparity);
    const size_t Dp = D <= 32 ? 32 : D <= 64 ? 64 : D <= 128 ? 128 : D <= 192 ? 192 : D <= 256 ? 256 : 512;   /* sgm_host.c's padded range */
    const size_t rows = (size_t)(r1 - r0) + (r1 - r0 < height ? 2 : 0);
    return (size_t)batch * (8 * rows * width * Dp + (size_t)64 * width * height + 4 * 3 * (size_t)width * Dp);
}

int sgm_tile_slots_needed(int world, int lead) { return (world > 1 ? world + 3 : 2) + lead; }

long sgm_tile_steps_total(long n_frames, int world, int lead) { return n_frames + world + 2 + lead; }

#define TRY(call) do { const int rc_ = (call); if (rc_ != 0) return rc_; } while (0)

int sgm_tile_step(const sgm_tile_engine* e, int rank, int world, int height, int slots, int lead, long step, long frames_known)
{
    const int r = rank, N = world;
    const long F = frames_known;
#define VALID(f) ((f) >= 0 && (f) < F)
#define SLOT(f) ((int)((f) % slots))
    if (!e || N < 1 || r < 0 || r >= N || lead < 0 || slots < sgm_tile_slots_needed(N, lead) || step < 0) return -1;
    const int lag = r > N - 1 - r ? r : N - 1 - r;
    int touched[8], n_touched = 0;
#define TOUCH(sl) do { int seen_ = 0; for (int i_ = 0; i_ < n_touched; ++i_) seen_ |= touched[i_] == (sl); if (!seen_) touched[n_touched++] = (sl); } while (0)

    if (VALID(step)) TRY(e->begin(e->user, SLOT(step), step));
    const long s = step - lead;                    /* everything below runs `lead` steps behind the begins */
    const long f = s - r, g = s - (N - 1 - r);     /* the frames whose forward / backward sweep reaches this rank now */
    for (int pass = 0; pass < 2; ++pass) {
        const int forward = pass == 0;
        const long fr = forward ? f : g;
        if (!VALID(fr)) continue;
        const int first = forward ? (r == 0) : (r == N - 1);
        const int last = forward ? (r == N - 1) : (r == 0);
        if (!first) TRY(e->import_boundary(e->user, SLOT(fr), forward));
        TRY(e->sweep(e->user, SLOT(fr), forward));
        if (!last) TRY(e->export_boundary(e->user, SLOT(fr), forward));
    }
    /* the exchange between step s and s + 1: boundary operations first, then the row gather (neighbouring ranks list the
     * operations between them in the same order) */
    if (N > 1) {
        sgm_tile_xop ops[4 + 64];
        int n = 0;
        if (N - 1 > 64) return -1;
        if (VALID(f) && r < N - 1) {
            ops[n++] = (sgm_tile_xop){SGM_XOP_SEND, SGM_XBUF_BOUNDARY, SLOT(f), 1, 0, 0, 0, r + 1};
            TOUCH(SLOT(f));
        }
        if (VALID(s + 1 - r) && r > 0) {
            ops[n++] = (sgm_tile_xop){SGM_XOP_RECV, SGM_XBUF_BOUNDARY, SLOT(s + 1 - r), 1, 1, 0, 0, r - 1};
            TOUCH(SLOT(s + 1 - r));
        }
        if (VALID(g) && r > 0) {
            ops[n++] = (sgm_tile_xop){SGM_XOP_SEND, SGM_XBUF_BOUNDARY, SLOT(g), 0, 0, 0, 0, r - 1};
            TOUCH(SLOT(g));
        }
        if (VALID(s + 1 - (N - 1 - r)) && r < N - 1) {
            ops[n++] = (sgm_tile_xop){SGM_XOP_RECV, SGM_XBUF_BOUNDARY, SLOT(s + 1 - (N - 1 - r)), 0, 1, 0, 0, r + 1};
            TOUCH(SLOT(s + 1 - (N - 1 - r)));
        }
        const long h = s - N;                      /* finished on every rank in an earlier step */
        if (VALID(h)) {
            const int owner = (int)(h % N);
            TOUCH(SLOT(h));
            int r0, r1;
            if (r != owner) {
                sgm_tile_rows(height, N, r, &r0, &r1);
                ops[n++] = (sgm_tile_xop){SGM_XOP_SEND, SGM_XBUF_ROWS, SLOT(h), 0, 0, r0, r1, owner};
            } else {
                for (int k = 0; k < N; ++k) {
                    if (k == r) continue;
                    sgm_tile_rows(height, N, k, &r0, &r1);
                    ops[n++] = (sgm_tile_xop){SGM_XOP_RECV, SGM_XBUF_ROWS, SLOT(h), 0, 0, r0, r1, k};
                }
            }
        }
        if (n > 0) {
            /* ascending slot order, as the Python schedule lists them (an engine may rely on nothing but the set) */
            for (int i = 1; i < n_touched; ++i)
                for (int j = i; j > 0 && touched[j - 1] > touched[j]; --j) { const int t = touched[j]; touched[j] = touched[j - 1]; touched[j - 1] = t; }
            TRY(e->exchange(e->user, ops, n, touched, n_touched));
        }
    }
    if (VALID(s - lag)) TRY(e->finish(e->user, SLOT(s - lag)));
    const long p = N > 1 ? s - N - 1 : s;          /* one rank: nothing to gather, the post pass follows finish */
    if (VALID(p) && p % N == r) TRY(e->post(e->user, SLOT(p), p));
    return 0;
#undef VALID
#undef SLOT
#undef TOUCH
}

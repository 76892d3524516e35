/*
 * sgm_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement of the reference SGM pipeline
 * (/root/reference/SemiGlobalMatching/SemiGlobalMatching/SemiGlobalMatching.c),
 * written from scratch with run-time sizes.  It is the checker for the HIP
 * path: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * may load it.  The product library (libsgm_mi355x.so) never links or calls it.
 *
 * Parity status: PINNED.  tests/test_oracle_*.py check every stage of this
 * restatement against (a) the reference compiled from its own sources into
 * oracle/_ref/ (when present), (b) the golden vectors under tests/golden/
 * produced by that build, (c) the nine stage digests of SURVEY.md 8(c) on the
 * cone pair, and (d) the reference's committed output Data/cone/im2.d.png.
 *
 * Semantics = the reference's, with its one piece of undefined behaviour
 * defined away: a diagonal path step whose pixel lies outside the image is
 * dropped (it is always the last step of its line) -- SURVEY.md Q6.
 */
#ifndef SGM_ORACLE_H
#define SGM_ORACLE_H

#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Same field order/types as the reference SGMOption (SemiGlobalMatching.h:24-40);
 * 28 bytes on x86-64 SysV. */
typedef struct {
    uint8_t  num_paths;
    uint16_t min_disparity;
    uint16_t max_disparity;
    bool     is_check_unique;
    float    uniqueness_ratio;
    bool     is_check_lr;
    float    lrcheck_thres;
    bool     is_remove_speckles;
    uint16_t min_speckle_area;
    int16_t  p1;
    int16_t  p2_init;
} sgmo_option;

/* ---- individual stages (each cites the reference lines it restates) ---- */

/* SemiGlobalMatching.c:134-159.  Border (2 px) is written as 0 (Q3). */
void sgmo_census5x5(const uint8_t* img, int W, int H, uint32_t* census);            /* into a zeroed buffer: border 0 */
void sgmo_census5x5_interior(const uint8_t* img, int W, int H, uint32_t* census);   /* the stores of ref :134-159 only */

/* Extension, pinned by this restatement only: the census transform for any odd window cw x ch of at most 64 pixels
 * (u64 words; equals sgmo_census5x5 for 5x5) and its Hamming cost. */
void sgmo_census_window(const uint8_t* img, int W, int H, int cw, int ch, uint64_t* census);
void sgmo_cost64(const uint64_t* cl, const uint64_t* cr, int W, int H, int dmin, int dmax, uint8_t* cost);

/* SemiGlobalMatching.c:161-196.  cost[(y*W+x)*D + (d-dmin)]. */
void sgmo_cost(const uint32_t* cl, const uint32_t* cr, int W, int H,
               int dmin, int dmax, uint8_t* cost);

/* Path geometry of one line (SemiGlobalMatching.c:238-255,281-323,359-367).
 * Writes the linear pixel index of every visited pixel (start pixel first) into pix[]
 * (capacity >= max(W,H)) and returns how many there are; an out-of-image step ends the
 * line (Q6).  dx,dy in {-1,0,1}, not both 0. */
int sgmo_path_walk(int W, int H, int dx, int dy, int line, int32_t* pix);

/* Number of lines of a direction (H for horizontal, W otherwise; .c:238). */
int sgmo_path_lines(int W, int H, int dx, int dy);

/* SemiGlobalMatching.c:229-372 for one direction.  Adds L_r into S (uint16, wraps).
 * If L_last != NULL it receives, per cell, the L_r of the LAST visit of that pixel in
 * this direction (cells never visited keep their previous content).
 * If visits != NULL (W*H bytes) it is incremented per pixel visit. */
void sgmo_aggregate_dir(const uint8_t* img, const uint8_t* cost, int W, int H, int D,
                        int p1, int p2_init, int dx, int dy,
                        uint16_t* S, uint8_t* L_last, uint8_t* visits);

/* SemiGlobalMatching.c:198-221: the fixed direction order. n_dirs = 8 (reference) or 4. */
void sgmo_aggregate_all(const uint8_t* img, const uint8_t* cost, int W, int H, int D,
                        int p1, int p2_init, int n_dirs, uint16_t* S);

/* SemiGlobalMatching.c:374-443.  right_view=0 -> left WTA, 1 -> right-view WTA. */
void sgmo_wta(const uint16_t* S, int W, int H, int dmin, int dmax,
              bool check_unique, float uniqueness_ratio, int right_view, float* disp);

/* SemiGlobalMatching.c:445-470 (in place on disp_left). */
void sgmo_lrcheck(float* disp_left, const float* disp_right, int W, int H, float thres);

/* Extension: the mirror image of the LR check with the right view as the reference view (in place on disp_right). */
void sgmo_lrcheck_right(float* disp_right, const float* disp_left, int W, int H, float thres);

/* SemiGlobalMatching.c:585-642 (diff_insame = 1 in the reference call, .c:115). */
void sgmo_remove_speckles(float* disp, int W, int H, float diff_insame, unsigned min_area);

/* SemiGlobalMatching.c:525-557 called with in == out (.c:120): raster-order recurrence (Q13). */
void sgmo_median3_inplace(float* disp, int W, int H);

/* main.c:92-117: min/max normalisation of valid disparities to 8 bit. */
void sgmo_normalize_u8(const float* disp, int W, int H, uint8_t* out);

/* ---- whole pipeline, mirroring SGM_Initialize / SGM_Reset / SGM_Match ---- */

typedef struct sgmo_ctx sgmo_ctx;

/* honor_num_paths: 0 = reference behaviour (num_paths ignored, always 8; Q1),
 * 1 = num_paths==4 runs only the first four directions. */
sgmo_ctx* sgmo_create(void);
void      sgmo_destroy(sgmo_ctx* c);
void      sgmo_set_honor_num_paths(sgmo_ctx* c, int honor);
/* extensions beyond the reference, each defined by this restatement alone:
 * census window cw x ch (odd, <= 64 pixels; 5x5 = reference; stages 0/1 are then uint64 words) and the view the
 * result is referenced to (0 left = reference; 1 right: stage 6.. hold the right view's map, LR-checked against the left) */
bool      sgmo_set_census_window(sgmo_ctx* c, int cw, int ch);
void      sgmo_set_reference_view(sgmo_ctx* c, int right);
bool      sgmo_initialize(sgmo_ctx* c, uint16_t width, uint16_t height, const sgmo_option* opt);
bool      sgmo_reset(sgmo_ctx* c, uint16_t width, uint16_t height, const sgmo_option* opt);
bool      sgmo_match(sgmo_ctx* c, const uint8_t* left, const uint8_t* right, float* disp_left);
/* The context's census buffers behave like the reference's statics (SemiGlobalMatching.h:67-68): zero when the
 * context is created, never cleared by sgmo_reset, written by sgmo_match in the interior only -- so after a Reset to another
 * shape the 2-pixel border (every pixel for W <= 5 or H <= 5) holds what earlier frames left at the same linear index, exactly
 * as a sequence of SGM_Reset / SGM_Match calls on the reference does.  sgmo_clear_census = a new process. */
void      sgmo_clear_census(sgmo_ctx* c);

/* Stage buffers of the last sgmo_match (valid until the next call / destroy).
 * which: 0 censusL(u32) 1 censusR(u32) 2 cost(u8) 3 aggr(u16) 4 dispL after WTA(f32)
 *        5 dispR(f32) 6 after LR(f32) 7 after speckle(f32) 8 final(f32). */
const void* sgmo_stage(const sgmo_ctx* c, int which, size_t* bytes);

/* Counters of the last match: [0] path lines ended by the out-of-image guard (Q6), [1] uint8 wraps of L_r (Q7). */
void sgmo_counters(const sgmo_ctx* c, uint64_t out[2]);

/* Synthetic stereo pair of SURVEY.md 8(d): LCG noise, 2x2 smoothing, slanted-plane disparity. */
void sgmo_synth_pair(int W, int H, int D, uint32_t seed, uint8_t* left, uint8_t* right);

#ifdef __cplusplus
}
#endif
#endif

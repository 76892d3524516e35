/*
 * sgm_mi355x.h -- C-ABI of libsgm_mi355x.so, the MI355X (gfx950) drop-in for the reference's
 * Semi-Global-Matching library.
 *
 * Part 1 is the reference boundary itself: the same three entry points with the same SGMOption
 * layout, argument meaning and error behaviour as
 *   /root/reference/SemiGlobalMatching/SemiGlobalMatching/SemiGlobalMatching.h:24-40 (SGMOption)
 *   /root/reference/SemiGlobalMatching/SemiGlobalMatching/SemiGlobalMatching.h:78-80 (functions)
 * so a caller such as the reference's main.c:72,83 links against this library unchanged.
 *
 * Part 2 are extensions the reference does not have (device-resident buffers, several
 * independent instances, stage read-back for parity tests).  Plain C types only.
 *
 * Results: disparity in pixels with sub-pixel fraction, invalid = +INFINITY
 * (SemiGlobalMatching.h:12), bit-identical to the reference's C code for every defined input
 * (see DESIGN.md "Parity contract" for the one undefined behaviour of the reference that is
 * defined here).
 */
#ifndef SGM_MI355X_H
#define SGM_MI355X_H

#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------------------------------
 * Part 1 -- the reference boundary
 * ---------------------------------------------------------------------------------------- */

/* replaces SemiGlobalMatching.h:24-40; 28 bytes, align 4 on x86-64 SysV */
typedef struct {
    uint8_t  num_paths;          /* ignored by the reference (always 8 directions); see SGM_SetHonorNumPaths */
    uint16_t min_disparity;
    uint16_t max_disparity;      /* search range is [min_disparity, max_disparity) */

    bool     is_check_unique;
    float    uniqueness_ratio;

    bool     is_check_lr;
    float    lrcheck_thres;

    bool     is_remove_speckles;
    uint16_t min_speckle_area;

    int16_t  p1;
    int16_t  p2_init;
} SGMOption;

/* replaces SemiGlobalMatching.h:78 / SemiGlobalMatching.c:37-66.
 * false for width==0, height==0, max_disparity<=min_disparity (as the reference) and, in addition,
 * when no gfx950 device is usable, a HIP call fails, or the disparity range exceeds
 * SGM_MAX_DISPARITY_RANGE (the reason is printed to stderr).  Sizes device buffers for this
 * shape and marks the aggregated-cost volume as zero. */
bool SGM_Initialize(uint16_t width, uint16_t height, const SGMOption* option);

/* replaces SemiGlobalMatching.h:79 / SemiGlobalMatching.c:128-132 (clear + Initialize). */
bool SGM_Reset(uint16_t width, uint16_t height, const SGMOption* option);

/* replaces SemiGlobalMatching.h:80 / SemiGlobalMatching.c:68-125.
 * img_left/img_right: host pointers, row-major uint8, stride = width, borrowed for the call.
 * disp_left: host pointer to width*height floats, fully overwritten.  Blocking.
 * false if not initialised or an image pointer is NULL (as the reference) or a HIP call failed.
 * Like the reference a second SGM_Match without SGM_Reset accumulates onto the
 * aggregated costs of the previous frame; call SGM_Reset per frame. */
bool SGM_Match(const uint8_t* img_left, const uint8_t* img_right, float* disp_left);

/* ------------------------------------------------------------------------------------------
 * Part 2 -- extensions
 * ---------------------------------------------------------------------------------------- */

#define SGM_MAX_DISPARITY_RANGE 512

/* The default instance used by Part 1 can be moved to another GPU of the node before
 * SGM_Initialize (default: device 0, or the SGM_DEVICE environment variable). */
bool SGM_SetDevice(int device_ordinal);

/* 0 (default) = reference behaviour: num_paths is ignored, all 8 directions run.
 * 1 = num_paths == 4 runs only the first four directions of SemiGlobalMatching.c:213-216.
 * Takes effect at the next SGM_Initialize / SGM_Reset. */
void SGM_SetHonorNumPaths(int honor);

/* Two options the reference does not have.  Neither is pinned by the reference: the CPU oracle
 * (oracle/sgm_oracle.c) defines them, "parity unpinned by the reference".  Both take effect at the next SGM_Initialize /
 * SGM_Reset; the defaults are the reference's behaviour.
 *   census window: any odd width x height of at most 64 pixels (e.g. 7x7, 9x7) instead of SemiGlobalMatching.c:134-159's
 *     5x5 -- same bit order (raster, first comparison in the highest bit, centre included), border of width/2 columns
 *     and height/2 rows zero, off-image cost 127.  Wide windows take the materialised-cost path (u64 census words, cost
 *     volume, volume-fed aggregation kernels): correct, but not the fused fast path of 5x5.
 *   reference view: 1 = the result is the RIGHT image's disparity map (the right-view winner-take-all of
 *     SemiGlobalMatching.c:395-408), validated by the mirror image of LRCheck (.c:445-470: right pixel x with disparity d
 *     must agree with the left map at column x + d), then speckle removal and median as usual.  0 = left (reference). */
bool SGM_SetCensusWindow(int width, int height);
void SGM_SetReferenceView(int right);

/* Same as SGM_Match but all three pointers are DEVICE pointers (HBM-resident frames) on the
 * instance's device.  Asynchronous on the instance's stream; SGM_Synchronize waits. */
bool SGM_MatchDevice(const uint8_t* d_left, const uint8_t* d_right, float* d_disp_left);
bool SGM_Synchronize(void);

/* Releases every device resource of the default instance (the reference has no counterpart). */
void SGM_Shutdown(void);

/* The one-call form BASELINE.json's north_star names ("sgm_compute(left, right, params -> disparity)"; the reference
 * itself has no such symbol, its boundary is the three functions above): SGM_Reset(width, height, option) followed by
 * SGM_Match(img_left, img_right, disp_left) on the default instance -- the per-frame sequence of main.c:72,83 and
 * of SURVEY.md Q14.  Host pointers, blocking, same results and the same true/false behaviour as the two calls. */
bool sgm_compute(const uint8_t* img_left, const uint8_t* img_right, uint16_t width, uint16_t height,
                 const SGMOption* option, float* disp_left);

/* ---- explicit instances: several frames in flight on one GPU, one per HIP stream ---- */
typedef struct sgm_instance sgm_instance;

sgm_instance* sgm_create(int device_ordinal);                 /* NULL on failure */
void          sgm_destroy(sgm_instance* s);
void          sgm_set_honor_num_paths(sgm_instance* s, int honor);
bool          sgm_set_census_window(sgm_instance* s, int width, int height);   /* see SGM_SetCensusWindow */
void          sgm_set_reference_view(sgm_instance* s, int right);              /* see SGM_SetReferenceView */
bool          sgm_initialize(sgm_instance* s, uint16_t width, uint16_t height, const SGMOption* option);
bool          sgm_reset(sgm_instance* s, uint16_t width, uint16_t height, const SGMOption* option);
bool          sgm_match(sgm_instance* s, const uint8_t* img_left, const uint8_t* img_right, float* disp_left);
bool          sgm_match_device(sgm_instance* s, const uint8_t* d_left, const uint8_t* d_right, float* d_disp_left);
bool          sgm_synchronize(sgm_instance* s);
/* Pipelined host-pointer matches (SGM_Match's H2D / kernels / D2H of SemiGlobalMatching.c:77-78,122 without the
 * blocking wait): sgm_match_async stages the images, queues the upload, the pipeline and the download on the instance's
 * stream and returns; the three buffers stay borrowed until sgm_match_wait (or the next sgm_match_async / sgm_initialize /
 * sgm_reset / sgm_destroy on the same instance, which wait implicitly) has handed the result over.  A caller that
 * round-robins frames over two or three instances overlaps the copies of one with the kernels of the others.
 * sgm_match == sgm_match_async + sgm_match_wait.  Buffers from sgm_host_alloc (page-locked) are used in place: no
 * staging copy on either side; any other host pointer is staged through the instance's own pinned buffers. */
bool          sgm_match_async(sgm_instance* s, const uint8_t* img_left, const uint8_t* img_right, float* disp_left);
bool          sgm_match_wait(sgm_instance* s);
void*         sgm_host_alloc(sgm_instance* s, size_t bytes);   /* page-locked host memory on the instance's device; NULL on failure */
void          sgm_host_free(sgm_instance* s, void* p);
/* Throughput option for a stream of matches on ONE instance: with sgm_set_overlap_post(s, 1) the post pass of a match (LR
 * check, speckle removal, median: latency-bound kernels that occupy a fraction of the GPU) runs on a second stream of the
 * instance, ordered behind the match's cost sum by an event, while sgm_stream(s) goes on with the census and aggregation of
 * the next match; the next cost sum waits for the post pass that still reads the shared maps.  Results are unchanged.  What
 * changes: the disparity map of sgm_match_device is complete after sgm_synchronize (or sgm_match_wait for the host-pointer
 * form), no longer in stream order of sgm_stream(s).  Off by default; ignored in row-tile mode. */
bool          sgm_set_overlap_post(sgm_instance* s, int enable);
/* Stage groups on streams -- and compute units -- of their own.  A match is three groups of kernels with different appetites:
 * census + path aggregation (VALU-bound, SemiGlobalMatching.c:82-94), cost sum + both winner-take-all passes (HBM-bound,
 * .c:94-105), LR check + speckle removal + median (latency-bound, a few workgroups, .c:109-120).  When several matches are in
 * flight on one GPU (two instances, or a stream of matches on one with sgm_set_overlap_post) kernels of different groups share
 * compute units and slow each other down far more than they gain from the sharing -- the median's serial kernel runs three
 * times longer next to an aggregation's waves than alone.  sgm_set_stage_cus(s, which, first, count) gives group `which` a HIP
 * stream of its own whose kernels run only on CUs [first, first + count) of EVERY XCD (MI355X: 32 CUs in each of 8 XCDs; each
 * XCD keeps its L2 and its path to HBM in play); the groups are ordered by events, results are unchanged.
 *   count > 0   own stream on those CUs        count == 0  own stream, all CUs       count < 0  back to the default
 *   which = SGM_STAGE_MAIN: the stream census + aggregation (and every group without a stream of its own) run on; re-created,
 *   so sgm_stream(s) changes: a caller that cached the handle (the tile pipeline's slots do) must query it again, the old one
 *   is destroyed.   SGM_STAGE_POST with count == 0 is sgm_set_overlap_post(s, 1).
 * As with sgm_set_overlap_post a result is complete after sgm_synchronize / sgm_match_wait, not in stream order of
 * sgm_stream(s).  Ignored in row-tile mode.  The instance must be idle (the call waits for it). */
enum { SGM_STAGE_MAIN = 0, SGM_STAGE_SUM = 1, SGM_STAGE_POST = 2 };
bool          sgm_set_stage_cus(sgm_instance* s, int which, int first_cu_per_xcd, int cus_per_xcd);
/* The same with a dispatch priority instead of a CU range: the group gets a stream of its own on all CUs whose waiting
 * workgroups the dispatcher serves before (priority < 0) or after (> 0) those of normal streams (clamped to the device's
 * range).  Replaces a CU range set before; a later sgm_set_stage_cus (count >= 0) replaces the priority stream in turn, and
 * count < 0 goes back to the default in either case. */
bool          sgm_set_stage_priority(sgm_instance* s, int which, int priority);
/* The HIP stream (hipStream_t as void*) the instance launches on, e.g. to record events. */
void*         sgm_stream(sgm_instance* s);
/* Whether the LAST match ran the fused last sweep (csrc/sgm_upsum.hip: the three upward directions computed inside the cost-sum /
 * winner-take-all kernel, their planes never written): image rows per workgroup of that kernel, 0 = the separate kernels.  It is used
 * for batches of whole frames with W > H, a padded range of 128, eight paths and P1 >= 0 when SGM_UPSUM allows it, and only by
 * matches that neither add to an earlier S (Q14) nor keep stages; results are identical either way. */
int           sgm_fused_sweep_rows(const sgm_instance* s);

/* Batches: after sgm_set_batch(s, n) and the next sgm_initialize / sgm_reset, every sgm_match /
 * sgm_match_device call processes n frames of the same shape stored back to back ([n][H][W] for the
 * images and the disparity output) -- each kernel of the pipeline covers all n frames in one launch,
 * which is how one GPU is filled when frames are small.  n = 1 (default) is the reference behaviour. */
bool          sgm_set_batch(sgm_instance* s, int frames);
/* which frame of the batch sgm_read_stage returns (default 0) */
void          sgm_select_frame(sgm_instance* s, int frame);

/* ---- row tiles: one frame over several GPUs (each GPU one instance, one process per GPU) ----
 * After sgm_set_rows(s, r0, r1) and the next sgm_initialize / sgm_reset the instance computes rows [r0, r1) of
 * the frame (r1 = 0 returns to whole frames).  Per frame, on every GPU (all calls asynchronous on sgm_stream):
 *
 *     sgm_tile_begin(s, d_left, d_right)          census, horizontal paths of the tile, anomalous diagonals
 *     forward sweep, tiles top to bottom:          backward sweep, tiles bottom to top:
 *       [sgm_tile_import_boundary(s, 1, buf)]        [sgm_tile_import_boundary(s, 0, buf)]   not at the frame edge
 *       sgm_tile_sweep(s, 1)                         sgm_tile_sweep(s, 0)
 *       sgm_tile_export_boundary(s, 1, buf)          sgm_tile_export_boundary(s, 0, buf)     -> next GPU
 *     sgm_tile_finish(s, d_disp)                   cost sum, WTA (both views), LR check -> rows [r0, r1) of d_disp
 *     (gather the rows of all GPUs into one [H][W] map)
 *     sgm_tile_post(s, d_disp)                     speckle removal + median on the whole frame
 *
 * d_left / d_right are the whole images (replicated); buf holds sgm_tile_boundary_bytes(s) bytes of device
 * memory: the path costs of one image row for the 3 directions of a sweep (1 with four paths).  The two
 * sweeps are independent of each other.  The result is bit-identical to sgm_match on one GPU.
 * With sgm_set_batch(s, B) every call covers the same rows of B frames ([B][H][W] images and maps; the hand-over buffer holds
 * B such rows, frame-major): one launch per stage for the B tiles -- a tile of a single frame leaves most of a GPU idle. */
bool   sgm_set_rows(sgm_instance* s, int row_begin, int row_end);
bool   sgm_tile_begin(sgm_instance* s, const uint8_t* d_left, const uint8_t* d_right);
size_t sgm_tile_boundary_bytes(const sgm_instance* s);
bool   sgm_tile_import_boundary(sgm_instance* s, int forward, const void* d_buf);
bool   sgm_tile_sweep(sgm_instance* s, int forward);
bool   sgm_tile_export_boundary(sgm_instance* s, int forward, void* d_buf);
bool   sgm_tile_finish(sgm_instance* s, float* d_disp_left);
bool   sgm_tile_post(sgm_instance* s, float* d_disp_left);

/* ---- test-platform arithmetic on device buffers ----
 * What the reference's host platform does with a returned map (HostScript_Server/depth_image.py:138-165 disparity_to_depth,
 * :276-319 compare_img), for maps that are already in HBM.  Restated from reading: that module imports cv2, which is not
 * installed where this library is built, so no reference-made vectors exist -- "parity unpinned" (tests compare with the
 * host restatement soc_project_stereo_matching_amd/platform.py).
 *   depth[mm] = float32(fx * baseline) / (disparity + doffs); a non-finite or zero denominator (invalid = +INF) -> NaN.
 *   compare (blocking): over the pixels finite in BOTH images: rmse = sqrt(mean((test - gt)^2)), bad_pixel_rate = share
 *   with |test - gt| > abs_thresh [mm], n_valid; (NaN, NaN, 0) if there is none.  Asynchronous / blocking on sgm_stream(s). */
bool   sgm_disparity_to_depth(sgm_instance* s, const float* d_disparity, size_t count, float fx, float baseline, float doffs,
                              float* d_depth);
bool   sgm_compare_depth(sgm_instance* s, const float* d_ground_truth, const float* d_test, size_t count, float abs_thresh,
                         double* rmse, double* bad_pixel_rate, uint64_t* n_valid);

/* ---- a test-platform frame end to end (the data formats either side of the path) ----
 * The server hands the board six byte planes per frame -- left B, G, R, right B, G, R, each h rows of w bytes
 * (HostScript_Server/server.py:105-131; received into frame_buffer.h:16-51 by tcp_perf_client.c:181-189) -- and expects h rows
 * of w float32 depth in mm back (message type 3, server.py:148-177).  The firmware's grey conversion is
 * (76 r + 150 g + 29 b) >> 8 (stereo_matching.c:18-25; stb's, behind main.c's image load, uses 77: weight_r selects).
 *   sgm_gray_from_planes   device buffers: three planes B, G, R of `count` bytes each at d_bgr -> d_gray; on sgm_stream(s).
 *   sgm_match_planes_async host buffers: queues H2D of the six planes (B frames of six with sgm_set_batch), both grey
 *                          conversions, the match, disparity -> depth (as sgm_disparity_to_depth) and D2H of the depth map,
 *                          then returns; sgm_match_wait hands the map over.  Pinned buffers (sgm_host_alloc) are used in
 *                          place.  The disparity map itself stays readable with sgm_read_stage(s, 8, ...).
 *   sgm_match_planes       = sgm_match_planes_async + sgm_match_wait. */
bool   sgm_gray_from_planes(sgm_instance* s, const uint8_t* d_bgr, size_t count, int weight_r, uint8_t* d_gray);
bool   sgm_match_planes_async(sgm_instance* s, const uint8_t* planes, float fx, float baseline, float doffs, float* depth);
bool   sgm_match_planes(sgm_instance* s, const uint8_t* planes, float fx, float baseline, float doffs, float* depth);

/* ---- stage read-back (parity tests; copies device -> host, blocking) ----
 * which: 0 census left (u32 [H][W])       1 census right (u32 [H][W])    (u64 words with a wide census window)
 *        2 matching cost (u8 [H][W][D])   3 aggregated cost S (u16 [H][W][D])
 *        4 left disparity after WTA       5 right-view disparity
 *        6 after LR check                 7 after speckle removal        8 final (all f32 [H][W])
 *        10..17 per-direction path cost L_r of direction (which-10) (u8 [H][W][D]; cells the
 *               direction never visits read 0, cells visited twice hold the last-but-one visit)
 * Returns the number of bytes written, 0 on error or if `capacity` is too small. */
size_t sgm_read_stage(sgm_instance* s, int which, void* host_out, size_t capacity);
size_t SGM_ReadStage(int which, void* host_out, size_t capacity);
/* Stages 4, 6 and 7 are overwritten in place by the following stage and stage 2 (the cost volume) is
 * normally never materialised (the aggregation kernel recomputes it from the census images); enable
 * keeping them (one extra kernel + three device-to-device copies per match) before the match whose
 * stages are read. */
void   sgm_keep_stages(sgm_instance* s, int enable);
void   SGM_KeepStages(int enable);

/* Per-kernel device time (ms) of the last match of the instance, measured with HIP events on
 * the instance's stream when timing is enabled.  names[i] points to static strings.
 * Returns the number of entries written (<= max_entries). */
void   sgm_enable_timing(sgm_instance* s, int enable);
int    sgm_last_timing(sgm_instance* s, const char** names, float* ms, int max_entries);
/* Mean and minimum per-kernel time over every match since timing was (re-)enabled and collected by
 * sgm_synchronize (up to 64 matches between two synchronizes are kept); *matches = how many. */
int    sgm_mean_timing(sgm_instance* s, const char** names, float* mean_ms, float* min_ms, int max_entries, long* matches);

/* Seeded synthetic stereo pair of SURVEY.md 8(d) (host buffers of width*height bytes each). */
void   SGM_SynthPair(int width, int height, int disparity_range, uint32_t seed, uint8_t* left, uint8_t* right);

/* Library build info, e.g. "sgm_mi355x 0.1 gfx950". */
const char* SGM_Version(void);

#ifdef __cplusplus
}
#endif
#endif /* SGM_MI355X_H */

/*
 * ref_harness_tail.c -- TEST INFRASTRUCTURE ONLY.  Compiled BEHIND the reference translation unit
 * in the same compilation (see build_ref.sh), so it can reach the reference's static stage
 * functions and its global instance `sgm`.  It replays the body of SGM_Match
 * (SemiGlobalMatching.c:77-122) one stage at a time and copies every intermediate buffer out.
 */
#include <string.h>

int ref_capacity(int out[3])
{
    out[0] = MAX_IMG_WIDTH; out[1] = MAX_IMG_HEIGHT; out[2] = MAX_DISPARITY_RANGE;
    return 0;
}

unsigned long ref_oob_count(void) { return ref_oob_dropped; }

/* The reference never writes the 2-px census border and relies on zero-initialised statics (Q3):
 * after a run at another width the buffers hold stale values there.  Call before SGM_Reset when
 * the shape changed. */
void ref_clear_census(void)
{
    memset(census_left_buffer, 0, sizeof census_left_buffer);
    memset(census_right_buffer, 0, sizeof census_right_buffer);
}

static int ref_fits(uint16_t w, uint16_t h, const SGMOption* o)
{
    return w <= MAX_IMG_WIDTH && h <= MAX_IMG_HEIGHT && (size_t)w * h <= (size_t)MAX_IMG_SIZE &&
           o->max_disparity > o->min_disparity && (o->max_disparity - o->min_disparity) <= MAX_DISPARITY_RANGE;
}

/* Every out pointer may be NULL.  Returns 0 on success.
 * first_dirs < 0: the reference's CostAggregation() as it stands (all eight CostAggregate calls, SemiGlobalMatching.c:213-220).
 * first_dirs = n: only the FIRST n of those eight calls, in the reference's order -- what "num_paths == 4" means in this project
 * (the reference never reads num_paths; the 4-path mode is defined as its first four calls, :213-216).  Every stage is
 * still the reference's own function; only the selection of CostAggregate calls is the harness's. */
static int run_stages(const uint8_t* left, const uint8_t* right, uint16_t w, uint16_t h, const SGMOption* opt, int first_dirs,
                      uint32_t* census_l, uint32_t* census_r, uint8_t* cost, uint16_t* aggr,
                      float* disp_l, float* disp_r, float* after_lr, float* after_speckle, float* final)
{
    if (!ref_fits(w, h, opt)) return -1;
    ref_oob_dropped = 0;
    /* the reference relies on zero-initialised statics for the census border (Q3) */
    ref_clear_census();
    if (!SGM_Reset(w, h, opt)) return -2;
    const size_t px = (size_t)w * h, cells = px * sgm.disp_range;
    sgm.img_left = left;
    sgm.img_right = right;
    census_transform_5x5(sgm.img_left, sgm.census_left);
    census_transform_5x5(sgm.img_right, sgm.census_right);
    if (census_l) memcpy(census_l, sgm.census_left, px * 4);
    if (census_r) memcpy(census_r, sgm.census_right, px * 4);
    ComputeCost(sgm.census_left, sgm.census_right, sgm.cost_init);
    if (cost) memcpy(cost, sgm.cost_init, cells);
    if (first_dirs < 0) CostAggregation();
    else {
        static const int8_t dir[8][2] = {{1, 0}, {-1, 0}, {0, 1}, {0, -1}, {1, 1}, {-1, -1}, {1, -1}, {-1, 1}};   /* .c:213-220 */
        for (int d = 0; d < first_dirs && d < 8; ++d) CostAggregate(sgm.img_left, sgm.cost_init, sgm.cost_aggr, dir[d][0], dir[d][1]);
    }
    if (aggr) memcpy(aggr, sgm.cost_aggr, cells * 2);
    ComputeDisparity(sgm.cost_aggr, sgm.disp_left, 0);
    if (disp_l) memcpy(disp_l, sgm.disp_left, px * 4);
    if (sgm.option.is_check_lr) {
        ComputeDisparity(sgm.cost_aggr, sgm.disp_right, 1);
        if (disp_r) memcpy(disp_r, sgm.disp_right, px * 4);
        LRCheck(sgm.disp_left, sgm.disp_right);
    }
    if (after_lr) memcpy(after_lr, sgm.disp_left, px * 4);
    if (sgm.option.is_remove_speckles) RemoveSpeckles(sgm.disp_left, 1);
    if (after_speckle) memcpy(after_speckle, sgm.disp_left, px * 4);
    MedianFilter(sgm.disp_left, sgm.disp_left, FILTER_WINDOW_SIZE);
    if (final) memcpy(final, sgm.disp_left, px * 4);
    return 0;
}

int ref_run_stages(const uint8_t* left, const uint8_t* right, uint16_t w, uint16_t h, const SGMOption* opt,
                   uint32_t* census_l, uint32_t* census_r, uint8_t* cost, uint16_t* aggr,
                   float* disp_l, float* disp_r, float* after_lr, float* after_speckle, float* final)
{
    return run_stages(left, right, w, h, opt, -1, census_l, census_r, cost, aggr, disp_l, disp_r, after_lr, after_speckle, final);
}

int ref_run_stages_first_dirs(const uint8_t* left, const uint8_t* right, uint16_t w, uint16_t h, const SGMOption* opt, int first_dirs,
                              uint32_t* census_l, uint32_t* census_r, uint8_t* cost, uint16_t* aggr,
                              float* disp_l, float* disp_r, float* after_lr, float* after_speckle, float* final)
{
    if (first_dirs < 0 || first_dirs > 8) return -3;
    return run_stages(left, right, w, h, opt, first_dirs, census_l, census_r, cost, aggr, disp_l, disp_r, after_lr, after_speckle, final);
}

/* One direction of CostAggregate on a caller-supplied cost volume, starting from S = 0. */
int ref_aggregate_dir(const uint8_t* img, const uint8_t* cost, uint16_t w, uint16_t h, const SGMOption* opt,
                      int dx, int dy, uint16_t* aggr)
{
    if (!ref_fits(w, h, opt)) return -1;
    ref_oob_dropped = 0;
    if (!SGM_Reset(w, h, opt)) return -2;
    const size_t cells = (size_t)w * h * sgm.disp_range;
    memcpy(sgm.cost_init, cost, cells);
    CostAggregate(img, sgm.cost_init, sgm.cost_aggr, (int8_t)dx, (int8_t)dy);
    memcpy(aggr, sgm.cost_aggr, cells * 2);
    return 0;
}

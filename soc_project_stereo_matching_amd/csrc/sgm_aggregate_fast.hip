#include "sgm_aggregate_impl.hpp"

// The aggregation kernels for ordinary penalties (0 <= P1 <= 31488, max(P1, P2_init) <= 223 -- the reference's defaults are 10 and
// 150, main.c:64-65): agg_step_nn with the FAST shortcuts (v_pk_minimum3_f16 as a three-way unsigned minimum, no uint8 mask away
// from the left border).  Their own translation unit so that the three step families build in parallel.
bool sgmd_aggregate_launch_fast(int lpp, int dpl, const AggArgs* a, int blocks, int pad, int hl, hipStream_t st)
{
    return launch_aggregate_key<2>(lpp, dpl, *a, blocks, pad != 0, hl, st);
}

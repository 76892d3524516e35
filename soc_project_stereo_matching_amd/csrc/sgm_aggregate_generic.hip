// The aggregation kernels with the generic step (any sign of P1).  Only negative P1 -- which the reference's C
// arithmetic defines and the parity tests cover, but no sane configuration uses -- runs these; the host then keeps to
// 16 lanes per pixel on every line (sgm_host.c), so only those combinations exist here.
#include "sgm_aggregate_impl.hpp"

bool sgmd_aggregate_launch_generic(int lpp, int dpl, const AggArgs* a, int blocks, int pad, hipStream_t st)
{
    return launch_aggregate_key<false>(lpp, dpl, *a, blocks, pad != 0, 0, st);
}

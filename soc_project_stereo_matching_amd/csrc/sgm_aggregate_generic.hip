// The aggregation kernels with the generic step (any sign of P1).  Only negative P1 -- which the reference's C
// arithmetic defines and the parity tests cover, but no sane configuration uses -- runs these; the host then keeps to
// 16 lanes per pixel on every line (sgm_host.c), so only those combinations exist here.
#include "sgm_aggregate_impl.hpp"

bool sgmd_aggregate_launch_generic(int lpp, int dpl, const AggArgs* a, int blocks, int pad, hipStream_t st)
{
    return launch_aggregate_key<0>(lpp, dpl, *a, blocks, pad != 0, 0, st);
}

// volume-fed kernels (wide census windows): generic step, 16 lanes per pixel
template <int DPL>
static void launch_aggregate_vol(const AggArgs& a, int blocks, bool pad, hipStream_t st)
{
    if (pad) hipLaunchKernelGGL((sgm_aggregate_k<DPL, true, 16, 0, false, true>), dim3(blocks), dim3(64), 0, st, a);
    else     hipLaunchKernelGGL((sgm_aggregate_k<DPL, false, 16, 0, false, true>), dim3(blocks), dim3(64), 0, st, a);
}
static bool launch_aggregate_vol_key(int dpl, const AggArgs& a, int blocks, bool pad, hipStream_t st)
{
    switch (dpl) {
    case 2: launch_aggregate_vol<2>(a, blocks, pad, st); return true;
    case 4: launch_aggregate_vol<4>(a, blocks, pad, st); return true;
    case 8: launch_aggregate_vol<8>(a, blocks, pad, st); return true;
    case 12: launch_aggregate_vol<12>(a, blocks, pad, st); return true;
    case 16: launch_aggregate_vol<16>(a, blocks, pad, st); return true;
    case 32: launch_aggregate_vol<32>(a, blocks, pad, st); return true;
    default: return false;
    }
}

// ... and the volume-fed kernels of the wide census windows (SURVEY.md 8f-4 extension)
bool sgmd_aggregate_launch_volume(int dpl, const AggArgs* a, int blocks, int pad, hipStream_t st)
{
    return launch_aggregate_vol_key(dpl, *a, blocks, pad != 0, st);
}

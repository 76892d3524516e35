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

// ... and the four anomalous diagonal lines of every frame as a launch of their own (sgm_aggregate_anom_k; the kernels of the
// non-negative-P1 step carry them inside their own launch): one instantiation per cell width and step
template <int DPL, int LPP, int NN, bool VOL>
static void launch_aggregate_anom_one(const AggArgs& a, bool pad, hipStream_t st)
{
    const dim3 grid(4 * a.B), block(64);
    if (pad) hipLaunchKernelGGL((sgm_aggregate_anom_k<DPL, true, LPP, NN, VOL>), grid, block, 0, st, a);
    else     hipLaunchKernelGGL((sgm_aggregate_anom_k<DPL, false, LPP, NN, VOL>), grid, block, 0, st, a);
}
// dp = disparities per cell (Dp); nn = the non-negative-P1 step may be used (never with a cost volume)
static bool launch_aggregate_anom(int dp, bool nn, bool from_volume, const AggArgs& a, bool pad, hipStream_t st)
{
#define SGM_ANOM_CASES(NN, VOL, L)                                                                             \
    switch (dp) {                                                                                              \
    case 32:  launch_aggregate_anom_one<L(32)::dpl, L(32)::lpp, NN, VOL>(a, pad, st); return true;             \
    case 64:  launch_aggregate_anom_one<L(64)::dpl, L(64)::lpp, NN, VOL>(a, pad, st); return true;             \
    case 128: launch_aggregate_anom_one<L(128)::dpl, L(128)::lpp, NN, VOL>(a, pad, st); return true;           \
    case 192: launch_aggregate_anom_one<L(192)::dpl, L(192)::lpp, NN, VOL>(a, pad, st); return true;           \
    case 256: launch_aggregate_anom_one<L(256)::dpl, L(256)::lpp, NN, VOL>(a, pad, st); return true;           \
    case 512: launch_aggregate_anom_one<L(512)::dpl, L(512)::lpp, NN, VOL>(a, pad, st); return true;           \
    default: return false;                                                                                     \
    }
#define SGM_ANOM_WIDE(dp) anom_layout<dp>
#define SGM_ANOM_16(dp) anom_layout16<dp>
    if (from_volume) { SGM_ANOM_CASES(0, true, SGM_ANOM_16) }
    if (nn) { SGM_ANOM_CASES(1, false, SGM_ANOM_WIDE) }
    SGM_ANOM_CASES(0, false, SGM_ANOM_16)
#undef SGM_ANOM_CASES
#undef SGM_ANOM_WIDE
#undef SGM_ANOM_16
}

bool sgmd_aggregate_launch_anom(int dp, const AggArgs* a, int pad, int from_volume, hipStream_t st)
{
    return launch_aggregate_anom(dp, a->p1 >= 0 && !from_volume, from_volume != 0, *a, pad != 0, st);
}

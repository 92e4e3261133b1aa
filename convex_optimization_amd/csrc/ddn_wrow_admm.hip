// Wave-per-row admm kernel (W = 4096, float).
#define DDN_WAVE_FAST 1
#include "ddn_launch.hpp"
#include "ddn_wrow.hpp"
template <>
int wrow_launch<AdmmP<float>>(hipStream_t st, int grid, const RowCommon<float>& c, const AdmmP<float>::Args& a) {
    static const int pd = getenv("DDN_ADMM_PD") ? atoi(getenv("DDN_ADMM_PD")) : 2;
    if (c.do_consume && c.do_produce && pd == 2) hipLaunchKernelGGL((wrow::k_wave_row<AdmmP<float>, 2, true, true>), dim3(grid), dim3(64), 0, st, c, a);
    else if (c.do_consume && c.do_produce && pd == 8) hipLaunchKernelGGL((wrow::k_wave_row<AdmmP<float>, 8, true, true>), dim3(grid), dim3(64), 0, st, c, a);
    else if (c.do_consume && c.do_produce && pd == 1) hipLaunchKernelGGL((wrow::k_wave_row<AdmmP<float>, 1, true, true>), dim3(grid), dim3(64), 0, st, c, a);
    else if (c.do_consume && c.do_produce) hipLaunchKernelGGL((wrow::k_wave_row<AdmmP<float>, 4, true, true>), dim3(grid), dim3(64), 0, st, c, a);
    else if (c.do_consume) hipLaunchKernelGGL((wrow::k_wave_row<AdmmP<float>, 4, true, false>), dim3(grid), dim3(64), 0, st, c, a);
    else hipLaunchKernelGGL((wrow::k_wave_row<AdmmP<float>, 4, false, true>), dim3(grid), dim3(64), 0, st, c, a);
    HIPCHK(hipGetLastError());
    return DDN_OK;
}

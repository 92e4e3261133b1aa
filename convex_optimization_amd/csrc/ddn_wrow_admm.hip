// Wave-per-row admm kernel (W = 4096, float).
#define DDN_WAVE_FAST 1
#include "ddn_launch.hpp"
#include "ddn_wrow.hpp"
template <>
int wrow_launch<AdmmP<float>>(hipStream_t st, int grid, const RowCommon<float>& c, const AdmmP<float>::Args& a) {
    if (c.do_consume && c.do_produce) hipLaunchKernelGGL((wrow::k_wave_row<AdmmP<float>, 4, true, true>), dim3(grid), dim3(64), 0, st, c, a);
    else if (c.do_consume) hipLaunchKernelGGL((wrow::k_wave_row<AdmmP<float>, 4, true, false>), dim3(grid), dim3(64), 0, st, c, a);
    else hipLaunchKernelGGL((wrow::k_wave_row<AdmmP<float>, 4, false, true>), dim3(grid), dim3(64), 0, st, c, a);
    HIPCHK(hipGetLastError());
    return DDN_OK;
}

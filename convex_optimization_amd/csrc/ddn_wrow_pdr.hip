// Wave-per-row pdr kernel (W = 4096, float).
#define DDN_WAVE_FAST 1
#include "ddn_launch.hpp"
#include "ddn_wrow.hpp"
template <>
int wrow_launch<PdrP<float>>(hipStream_t st, int grid, const RowCommon<float>& c, const PdrP<float>::Args& a) {
    static const int pd = getenv("DDN_ROW_PD") ? atoi(getenv("DDN_ROW_PD")) : 8;
    static const int fe = getenv("DDN_ROW_FE") ? atoi(getenv("DDN_ROW_FE")) : 1;
    if (c.do_consume && c.do_produce && fe == 2) hipLaunchKernelGGL((wrow::k_wave_row<PdrP<float>, 8, true, true, 2>), dim3(grid), dim3(64), 0, st, c, a);
    else if (c.do_consume && c.do_produce && fe == 4) hipLaunchKernelGGL((wrow::k_wave_row<PdrP<float>, 8, true, true, 4>), dim3(grid), dim3(64), 0, st, c, a);
    else if (c.do_consume && c.do_produce && fe == 64) hipLaunchKernelGGL((wrow::k_wave_row<PdrP<float>, 8, true, true, 64>), dim3(grid), dim3(64), 0, st, c, a);
    else if (c.do_consume && c.do_produce && pd == 2) hipLaunchKernelGGL((wrow::k_wave_row<PdrP<float>, 2, true, true>), dim3(grid), dim3(64), 0, st, c, a);
    else if (c.do_consume && c.do_produce && pd == 4) hipLaunchKernelGGL((wrow::k_wave_row<PdrP<float>, 4, true, true>), dim3(grid), dim3(64), 0, st, c, a);
    else if (c.do_consume && c.do_produce && pd == 12) hipLaunchKernelGGL((wrow::k_wave_row<PdrP<float>, 12, true, true>), dim3(grid), dim3(64), 0, st, c, a);
    else if (c.do_consume && c.do_produce) hipLaunchKernelGGL((wrow::k_wave_row<PdrP<float>, 8, true, true>), dim3(grid), dim3(64), 0, st, c, a);
    else if (c.do_consume) hipLaunchKernelGGL((wrow::k_wave_row<PdrP<float>, 8, true, false>), dim3(grid), dim3(64), 0, st, c, a);
    else hipLaunchKernelGGL((wrow::k_wave_row<PdrP<float>, 8, false, true>), dim3(grid), dim3(64), 0, st, c, a);
    HIPCHK(hipGetLastError());
    return DDN_OK;
}

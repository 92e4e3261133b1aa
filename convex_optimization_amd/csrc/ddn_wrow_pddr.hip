// Wave-per-row pddr kernel (W = 4096, float).
#define DDN_WAVE_FAST 1
#include "ddn_launch.hpp"
#include "ddn_wrow.hpp"
template <>
int wrow_launch<PddrP<float>>(hipStream_t st, int grid, const RowCommon<float>& c, const PddrP<float>::Args& a) {
    static const int pd = getenv("DDN_ROW_PD") ? atoi(getenv("DDN_ROW_PD")) : 8;
    if (c.do_consume && c.do_produce && pd == 2) hipLaunchKernelGGL((wrow::k_wave_row<PddrP<float>, 2, true, true>), dim3(grid), dim3(64), 0, st, c, a);
    else if (c.do_consume && c.do_produce && pd == 4) hipLaunchKernelGGL((wrow::k_wave_row<PddrP<float>, 4, true, true>), dim3(grid), dim3(64), 0, st, c, a);
    else if (c.do_consume && c.do_produce && pd == 12) hipLaunchKernelGGL((wrow::k_wave_row<PddrP<float>, 12, true, true>), dim3(grid), dim3(64), 0, st, c, a);
    else if (c.do_consume && c.do_produce) hipLaunchKernelGGL((wrow::k_wave_row<PddrP<float>, 8, true, true>), dim3(grid), dim3(64), 0, st, c, a);
    else if (c.do_consume) hipLaunchKernelGGL((wrow::k_wave_row<PddrP<float>, 8, true, false>), dim3(grid), dim3(64), 0, st, c, a);
    else hipLaunchKernelGGL((wrow::k_wave_row<PddrP<float>, 8, false, true>), dim3(grid), dim3(64), 0, st, c, a);
    HIPCHK(hipGetLastError());
    return DDN_OK;
}

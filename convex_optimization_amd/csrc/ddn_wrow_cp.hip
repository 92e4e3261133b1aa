// Wave-per-row Chambolle-Pock kernels (W = 4096, float).
#define DDN_WAVE_FAST 1
#include "ddn_launch.hpp"
#include "ddn_wrow.hpp"
int wrow_cp_launch(hipStream_t st, int grid, const RowCommon<float>& c, const CpArgs<float>& a) {
    if (a.step == 0) hipLaunchKernelGGL((wrow::k_wave_cp<8, 0, true, true>), dim3(grid), dim3(64), 0, st, c, a);
    else if (!c.do_consume) hipLaunchKernelGGL((wrow::k_wave_cp<8, 1, false, true>), dim3(grid), dim3(64), 0, st, c, a);
    else if (c.do_produce) hipLaunchKernelGGL((wrow::k_wave_cp<8, 1, true, true>), dim3(grid), dim3(64), 0, st, c, a);
    else hipLaunchKernelGGL((wrow::k_wave_cp<8, 1, true, false>), dim3(grid), dim3(64), 0, st, c, a);
    HIPCHK(hipGetLastError());
    return DDN_OK;
}

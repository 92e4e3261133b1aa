// Wave-per-row pddr kernels (W = 4096, float): full iteration, consume-only (last launch) and
// produce-only (first launch) variants.  Load-pipeline depth 8: measured best (the kernel is not
// latency-bound on the iterate loads; deeper pipelines only cost registers).
#define DDN_WAVE_FAST 1
#include "ddn_launch.hpp"
#include "ddn_wrow.hpp"
template <>
int wrow_launch<PddrP<float>>(hipStream_t st, int grid, const RowCommon<float>& c, const PddrP<float>::Args& a) {
    if (c.do_consume && c.do_produce) hipLaunchKernelGGL((wrow::k_wave_row<PddrP<float>, 8, true, true, 2>), dim3(grid), dim3(64), 0, st, c, a);
    else if (c.do_consume) hipLaunchKernelGGL((wrow::k_wave_row<PddrP<float>, 8, true, false>), dim3(grid), dim3(64), 0, st, c, a);
    else hipLaunchKernelGGL((wrow::k_wave_row<PddrP<float>, 8, false, true>), dim3(grid), dim3(64), 0, st, c, a);
    HIPCHK(hipGetLastError());
    return DDN_OK;
}

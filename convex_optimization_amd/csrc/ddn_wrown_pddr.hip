// Wave-per-row-group pddr kernels for W = 512, 1024, 2048 (float): 64 / (W / 64) rows per wavefront
// (ddn_wrown.hpp); full iteration, consume-only and produce-only variants.
#define DDN_WAVE_FAST 1
#ifndef DDN_NO_NT_ITER
#define DDN_NT_ITER 1  // non-temporal iterate accesses (ddn_wrow.hpp, ldi / sti)
#endif
#include "ddn_launch.hpp"
#include "ddn_wrown.hpp"

template <>
int wrown_launch<PddrP<float>>(hipStream_t st, int grid, const RowCommon<float>& c, const PddrP<float>::Args& a) {
    return wrow::wrown_dispatch<PddrP<float>, 8, 2>(st, grid, c, a);
}

// Wave-per-row-group Chambolle-Pock kernels for W = 512, 1024, 2048 (float): 2 * 64 / (W / 64) rows per
// wavefront (ddn_wrown.hpp).
#define DDN_WAVE_FAST 1
#ifndef DDN_NO_NT_ITER
#define DDN_NT_ITER 1  // non-temporal iterate accesses (ddn_wrow.hpp, ldi / sti)
#endif
#include "ddn_launch.hpp"
#include "ddn_wrown.hpp"

int wrown_cp_launch(hipStream_t st, int grid, const RowCommon<float>& c, const CpArgs<float>& a) {
    switch (c.W) {
        case 512: return wrow::wrown_cp_go<8, 8>(st, grid, c, a);
        case 1024: return wrow::wrown_cp_go<8, 16>(st, grid, c, a);
        case 2048: return wrow::wrown_cp_go<8, 32>(st, grid, c, a);
        default: return DDN_E_UNSUPPORTED;
    }
}

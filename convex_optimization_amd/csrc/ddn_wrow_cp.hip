// Wave-per-row Chambolle-Pock kernels (W = 4096, float); one wave (one row pair) per workgroup with
// 16-column spectral tiles, four with 4-column tiles.
#define DDN_WAVE_FAST 1
#ifndef DDN_NO_NT_ITER
#define DDN_NT_ITER 1  // non-temporal iterate accesses (ddn_wrow.hpp, ldi / sti)
#endif
#include "ddn_launch.hpp"
#include "ddn_wrow.hpp"

template <int WPW>
static int go(hipStream_t st, int grid, const RowCommon<float>& c, const CpArgs<float>& a) {
    const size_t lds = (size_t)WPW * WFFT_TILE_FLOATS * sizeof(float);
    if (a.step == 0) return launch_row_fn(st, wrow::k_wave_cp<8, 0, true, true, WPW>, grid, 64 * WPW, lds, c, a);
    if (!c.do_consume) return launch_row_fn(st, wrow::k_wave_cp<8, 1, false, true, WPW>, grid, 64 * WPW, lds, c, a);
    if (c.do_produce) return launch_row_fn(st, wrow::k_wave_cp<8, 1, true, true, WPW>, grid, 64 * WPW, lds, c, a);
    return launch_row_fn(st, wrow::k_wave_cp<8, 1, true, false, WPW>, grid, 64 * WPW, lds, c, a);
}

int wrow_cp_launch(hipStream_t st, int grid, const RowCommon<float>& c, const CpArgs<float>& a) {
    return go<1>(st, grid, c, a);
}

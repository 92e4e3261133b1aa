// Wave-per-row admm kernels (W = 4096, float): full iteration, consume-only (last launch) and
// produce-only (first launch) variants; one wave per workgroup with 16-column spectral tiles, four
// (rows 4g .. 4g+3) with 4-column tiles.  Load-pipeline depth 2: measured best (the kernel is not
// latency-bound on the iterate loads; deeper pipelines only cost registers).
#define DDN_WAVE_FAST 1
#ifndef DDN_NO_NT_ITER
#define DDN_NT_ITER 1  // non-temporal iterate accesses (ddn_wrow.hpp, ldi / sti)
#endif
#include "ddn_launch.hpp"
#include "ddn_wrow.hpp"

template <int WPW>
static int go(hipStream_t st, int grid, const RowCommon<float>& c, const AdmmP<float>::Args& a) {
    const size_t lds = (size_t)WPW * WFFT_TILE_FLOATS * sizeof(float);
    if (c.do_consume && c.do_produce) return launch_row_fn(st, wrow::k_wave_row<AdmmP<float>, 2, true, true, 1, WPW, 16>, grid, 64 * WPW, lds, c, a);
    if (c.do_consume) return launch_row_fn(st, wrow::k_wave_row<AdmmP<float>, 2, true, false, 1, WPW>, grid, 64 * WPW, lds, c, a);
    return launch_row_fn(st, wrow::k_wave_row<AdmmP<float>, 2, false, true, 1, WPW>, grid, 64 * WPW, lds, c, a);
}

// interleaved half-spectra (16-column tiles only)
static int go_ilv(hipStream_t st, int grid, const RowCommon<float>& c, const AdmmP<float>::Args& a) {
    const size_t lds = (size_t)WFFT_TILE_FLOATS * sizeof(float);
    if (c.do_consume && c.do_produce) return launch_row_fn(st, wrow::k_wave_row<AdmmP<float>, 2, true, true, 1, 1, 16, true>, grid, 64, lds, c, a);
    if (c.do_consume) return launch_row_fn(st, wrow::k_wave_row<AdmmP<float>, 2, true, false, 1, 1, 0, true>, grid, 64, lds, c, a);
    return launch_row_fn(st, wrow::k_wave_row<AdmmP<float>, 2, false, true, 1, 1, 0, true>, grid, 64, lds, c, a);
}

template <>
int wrow_launch<AdmmP<float>>(hipStream_t st, int grid, const RowCommon<float>& c, const AdmmP<float>::Args& a) {
    if (c.ilv) return c.lay.cs == 4 ? go_ilv(st, grid, c, a) : DDN_E_UNSUPPORTED;
    return go<1>(st, grid, c, a);
}

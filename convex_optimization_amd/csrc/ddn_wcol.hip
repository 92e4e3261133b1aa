// Wave-per-column single-pass column stage (H = 4096, float).
#define DDN_WAVE_FAST 1
#include "ddn_launch.hpp"
#include "ddn_wcol.hpp"

template <int MODE, bool PACKED, bool ILV = false, bool RB = false>
static int wcol_go(hipStream_t st, int grid, const wcol::Args& a) {
    constexpr int waves = MODE == COL_RES2 ? 8 : 4;
    auto fn = wcol::k_wave_col<MODE, PACKED, ILV, RB>;
    int rc = ensure_lds_attr((const void*)fn);
    if (rc) return rc;
    hipLaunchKernelGGL(fn, dim3(grid), dim3(64 * waves), waves * wcol::kLdsBytes + wcol::kLdsExtra, st, a);
    HIPCHK(hipGetLastError());
    return DDN_OK;
}

int wcol_launch(hipStream_t st, int mode, const wcol::Args& a) {
    const int nq = (a.Wh - (a.nyq_packed ? 1 : 0) + 3) >> 2;
    const int grid = a.batch * (a.lay.cs == 2 ? nq : 32 * ((a.lay.ntile + 7) / 8));
    if (a.rb_extra) {
        // rows stored in per-source blocks (row-slab decomposition): plain columns, separate spectra
        if (a.ilv || a.nyq_packed || a.nyqd || a.lay.cs != 4) return DDN_E_UNSUPPORTED;
        return mode == COL_RES2 ? wcol_go<COL_RES2, false, false, true>(st, grid, a) : wcol_go<COL_MUL, false, false, true>(st, grid, a);
    }
    if (a.ilv) {
        if (mode != COL_RES2 || a.nyq_packed || a.lay.cs != 4) return DDN_E_UNSUPPORTED;
        return wcol_go<COL_RES2, false, true>(st, grid, a);
    }
    if (mode == COL_RES2) return a.nyq_packed ? wcol_go<COL_RES2, true>(st, grid, a) : wcol_go<COL_RES2, false>(st, grid, a);
    return a.nyq_packed ? wcol_go<COL_MUL, true>(st, grid, a) : wcol_go<COL_MUL, false>(st, grid, a);
}

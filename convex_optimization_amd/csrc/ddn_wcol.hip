// Wave-per-column single-pass column stage (H = 4096, float).
#define DDN_WAVE_FAST 1
#include "ddn_launch.hpp"
#include "ddn_wcol.hpp"

template <int MODE, bool ILV = false, bool RB = false>
static int wcol_go(hipStream_t st, int grid, const wcol::Args& a) {
    constexpr int waves = MODE == COL_RES2 ? 8 : 4;
    auto fn = wcol::k_wave_col<MODE, ILV, RB>;
    int rc = ensure_lds_attr((const void*)fn);
    if (rc) return rc;
    hipLaunchKernelGGL(fn, dim3(grid), dim3(64 * waves), waves * wcol::kLdsBytes, st, a);
    HIPCHK(hipGetLastError());
    return DDN_OK;
}

int wcol_launch(hipStream_t st, int mode, const wcol::Args& a) {
    if (a.lay.cs != 4) return fail(DDN_E_UNSUPPORTED, "wcol: 16-column spectral tiles only");
    const int grid = a.batch * ((a.Wh + 3) >> 2);   // one workgroup per quarter tile that holds a column
    if (a.rb_extra) {
        // rows stored in per-source blocks (row-slab decomposition): separate spectra
        if (a.ilv || a.nyqd) return fail(DDN_E_UNSUPPORTED, "wcol: row blocks go with separate spectra");
        return mode == COL_RES2 ? wcol_go<COL_RES2, false, true>(st, grid, a) : wcol_go<COL_MUL, false, true>(st, grid, a);
    }
    if (a.ilv) {
        if (mode != COL_RES2) return fail(DDN_E_UNSUPPORTED, "wcol: interleaved spectra are a resolvent-mode layout");
        return wcol_go<COL_RES2, true>(st, grid, a);
    }
    return mode == COL_RES2 ? wcol_go<COL_RES2>(st, grid, a) : wcol_go<COL_MUL>(st, grid, a);
}

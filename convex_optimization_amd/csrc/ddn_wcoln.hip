// Wave-per-column-group single-pass column stage (H = 512, 1024, 2048, float; ddn_wcoln.hpp).
#define DDN_WAVE_FAST 1
#include "ddn_launch.hpp"
#include "ddn_wcoln.hpp"

template <int MODE, int R, int NP, bool ILV = false, bool RB = false>
static int wcoln_go1(hipStream_t st, const wcol::Args& a) {
    constexpr int waves = (MODE == COL_RES2 ? 2 : 1) * NP;
    constexpr int M = 64 / R;
    if ((1 << a.lay.cs) < M * NP || a.lay.cs > 4) return fail(DDN_E_UNSUPPORTED, "wcoln: the spectral tile width does not hold a whole column group");
    const int G = (1 << a.lay.cs) / (M * NP);
    auto fn = wcol::k_wave_colN<MODE, R, NP, ILV, RB>;
    int rc = ensure_lds_attr((const void*)fn);
    if (rc) return rc;
    const int grid = a.batch * 8 * G * ((a.lay.ntile + 7) / 8);
    hipLaunchKernelGGL(fn, dim3(grid), dim3(64 * waves), waves * WFFT_TILE_FLOATS * sizeof(float), st, a);
    HIPCHK(hipGetLastError());
    return DDN_OK;
}

template <int MODE, int R>
static int wcoln_go(hipStream_t st, const wcol::Args& a) {
    // groups per workgroup.  Resolvent mode: one and two measured the same within 2-4 % for H = 512 and 1024; two are
    // 8 % faster for H = 2048 (16 bytes of a line per group).  Multiplier mode (one spectrum: a group's piece of a row
    // is 8 M bytes, a quarter of a line at H = 1024): two, 6 % faster there (768 planes of 1024^2: 0.574 -> 0.540 ms
    // per launch; four: 0.534).  DDN_COLN_NP overrides, for measurements.
    static const int np = getenv("DDN_COLN_NP") ? atoi(getenv("DDN_COLN_NP")) : ((R == 32 || MODE == COL_MUL) ? 2 : 1);
    constexpr int M = 64 / R;
    if (a.rb_extra) {
        // rows stored in per-source blocks (row-slab decomposition): separate spectra, one group per workgroup
        if (a.ilv) return fail(DDN_E_UNSUPPORTED, "wcoln: row blocks go with separate spectra");
        return wcoln_go1<MODE, R, 1, false, true>(st, a);
    }
    if constexpr (MODE == COL_RES2) {
        if (a.ilv) {
            if (np >= 2 && (1 << a.lay.cs) >= 2 * M) return wcoln_go1<MODE, R, 2, true>(st, a);
            return wcoln_go1<MODE, R, 1, true>(st, a);
        }
    }
    if (np == 4 && (1 << a.lay.cs) >= 4 * M) return wcoln_go1<MODE, R, 4>(st, a);
    if (np >= 2 && (1 << a.lay.cs) >= 2 * M) return wcoln_go1<MODE, R, 2>(st, a);
    return wcoln_go1<MODE, R, 1>(st, a);
}

int wcoln_launch(hipStream_t st, int mode, const wcol::Args& a) {
    const bool res = mode == COL_RES2;
    switch (a.Hfft ? a.Hfft : a.lay.H) {
        case 512: return res ? wcoln_go<COL_RES2, 8>(st, a) : wcoln_go<COL_MUL, 8>(st, a);
        case 1024: return res ? wcoln_go<COL_RES2, 16>(st, a) : wcoln_go<COL_MUL, 16>(st, a);
        case 2048: return res ? wcoln_go<COL_RES2, 32>(st, a) : wcoln_go<COL_MUL, 32>(st, a);
        default: return fail(DDN_E_UNSUPPORTED, "wcoln: column length is not 512, 1024 or 2048");
    }
}

// Wave-per-column-group single-pass column stage (H = 512, 1024, 2048, float; ddn_wcoln.hpp).
#define DDN_WAVE_FAST 1
#include "ddn_launch.hpp"
#include "ddn_wcoln.hpp"

template <int MODE, int R>
static int wcoln_go(hipStream_t st, const wcol::Args& a) {
    constexpr int waves = MODE == COL_RES2 ? 2 : 1;
    constexpr int M = 64 / R;
    if ((1 << a.lay.cs) < M || a.lay.cs > 4) return DDN_E_UNSUPPORTED;
    const int G = (1 << a.lay.cs) / M;
    auto fn = wcol::k_wave_colN<MODE, R>;
    int rc = ensure_lds_attr((const void*)fn);
    if (rc) return rc;
    const int grid = a.batch * 8 * G * ((a.lay.ntile + 7) / 8);
    hipLaunchKernelGGL(fn, dim3(grid), dim3(64 * waves), waves * WFFT_TILE_FLOATS * sizeof(float), st, a);
    HIPCHK(hipGetLastError());
    return DDN_OK;
}

int wcoln_launch(hipStream_t st, int mode, const wcol::Args& a) {
    const bool res = mode == COL_RES2;
    switch (a.lay.H) {
        case 512: return res ? wcoln_go<COL_RES2, 8>(st, a) : wcoln_go<COL_MUL, 8>(st, a);
        case 1024: return res ? wcoln_go<COL_RES2, 16>(st, a) : wcoln_go<COL_MUL, 16>(st, a);
        case 2048: return res ? wcoln_go<COL_RES2, 32>(st, a) : wcoln_go<COL_MUL, 32>(st, a);
        default: return DDN_E_UNSUPPORTED;
    }
}

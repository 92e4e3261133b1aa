// Wave-per-column single-pass column stage (H = 4096, float).
#define DDN_WAVE_FAST 1
#include "ddn_launch.hpp"
#include "ddn_wcol.hpp"

int wcol_launch(hipStream_t st, int mode, const wcol::Args& a) {
    const int grid = a.batch * 32 * ((a.lay.ntile + 7) / 8);
    int rc;
    if (mode == COL_RES2) {
        if ((rc = ensure_lds_attr((const void*)wcol::k_wave_col<COL_RES2>))) return rc;
        hipLaunchKernelGGL((wcol::k_wave_col<COL_RES2>), dim3(grid), dim3(512), 8 * wcol::kLdsBytes, st, a);
    } else {
        if ((rc = ensure_lds_attr((const void*)wcol::k_wave_col<COL_MUL>))) return rc;
        hipLaunchKernelGGL((wcol::k_wave_col<COL_MUL>), dim3(grid), dim3(256), 4 * wcol::kLdsBytes, st, a);
    }
    HIPCHK(hipGetLastError());
    return DDN_OK;
}

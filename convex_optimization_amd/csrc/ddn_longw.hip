// Launchers of the wave-resident frequency-side kernels of the split row transform (M = 4096, float; ddn_longw.hpp).
#define DDN_WAVE_FAST 1
#include "ddn_launch.hpp"
#include "ddn_longw.hpp"

namespace {
template <int R, bool INV>
int go(hipStream_t st, int num_cu, const LongArgs<float>& a) {
    auto fn = INV ? dlongw::k_long_inv_w<R> : dlongw::k_long_fwd_w<R>;
    int rc = ensure_lds_attr((const void*)fn);
    if (rc) return rc;
    constexpr int waves = R / 2 + 1;
    const size_t nrows = (size_t)a.c.batch * a.c.H;
    const int grid = (int)std::max<size_t>(1, std::min<size_t>(nrows, (size_t)num_cu * 8));
    hipLaunchKernelGGL(fn, dim3(grid), dim3(64 * waves), dlongw::Geo<R>::lds_bytes(), st, a);
    HIPCHK(hipGetLastError());
    return DDN_OK;
}
}  // namespace

bool long_wave_ok(const LongArgs<float>& a) { return a.M == 4096 && a.R <= 8 && a.wtw != nullptr && !a.c.ilv && a.c.lay.cs <= 6; }

int long_wave_launch(hipStream_t st, int num_cu, const LongArgs<float>& a, bool inverse) {
    switch (a.R) {
        case 2: return inverse ? go<2, true>(st, num_cu, a) : go<2, false>(st, num_cu, a);
        case 4: return inverse ? go<4, true>(st, num_cu, a) : go<4, false>(st, num_cu, a);
        case 8: return inverse ? go<8, true>(st, num_cu, a) : go<8, false>(st, num_cu, a);
        default: return fail(DDN_E_UNSUPPORTED, "long rows: the wave-resident transform takes splits of 2, 4 or 8");
    }
}

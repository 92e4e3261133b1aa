// Launchers of the long-row path (rows wider than 8192 pixels; ddn_long.hpp), both dtypes.
#include "ddn_launch.hpp"
#include "ddn_long.hpp"

namespace {

template <typename T>
int long_inv(hipStream_t st, int num_cu, const LongArgs<T>& a) {
    if constexpr (sizeof(T) == 4) {
        if (long_wave_ok(a)) return long_wave_launch(st, num_cu, a, true);
    }
    auto fn = dlong::k_long_inv<T>;
    int rc = ensure_lds_attr((const void*)fn);
    if (rc) return rc;
    const int grid = dlong::long_grid((size_t)a.c.batch * a.c.H, a.narr * (a.R / 2 + 1), num_cu);
    hipLaunchKernelGGL(fn, dim3(grid), dim3(256), row_lds_bytes<T>(a.M, 1), st, a);
    HIPCHK(hipGetLastError());
    return DDN_OK;
}
template <typename T>
int long_fwd(hipStream_t st, int num_cu, const LongArgs<T>& a) {
    if constexpr (sizeof(T) == 4) {
        if (long_wave_ok(a)) return long_wave_launch(st, num_cu, a, false);
    }
    auto fn = dlong::k_long_fwd<T>;
    int rc = ensure_lds_attr((const void*)fn);
    if (rc) return rc;
    const int grid = dlong::long_grid((size_t)a.c.batch * a.c.H, a.R / 2 + 1, num_cu);
    hipLaunchKernelGGL(fn, dim3(grid), dim3(256), row_lds_bytes<T>(a.M, 1), st, a);
    HIPCHK(hipGetLastError());
    return DDN_OK;
}
inline int pix_grid(int num_cu, size_t nrows) { return (int)std::max<size_t>(1, std::min<size_t>(nrows, (size_t)num_cu * 8)); }

template <typename T, class P, int R>
int pix(hipStream_t st, int num_cu, const LongArgs<T>& a, const typename P::Args& pa) {
    constexpr int VMAX = 4;     // the width of the policies' vector accesses (four pixels: 16 bytes in float, 2 x 16 in double)
    const dim3 grid(pix_grid(num_cu, (size_t)a.c.batch * a.c.H));
    if (a.vec == VMAX && a.M % VMAX == 0) hipLaunchKernelGGL((dlong::k_long_pix<T, P, R, VMAX>), grid, dim3(256), 0, st, a, pa);
    else hipLaunchKernelGGL((dlong::k_long_pix<T, P, R, 1>), grid, dim3(256), 0, st, a, pa);
    HIPCHK(hipGetLastError());
    return DDN_OK;
}
template <typename T, int R>
int pix_cp(hipStream_t st, int num_cu, const LongArgs<T>& a, const CpArgs<T>& pa) {
    hipLaunchKernelGGL((dlong::k_long_cp<T, R>), dim3(pix_grid(num_cu, (size_t)a.c.batch * a.c.H)), dim3(256), 0, st, a, pa);
    HIPCHK(hipGetLastError());
    return DDN_OK;
}
template <typename T, int R>
int pix_plain(hipStream_t st, int num_cu, const LongArgs<T>& a, const T* src, T* dst, T scale) {
    hipLaunchKernelGGL((dlong::k_long_plain<T, R>), dim3(pix_grid(num_cu, (size_t)a.c.batch * a.c.H)), dim3(256), 0, st, a, src, dst, scale);
    HIPCHK(hipGetLastError());
    return DDN_OK;
}

}  // namespace

#define DDN_LONG_R(call)                                                          \
    switch (a.R) {                                                                \
        case 2: rc = call(2); break;                                              \
        case 4: rc = call(4); break;                                              \
        case 8: rc = call(8); break;                                              \
        case 16: rc = call(16); break;                                            \
        default: return fail(DDN_E_UNSUPPORTED, "long rows: unsupported split");  \
    }

template <typename T, class P>
int long_row_launch(hipStream_t st, int num_cu, const LongArgs<T>& a, const typename P::Args& pa) {
    int rc;
    if (a.c.do_consume && (rc = long_inv<T>(st, num_cu, a))) return rc;
#define DDN_CALL(RR) pix<T, P, RR>(st, num_cu, a, pa)
    DDN_LONG_R(DDN_CALL)
#undef DDN_CALL
    if (rc) return rc;
    if (a.c.do_produce && (rc = long_fwd<T>(st, num_cu, a))) return rc;
    return DDN_OK;
}

template <typename T>
int long_cp_launch(hipStream_t st, int num_cu, const LongArgs<T>& a, const CpArgs<T>& pa) {
    int rc;
    if (a.c.do_consume && (rc = long_inv<T>(st, num_cu, a))) return rc;
#define DDN_CALL(RR) pix_cp<T, RR>(st, num_cu, a, pa)
    DDN_LONG_R(DDN_CALL)
#undef DDN_CALL
    if (rc) return rc;
    if (a.c.do_produce && (rc = long_fwd<T>(st, num_cu, a))) return rc;
    return DDN_OK;
}

// src != null: real rows -> half-spectra in a.c.S0; else half-spectra -> dst * scale
template <typename T>
int long_plain_launch(hipStream_t st, int num_cu, const LongArgs<T>& a, const T* src, T* dst, T scale) {
    int rc;
    if (!src && (rc = long_inv<T>(st, num_cu, a))) return rc;
#define DDN_CALL(RR) pix_plain<T, RR>(st, num_cu, a, src, dst, scale)
    DDN_LONG_R(DDN_CALL)
#undef DDN_CALL
    if (rc) return rc;
    if (src && (rc = long_fwd<T>(st, num_cu, a))) return rc;
    return DDN_OK;
}

#define DDN_INST_LONG(T)                                                                                                  \
    template int long_row_launch<T, PdrP<T>>(hipStream_t, int, const LongArgs<T>&, const PdrP<T>::Args&);                 \
    template int long_row_launch<T, PddrP<T>>(hipStream_t, int, const LongArgs<T>&, const PddrP<T>::Args&);               \
    template int long_row_launch<T, AdmmP<T>>(hipStream_t, int, const LongArgs<T>&, const AdmmP<T>::Args&);               \
    template int long_cp_launch<T>(hipStream_t, int, const LongArgs<T>&, const CpArgs<T>&);                                \
    template int long_plain_launch<T>(hipStream_t, int, const LongArgs<T>&, const T*, T*, T);
DDN_INST_LONG(float)
DDN_INST_LONG(double)

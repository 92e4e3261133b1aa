// Launcher of the whole-row kernels (rows of 16384 / 32768 pixels, float; ddn_wholerow.hpp).
#ifndef DDN_NO_NT_ITER
#define DDN_NT_ITER 1  // non-temporal iterate accesses (ddn_kernels.hpp, vload / vstore)
#endif
#include "ddn_launch.hpp"
#include "ddn_wholerow.hpp"

bool whole_row_width(int W) { return W == 8192 || W == 16384 || W == 32768; }

template <class P, int NH, int NT>
static int go1(hipStream_t st, int grid, const WholeRowArgs& a, const typename P::Args& pa) {
    auto fn = wholerow::k_row_whole<P, NH, NT>;
    int rc = ensure_lds_attr((const void*)fn);
    if (rc) return rc;
    hipLaunchKernelGGL(fn, dim3(grid), dim3(NT), wholerow::lds_bytes<NH>(), st, a, pa);
    HIPCHK(hipGetLastError());
    return DDN_OK;
}
// 512 threads: a thread's 64 pixels of the row fit its 256 registers beside the transforms' working set (1024 threads
// of 128 registers spill and were 13 % slower at 32768 pixels)
template <class P, int NH>
static int go(hipStream_t st, int grid, const WholeRowArgs& a, const typename P::Args& pa) {
    return go1<P, NH, 512>(st, grid, a, pa);
}

template <class P>
int whole_row_launch(hipStream_t st, int num_cu, const WholeRowArgs& a, const typename P::Args& pa) {
    const size_t nrows = (size_t)a.c.batch * a.c.H;
    const int grid = (int)std::max<size_t>(1, std::min<size_t>(nrows, (size_t)num_cu));   // 128 KB of LDS: one workgroup per CU
    if (a.c.W == 32768) return go<P, 16384>(st, grid, a, pa);
    if (a.c.W == 16384) return go<P, 8192>(st, grid, a, pa);
    if (a.c.W == 8192) {
        // 35 KB of LDS and 256 threads: several rows per CU
        const int g2 = (int)std::max<size_t>(1, std::min<size_t>(nrows, (size_t)num_cu * 4));
        return go1<P, 4096, 256>(st, g2, a, pa);
    }
    return fail(DDN_E_UNSUPPORTED, "whole-row kernel: rows of 8192, 16384 or 32768 pixels only");
}

template int whole_row_launch<PdrP<float>>(hipStream_t, int, const WholeRowArgs&, const PdrP<float>::Args&);
template int whole_row_launch<PddrP<float>>(hipStream_t, int, const WholeRowArgs&, const PddrP<float>::Args&);
template int whole_row_launch<AdmmP<float>>(hipStream_t, int, const WholeRowArgs&, const AdmmP<float>::Args&);

template <int NH, int NT>
static int go_cp(hipStream_t st, int grid, const WholeRowArgs& a, const CpArgs<float>& pa) {
    auto fn = wholerow::k_row_whole_cp<NH, NT>;
    int rc = ensure_lds_attr((const void*)fn);
    if (rc) return rc;
    hipLaunchKernelGGL(fn, dim3(grid), dim3(NT), wholerow::lds_bytes<NH>(), st, a, pa);
    HIPCHK(hipGetLastError());
    return DDN_OK;
}

int whole_row_cp_launch(hipStream_t st, int num_cu, const WholeRowArgs& a, const CpArgs<float>& pa) {
    const size_t nrows = (size_t)a.c.batch * a.c.H;
    const int grid = (int)std::max<size_t>(1, std::min<size_t>(nrows, (size_t)num_cu));
    if (a.c.W == 32768) return go_cp<16384, 512>(st, grid, a, pa);
    if (a.c.W == 16384) return go_cp<8192, 512>(st, grid, a, pa);
    if (a.c.W == 8192) return go_cp<4096, 256>(st, (int)std::max<size_t>(1, std::min<size_t>(nrows, (size_t)num_cu * 4)), a, pa);
    return fail(DDN_E_UNSUPPORTED, "whole-row kernel: rows of 8192, 16384 or 32768 pixels only");
}

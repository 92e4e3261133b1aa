// Argument block and launcher declarations of the long-row path (ddn_long.hpp / ddn_long.hip), kept apart from the
// kernels so that the host translation unit does not compile them.
#pragma once
#include <hip/hip_runtime.h>

#include "ddn_kernels.hpp"

template <typename T>
struct LongArgs {
    RowCommon<T> c;      // H = rows of this launch per image, W, batch, lay, S0, S1, stop, partials, do_consume, do_produce
    cx<T>* Y;            // [3 arrays][batch * H rows][R/2 + 1][M]
    const cx<T>* twM;    // twiddle table of length M (layout of make_table)
    const cx<T>* ed;     // row-spectral D: eD[k], k = 0 .. W/2
    FftDesc fm;          // the length-M transform
    const cx<float>* wtw; // per-lane twiddles of the register-resident 4096-point transform (M = 4096, float) or null
    int R, M;
    int vec;             // pixels per access of the pixel kernels: 1, or 4 when every array is 16-byte aligned and W % 4 == 0
    int narr;            // arrays crossing per direction: 3 (u, K u, D u) / (re, im, s) for the resolvent solvers, 1 otherwise
    // Chambolle-Pock with a row-spectral D (injected eigen arrays): D z and D^T(y1 + y2) cross the row transform as a second
    // spectrum in S1 that bypasses the column stage (hence the factor 1/W), as in k_cp_row<SPEC>:
    int cp_spec;         //   the pixel kernel takes D z from array 1 (dual step) and hands y1 + y2 on as array 1
    int inv_add_s1;      //   k_long_inv: array 0 = S0 + S1 (primal step: K^T y0 + D^T(y1 + y2))
    int fwd_mode;        //   k_long_fwd: 0 plain; 1: S0 = F(a0) and S1 = eD F(a0) / W (primal step: z and D z);
                         //   2: S0 = F(a0), S1 = conj(eD) F(a1) / W (dual step: y0 and D^T(y1 + y2))
};

template <typename T, class P>
int long_row_launch(hipStream_t st, int num_cu, const LongArgs<T>& a, const typename P::Args& pa);
template <typename T>
int long_cp_launch(hipStream_t st, int num_cu, const LongArgs<T>& a, const CpArgs<T>& pa);
template <typename T>
int long_plain_launch(hipStream_t st, int num_cu, const LongArgs<T>& a, const T* src, T* dst, T scale);

// M = 4096, float: the frequency-side kernels with wave-resident transforms (ddn_longw.hip)
bool long_wave_ok(const LongArgs<float>& a);
int long_wave_launch(hipStream_t st, int num_cu, const LongArgs<float>& a, bool inverse);

// Rows of 16384 / 32768 pixels, float, two-tap D: one workgroup per row with the row in LDS (ddn_wholerow.hpp),
// no intermediate array in memory.
struct WholeRowArgs {
    RowCommon<float> c;        // twW: the table exp(-2 pi i k / W); S0, S1, lay, stop, partials, do_consume, do_produce
    const cx<float>* twHalf;   // twiddle table of the length-W/2 transform (layout of make_table)
};
bool whole_row_width(int W);
template <class P>
int whole_row_launch(hipStream_t st, int num_cu, const WholeRowArgs& a, const typename P::Args& pa);
int whole_row_cp_launch(hipStream_t st, int num_cu, const WholeRowArgs& a, const CpArgs<float>& pa);

// Shared host-side helpers of libddn's translation units: error string, HIP error check, the
// dynamic-LDS attribute, and the launchers of the fused row kernels.  The row kernels are explicitly
// instantiated in ddn_row_<solver>_<dtype>.hip (one translation unit each, compiled in parallel);
// ddn_api.hip only references them through `extern template` declarations.
#pragma once
#include <hip/hip_runtime.h>

#include <mutex>
#include <set>
#include <string>
#include <utility>

#include "../../include/ddn.h"
#include "ddn_kernels.hpp"
#include "ddn_wcol_args.hpp"

inline std::string& ddn_err() {
    thread_local std::string e;
    return e;
}
inline int fail(int code, const std::string& msg) {
    ddn_err() = msg;
    return code;
}

#define HIPCHK(expr)                                                                                   \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess) return fail(DDN_E_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

constexpr size_t kLdsMax = 160 * 1024;

// The attribute is set once per (device, kernel): a process that drives several GPUs loads the code object on each.
inline int ensure_lds_attr(const void* fn) {
    static std::mutex mu;
    static std::set<std::pair<int, const void*>> done;
    int dev = 0;
    HIPCHK(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(mu);
    if (done.count({dev, fn})) return DDN_OK;
    HIPCHK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsMax));
    done.insert({dev, fn});
    return DDN_OK;
}

template <typename K, typename C_, typename A_>
int launch_row_fn(hipStream_t st, K fn, int grid, int threads, size_t lds, const C_& c, const A_& a) {
    int rc = ensure_lds_attr((const void*)fn);
    if (rc) return rc;
    hipLaunchKernelGGL(fn, dim3(grid), dim3(threads), lds, st, c, a);
    HIPCHK(hipGetLastError());
    return DDN_OK;
}

// V = 4: 16-byte accesses, four pixel groups per thread, 256-thread workgroups, two per CU.  Float
// kernels for W = 512 / 1024 / 2048 / 4096 use the compile-time specialised transform and hold the
// whole row's global loads in registers (PF).  V = 1: any W up to 8192, scalar accesses.
// (two wavefronts per SIMD: with the whole row's loads held in registers these kernels need ~250 of them; compiled
// for four per SIMD they kept 0.3-0.5 KB per lane in scratch memory and one 512 x 512 image ran 17 % slower)
// Row widths with a compile-time transform in the float LDS kernels beside the powers of two (common image widths;
// every entry costs four kernels of build time).  DDN_ROWW_EXTRA(X) applies X to each.
#ifndef DDN_ROWW_EXTRA
#define DDN_ROWW_EXTRA(X) X(480) X(600) X(640) X(720) X(768) X(800) X(900) X(960) X(1000) X(1080) X(1200) X(1280) X(1440) X(1536) X(1600) \
    X(1920) X(2000) X(2160) X(2400) X(2560) X(2880) X(3000) X(3024) X(3072) X(3200) X(3600) X(3840) X(4000) X(4032)
#endif
// Float rows of 4097 ... 8192 pixels with a compile-time transform: photo and video widths up to 24 megapixels and 8K
// UHD.  Eight pixels per thread, up to 1024 threads (128 registers, twelve to sixteen wavefronts per CU), unless
// DDN_WIDE16 asks for sixteen per thread and 512 threads (256 registers, eight wavefronts: 6000 x 4000 20.8 k against
// 22.7 k Mpx-it/s).
#ifndef DDN_ROWW_WIDE
#define DDN_ROWW_WIDE(X) X(4608) X(5120) X(5184) X(6000) X(6144) X(7680) X(8000)
#endif
inline bool roww_wide_fixed_width(int W) {
#define DDN_X(w) if (W == w) return true;
    DDN_ROWW_WIDE(DDN_X)
#undef DDN_X
    return false;
}
inline bool roww_fixed_width(int W) {
    if (W == 512 || W == 1024 || W == 2048 || W == 4096) return true;
#define DDN_X(w) if (W == w) return true;
    DDN_ROWW_EXTRA(DDN_X)
#undef DDN_X
    return false;
}
#ifndef DDN_ROWW_MINW
#define DDN_ROWW_MINW 2
#endif
// dense: the image has several rounds of row workgroups per CU -- the variant compiled for four wavefronts per SIMD (128
// registers, loads issued where they are used) keeps twice the workgroups resident, and one's transform phases
// overlap another's memory phases (one 3840 x 3840 image, primal DR row launch 0.358 -> 0.266 ms); the two-per-SIMD
// variant with the whole row's loads in registers is for images whose rows are all in flight at once.
template <typename T, class P, int WF>
int row_launch_w(hipStream_t st, int grid, int threads, size_t lds, bool dense, const RowCommon<T>& c, const typename P::Args& a) {
    if (dense) return launch_row_fn(st, k_solver_row<T, P, 4, 2, false, 0, WF, 512, 4>, grid, threads, lds, c, a);
    return launch_row_fn(st, k_solver_row<T, P, 4, 2, false, 1, WF, 512, DDN_ROWW_MINW>, grid, threads, lds, c, a);
}

// Double, W = 512 / 1024 / 2048 / 4096: the compile-time specialised transform as well (radix-8 passes, W / 8 threads,
// one workgroup of up to eight wavefronts per CU at ~230 registers).  The run-time-radix transform in double does
// not fit a lane's 256 registers -- its kernels keep 2.5 KB per lane in scratch memory -- and stays for the other widths.
template <typename T, class P, int WF>
int row_launch_d(hipStream_t st, int grid, int threads, size_t lds, bool spec, const RowCommon<T>& c, const typename P::Args& a) {
    if (spec) return launch_row_fn(st, k_solver_row<T, P, 4, 2, true, 0, WF, 512, 2>, grid, threads, lds, c, a);
    return launch_row_fn(st, k_solver_row<T, P, 4, 2, false, 0, WF, 512, 2>, grid, threads, lds, c, a);
}

// Wave-per-row kernels (W = 4096, float, 2-tap D; ddn_wrow.hpp), instantiated in ddn_wrow_*.hip.
template <class P>
int wrow_launch(hipStream_t st, int grid, const RowCommon<float>& c, const typename P::Args& a);
int wrow_cp_launch(hipStream_t st, int grid, const RowCommon<float>& c, const CpArgs<float>& a);
// The same for W = 512, 1024, 2048 (several rows per wavefront; ddn_wrown.hpp), in ddn_wrown_*.hip.
template <class P>
int wrown_launch(hipStream_t st, int grid, const RowCommon<float>& c, const typename P::Args& a);
int wrown_cp_launch(hipStream_t st, int grid, const RowCommon<float>& c, const CpArgs<float>& a);
// Wave-per-column single-pass column stage (H = 4096, float; ddn_wcol.hpp), in ddn_wcol.hip.
int wcol_launch(hipStream_t st, int mode, const wcol::Args& a);
// k_col<float, COL_RES2 / COL_MUL, n> with the compile-time transform of n rows, or null when n is not in the list
// (ddn_colfix.hpp)
const void* col_fixed_res2_f32(int n);
const void* col_fixed_mul_f32(int n);
const void* col_fixed_fwd2_f32(int n);   // COL_FWD2 / COL_FWD: the short factor of a split column
const void* col_fixed_fwd_f32(int n);
// The same for H = 512, 1024, 2048 (several adjacent columns per wavefront; ddn_wcoln.hpp), in ddn_wcoln.hip.
int wcoln_launch(hipStream_t st, int mode, const wcol::Args& a);

// SSIM (core/loss.py:39-110) without allocation or synchronisation, for use inside the solver loops (ddn_ssim.hip)
size_t ssim_partial_count(int H, int W, int planes);
int ssim_launch(int dtype, const void* x, const void* y, int H, int W, int planes, double max_val, const float* g11, double* partial_dev, double* out_dev,
                int out_stride, int clip_x, hipStream_t st);
int ssim_commit_launch(const double* value, double* slot, int slot_stride, int* stop, int batch, int iter, int first_iter, double tol, int test, hipStream_t st);

template <typename T, class P>
int row_launch(hipStream_t st, int grid, int threads, size_t lds, int V, const RowCommon<T>& c, const typename P::Args& a) {
    const bool spec = c.eDrow != nullptr;
    if constexpr (sizeof(T) == 4) {
        if (V == 0) return c.W == 4096 ? wrow_launch<P>(st, grid, c, a) : wrown_launch<P>(st, grid, c, a);
    }
    const bool dense = V == 5;   // V == 5: as 4, with the four-wavefronts-per-SIMD variant of the compile-time-width float kernels
    if (V == 5) V = 4;
    if (V == 4) {
        if constexpr (sizeof(T) == 8) {
            switch (c.W) {
                case 512: return row_launch_d<T, P, 512>(st, grid, threads, lds, spec, c, a);
                case 1024: return row_launch_d<T, P, 1024>(st, grid, threads, lds, spec, c, a);
                case 2048: return row_launch_d<T, P, 2048>(st, grid, threads, lds, spec, c, a);
                case 4096: return row_launch_d<T, P, 4096>(st, grid, threads, lds, spec, c, a);
                default: break;
            }
        }
        if (spec) return launch_row_fn(st, k_solver_row<T, P, 4, 4, true, 0, 0, 256, 2>, grid, threads, lds, c, a);
        if constexpr (sizeof(T) == 4) {
            switch (c.W) {
                case 512: return row_launch_w<T, P, 512>(st, grid, threads, lds, dense, c, a);
                case 1024: return row_launch_w<T, P, 1024>(st, grid, threads, lds, dense, c, a);
                case 2048: return row_launch_w<T, P, 2048>(st, grid, threads, lds, dense, c, a);
                case 4096: return row_launch_w<T, P, 4096>(st, grid, threads, lds, dense, c, a);
#define DDN_X(w) case w: return row_launch_w<T, P, w>(st, grid, threads, lds, dense, c, a);
                DDN_ROWW_EXTRA(DDN_X)
#undef DDN_X
                default: break;
            }
        }
        // float, any width: one, two or four pixel groups per thread -- the launch shape (row_cfg) gives a row as many
        // threads as the image's rows leave wavefronts for; the kernel with the fewest groups that covers the row is taken
        if constexpr (sizeof(T) == 4) {
            if (threads <= 512 && threads * 4 >= c.W) return launch_row_fn(st, k_solver_row<T, P, 4, 1, false, 0, 0, 512, 2>, grid, threads, lds, c, a);
            if (threads <= 512 && threads * 8 >= c.W) return launch_row_fn(st, k_solver_row<T, P, 4, 2, false, 0, 0, 512, 2>, grid, threads, lds, c, a);
        }
        if constexpr (sizeof(T) == 4) {
            if (threads > 256) {   // rows of 4097 ... 8192 pixels
                switch (c.W) {
#define DDN_X(w) case w: return threads > 512 ? launch_row_fn(st, k_solver_row<T, P, 4, 2, false, 0, w, 1024, 1>, grid, threads, lds, c, a) \
                                      : launch_row_fn(st, k_solver_row<T, P, 4, 4, false, 0, w, 512, 2>, grid, threads, lds, c, a);
                    DDN_ROWW_WIDE(DDN_X)
#undef DDN_X
                    default: break;
                }
                return launch_row_fn(st, k_solver_row<T, P, 4, 4, false, 0, 0, 512, 2>, grid, threads, lds, c, a);
            }
        }
        return launch_row_fn(st, k_solver_row<T, P, 4, 4, false, 0, 0, 256, 2>, grid, threads, lds, c, a);
    }
    if (spec) return launch_row_fn(st, k_solver_row<T, P, 1, 8, true, 0, 0, 1024, 1>, grid, threads, lds, c, a);
    return launch_row_fn(st, k_solver_row<T, P, 1, 8, false, 0, 0, 1024, 1>, grid, threads, lds, c, a);
}

template <typename T, int WF>
int cp_launch_w(hipStream_t st, int grid, int threads, size_t lds, bool dense, const RowCommon<T>& c, const CpArgs<T>& a) {
    if (dense) return launch_row_fn(st, k_cp_row<T, 4, 2, false, WF, 512, 4>, grid, threads, lds, c, a);
    return launch_row_fn(st, k_cp_row<T, 4, 2, false, WF, 512, DDN_ROWW_MINW>, grid, threads, lds, c, a);
}

template <typename T, int WF>
int cp_launch_d(hipStream_t st, int grid, int threads, size_t lds, bool spec, const RowCommon<T>& c, const CpArgs<T>& a) {
    if (spec) return launch_row_fn(st, k_cp_row<T, 4, 2, true, WF, 512, 2>, grid, threads, lds, c, a);
    return launch_row_fn(st, k_cp_row<T, 4, 2, false, WF, 512, 2>, grid, threads, lds, c, a);
}

template <typename T>
int cp_launch(hipStream_t st, int grid, int threads, size_t lds, int V, const RowCommon<T>& c, const CpArgs<T>& a) {
    const bool spec = c.eDrow != nullptr;
    if constexpr (sizeof(T) == 4) {
        if (V == 0) return c.W == 4096 ? wrow_cp_launch(st, grid, c, a) : wrown_cp_launch(st, grid, c, a);
    }
    const bool dense = V == 5;
    if (V == 5) V = 4;
    if (V == 4) {
        if constexpr (sizeof(T) == 8) {
            switch (c.W) {
                case 512: return cp_launch_d<T, 512>(st, grid, threads, lds, spec, c, a);
                case 1024: return cp_launch_d<T, 1024>(st, grid, threads, lds, spec, c, a);
                case 2048: return cp_launch_d<T, 2048>(st, grid, threads, lds, spec, c, a);
                case 4096: return cp_launch_d<T, 4096>(st, grid, threads, lds, spec, c, a);
                default: break;
            }
        }
        if (spec) return launch_row_fn(st, k_cp_row<T, 4, 4, true, 0, 256, 2>, grid, threads, lds, c, a);
        if constexpr (sizeof(T) == 4) {
            switch (c.W) {
                case 512: return cp_launch_w<T, 512>(st, grid, threads, lds, dense, c, a);
                case 1024: return cp_launch_w<T, 1024>(st, grid, threads, lds, dense, c, a);
                case 2048: return cp_launch_w<T, 2048>(st, grid, threads, lds, dense, c, a);
                case 4096: return cp_launch_w<T, 4096>(st, grid, threads, lds, dense, c, a);
#define DDN_X(w) case w: return cp_launch_w<T, w>(st, grid, threads, lds, dense, c, a);
                DDN_ROWW_EXTRA(DDN_X)
#undef DDN_X
                default: break;
            }
        }
        if constexpr (sizeof(T) == 4) {
            if (threads <= 512 && threads * 4 >= c.W) return launch_row_fn(st, k_cp_row<T, 4, 1, false, 0, 512, 2>, grid, threads, lds, c, a);
            if (threads <= 512 && threads * 8 >= c.W) return launch_row_fn(st, k_cp_row<T, 4, 2, false, 0, 512, 2>, grid, threads, lds, c, a);
        }
        if constexpr (sizeof(T) == 4) {
            if (threads > 256) {
                switch (c.W) {
#define DDN_X(w) case w: return threads > 512 ? launch_row_fn(st, k_cp_row<T, 4, 2, false, w, 1024, 1>, grid, threads, lds, c, a) \
                                      : launch_row_fn(st, k_cp_row<T, 4, 4, false, w, 512, 2>, grid, threads, lds, c, a);
                    DDN_ROWW_WIDE(DDN_X)
#undef DDN_X
                    default: break;
                }
                return launch_row_fn(st, k_cp_row<T, 4, 4, false, 0, 512, 2>, grid, threads, lds, c, a);
            }
        }
        return launch_row_fn(st, k_cp_row<T, 4, 4, false, 0, 256, 2>, grid, threads, lds, c, a);
    }
    if (spec) return launch_row_fn(st, k_cp_row<T, 1, 8, true, 0, 1024, 1>, grid, threads, lds, c, a);
    return launch_row_fn(st, k_cp_row<T, 1, 8, false, 0, 1024, 1>, grid, threads, lds, c, a);
}

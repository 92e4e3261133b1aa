// Fused SSIM (core/loss.py:39-110 of the reference): 11 x 11 Gaussian window (sigma 1.5, zero padding)
// applied to x, y, x*x, y*y, x*y, the SSIM map and its mean, in one pass over the two images.
// A workgroup produces a 16 x 64 tile of the map: it stages the (16+10) x (64+10) input patch of both
// images in LDS, runs the horizontal taps of the five maps into LDS, then the vertical taps per output
// pixel, and sums the map in double.  The window is separable (the reference builds it as an outer
// product); per image the partial sums are reduced in a fixed order by a second kernel.
#include <algorithm>

#include "ddn_launch.hpp"

namespace {

constexpr int TH = 16, TW = 64, R = 5, PH = TH + 2 * R, PW = TW + 2 * R;

struct SsimWin {
    float g[11];       // separable factor (float path)
    float w2[121];     // the reference's window: float32(g[i] * g[j]) (double path, tap for tap)
};

template <typename T>
__global__ void __launch_bounds__(256) k_ssim_tile(const T* x, const T* y, int H, int W, int tiles_x, int tiles_y, SsimWin win, double c1, double c2, double* partial, int clip_x) {
    __shared__ T sx[PH][PW + 1], sy[PH][PW + 1];
    __shared__ T hm[5][PH][TW + 1];
    __shared__ double red[64];
    const int tid = threadIdx.x;
    int b = blockIdx.x;
    const int tx = b % tiles_x;
    b /= tiles_x;
    const int ty = b % tiles_y;
    const int plane = b / tiles_y;
    const T* xp = x + (size_t)plane * H * W;
    const T* yp = y + (size_t)plane * H * W;
    const int r0 = ty * TH - R, c0 = tx * TW - R;
    for (int i = tid; i < PH * PW; i += 256) {
        const int pr = i / PW, pc = i - pr * PW;
        const int r = r0 + pr, c = c0 + pc;
        const bool in = r >= 0 && r < H && c >= 0 && c < W;
        T vx0 = in ? xp[(size_t)r * W + c] : (T)0;
        if (clip_x) vx0 = vx0 < (T)0 ? (T)0 : (vx0 > (T)1 ? (T)1 : vx0);   // the solver's clip(p) (primal_dual_dr/algorithm.py:104)
        sx[pr][pc] = vx0;
        sy[pr][pc] = in ? yp[(size_t)r * W + c] : (T)0;
    }
    __syncthreads();
    if constexpr (sizeof(T) == 8) {
        // float64: the reference's window is the float32-rounded outer product, which is not exactly
        // separable; evaluate its 121 taps as they are (parity to 1e-11; not a performance path)
        double acc2 = 0;
        for (int i = tid; i < TH * TW; i += 256) {
            const int orow = i / TW, oc = i - orow * TW;
            const int r = ty * TH + orow, c = tx * TW + oc;
            if (r < H && c < W) {
                T m[5] = {0, 0, 0, 0, 0};
                for (int u = 0; u < 11; ++u)
#pragma unroll
                    for (int v = 0; v < 11; ++v) {
                        const T g = (T)win.w2[u * 11 + v];
                        const T vx = sx[orow + u][oc + v], vy = sy[orow + u][oc + v];
                        m[0] += g * vx; m[1] += g * vy; m[2] += g * (vx * vx); m[3] += g * (vy * vy); m[4] += g * (vx * vy);
                    }
                const T mux = m[0], muy = m[1];
                const T vx = m[2] - mux * mux, vy = m[3] - muy * muy, cxy = m[4] - mux * muy;
                const T num = ((T)2 * mux * muy + (T)c1) * ((T)2 * cxy + (T)c2);
                const T den = (mux * mux + muy * muy + (T)c1) * (vx + vy + (T)c2);
                acc2 += (double)(num / den);
            }
        }
        for (int off = 32; off >= 1; off >>= 1) acc2 += __shfl_xor(acc2, off, 64);
        if ((tid & 63) == 0) red[tid >> 6] = acc2;
        __syncthreads();
        if (tid == 0) partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
        return;
    }
    // horizontal taps: PH x TW outputs of five maps
    for (int i = tid; i < PH * TW; i += 256) {
        const int pr = i / TW, oc = i - pr * TW;
        T a0 = 0, a1 = 0, a2 = 0, a3 = 0, a4 = 0;
#pragma unroll
        for (int k = 0; k < 11; ++k) {
            const T g = (T)win.g[k];
            const T vx = sx[pr][oc + k], vy = sy[pr][oc + k];
            a0 += g * vx;
            a1 += g * vy;
            a2 += g * (vx * vx);
            a3 += g * (vy * vy);
            a4 += g * (vx * vy);
        }
        hm[0][pr][oc] = a0; hm[1][pr][oc] = a1; hm[2][pr][oc] = a2; hm[3][pr][oc] = a3; hm[4][pr][oc] = a4;
    }
    __syncthreads();
    double acc = 0;
    for (int i = tid; i < TH * TW; i += 256) {
        const int orow = i / TW, oc = i - orow * TW;
        const int r = ty * TH + orow, c = tx * TW + oc;
        if (r < H && c < W) {
            T m[5] = {0, 0, 0, 0, 0};
#pragma unroll
            for (int k = 0; k < 11; ++k) {
                const T g = (T)win.g[k];
#pragma unroll
                for (int q = 0; q < 5; ++q) m[q] += g * hm[q][orow + k][oc];
            }
            const T mux = m[0], muy = m[1];
            const T vx = m[2] - mux * mux, vy = m[3] - muy * muy, cxy = m[4] - mux * muy;
            const T num = ((T)2 * mux * muy + (T)c1) * ((T)2 * cxy + (T)c2);
            const T den = (mux * mux + muy * muy + (T)c1) * (vx + vy + (T)c2);
            acc += (double)(num / den);
        }
    }
    // workgroup sum in a fixed order
    for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off, 64);
    if ((tid & 63) == 0) red[tid >> 6] = acc;
    __syncthreads();
    if (tid == 0) partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// Any window size (the reference's `window_size` argument, core/loss.py:39,75-81): direct evaluation of the
// ws x ws taps float32(g[u] g[v]) under zero padding ws / 2, one thread per entry of the map -- which has
// H + 2 (ws / 2) - ws + 1 rows, that is H + 1 for an even window, as the reference's conv2d calls produce.
// Sums in double; not a performance path (the solvers' loops use the 11-tap kernel above).
constexpr int kMaxWin = 63;
struct SsimWinAny {
    float g[kMaxWin];
    int ws;
};
template <typename T>
__global__ void __launch_bounds__(256) k_ssim_any(const T* x, const T* y, int H, int W, int OH, int OW, SsimWinAny win, double c1, double c2, double* partial, int per_plane) {
    __shared__ double red[4];
    const int tid = threadIdx.x;
    const int plane = blockIdx.x / per_plane, blk = blockIdx.x - plane * per_plane;
    const T* xp = x + (size_t)plane * H * W;
    const T* yp = y + (size_t)plane * H * W;
    const int ws = win.ws, pad = ws / 2;
    double acc = 0;
    for (size_t i = (size_t)blk * 256 + tid; i < (size_t)OH * OW; i += (size_t)per_plane * 256) {
        const int orow = (int)(i / OW), oc = (int)(i - (size_t)orow * OW);
        double m[5] = {0, 0, 0, 0, 0};
        for (int u = 0; u < ws; ++u) {
            const int r = orow + u - pad;
            if (r < 0 || r >= H) continue;
            for (int v = 0; v < ws; ++v) {
                const int c = oc + v - pad;
                if (c < 0 || c >= W) continue;
                const double g = (double)(win.g[u] * win.g[v]);       // the float32 product, as g[:, None] * g[None, :]
                const double vx = (double)xp[(size_t)r * W + c], vy = (double)yp[(size_t)r * W + c];
                m[0] += g * vx; m[1] += g * vy; m[2] += g * (vx * vx); m[3] += g * (vy * vy); m[4] += g * (vx * vy);
            }
        }
        const double mux = m[0], muy = m[1];
        const double vx = m[2] - mux * mux, vy = m[3] - muy * muy, cxy = m[4] - mux * muy;
        acc += ((2 * mux * muy + c1) * (2 * cxy + c2)) / ((mux * mux + muy * muy + c1) * (vx + vy + c2));
    }
    for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off, 64);
    if ((tid & 63) == 0) red[tid >> 6] = acc;
    __syncthreads();
    if (tid == 0) partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

__global__ void k_ssim_final(const double* partial, int per_plane, double npix, double* out, int out_stride) {
    __shared__ double red[256];
    const int plane = blockIdx.x, tid = threadIdx.x;
    double a = 0;
    for (int i = tid; i < per_plane; i += 256) a += partial[(size_t)plane * per_plane + i];
    red[tid] = a;
    __syncthreads();
    for (int s = 128; s >= 1; s >>= 1) {
        if (tid < s) red[tid] += red[tid + s];
        __syncthreads();
    }
    if (tid == 0) out[(size_t)plane * out_stride] = red[0] / npix;
}

// X^T w for the k x k blind-kernel step (op_math/python_code/apply_x_trans.py:4-46), k odd:
// out[i][j] = sum_{p,q} w[p][q] x[p + i - k/2][q + j - k/2], x zero outside (the reference correlates x
// with a one-hot kernel at (i, j) under zero padding k/2 and sums the product with w).  One thread
// per output entry, double accumulation.
template <typename T>
__global__ void k_apply_x_trans(const T* __restrict__ w, const T* __restrict__ x, int k, T* __restrict__ out) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= k * k) return;
    const int i = idx / k, j = idx - i * k, c = k / 2;
    double acc = 0.0;
    for (int p = 0; p < k; ++p) {
        const int xr = p + i - c;
        if (xr < 0 || xr >= k) continue;
        for (int q = 0; q < k; ++q) {
            const int xc = q + j - c;
            if (xc < 0 || xc >= k) continue;
            acc += (double)w[p * k + q] * (double)x[xr * k + xc];
        }
    }
    out[idx] = (T)acc;
}

// In-loop use: value[image] -> this iteration's slot of the caller's record, and the stop test, unless the image is
// already frozen (it keeps its last record, as with the other loss kinds).
__global__ void k_ssim_commit(const double* value, double* slot, int slot_stride, int* stop, int batch, int iter, int first_iter, double tol, int test) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= batch) return;
    if (stop && stop[i] >= 0) return;
    const double v = value[i];
    slot[(size_t)i * slot_stride] = v;
    if (test && stop && iter >= first_iter && v < tol) stop[i] = iter;
}

}  // namespace

size_t ssim_partial_count(int H, int W, int planes) {
    return (size_t)((W + TW - 1) / TW) * ((H + TH - 1) / TH) * planes;
}

// The two SSIM kernels on `st`, no allocation, no synchronisation: partial_dev holds ssim_partial_count doubles,
// out_dev[plane * out_stride] receives the mean SSIM of each plane.
int ssim_launch(int dtype, const void* x, const void* y, int H, int W, int planes, double max_val, const float* g11, double* partial_dev, double* out_dev,
                int out_stride, int clip_x, hipStream_t st) {
    SsimWin win;
    for (int i = 0; i < 11; ++i) win.g[i] = g11[i];
    for (int i = 0; i < 11; ++i)
        for (int j = 0; j < 11; ++j) win.w2[i * 11 + j] = g11[i] * g11[j];  // float32 product, as g[:, None] * g[None, :]
    const int tiles_x = (W + TW - 1) / TW, tiles_y = (H + TH - 1) / TH;
    const int per_plane = tiles_x * tiles_y;
    const double c1 = (0.01 * max_val) * (0.01 * max_val), c2 = (0.03 * max_val) * (0.03 * max_val);
    if (dtype == DDN_F32)
        hipLaunchKernelGGL(k_ssim_tile<float>, dim3(per_plane * planes), dim3(256), 0, st, (const float*)x, (const float*)y, H, W, tiles_x, tiles_y, win, c1, c2, partial_dev, clip_x);
    else
        hipLaunchKernelGGL(k_ssim_tile<double>, dim3(per_plane * planes), dim3(256), 0, st, (const double*)x, (const double*)y, H, W, tiles_x, tiles_y, win, c1, c2, partial_dev, clip_x);
    hipLaunchKernelGGL(k_ssim_final, dim3(planes), dim3(256), 0, st, partial_dev, per_plane, (double)H * (double)W, out_dev, out_stride);
    HIPCHK(hipGetLastError());
    return DDN_OK;
}

int ssim_commit_launch(const double* value, double* slot, int slot_stride, int* stop, int batch, int iter, int first_iter, double tol, int test, hipStream_t st) {
    hipLaunchKernelGGL(k_ssim_commit, dim3((batch + 63) / 64), dim3(64), 0, st, value, slot, slot_stride, stop, batch, iter, first_iter, tol, test);
    HIPCHK(hipGetLastError());
    return DDN_OK;
}

extern "C" int ddn_apply_x_trans(int dtype, const void* w_dev, const void* x_dev, int kernel_size, void* out_dev, void* stream) {
    if (!w_dev || !x_dev || !out_dev || kernel_size < 1) return fail(DDN_E_ARG, "apply_x_trans: bad arguments");
    if (kernel_size % 2 == 0) return fail(DDN_E_UNSUPPORTED, "apply_x_trans: kernel_size must be odd (the reference's padded correlation is (k+1) x (k+1) for even k and cannot be multiplied with w)");
    hipStream_t st = (hipStream_t)stream;
    const int n = kernel_size * kernel_size;
    if (dtype == DDN_F32)
        hipLaunchKernelGGL(k_apply_x_trans<float>, dim3((n + 255) / 256), dim3(256), 0, st, (const float*)w_dev, (const float*)x_dev, kernel_size, (float*)out_dev);
    else
        hipLaunchKernelGGL(k_apply_x_trans<double>, dim3((n + 255) / 256), dim3(256), 0, st, (const double*)w_dev, (const double*)x_dev, kernel_size, (double*)out_dev);
    HIPCHK(hipGetLastError());
    return DDN_OK;
}

extern "C" int ddn_ssim(int dtype, const void* x, const void* y, int H, int W, int planes, double max_val, const float* g11_host, double* out_dev, void* stream) {
    if (!x || !y || !out_dev || !g11_host || H < 1 || W < 1 || planes < 1) return fail(DDN_E_ARG, "ssim: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    double* part = nullptr;
    HIPCHK(hipMalloc((void**)&part, sizeof(double) * ssim_partial_count(H, W, planes)));
    const int rc = ssim_launch(dtype, x, y, H, W, planes, max_val, g11_host, part, out_dev, 1, 0, st);
    const hipError_t e = hipStreamSynchronize(st);
    (void)hipFree(part);
    if (rc) return rc;
    if (e != hipSuccess) return fail(DDN_E_HIP, hipGetErrorString(e));
    return DDN_OK;
}

extern "C" int ddn_ssim_window(int dtype, const void* x, const void* y, int H, int W, int planes, double max_val, const float* g_host, int window_size,
                               double* out_dev, void* stream) {
    if (!x || !y || !out_dev || !g_host || H < 1 || W < 1 || planes < 1 || window_size < 1) return fail(DDN_E_ARG, "ssim_window: bad arguments");
    if (window_size > kMaxWin) return fail(DDN_E_UNSUPPORTED, "ssim_window: windows of more than 63 taps are not supported");
    hipStream_t st = (hipStream_t)stream;
    SsimWinAny win;
    win.ws = window_size;
    for (int i = 0; i < kMaxWin; ++i) win.g[i] = i < window_size ? g_host[i] : 0.f;
    const int OH = H + 2 * (window_size / 2) - window_size + 1, OW = W + 2 * (window_size / 2) - window_size + 1;
    const int per_plane = (int)std::min<size_t>(1024, ((size_t)OH * OW + 255) / 256);
    double* part = nullptr;
    HIPCHK(hipMalloc((void**)&part, sizeof(double) * (size_t)per_plane * planes));
    const double c1 = (0.01 * max_val) * (0.01 * max_val), c2 = (0.03 * max_val) * (0.03 * max_val);
    if (dtype == DDN_F32)
        hipLaunchKernelGGL(k_ssim_any<float>, dim3(per_plane * planes), dim3(256), 0, st, (const float*)x, (const float*)y, H, W, OH, OW, win, c1, c2, part, per_plane);
    else
        hipLaunchKernelGGL(k_ssim_any<double>, dim3(per_plane * planes), dim3(256), 0, st, (const double*)x, (const double*)y, H, W, OH, OW, win, c1, c2, part, per_plane);
    hipLaunchKernelGGL(k_ssim_final, dim3(planes), dim3(256), 0, st, part, per_plane, (double)OH * (double)OW, out_dev, 1);
    const hipError_t le = hipGetLastError();
    const hipError_t e = hipStreamSynchronize(st);
    (void)hipFree(part);
    if (le != hipSuccess) return fail(DDN_E_HIP, hipGetErrorString(le));
    if (e != hipSuccess) return fail(DDN_E_HIP, hipGetErrorString(e));
    return DDN_OK;
}

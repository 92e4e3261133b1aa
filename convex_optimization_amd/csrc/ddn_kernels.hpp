// gfx950 kernels of the deblur_denoise iteration hot path.
//
// Data layout in HBM
//   images / iterates   real [batch][H][W], row-major
//   half-row-spectra    complex [batch][H][Wp]: for image row i, entries k2 = 0..W/2 of the
//                       1-D transform of that row (natural k2 order); Wp = W/2+1 rounded up to 32
//   multipliers (eK, G) [H][Wp] with the ROW index in column-transform position order
//                       (digit-reversed k1), shared by all images
//
// One 2-D real transform pair = a "row kernel" (one workgroup per image row or row pair: packs two
// real sequences into one complex length-W transform in LDS, separates the two half-spectra by
// Hermitian symmetry) and a "column kernel" (one workgroup per tile of C adjacent columns: forward
// transform along H in LDS, pointwise spectral algebra, inverse transform, in place).
// Every solver iteration is: column kernel -> fused row kernel (inverse row transform, D / D^T
// stencils, prox maps, relaxation and dual updates, loss partial sums, forward row transform).
#pragma once
#include "ddn_fft.hpp"

// --------------------------------------------------------------------------- small helpers
template <typename T>
__device__ __forceinline__ T t_abs(T v) { return v < 0 ? -v : v; }
template <typename T>
__device__ __forceinline__ T clip01(T v) { return v < (T)0 ? (T)0 : (v > (T)1 ? (T)1 : v); }
// prox_l1 (core/proximal_operators.py:8): sign(x) * max(|x| - lam, 0)
template <typename T>
__device__ __forceinline__ T soft(T v, T lam) {
    T a = t_abs(v) - lam;
    a = a > (T)0 ? a : (T)0;
    return v < (T)0 ? -a : a;
}
__device__ __forceinline__ float t_sqrt(float v) { return sqrtf(v); }
__device__ __forceinline__ double t_sqrt(double v) { return sqrt(v); }
// prox_iso (core/proximal_operators.py:22-50): per-pixel 2-vector shrink, zero where norm <= lam
template <typename T>
__device__ __forceinline__ void iso(T a, T c, T lam, T& oa, T& oc) {
    const T n = t_sqrt(a * a + c * c);
    const T f = n > lam ? (T)1 - lam / n : (T)0;
    oa = f * a;
    oc = f * c;
}

template <typename T, int V>
__device__ __forceinline__ void vload(const T* p, T (&r)[V]) {
    if constexpr (V == 1) {
        r[0] = p[0];
    } else if constexpr (sizeof(T) == 4) {
        const float4 t = *reinterpret_cast<const float4*>(p);
        r[0] = t.x; r[1] = t.y; r[2] = t.z; r[3] = t.w;
    } else {
        const double2 a = *reinterpret_cast<const double2*>(p);
        const double2 b = *(reinterpret_cast<const double2*>(p) + 1);
        r[0] = a.x; r[1] = a.y; r[2] = b.x; r[3] = b.y;
    }
}
template <typename T, int V>
__device__ __forceinline__ void vstore(T* p, const T (&r)[V]) {
    if constexpr (V == 1) {
        p[0] = r[0];
    } else if constexpr (sizeof(T) == 4) {
        *reinterpret_cast<float4*>(p) = make_float4(r[0], r[1], r[2], r[3]);
    } else {
        *reinterpret_cast<double2*>(p) = make_double2(r[0], r[1]);
        *(reinterpret_cast<double2*>(p) + 1) = make_double2(r[2], r[3]);
    }
}

// Sum NS values over the workgroup; result valid in thread 0.  red: LDS scratch of >= 16*NS doubles.
template <int NS>
__device__ __forceinline__ void block_reduce(double (&v)[NS], double* red, int tid, int nt) {
#pragma unroll
    for (int s = 0; s < NS; ++s) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v[s] += __shfl_down(v[s], off, 64);
    }
    const int wave = tid >> 6, lane = tid & 63, nw = (nt + 63) >> 6;
    if (lane == 0) {
#pragma unroll
        for (int s = 0; s < NS; ++s) red[wave * NS + s] = v[s];
    }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < nw; ++w) {
#pragma unroll
            for (int s = 0; s < NS; ++s) v[s] += red[w * NS + s];
        }
    }
    __syncthreads();
}

template <typename T>
__device__ __forceinline__ void load_table(cx<T>* dst, const cx<T>* src, int n, int tid, int nt) {
    for (int i = tid; i < n; i += nt) dst[i] = src[i];
}

// ------------------------------------------------------------------- row transform helpers
// Pack the half-spectra of two real rows a, b (srcA, srcB; either may be null = zero) into the
// full spectrum of a + i b, at digit-reversed LDS positions.  No barrier inside.
template <typename T>
__device__ __forceinline__ void row_pack(cx<T>* lds, const FftDesc& fw, int W, const cx<T>* srcA, const cx<T>* srcB, int tid, int nt,
                                         const cx<T>* addA = nullptr, const cx<T>* addB = nullptr) {
    const int half = W >> 1;
    for (int k = tid; k <= half; k += nt) {
        cx<T> U = srcA ? srcA[k] : mk<T>(0, 0);
        cx<T> V = srcB ? srcB[k] : mk<T>(0, 0);
        if (addA) U = U + addA[k];
        if (addB) V = V + addB[k];
        const int km = (k == 0) ? 0 : W - k;
        if (km == k) {
            lds[SW(pos_of_freq(fw, k))] = mk<T>(U.x, V.x);  // DC / Nyquist of a real row: imaginary part dropped
        } else {
            lds[SW(pos_of_freq(fw, k))] = mk<T>(U.x - V.y, U.y + V.x);
            lds[SW(pos_of_freq(fw, km))] = mk<T>(U.x + V.y, V.x - U.y);
        }
    }
}
// As row_pack but a single source multiplied by a row table (row-spectral D: U[k] * eD[k]).
template <typename T>
__device__ __forceinline__ void row_pack_mul(cx<T>* lds, const FftDesc& fw, int W, const cx<T>* srcA, const cx<T>* rowmul, int tid, int nt) {
    const int half = W >> 1;
    for (int k = tid; k <= half; k += nt) {
        cx<T> U = srcA[k] * rowmul[k];
        const int km = (k == 0) ? 0 : W - k;
        if (km == k) {
            lds[SW(pos_of_freq(fw, k))] = mk<T>(U.x, 0);
        } else {
            lds[SW(pos_of_freq(fw, k))] = U;
            lds[SW(pos_of_freq(fw, km))] = cconj(U);
        }
    }
}
// LDS holds the digit-reversed spectrum Z of a + i b.  Separate A = F(a), B = F(b) for k2 = 0..W/2
// and store them (null destination = skip).  mulA != null: dstA[k] (+)= conj(mulA[k]) * A instead
// (row-spectral D^T); accA adds to what dstA already holds.
template <typename T>
__device__ __forceinline__ void row_unpack(const cx<T>* lds, const FftDesc& fw, int W, cx<T>* dstA, cx<T>* dstB,
                                           const cx<T>* mulA, bool accA, int tid, int nt) {
    const int half = W >> 1;
    for (int k = tid; k <= half; k += nt) {
        const int km = (k == 0) ? 0 : W - k;
        const cx<T> Z1 = lds[SW(pos_of_freq(fw, k))];
        const cx<T> Z2 = lds[SW(pos_of_freq(fw, km))];
        if (dstA) {
            cx<T> A = mk<T>((T)0.5 * (Z1.x + Z2.x), (T)0.5 * (Z1.y - Z2.y));
            if (mulA) A = mulc(A, mulA[k]);
            if (accA) A = A + dstA[k];
            dstA[k] = A;
        }
        if (dstB) dstB[k] = mk<T>((T)0.5 * (Z1.y + Z2.y), (T)0.5 * (Z2.x - Z1.x));
    }
}

// ----------------------------------------------------------------------------- loss finalize
struct LossFin {
    const double* partials;  // [batch][rows_per_plane][4]
    double* ring;            // this iteration's slot: [batch][8]; may be null
    int* stop;               // [batch]; -1 = running, else iteration count at which frozen; may be null
    int rows_per_plane;
    int tol_kind;            // DDN_LOSS_*; 0 = never stop
    int iter;                // iteration count this slot describes (1-based)
    int first_iter;          // tolerance tested only for iter >= first_iter
    double tol;
    double npix;
    int enabled;
};

__device__ __forceinline__ void loss_finalize(const LossFin& lf, int plane, double* red, int tid, int nt) {
    if (lf.stop && lf.stop[plane] >= 0) return;  // frozen image keeps its last record
    double v[4] = {0, 0, 0, 0};
    const double* p = lf.partials + (size_t)plane * lf.rows_per_plane * 4;
    for (int r = tid; r < lf.rows_per_plane; r += nt) {
#pragma unroll
        for (int s = 0; s < 4; ++s) v[s] += p[r * 4 + s];
    }
    block_reduce<4>(v, red, tid, nt);
    if (tid == 0) {
        double out[8];
        for (int h = 0; h < 2; ++h) {
            const double s1 = v[2 * h], s2 = v[2 * h + 1];
            const double mse = s2 / lf.npix;
            out[4 * h + 0] = s1;
            out[4 * h + 1] = sqrt(s2);
            out[4 * h + 2] = mse;
            out[4 * h + 3] = (mse == 0.0) ? (double)INFINITY : 10.0 * log10(1.0 / mse);
        }
        if (lf.ring) {
            for (int i = 0; i < 8; ++i) lf.ring[plane * 8 + i] = out[i];
        }
        if (lf.stop && lf.tol_kind > 0 && lf.iter >= lf.first_iter) {
            if (out[lf.tol_kind - 1] < lf.tol) lf.stop[plane] = lf.iter;
        }
    }
}

__global__ void k_loss_finalize(LossFin lf) {
    __shared__ double red[16 * 4];
    loss_finalize(lf, blockIdx.x, red, threadIdx.x, blockDim.x);
}

// ------------------------------------------------------------------------------ column kernel
#define COL_RES2 0  // (S0,S1) <- ( (S0 + conj(e) S1) g ,  e (S0 + conj(e) S1) g )
#define COL_MUL 1   // S0 <- scale * m * S0   (conj_mult: conj(m))
#define COL_MULR 2  // S0 <- g * S0            (real or complex g per gcplx)
#define COL_FWD 3   // S0 <- scale * forward transform only (rows left in position order)

template <typename T>
struct ColArgs {
    cx<T>* S0;
    cx<T>* S1;
    const cx<T>* mult;   // complex multiplier [H][Wp], position order
    const T* greal;      // real resolvent multiplier [H][Wp]
    const cx<T>* gcplx;  // complex resolvent multiplier (injected eigen arrays)
    const cx<T>* tw;     // table exp(-2 pi i e / n), n = fd.n
    const cx<T>* twH;    // table for H (outer twiddles of the split transform)
    FftDesc fd;          // transform done by this launch
    FftDesc fa;          // first-stage factor of a split transform (outer != 0)
    int H, Wh, Wp, batch;
    int cshift, ntile;
    int nseg, row_stride, seg_stride;
    int outer;      // split transform, middle stage: bit0 = w_H^(b*ka) on load, bit1 = conj of it on store
    int dir;        // COL_FWD only: 0 forward, 1 inverse (strided short transforms of a split)
    int conj_mult;
    T scale;
    LossFin lf;
};

template <typename T, int MODE>
__global__ void __launch_bounds__(512) k_col(ColArgs<T> a) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int tid = threadIdx.x, nt = blockDim.x;
    const int n = a.fd.n, C = 1 << a.cshift;
    const int arr = lds_round(n << a.cshift);
    constexpr int NARR = (MODE == COL_RES2) ? 2 : 1;
    cx<T>* lds = reinterpret_cast<cx<T>*>(smem);
    cx<T>* tw = lds + NARR * arr;
    double* red = reinterpret_cast<double*>(tw + n);

    int bid = blockIdx.x;
    const int tile = bid % a.ntile;
    bid /= a.ntile;
    const int seg = bid % a.nseg;
    const int plane = bid / a.nseg;

    if (a.lf.enabled && tile == 0 && seg == 0) loss_finalize(a.lf, plane, red, tid, nt);
    if (a.lf.stop) {
        // image frozen by an earlier launch: nothing to transform.  (An image frozen by this very
        // launch may still be transformed once; harmless, its row kernels are skipped afterwards.)
        const int s = a.lf.stop[plane];
        if (s >= 0 && s < a.lf.iter) return;
    }

    load_table(tw, a.tw, n, tid, nt);
    const int c0 = tile << a.cshift;
    const size_t pbase = (size_t)plane * a.H;
    const int ka = a.outer ? freq_of_pos(a.fa, seg) : 0;
    const int total = n << a.cshift;
    for (int i = tid; i < total; i += nt) {
        const int c = i & (C - 1), e = i >> a.cshift;
        const size_t g = (pbase + (size_t)seg * a.seg_stride + (size_t)e * a.row_stride) * a.Wp + c0 + c;
        cx<T> v0 = a.S0[g];
        cx<T> v1;
        if (NARR == 2) v1 = a.S1[g];
        if (a.outer & 1) {
            const cx<T> w = a.twH[e * ka];
            v0 = v0 * w;
            if (NARR == 2) v1 = v1 * w;
        }
        lds[SW(i)] = v0;
        if (NARR == 2) lds[arr + SW(i)] = v1;
    }
    __syncthreads();
    if (MODE == COL_FWD && a.dir == 1) {
        fft_inverse<T>(lds, NARR, arr, tw, a.fd, a.cshift, tid, nt);
    } else {
        fft_forward<T>(lds, NARR, arr, tw, a.fd, a.cshift, tid, nt);
    }
    if (MODE != COL_FWD) {
        for (int i = tid; i < total; i += nt) {
            const int c = i & (C - 1), e = i >> a.cshift;
            const size_t mrow = (size_t)seg * a.seg_stride + (size_t)e * a.row_stride;
            const size_t gi = mrow * a.Wp + c0 + c;
            const int li = SW(i);
            if (MODE == COL_RES2) {
                const cx<T> ek = a.mult[gi];
                cx<T> U = lds[li] + mulc(lds[arr + li], ek);
                if (a.gcplx) U = U * a.gcplx[gi]; else U = U * a.greal[gi];
                lds[li] = U;
                lds[arr + li] = ek * U;
            } else if (MODE == COL_MUL) {
                const cx<T> m = a.mult[gi];
                const cx<T> x = lds[li];
                lds[li] = (a.conj_mult ? mulc(x, m) : x * m) * a.scale;
            } else {
                if (a.gcplx) lds[li] = lds[li] * a.gcplx[gi]; else lds[li] = lds[li] * a.greal[gi];
            }
        }
        __syncthreads();
        fft_inverse<T>(lds, NARR, arr, tw, a.fd, a.cshift, tid, nt);
    }
    for (int i = tid; i < total; i += nt) {
        const int c = i & (C - 1), e = i >> a.cshift;
        const size_t g = (pbase + (size_t)seg * a.seg_stride + (size_t)e * a.row_stride) * a.Wp + c0 + c;
        cx<T> v0 = lds[SW(i)];
        cx<T> v1;
        if (NARR == 2) v1 = lds[arr + SW(i)];
        if (MODE == COL_FWD) v0 = v0 * a.scale;
        if (a.outer & 2) {
            const cx<T> w = a.twH[e * ka];
            v0 = mulc(v0, w);
            if (NARR == 2) v1 = mulc(v1, w);
        }
        a.S0[g] = v0;
        if (NARR == 2) a.S1[g] = v1;
    }
}

// ----------------------------------------------------------------- arguments of row kernels
template <typename T>
struct RowCommon {
    int H, W, Wp, batch;
    int V_groups;        // W / V
    FftDesc fw;
    const cx<T>* twW;    // global table for W
    cx<T>* S0;
    cx<T>* S1;
    const cx<T>* eDrow;  // row-spectral D table (W/2+1 entries) or null = 2-tap stencils
    const int* stop;     // per image, may be null
    double* partials;    // [batch*H][4]
    int do_consume, do_produce;
};

// LDS carve-up shared by all row kernels: data | twiddles | edge values | reduction scratch
template <typename T>
struct RowLds {
    cx<T>* data;
    cx<T>* tw;
    T* edge;
    double* red;
    __device__ __forceinline__ RowLds(unsigned char* smem, int W) {
        data = reinterpret_cast<cx<T>*>(smem);
        tw = data + lds_round(W);
        red = reinterpret_cast<double*>(tw + W);
        edge = reinterpret_cast<T*>(red + 16 * 4);
    }
};
template <typename T>
__host__ __device__ inline size_t row_lds_bytes(int W, int V) {
    return (size_t)(lds_round(W) + W) * sizeof(cx<T>) + 16 * 4 * sizeof(double) + (size_t)(W / V + 8) * sizeof(T);
}

// Run a second inverse transform of (U * eD) to obtain D applied to the first packed row
// (row-spectral D).  On return data[j].x = (D a)[j]; the previous contents are lost.
template <typename T>
__device__ __forceinline__ void row_spectral_D(RowLds<T>& L, const RowCommon<T>& c, const cx<T>* srcA, int tid, int nt) {
    row_pack_mul(L.data, c.fw, c.W, srcA, c.eDrow, tid, nt);
    __syncthreads();
    fft_inverse<T>(L.data, 1, 0, L.tw, c.fw, 0, tid, nt);
}

// ------------------------------------------------------------------------ primal DR row kernel
// primal_dr/algorithm.py:94-129.  consume: finish iteration with u = F^-1 U, v0 = F^-1 (eK U):
//   z1 += rho (u - x), z2 += rho ([v0, Du, Du] - y) with x = clip(z1), y = prox(z2) recomputed;
// produce: x, y of the new z, a = 2y - z2, transform inputs p = 2x - z1 + D^T(a1 + a2), q = a0.
template <typename T>
struct PdrArgs {
    RowCommon<T> c;
    T* z1; T* z20; T* z21; T* z22;
    const T* b;
    T t, rho, tg;  // tg = t*gamma
    T l2div;       // 2t + 1
    int objective;
};

template <typename T>
__device__ __forceinline__ void pdr_xy(const PdrArgs<T>& a, T z1, T z20, T z21, T z22, T b, T& x, T& y0, T& y1, T& y2) {
    x = clip01(z1);
    if (a.objective == 0) y0 = b + soft(z20 - b, a.t); else y0 = b + (z20 - b) / a.l2div;
    iso(z21, z22, a.tg, y1, y2);
}

template <typename T, int V, int NG, int MAXT>
__global__ void __launch_bounds__(MAXT) k_pdr_row(PdrArgs<T> a) {
    extern __shared__ __align__(16) unsigned char smem[];
    const RowCommon<T>& c = a.c;
    const int tid = threadIdx.x, nt = blockDim.x, W = c.W;
    RowLds<T> L(smem, W);
    load_table(L.tw, c.twW, W, tid, nt);
    const bool spec = c.eDrow != nullptr;
    const int nrows = c.batch * c.H;
    for (int row = blockIdx.x; row < nrows; row += gridDim.x) {
        const int plane = row / c.H;
        if (c.stop && c.stop[plane] >= 0) continue;
        const size_t ro = (size_t)row * W, so = (size_t)row * c.Wp;
        T u[NG][V], k0[NG][V], du[NG][V];
        __syncthreads();
        if (c.do_consume) {
            if (spec) {
                row_spectral_D(L, c, c.S0 + so, tid, nt);
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    const int j = (tid + g * nt) * V;
                    if (j < W) {
#pragma unroll
                        for (int v = 0; v < V; ++v) du[g][v] = L.data[SW(j + v)].x;
                    }
                }
                __syncthreads();
            }
            row_pack(L.data, c.fw, W, c.S0 + so, c.S1 + so, tid, nt);
            __syncthreads();
            fft_inverse<T>(L.data, 1, 0, L.tw, c.fw, 0, tid, nt);
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                const int j = (tid + g * nt) * V;
                if (j < W) {
                    T prev = L.data[SW(j == 0 ? W - 1 : j - 1)].x;
#pragma unroll
                    for (int v = 0; v < V; ++v) {
                        const cx<T> e = L.data[SW(j + v)];
                        u[g][v] = e.x;
                        k0[g][v] = e.y;
                        if (!spec) du[g][v] = e.x - prev;
                        prev = e.x;
                    }
                }
            }
        }
        T pb[NG][V], q0[NG][V], sv[NG][V];
        double acc[4] = {0, 0, 0, 0};
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const int j = (tid + g * nt) * V;
            if (j < W) {
                T z1[V], z20[V], z21[V], z22[V], b[V];
                vload<T, V>(a.z1 + ro + j, z1);
                vload<T, V>(a.z20 + ro + j, z20);
                vload<T, V>(a.z21 + ro + j, z21);
                vload<T, V>(a.z22 + ro + j, z22);
                vload<T, V>(a.b + ro + j, b);
                if (c.do_consume) {
#pragma unroll
                    for (int v = 0; v < V; ++v) {
                        T x, y0, y1, y2;
                        pdr_xy(a, z1[v], z20[v], z21[v], z22[v], b[v], x, y0, y1, y2);
                        const T n1 = z1[v] + a.rho * (u[g][v] - x);
                        const T d1 = n1 - z1[v], d2 = n1 - b[v];
                        acc[0] += (double)t_abs(d1); acc[1] += (double)d1 * (double)d1;
                        acc[2] += (double)t_abs(d2); acc[3] += (double)d2 * (double)d2;
                        z1[v] = n1;
                        z20[v] = z20[v] + a.rho * (k0[g][v] - y0);
                        z21[v] = z21[v] + a.rho * (du[g][v] - y1);
                        z22[v] = z22[v] + a.rho * (du[g][v] - y2);
                    }
                    vstore<T, V>(a.z1 + ro + j, z1);
                    vstore<T, V>(a.z20 + ro + j, z20);
                    vstore<T, V>(a.z21 + ro + j, z21);
                    vstore<T, V>(a.z22 + ro + j, z22);
                }
                if (c.do_produce) {
#pragma unroll
                    for (int v = 0; v < V; ++v) {
                        T x, y0, y1, y2;
                        pdr_xy(a, z1[v], z20[v], z21[v], z22[v], b[v], x, y0, y1, y2);
                        pb[g][v] = (T)2 * x - z1[v];
                        q0[g][v] = (T)2 * y0 - z20[v];
                        sv[g][v] = ((T)2 * y1 - z21[v]) + ((T)2 * y2 - z22[v]);
                    }
                    L.edge[tid + g * nt] = sv[g][0];
                }
            }
        }
        if (c.do_consume) {
            block_reduce<4>(acc, L.red, tid, nt);
            if (tid == 0) {
#pragma unroll
                for (int s = 0; s < 4; ++s) c.partials[(size_t)row * 4 + s] = acc[s];
            }
        }
        __syncthreads();
        if (c.do_produce) {
            if (spec) {
                // first transform: s alone, D^T applied as conj(eD) in the row spectrum
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    const int j = (tid + g * nt) * V;
                    if (j < W) {
#pragma unroll
                        for (int v = 0; v < V; ++v) L.data[SW(j + v)] = mk<T>(sv[g][v], 0);
                    }
                }
                __syncthreads();
                fft_forward<T>(L.data, 1, 0, L.tw, c.fw, 0, tid, nt);
                row_unpack<T>(L.data, c.fw, W, c.S0 + so, nullptr, c.eDrow, false, tid, nt);
                __syncthreads();
            }
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                const int j = (tid + g * nt) * V;
                if (j < W) {
                    const int grp = tid + g * nt;
                    const T snext = L.edge[grp + 1 == c.V_groups ? 0 : grp + 1];
#pragma unroll
                    for (int v = 0; v < V; ++v) {
                        const T nx = (v + 1 < V) ? sv[g][(v + 1) % V] : snext;
                        const T re = spec ? pb[g][v] : pb[g][v] + (sv[g][v] - nx);
                        L.data[SW(j + v)] = mk<T>(re, q0[g][v]);
                    }
                }
            }
            __syncthreads();
            fft_forward<T>(L.data, 1, 0, L.tw, c.fw, 0, tid, nt);
            row_unpack<T>(L.data, c.fw, W, c.S0 + so, c.S1 + so, nullptr, spec, tid, nt);
        }
    }
}

// ---------------------------------------------------------------- primal-dual DR row kernel
// primal_dual_dr/algorithm.py:65-101.
template <typename T>
struct PddrArgs {
    RowCommon<T> c;
    T* p; T* q0; T* q1; T* q2;
    const T* b;
    T t, it, rho, itg;  // it = 1/t, itg = gamma/t
};

template <typename T>
__device__ __forceinline__ void pddr_xz(const PddrArgs<T>& a, T p, T q0, T q1, T q2, T b, T& x, T& z0, T& z1, T& z2) {
    x = clip01(p);
    z0 = q0 - a.t * b - a.t * soft(a.it * q0 - b, a.it);
    T i1, i2;
    iso(a.it * q1, a.it * q2, a.itg, i1, i2);
    z1 = q1 - a.t * i1;
    z2 = q2 - a.t * i2;
}

template <typename T, int V, int NG, int MAXT>
__global__ void __launch_bounds__(MAXT) k_pddr_row(PddrArgs<T> a) {
    extern __shared__ __align__(16) unsigned char smem[];
    const RowCommon<T>& c = a.c;
    const int tid = threadIdx.x, nt = blockDim.x, W = c.W;
    RowLds<T> L(smem, W);
    load_table(L.tw, c.twW, W, tid, nt);
    const bool spec = c.eDrow != nullptr;
    const int nrows = c.batch * c.H;
    for (int row = blockIdx.x; row < nrows; row += gridDim.x) {
        const int plane = row / c.H;
        if (c.stop && c.stop[plane] >= 0) continue;
        const size_t ro = (size_t)row * W, so = (size_t)row * c.Wp;
        T w[NG][V], kw[NG][V], dw[NG][V];
        __syncthreads();
        if (c.do_consume) {
            if (spec) {
                row_spectral_D(L, c, c.S0 + so, tid, nt);
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    const int j = (tid + g * nt) * V;
                    if (j < W) {
#pragma unroll
                        for (int v = 0; v < V; ++v) dw[g][v] = L.data[SW(j + v)].x;
                    }
                }
                __syncthreads();
            }
            row_pack(L.data, c.fw, W, c.S0 + so, c.S1 + so, tid, nt);
            __syncthreads();
            fft_inverse<T>(L.data, 1, 0, L.tw, c.fw, 0, tid, nt);
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                const int j = (tid + g * nt) * V;
                if (j < W) {
                    T prev = L.data[SW(j == 0 ? W - 1 : j - 1)].x;
#pragma unroll
                    for (int v = 0; v < V; ++v) {
                        const cx<T> e = L.data[SW(j + v)];
                        w[g][v] = e.x;
                        kw[g][v] = e.y;
                        if (!spec) dw[g][v] = e.x - prev;
                        prev = e.x;
                    }
                }
            }
        }
        T pb[NG][V], qin[NG][V], sv[NG][V];
        double acc[4] = {0, 0, 0, 0};
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const int j = (tid + g * nt) * V;
            if (j < W) {
                T p[V], q0[V], q1[V], q2[V], b[V];
                vload<T, V>(a.p + ro + j, p);
                vload<T, V>(a.q0 + ro + j, q0);
                vload<T, V>(a.q1 + ro + j, q1);
                vload<T, V>(a.q2 + ro + j, q2);
                vload<T, V>(a.b + ro + j, b);
                if (c.do_consume) {
#pragma unroll
                    for (int v = 0; v < V; ++v) {
                        T x, z0, z1, z2;
                        pddr_xz(a, p[v], q0[v], q1[v], q2[v], b[v], x, z0, z1, z2);
                        const T o0 = (T)2 * z0 - q0[v], o1 = (T)2 * z1 - q1[v], o2 = (T)2 * z2 - q2[v];
                        p[v] = p[v] + a.rho * (w[g][v] - x);
                        q0[v] = q0[v] + a.rho * ((a.t * kw[g][v] + o0) - z0);
                        q1[v] = q1[v] + a.rho * ((a.t * dw[g][v] + o1) - z1);
                        q2[v] = q2[v] + a.rho * ((a.t * dw[g][v] + o2) - z2);
                        const T d = clip01(p[v]) - b[v];
                        acc[0] += (double)t_abs(d); acc[1] += (double)d * (double)d;
                    }
                    vstore<T, V>(a.p + ro + j, p);
                    vstore<T, V>(a.q0 + ro + j, q0);
                    vstore<T, V>(a.q1 + ro + j, q1);
                    vstore<T, V>(a.q2 + ro + j, q2);
                }
                if (c.do_produce) {
#pragma unroll
                    for (int v = 0; v < V; ++v) {
                        T x, z0, z1, z2;
                        pddr_xz(a, p[v], q0[v], q1[v], q2[v], b[v], x, z0, z1, z2);
                        pb[g][v] = (T)2 * x - p[v];
                        qin[g][v] = -a.t * ((T)2 * z0 - q0[v]);
                        sv[g][v] = -a.t * (((T)2 * z1 - q1[v]) + ((T)2 * z2 - q2[v]));
                    }
                    L.edge[tid + g * nt] = sv[g][0];
                }
            }
        }
        if (c.do_consume) {
            block_reduce<4>(acc, L.red, tid, nt);
            if (tid == 0) {
#pragma unroll
                for (int s = 0; s < 4; ++s) c.partials[(size_t)row * 4 + s] = acc[s];
            }
        }
        __syncthreads();
        if (c.do_produce) {
            if (spec) {
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    const int j = (tid + g * nt) * V;
                    if (j < W) {
#pragma unroll
                        for (int v = 0; v < V; ++v) L.data[SW(j + v)] = mk<T>(sv[g][v], 0);
                    }
                }
                __syncthreads();
                fft_forward<T>(L.data, 1, 0, L.tw, c.fw, 0, tid, nt);
                row_unpack<T>(L.data, c.fw, W, c.S0 + so, nullptr, c.eDrow, false, tid, nt);
                __syncthreads();
            }
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                const int j = (tid + g * nt) * V;
                if (j < W) {
                    const int grp = tid + g * nt;
                    const T snext = L.edge[grp + 1 == c.V_groups ? 0 : grp + 1];
#pragma unroll
                    for (int v = 0; v < V; ++v) {
                        const T nx = (v + 1 < V) ? sv[g][(v + 1) % V] : snext;
                        const T re = spec ? pb[g][v] : pb[g][v] + (sv[g][v] - nx);
                        L.data[SW(j + v)] = mk<T>(re, qin[g][v]);
                    }
                }
            }
            __syncthreads();
            fft_forward<T>(L.data, 1, 0, L.tw, c.fw, 0, tid, nt);
            row_unpack<T>(L.data, c.fw, W, c.S0 + so, c.S1 + so, nullptr, spec, tid, nt);
        }
    }
}

// ------------------------------------------------------------------------------ ADMM row kernel
// admm/algorithm.py:69-115.  consume: x = F^-1 X, kx = F^-1 (eK X), dx = D x, then the u, y, w, z
// updates; produce: transform inputs (u - w/t) + D^T((y1+y2) - (z1+z2)/t) and y0 - z0/t.
template <typename T>
struct AdmmArgs {
    RowCommon<T> c;
    T* u; T* w; T* y0; T* y1; T* y2; T* z0; T* z1; T* z2;
    const T* b;
    T* sol;  // when non-null the consume step stores x (the look-ahead solution of admm:118-132)
    T t, it, rho, omr, itg, l2div;  // it = 1/t, omr = 1 - rho, itg = gamma/t, l2div = 2/t + 1
    int objective;
    int sol_only;  // consume computes x and the loss only (final solve, admm:160-173)
};

template <typename T, int V, int NG, int MAXT>
__global__ void __launch_bounds__(MAXT) k_admm_row(AdmmArgs<T> a) {
    extern __shared__ __align__(16) unsigned char smem[];
    const RowCommon<T>& c = a.c;
    const int tid = threadIdx.x, nt = blockDim.x, W = c.W;
    RowLds<T> L(smem, W);
    load_table(L.tw, c.twW, W, tid, nt);
    const bool spec = c.eDrow != nullptr;
    const int nrows = c.batch * c.H;
    for (int row = blockIdx.x; row < nrows; row += gridDim.x) {
        const int plane = row / c.H;
        if (c.stop && c.stop[plane] >= 0) continue;
        const size_t ro = (size_t)row * W, so = (size_t)row * c.Wp;
        T x[NG][V], kx[NG][V], dx[NG][V];
        __syncthreads();
        if (c.do_consume) {
            if (spec && !a.sol_only) {
                row_spectral_D(L, c, c.S0 + so, tid, nt);
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    const int j = (tid + g * nt) * V;
                    if (j < W) {
#pragma unroll
                        for (int v = 0; v < V; ++v) dx[g][v] = L.data[SW(j + v)].x;
                    }
                }
                __syncthreads();
            }
            row_pack(L.data, c.fw, W, c.S0 + so, c.S1 + so, tid, nt);
            __syncthreads();
            fft_inverse<T>(L.data, 1, 0, L.tw, c.fw, 0, tid, nt);
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                const int j = (tid + g * nt) * V;
                if (j < W) {
                    T prev = L.data[SW(j == 0 ? W - 1 : j - 1)].x;
#pragma unroll
                    for (int v = 0; v < V; ++v) {
                        const cx<T> e = L.data[SW(j + v)];
                        x[g][v] = e.x;
                        kx[g][v] = e.y;
                        if (!spec) dx[g][v] = e.x - prev;
                        prev = e.x;
                    }
                }
            }
        }
        T pb[NG][V], qin[NG][V], sv[NG][V];
        double acc[4] = {0, 0, 0, 0};
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const int j = (tid + g * nt) * V;
            if (j < W) {
                T b[V];
                vload<T, V>(a.b + ro + j, b);
                if (c.do_consume) {
#pragma unroll
                    for (int v = 0; v < V; ++v) {
                        const T d = x[g][v] - b[v];
                        acc[0] += (double)t_abs(d); acc[1] += (double)d * (double)d;
                    }
                    if (a.sol) vstore<T, V>(a.sol + ro + j, x[g]);
                }
                if (a.sol_only) continue;
                T u[V], w[V], y0[V], y1[V], y2[V], z0[V], z1[V], z2[V];
                vload<T, V>(a.u + ro + j, u);
                vload<T, V>(a.w + ro + j, w);
                vload<T, V>(a.y0 + ro + j, y0);
                vload<T, V>(a.y1 + ro + j, y1);
                vload<T, V>(a.y2 + ro + j, y2);
                vload<T, V>(a.z0 + ro + j, z0);
                vload<T, V>(a.z1 + ro + j, z1);
                vload<T, V>(a.z2 + ro + j, z2);
                if (c.do_consume) {
#pragma unroll
                    for (int v = 0; v < V; ++v) {
                        const T un = clip01(a.rho * x[g][v] + a.omr * u[v] + a.it * w[v]);
                        const T pp0 = a.rho * kx[g][v] + a.omr * y0[v] + a.it * z0[v];
                        const T pp1 = a.rho * dx[g][v] + a.omr * y1[v] + a.it * z1[v];
                        const T pp2 = a.rho * dx[g][v] + a.omr * y2[v] + a.it * z2[v];
                        T y0n, y1n, y2n;
                        if (a.objective == 0) y0n = b[v] + soft(pp0 - b[v], a.it); else y0n = b[v] + (pp0 - b[v]) / a.l2div;
                        iso(pp1, pp2, a.itg, y1n, y2n);
                        w[v] = w[v] + a.t * (x[g][v] - un);
                        z0[v] = z0[v] + a.t * (kx[g][v] - y0n);
                        z1[v] = z1[v] + a.t * (dx[g][v] - y1n);
                        z2[v] = z2[v] + a.t * (dx[g][v] - y2n);
                        u[v] = un; y0[v] = y0n; y1[v] = y1n; y2[v] = y2n;
                    }
                    vstore<T, V>(a.u + ro + j, u);
                    vstore<T, V>(a.w + ro + j, w);
                    vstore<T, V>(a.y0 + ro + j, y0);
                    vstore<T, V>(a.y1 + ro + j, y1);
                    vstore<T, V>(a.y2 + ro + j, y2);
                    vstore<T, V>(a.z0 + ro + j, z0);
                    vstore<T, V>(a.z1 + ro + j, z1);
                    vstore<T, V>(a.z2 + ro + j, z2);
                }
                if (c.do_produce) {
#pragma unroll
                    for (int v = 0; v < V; ++v) {
                        pb[g][v] = u[v] - a.it * w[v];
                        qin[g][v] = y0[v] - a.it * z0[v];
                        sv[g][v] = (y1[v] + y2[v]) - a.it * (z1[v] + z2[v]);
                    }
                    L.edge[tid + g * nt] = sv[g][0];
                }
            }
        }
        if (c.do_consume) {
            block_reduce<4>(acc, L.red, tid, nt);
            if (tid == 0) {
#pragma unroll
                for (int s = 0; s < 4; ++s) c.partials[(size_t)row * 4 + s] = acc[s];
            }
        }
        __syncthreads();
        if (c.do_produce && !a.sol_only) {
            if (spec) {
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    const int j = (tid + g * nt) * V;
                    if (j < W) {
#pragma unroll
                        for (int v = 0; v < V; ++v) L.data[SW(j + v)] = mk<T>(sv[g][v], 0);
                    }
                }
                __syncthreads();
                fft_forward<T>(L.data, 1, 0, L.tw, c.fw, 0, tid, nt);
                row_unpack<T>(L.data, c.fw, W, c.S0 + so, nullptr, c.eDrow, false, tid, nt);
                __syncthreads();
            }
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                const int j = (tid + g * nt) * V;
                if (j < W) {
                    const int grp = tid + g * nt;
                    const T snext = L.edge[grp + 1 == c.V_groups ? 0 : grp + 1];
#pragma unroll
                    for (int v = 0; v < V; ++v) {
                        const T nx = (v + 1 < V) ? sv[g][(v + 1) % V] : snext;
                        const T re = spec ? pb[g][v] : pb[g][v] + (sv[g][v] - nx);
                        L.data[SW(j + v)] = mk<T>(re, qin[g][v]);
                    }
                }
            }
            __syncthreads();
            fft_forward<T>(L.data, 1, 0, L.tw, c.fw, 0, tid, nt);
            row_unpack<T>(L.data, c.fw, W, c.S0 + so, c.S1 + so, nullptr, spec, tid, nt);
        }
    }
}

// ------------------------------------------------------------------ Chambolle-Pock row kernels
// chambolle_pock/algorithm.py:58-92.  A workgroup owns two image rows (the two real slots of the
// complex transform).  STEP 0 ("dual"): consume Kz, update y (cp:60-76), produce the transform input
// y0.  STEP 1 ("primal"): consume K^T y0, update x and z (cp:79-87), produce the transform input z.
// With row-spectral D the dual step also needs D z and the primal step D^T(y1+y2) in the row
// spectrum; those use one extra transform each.
template <typename T>
struct CpArgs {
    RowCommon<T> c;
    T* x; T* z; T* y0; T* y1; T* y2;
    const T* b;
    T t, s, is, isg, l2div;  // is = 1/s, isg = gamma/s, l2div = 2/s + 1
    int objective;
    int step;
};

template <typename T, int V, int NG, int MAXT>
__global__ void __launch_bounds__(MAXT) k_cp_row(CpArgs<T> a) {
    extern __shared__ __align__(16) unsigned char smem[];
    const RowCommon<T>& c = a.c;
    const int tid = threadIdx.x, nt = blockDim.x, W = c.W, H = c.H;
    RowLds<T> L(smem, W);
    load_table(L.tw, c.twW, W, tid, nt);
    const bool spec = c.eDrow != nullptr;
    const int ppp = (H + 1) >> 1;  // row pairs per image
    const int npairs = c.batch * ppp;
    for (int pr = blockIdx.x; pr < npairs; pr += gridDim.x) {
        const int plane = pr / ppp;
        if (c.stop && c.stop[plane] >= 0) continue;
        const int r0 = (pr - plane * ppp) * 2;
        const int nslot = (r0 + 1 < H) ? 2 : 1;
        const size_t row0 = (size_t)plane * H + r0;
        T in[2][NG][V];  // consumed transform output per slot
        T aux[2][NG][V]; // row-spectral D z (dual step)
        __syncthreads();
        if (c.do_consume) {
            if (spec && a.step == 0) {
                // D z for both rows: transform z rows, multiply by eD, inverse
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    const int j = (tid + g * nt) * V;
                    if (j < W) {
                        T za[V], zb[V];
                        vload<T, V>(a.z + row0 * W + j, za);
                        if (nslot == 2) vload<T, V>(a.z + (row0 + 1) * W + j, zb);
#pragma unroll
                        for (int v = 0; v < V; ++v) L.data[SW(j + v)] = mk<T>(za[v], nslot == 2 ? zb[v] : (T)0);
                    }
                }
                __syncthreads();
                fft_forward<T>(L.data, 1, 0, L.tw, c.fw, 0, tid, nt);
                // multiply the two separated spectra by eD and repack in place: since eD is applied to
                // both slots alike, Z' [k] = eD[k] Z[k] for k <= W/2 and conj-mirrored table above.
                {
                    const int half = W >> 1;
                    for (int k = tid; k <= half; k += nt) {
                        const int km = (k == 0) ? 0 : W - k;
                        const int p1 = SW(pos_of_freq(c.fw, k)), p2 = SW(pos_of_freq(c.fw, km));
                        const cx<T> e = c.eDrow[k];
                        const cx<T> Z1 = L.data[p1], Z2 = L.data[p2];
                        L.data[p1] = Z1 * e;
                        if (km != k) L.data[p2] = Z2 * cconj(e);
                    }
                }
                __syncthreads();
                fft_inverse<T>(L.data, 1, 0, L.tw, c.fw, 0, tid, nt);
                const T sc = (T)1 / (T)W;
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    const int j = (tid + g * nt) * V;
                    if (j < W) {
#pragma unroll
                        for (int v = 0; v < V; ++v) {
                            const cx<T> e = L.data[SW(j + v)];
                            aux[0][g][v] = e.x * sc;
                            aux[1][g][v] = e.y * sc;
                        }
                    }
                }
                __syncthreads();
            }
            const bool addS1 = spec && a.step == 1;
            row_pack(L.data, c.fw, W, c.S0 + row0 * c.Wp, nslot == 2 ? c.S0 + (row0 + 1) * c.Wp : nullptr, tid, nt,
                     addS1 ? c.S1 + row0 * c.Wp : nullptr, (addS1 && nslot == 2) ? c.S1 + (row0 + 1) * c.Wp : nullptr);
            __syncthreads();
            fft_inverse<T>(L.data, 1, 0, L.tw, c.fw, 0, tid, nt);
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                const int j = (tid + g * nt) * V;
                if (j < W) {
#pragma unroll
                    for (int v = 0; v < V; ++v) {
                        const cx<T> e = L.data[SW(j + v)];
                        in[0][g][v] = e.x;
                        in[1][g][v] = e.y;
                    }
                }
            }
        }
        T out[2][NG][V];
        double acc[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
#pragma unroll
        for (int sl = 0; sl < 2; ++sl) {
            if (sl < nslot) {
                const size_t ro = (row0 + sl) * W;
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    const int j = (tid + g * nt) * V;
                    if (j < W) {
                        if (a.step == 0) {
                            // dual step
                            T z[V], y0[V], y1[V], y2[V], b[V];
                            vload<T, V>(a.z + ro + j, z);
                            vload<T, V>(a.y0 + ro + j, y0);
                            vload<T, V>(a.y1 + ro + j, y1);
                            vload<T, V>(a.y2 + ro + j, y2);
                            vload<T, V>(a.b + ro + j, b);
                            T prev = a.z[ro + (j == 0 ? W - 1 : j - 1)];
#pragma unroll
                            for (int v = 0; v < V; ++v) {
                                const T dz = spec ? aux[sl][g][v] : z[v] - prev;
                                prev = z[v];
                                const T g0 = y0[v] + a.s * in[sl][g][v];
                                const T g1 = y1[v] + a.s * dz;
                                const T g2 = y2[v] + a.s * dz;
                                T pr0;
                                if (a.objective == 0) pr0 = soft(g0 / a.s - b[v], a.is); else pr0 = (g0 / a.s - b[v]) / a.l2div;
                                y0[v] = g0 - a.s * b[v] - a.s * pr0;
                                T i1, i2;
                                iso(g1 / a.s, g2 / a.s, a.isg, i1, i2);
                                y1[v] = g1 - a.s * i1;
                                y2[v] = g2 - a.s * i2;
                                out[sl][g][v] = y0[v];
                            }
                            vstore<T, V>(a.y0 + ro + j, y0);
                            vstore<T, V>(a.y1 + ro + j, y1);
                            vstore<T, V>(a.y2 + ro + j, y2);
                        } else {
                            // primal step
                            T z[V];
                            if (c.do_consume) {
                                T x[V], y1[V], y2[V], b[V];
                                vload<T, V>(a.x + ro + j, x);
                                vload<T, V>(a.b + ro + j, b);
                                T dty[V];
                                if (spec) {
#pragma unroll
                                    for (int v = 0; v < V; ++v) dty[v] = 0;  // included spectrally in `in`
                                } else {
                                    vload<T, V>(a.y1 + ro + j, y1);
                                    vload<T, V>(a.y2 + ro + j, y2);
                                    const int jn = (j + V == W) ? 0 : j + V;
                                    const T snext = a.y1[ro + jn] + a.y2[ro + jn];
#pragma unroll
                                    for (int v = 0; v < V; ++v) {
                                        const T sc = y1[v] + y2[v];
                                        const T sn = (v + 1 < V) ? (y1[(v + 1) % V] + y2[(v + 1) % V]) : snext;
                                        dty[v] = sc - sn;
                                    }
                                }
#pragma unroll
                                for (int v = 0; v < V; ++v) {
                                    const T aty = in[sl][g][v] + dty[v];
                                    const T xn = clip01(x[v] - a.t * aty);
                                    z[v] = (T)2 * xn - x[v];
                                    x[v] = xn;
                                    const T d = xn - b[v];
                                    acc[sl][0] += (double)t_abs(d); acc[sl][1] += (double)d * (double)d;
                                }
                                vstore<T, V>(a.x + ro + j, x);
                                vstore<T, V>(a.z + ro + j, z);
                            } else {
                                vload<T, V>(a.z + ro + j, z);
                            }
#pragma unroll
                            for (int v = 0; v < V; ++v) out[sl][g][v] = z[v];
                        }
                    }
                }
            }
        }
        if (a.step == 1 && c.do_consume) {
            double accs[4];
            for (int sl = 0; sl < 2; ++sl) {
#pragma unroll
                for (int s = 0; s < 4; ++s) accs[s] = acc[sl][s];
                block_reduce<4>(accs, L.red, tid, nt);
                if (tid == 0 && sl < nslot) {
#pragma unroll
                    for (int s = 0; s < 4; ++s) c.partials[(row0 + sl) * 4 + s] = accs[s];
                }
            }
        }
        __syncthreads();
        if (c.do_produce) {
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                const int j = (tid + g * nt) * V;
                if (j < W) {
#pragma unroll
                    for (int v = 0; v < V; ++v) L.data[SW(j + v)] = mk<T>(out[0][g][v], nslot == 2 ? out[1][g][v] : (T)0);
                }
            }
            __syncthreads();
            fft_forward<T>(L.data, 1, 0, L.tw, c.fw, 0, tid, nt);
            row_unpack<T>(L.data, c.fw, W, c.S0 + row0 * c.Wp, nslot == 2 ? c.S0 + (row0 + 1) * c.Wp : nullptr, nullptr, false, tid, nt);
            if (spec && a.step == 0) {
                // second output of the dual step: row spectrum of (y1 + y2) times conj(eD), into S1;
                // the column kernel leaves S1 alone and the primal step adds its inverse row transform.
                __syncthreads();
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    const int j = (tid + g * nt) * V;
                    if (j < W) {
                        T ya[V], yb[V], yc[V], yd[V];
                        vload<T, V>(a.y1 + row0 * W + j, ya);
                        vload<T, V>(a.y2 + row0 * W + j, yb);
                        if (nslot == 2) {
                            vload<T, V>(a.y1 + (row0 + 1) * W + j, yc);
                            vload<T, V>(a.y2 + (row0 + 1) * W + j, yd);
                        }
#pragma unroll
                        for (int v = 0; v < V; ++v) L.data[SW(j + v)] = mk<T>(ya[v] + yb[v], nslot == 2 ? yc[v] + yd[v] : (T)0);
                    }
                }
                __syncthreads();
                fft_forward<T>(L.data, 1, 0, L.tw, c.fw, 0, tid, nt);
                {
                    const int half = W >> 1;
                    for (int k = tid; k <= half; k += nt) {
                        const int km = (k == 0) ? 0 : W - k;
                        const cx<T> Z1 = L.data[SW(pos_of_freq(c.fw, k))], Z2 = L.data[SW(pos_of_freq(c.fw, km))];
                        const cx<T> e = c.eDrow[k];
                        const T hw = (T)0.5 / (T)W;  // 1/W: this spectrum skips the normalised column stage
                        const cx<T> A = mulc(mk<T>(hw * (Z1.x + Z2.x), hw * (Z1.y - Z2.y)), e);
                        const cx<T> B = mulc(mk<T>(hw * (Z1.y + Z2.y), hw * (Z2.x - Z1.x)), e);
                        c.S1[row0 * c.Wp + k] = A;
                        if (nslot == 2) c.S1[(row0 + 1) * c.Wp + k] = B;
                    }
                }
            }
        }
    }
}

// ------------------------------------------------------------ plain row transforms (row pairs)
template <typename T>
struct PlainRowArgs {
    int H, W, Wp, batch;
    FftDesc fw;
    const cx<T>* twW;
    cx<T>* S;       // half-row-spectra [batch][H][Wp]
    const T* src;   // real -> spectrum
    T* dst;         // spectrum -> real
    T scale;        // applied to the real output
    const T* add;   // optional: dst = scale * result + add
};

template <typename T>
__global__ void __launch_bounds__(1024) k_rows_r2c(PlainRowArgs<T> a) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int tid = threadIdx.x, nt = blockDim.x, W = a.W, H = a.H;
    RowLds<T> L(smem, W);
    load_table(L.tw, a.twW, W, tid, nt);
    const int ppp = (H + 1) >> 1, npairs = a.batch * ppp;
    for (int pr = blockIdx.x; pr < npairs; pr += gridDim.x) {
        const int plane = pr / ppp, r0 = (pr - plane * ppp) * 2;
        const bool two = r0 + 1 < H;
        const size_t row0 = (size_t)plane * H + r0;
        __syncthreads();
        for (int j = tid; j < W; j += nt) L.data[SW(j)] = mk<T>(a.src[row0 * W + j], two ? a.src[(row0 + 1) * W + j] : (T)0);
        __syncthreads();
        fft_forward<T>(L.data, 1, 0, L.tw, a.fw, 0, tid, nt);
        row_unpack<T>(L.data, a.fw, W, a.S + row0 * a.Wp, two ? a.S + (row0 + 1) * a.Wp : nullptr, nullptr, false, tid, nt);
    }
}

template <typename T>
__global__ void __launch_bounds__(1024) k_rows_c2r(PlainRowArgs<T> a) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int tid = threadIdx.x, nt = blockDim.x, W = a.W, H = a.H;
    RowLds<T> L(smem, W);
    load_table(L.tw, a.twW, W, tid, nt);
    const int ppp = (H + 1) >> 1, npairs = a.batch * ppp;
    for (int pr = blockIdx.x; pr < npairs; pr += gridDim.x) {
        const int plane = pr / ppp, r0 = (pr - plane * ppp) * 2;
        const bool two = r0 + 1 < H;
        const size_t row0 = (size_t)plane * H + r0;
        __syncthreads();
        row_pack(L.data, a.fw, W, a.S + row0 * a.Wp, two ? a.S + (row0 + 1) * a.Wp : nullptr, tid, nt);
        __syncthreads();
        fft_inverse<T>(L.data, 1, 0, L.tw, a.fw, 0, tid, nt);
        for (int j = tid; j < W; j += nt) {
            const cx<T> e = L.data[SW(j)];
            T v0 = e.x * a.scale, v1 = e.y * a.scale;
            if (a.add) {
                v0 += a.add[row0 * W + j];
                if (two) v1 += a.add[(row0 + 1) * W + j];
            }
            a.dst[row0 * W + j] = v0;
            if (two) a.dst[(row0 + 1) * W + j] = v1;
        }
    }
}

// ------------------------------------------------------------------- elementwise kernels
// Horizontal circular differences (mm:55-59): (Dx)[j] = x[j] - x[j-1], (D^T y)[j] = y[j] - y[j+1].
template <typename T>
__global__ void k_stencil(const T* x, T* y, int W, size_t n, int transpose) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const size_t row = i / W;
        const int j = (int)(i - row * W);
        const int jn = transpose ? (j + 1 == W ? 0 : j + 1) : (j == 0 ? W - 1 : j - 1);
        y[i] = x[i] - x[row * W + jn];
    }
}

template <typename T>
__global__ void k_prox(int kind, const T* x, T* y, size_t n, T lam, const double* norm2) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        if (kind == 0) y[i] = soft(x[i], lam);
        else if (kind == 1) y[i] = x[i] / ((T)2 * lam + (T)1);
        else if (kind == 2) {
            T oa, oc;
            iso(x[i], x[i + n], lam, oa, oc);
            y[i] = oa;
            y[i + n] = oc;
        } else if (kind == 3) y[i] = clip01(x[i]);
        else {
            // prox_l2 (prox:10-14): whole-tensor norm
            const T nrm = (T)sqrt(norm2[0]);
            y[i] = nrm > lam ? ((T)1 - lam / nrm) * x[i] : (T)0;
        }
    }
}

// sums of |x-y| and (x-y)^2 per image; one block per (image, chunk); deterministic two-level sum
template <typename T>
__global__ void k_reduce_partial(const T* x, const T* y, size_t n_per_image, int chunks, double* partial) {
    __shared__ double red[16 * 2];
    const int image = blockIdx.x / chunks, chunk = blockIdx.x % chunks;
    const size_t per = (n_per_image + chunks - 1) / chunks;
    const size_t lo = (size_t)chunk * per, hi = lo + per < n_per_image ? lo + per : n_per_image;
    double v[2] = {0, 0};
    for (size_t i = lo + threadIdx.x; i < hi; i += blockDim.x) {
        const double d = (double)x[image * n_per_image + i] - (y ? (double)y[image * n_per_image + i] : 0.0);
        v[0] += d < 0 ? -d : d;
        v[1] += d * d;
    }
    block_reduce<2>(v, red, threadIdx.x, blockDim.x);
    if (threadIdx.x == 0) {
        partial[blockIdx.x * 2] = v[0];
        partial[blockIdx.x * 2 + 1] = v[1];
    }
}
__global__ void k_reduce_final(const double* partial, int chunks, double npix, double* out4) {
    const int image = blockIdx.x;
    if (threadIdx.x == 0) {
        double s1 = 0, s2 = 0;
        for (int c = 0; c < chunks; ++c) {
            s1 += partial[(image * chunks + c) * 2];
            s2 += partial[(image * chunks + c) * 2 + 1];
        }
        const double mse = s2 / npix;
        out4[image * 4 + 0] = s1;
        out4[image * 4 + 1] = sqrt(s2);
        out4[image * 4 + 2] = mse;
        out4[image * 4 + 3] = mse == 0.0 ? (double)INFINITY : 10.0 * log10(1.0 / mse);
    }
}

// ------------------------------------------------------------------------- set-up kernels
// Position of natural column frequency k1 under the (possibly split) column transform.
struct ColMap {
    FftDesc fa, fb;
    int A, B;
};
__device__ __forceinline__ int colpos_of_freq(const ColMap& m, int k1) {
    if (m.A == 1) return pos_of_freq(m.fb, k1);
    return pos_of_freq(m.fa, k1 % m.A) * m.B + pos_of_freq(m.fb, k1 / m.A);
}
__device__ __forceinline__ int colfreq_of_pos(const ColMap& m, int r) {
    if (m.A == 1) return freq_of_pos(m.fb, r);
    return freq_of_pos(m.fa, r / m.B) + m.A * freq_of_pos(m.fb, r % m.B);
}

// Import a full natural-order H x W complex multiplier (as the reference stores eigen arrays):
// Hermitian-symmetrise in double, optionally take the reciprocal first, scale, keep k2 <= W/2 and
// store rows in column-transform position order.  TI = float or double component type of the input.
template <typename T, typename TI>
__global__ void k_import_mult(const cx<TI>* in, cx<T>* out, T* out_real, int H, int W, int Wh, int Wp, ColMap cm, int recip, int conj_in, double scale) {
    const size_t total = (size_t)H * Wh;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int r = (int)(i / Wh), k2 = (int)(i - (size_t)r * Wh);
        const int k1 = colfreq_of_pos(cm, r);
        const int m1 = k1 == 0 ? 0 : H - k1, m2 = k2 == 0 ? 0 : W - k2;
        const cx<TI> a0 = in[(size_t)k1 * W + k2], b0 = in[(size_t)m1 * W + m2];
        double ax = a0.x, ay = a0.y, bx = b0.x, by = b0.y;
        if (recip) {
            const double da = ax * ax + ay * ay, db = bx * bx + by * by;
            ax = ax / da; ay = -ay / da; bx = bx / db; by = -by / db;
        }
        if (conj_in) { ay = -ay; by = -by; }
        const double rx = 0.5 * (ax + bx) * scale, ry = 0.5 * (ay - by) * scale;
        if (out) out[(size_t)r * Wp + k2] = mk<T>((T)rx, (T)ry);
        if (out_real) out_real[(size_t)r * Wp + k2] = (T)rx;
    }
}

// G[r][k2] = scale / (1 + cK |eK|^2 + cD ed2[k2])  (mm:67-72 in real arithmetic)
template <typename T>
__global__ void k_make_G(const cx<T>* eK, const T* ed2, T* G, int H, int Wh, int Wp, T cK, T cD, T scale) {
    const size_t total = (size_t)H * Wh;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int r = (int)(i / Wh), k2 = (int)(i - (size_t)r * Wh);
        const cx<T> e = eK[(size_t)r * Wp + k2];
        G[(size_t)r * Wp + k2] = scale / ((T)1 + cK * (e.x * e.x + e.y * e.y) + cD * ed2[k2]);
    }
}
// mult[r][k2] = 1 + |eK|^2 + 2 ed2  (apply_Mat, mm:106-108)
template <typename T>
__global__ void k_make_mat(const cx<T>* eK, const T* ed2, cx<T>* out, int H, int Wh, int Wp) {
    const size_t total = (size_t)H * Wh;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int r = (int)(i / Wh), k2 = (int)(i - (size_t)r * Wh);
        const cx<T> e = eK[(size_t)r * Wp + k2];
        out[(size_t)r * Wp + k2] = mk<T>((T)1 + (e.x * e.x + e.y * e.y) + (T)2 * ed2[k2], (T)0);
    }
}
// broadcast a row table to every row: out[r][k2] = tab[k2]
template <typename T>
__global__ void k_bcast_row(const cx<T>* tab, cx<T>* out, int H, int Wh, int Wp) {
    const size_t total = (size_t)H * Wh;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int r = (int)(i / Wh), k2 = (int)(i - (size_t)r * Wh);
        out[(size_t)r * Wp + k2] = tab[k2];
    }
}

// Export a half-plane, position-ordered multiplier to the reference's full natural layout.
template <typename T>
__global__ void k_export_full(const cx<T>* half, cx<T>* out, int H, int W, int Wp, ColMap cm) {
    const size_t total = (size_t)H * W;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int k1 = (int)(i / W), k2 = (int)(i - (size_t)k1 * W);
        if (k2 <= W / 2) {
            out[i] = half[(size_t)colpos_of_freq(cm, k1) * Wp + k2];
        } else {
            const int m1 = k1 == 0 ? 0 : H - k1;
            out[i] = cconj(half[(size_t)colpos_of_freq(cm, m1) * Wp + (W - k2)]);
        }
    }
}

__global__ void k_iters_done(const int* stop, int* out, int n, int n_iters) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) out[i] = stop[i] >= 0 ? stop[i] : n_iters;
}

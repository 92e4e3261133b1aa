// Rows longer than one workgroup's LDS holds (W > 8192: the 16384- and 32768-pixel rows of very large images).
//
// The length-W row transform is split four-step style, W = R x M with M <= 4096 and R in {2, 4, 8, 16}:
// pixel n = M j + m, frequency k = R kappa + q,
//     Z[R kappa + q] = sum_m w_M^(m kappa) [ w_W^(m q) sum_j z[M j + m] w_R^(j q) ].
// Three kernels replace the fused row kernel of the shorter rows (the reference transforms any row length in
// one fft2 call: apply_periodic_conv_2d.py:28, multiplying_matrix.py:110-112):
//   k_long_inv   one workgroup per (row, array, q <= R/2): gathers Z[R kappa + q] of one real array from its
//                half-spectrum (Hermitian extension for k > W/2), length-M inverse transform in LDS, -> Y[array][row][q][m];
//   k_long_pix / k_long_cp / k_long_plain   one thread per (row, m): the R-point transforms over q / j in registers,
//                the twiddle w_W^(m q), and between them the solver's per-pixel step on the R pixels M j + m;
//   k_long_fwd   one workgroup per (row, q <= R/2): length-M forward transforms of the produced arrays -> half-spectra
//                (0 < q < R/2 also writes the terms of residue R - q, the conjugates of its upper half).
// Every array is real, so only q = 0 .. R/2 exist on the frequency side (the rest are conjugates).
// The two real arrays of an iteration are transformed separately here (no packing into one complex transform), and
// D / D^T are applied as row-spectral multipliers eD[k] = 1 - w_W^k (a third array each way) instead of as stencils,
// so the pixel kernel needs no neighbour from another thread.  This path trades bandwidth for simplicity: it exists
// so that images wider than 8192 pixels run at all (row slabs of one 32768 x 32768 image), not to be fast.
#pragma once
#include "ddn_kernels.hpp"
#include "ddn_long_args.hpp"

namespace dlong {

template <typename T>
__device__ __forceinline__ cx<T> unit(long e, int n) {   // exp(-2 pi i e / n), 0 <= e < n
    T s, c;
    if constexpr (sizeof(T) == 4) sincospif(-2.0f * (float)e / (float)n, &s, &c);
    else sincospi(-2.0 * (double)e / (double)n, &s, &c);
    return mk<T>(c, s);
}
template <typename T>
__device__ __forceinline__ size_t y_index(const LongArgs<T>& a, int arr, size_t row, int q, int m) {
    return ((((size_t)arr * ((size_t)a.c.batch * a.c.H) + row) * (a.R / 2 + 1)) + q) * (size_t)a.M + m;
}

// Work split of the two frequency-side kernels.  The `per_row` workgroups of one row (one per residue q, and per
// array) touch interleaved elements of the same cache lines of the row's half-spectra: element R kappa + q sits
// between those of the other residues.  Workgroup b runs on XCD b % 8 and only that XCD's L2 can serve a line to
// several of them (or merge their partial-line writes), so the per_row workgroups of a row take consecutive slots
// of ONE XCD and run at the same time.  The grid is 8 * per_row * S workgroups (long_grid below).
struct RowSlot {
    size_t row, step;
    int sub;
};
__device__ __forceinline__ RowSlot row_slot(int per_row) {
    const int x = blockIdx.x & 7, j = blockIdx.x >> 3, S = (int)(gridDim.x >> 3) / per_row;
    RowSlot r;
    r.sub = j % per_row;
    r.row = (size_t)(j / per_row) * 8 + x;
    r.step = (size_t)8 * S;
    return r;
}
inline int long_grid(size_t nrows, int per_row, int num_cu) {
    const size_t want = std::max<size_t>(1, (size_t)num_cu / per_row);            // slots per XCD: about 8 workgroups per CU in all
    const size_t S = std::max<size_t>(1, std::min<size_t>(want, (nrows + 7) / 8));
    return (int)(8 * per_row * S);
}

// ---- frequency side, inverse: half-spectra -> Y -----------------------------------------------------------
// array 0: S0 (u); array 1: S1 (K u); array 2: eD * S0 (D u).  narr = 1: S0 only.
template <typename T>
__global__ void __launch_bounds__(256) k_long_inv(LongArgs<T> a) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int tid = threadIdx.x, nt = blockDim.x, W = a.c.W, M = a.M, R = a.R, nq = R / 2 + 1;
    RowLds<T> L(smem, M);
    const Tw<T> tw = tw_setup(L.tw, a.twM, M, tid, nt, tw_direct_rows<T>());
    const size_t nrows = (size_t)a.c.batch * a.c.H;
    const RowSlot rs = row_slot(a.narr * nq);
    const int q = rs.sub % nq, arr = rs.sub / nq;
    for (size_t row = rs.row; row < nrows; row += rs.step) {
        const int plane = (int)(row / a.c.H);
        if (a.c.stop && a.c.stop[plane] >= 0) continue;
        const SpecRow sr = spec_row(a.c.lay, plane, (int)(row - (size_t)plane * a.c.H));
        const cx<T>* const src = arr == 1 ? a.c.S1 : a.c.S0;
        __syncthreads();
        for (int kap = tid; kap < M; kap += nt) {
            const int k = R * kap + q;
            const int kk = k <= W / 2 ? k : W - k;
            cx<T> z = src[sr.at(kk)];
            if (a.inv_add_s1) z = z + a.c.S1[sr.at(kk)];
            if (arr == 2) z = z * a.ed[kk];
            if (kk == 0 || 2 * kk == W) z.y = (T)0;          // mean and Nyquist term of a real row are real
            if (k > W / 2) z = cconj(z);
            L.data[SW(pos_of_freq(a.fm, kap))] = z;
        }
        __syncthreads();
        fft_inverse<T>(L.data, 1, 0, tw, a.fm, 0, tid, nt);
        cx<T>* const y = a.Y + y_index(a, arr, row, q, 0);
        for (int m = tid; m < M; m += nt) y[m] = L.data[SW(m)];
    }
}

// ---- frequency side, forward: Y -> half-spectra -------------------------------------------------------------
// narr = 3: S0 <- F(re) + conj(eD) F(s) (arrays 0 and 2), S1 <- F(im) (array 1).  narr = 1: S0 <- F(array 0).
template <typename T>
__global__ void __launch_bounds__(256) k_long_fwd(LongArgs<T> a) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int tid = threadIdx.x, nt = blockDim.x, W = a.c.W, M = a.M, R = a.R, nq = R / 2 + 1;
    RowLds<T> L(smem, M);
    const Tw<T> tw = tw_setup(L.tw, a.twM, M, tid, nt, tw_direct_rows<T>());
    const size_t nrows = (size_t)a.c.batch * a.c.H;
    const RowSlot rs = row_slot(nq);
    const int q = rs.sub;
    for (size_t row = rs.row; row < nrows; row += rs.step) {
        const int plane = (int)(row / a.c.H);
        if (a.c.stop && a.c.stop[plane] >= 0) continue;
        const SpecRow sr = spec_row(a.c.lay, plane, (int)(row - (size_t)plane * a.c.H));
        // passes: (array, destination, mode) with mode 0 store, 1 store times conj(eD), 2 add
        // further modes (Chambolle-Pock with a row-spectral D, LongArgs::fwd_mode): 3 store and also S1 = z eD / W,
        // 4 store times conj(eD) / W
        const int npass = a.narr == 3 ? 3 : (a.fwd_mode == 2 ? 2 : 1);
        const T invW = (T)1 / (T)W;
        for (int ps = 0; ps < npass; ++ps) {
            const int arr = a.narr == 3 ? (ps == 0 ? 2 : (ps == 1 ? 0 : 1)) : ps;
            const int mode = a.narr == 3 ? (ps == 0 ? 1 : (ps == 1 ? 2 : 0)) : (a.fwd_mode == 1 ? 3 : (a.fwd_mode == 2 && ps == 1 ? 4 : 0));
            cx<T>* const dst = ((a.narr == 3 && ps == 2) || mode == 4) ? a.c.S1 : a.c.S0;
            const cx<T>* const y = a.Y + y_index(a, arr, row, q, 0);
            __syncthreads();
            for (int m = tid; m < M; m += nt) L.data[SW(m)] = y[m];
            __syncthreads();
            fft_forward<T>(L.data, 1, 0, tw, a.fm, 0, tid, nt);
            for (int kap = tid; kap < M; kap += nt) {
                // residues 0 and R/2 hold both halves of their own terms: keep k <= W/2.  A residue in between
                // also produces, as conjugates, the terms of residue R - q, which no workgroup transforms.
                int k = R * kap + q;
                const bool upper = k > W / 2;
                if (upper && (q == 0 || 2 * q == R)) continue;
                cx<T> z = L.data[SW(pos_of_freq(a.fm, kap))];
                if (upper) { k = W - k; z = cconj(z); }
                const size_t o = sr.at(k);
                if (mode == 1) z = mulc(z, a.ed[k]);
                if (mode == 2) z = z + dst[o];
                if (mode == 3) a.c.S1[o] = (z * a.ed[k]) * invW;
                if (mode == 4) z = mulc(z, a.ed[k]) * invW;
                dst[o] = z;
            }
        }
    }
}

// ---- pixel side helpers ------------------------------------------------------------------------------------
// R real values at one m <-> the R/2 + 1 complex terms y[q] (twiddle w_W^(m q) included).
template <typename T, int R>
__device__ __forceinline__ void to_pixels(const LongArgs<T>& a, int arr, size_t row, int m, const cx<T> (&twd)[R / 2 + 1], T (&out)[R]) {
    cx<T> v[R];
#pragma unroll
    for (int q = 0; q <= R / 2; ++q) v[q] = mulc(a.Y[y_index(a, arr, row, q, m)], twd[q]);   // conj twiddle on the way back
#pragma unroll
    for (int q = R / 2 + 1; q < R; ++q) v[q] = cconj(v[R - q]);
    v[0].y = (T)0;
    v[R / 2].y = (T)0;
    Dft<T, R, true>::run(v);
#pragma unroll
    for (int j = 0; j < R; ++j) out[j] = v[j].x;
}
template <typename T, int R>
__device__ __forceinline__ void from_pixels(const LongArgs<T>& a, int arr, size_t row, int m, const cx<T> (&twd)[R / 2 + 1], const T (&in)[R]) {
    cx<T> v[R];
#pragma unroll
    for (int j = 0; j < R; ++j) v[j] = mk<T>(in[j], (T)0);
    Dft<T, R, false>::run(v);
#pragma unroll
    for (int q = 0; q <= R / 2; ++q) a.Y[y_index(a, arr, row, q, m)] = v[q] * twd[q];
}
template <typename T, int R>
__device__ __forceinline__ void twiddles(const LongArgs<T>& a, int m, cx<T> (&twd)[R / 2 + 1]) {
#pragma unroll
    for (int q = 0; q <= R / 2; ++q) twd[q] = unit<T>((long)m * q, a.c.W);
}

// V consecutive complex values as one aligned access (two 16-byte accesses for four float pairs)
template <typename T, int V>
struct alignas((sizeof(cx<T>) * V) > 16 ? 16 : (sizeof(cx<T>) * V)) CxVec {
    cx<T> e[V];
};

// ---- the resolvent solvers (primal DR, primal-dual DR, ADMM): policy step on the R pixels of every m ------------
// A thread owns V consecutive m: its accesses of Y are V complex values wide and the policy's accesses of the
// iterate arrays V pixels wide (16 bytes for V = 4 floats).  V > 1 needs 16-byte aligned arrays and M % V == 0.
template <typename T, class P, int R, int V>
__global__ void __launch_bounds__(256) k_long_pix(LongArgs<T> a, typename P::Args pa) {
    __shared__ double red[16 * 4];
    constexpr int NQ = R / 2 + 1;
    const int tid = threadIdx.x, nt = blockDim.x, M = a.M, W = a.c.W;
    const size_t nrows = (size_t)a.c.batch * a.c.H;
    for (size_t row = blockIdx.x; row < nrows; row += gridDim.x) {
        const int plane = (int)(row / a.c.H);
        if (a.c.stop && a.c.stop[plane] >= 0) continue;
        double acc[4] = {0, 0, 0, 0};
        for (int m = tid * V; m < M; m += nt * V) {
            cx<T> twd[V][NQ];
#pragma unroll
            for (int v = 0; v < V; ++v) twiddles<T, R>(a, m + v, twd[v]);
            T px[3][R][V];     // (u, K u, D u) in, then (re, im, s) out: [array][j][v]
            if (a.c.do_consume) {
#pragma unroll
                for (int arr = 0; arr < 3; ++arr) {
                    CxVec<T, V> y[NQ];
#pragma unroll
                    for (int q = 0; q < NQ; ++q) y[q] = *reinterpret_cast<const CxVec<T, V>*>(a.Y + y_index(a, arr, row, q, m));
#pragma unroll
                    for (int v = 0; v < V; ++v) {
                        cx<T> z[R];
#pragma unroll
                        for (int q = 0; q < NQ; ++q) z[q] = mulc(y[q].e[v], twd[v][q]);   // conj twiddle on the way back
#pragma unroll
                        for (int q = NQ; q < R; ++q) z[q] = cconj(z[R - q]);
                        z[0].y = (T)0;
                        z[R / 2].y = (T)0;
                        Dft<T, R, true>::run(z);
#pragma unroll
                        for (int j = 0; j < R; ++j) px[arr][j][v] = z[j].x;
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < R; ++j) {
                const size_t o = row * (size_t)W + (size_t)M * j + m;
                T r1[V], i1[V], s1[V];
                P::template step<V>(pa, a.c.do_consume != 0, a.c.do_produce != 0, o, px[0][j], px[1][j], px[2][j], r1, i1, s1, acc);
#pragma unroll
                for (int v = 0; v < V; ++v) { px[0][j][v] = r1[v]; px[1][j][v] = i1[v]; px[2][j][v] = s1[v]; }
            }
            if (a.c.do_produce) {
#pragma unroll
                for (int arr = 0; arr < 3; ++arr) {
                    CxVec<T, V> y[NQ];
#pragma unroll
                    for (int v = 0; v < V; ++v) {
                        cx<T> z[R];
#pragma unroll
                        for (int j = 0; j < R; ++j) z[j] = mk<T>(px[arr][j][v], (T)0);
                        Dft<T, R, false>::run(z);
#pragma unroll
                        for (int q = 0; q < NQ; ++q) y[q].e[v] = z[q] * twd[v][q];
                    }
#pragma unroll
                    for (int q = 0; q < NQ; ++q) *reinterpret_cast<CxVec<T, V>*>(a.Y + y_index(a, arr, row, q, m)) = y[q];
                }
            }
        }
        if (a.c.do_consume) {
            block_reduce<4>(acc, red, tid, nt);
            if (tid == 0) {
#pragma unroll
                for (int q = 0; q < 4; ++q) a.c.partials[row * 4 + q] = acc[q];
            }
        }
    }
}

// ---- Chambolle-Pock (cp:58-92): one array each way; D z and D^T(y1 + y2) are differences of arrays the step only reads
template <typename T, int R>
__global__ void __launch_bounds__(256) k_long_cp(LongArgs<T> a, CpArgs<T> p) {
    __shared__ double red[16 * 4];
    const int tid = threadIdx.x, nt = blockDim.x, M = a.M, W = a.c.W;
    const size_t nrows = (size_t)a.c.batch * a.c.H;
    for (size_t row = blockIdx.x; row < nrows; row += gridDim.x) {
        double acc[4] = {0, 0, 0, 0};
        const size_t ro = row * (size_t)W;
        for (int m = tid; m < M; m += nt) {
            cx<T> twd[R / 2 + 1];
            twiddles<T, R>(a, m, twd);
            T in[R], out[R], in1[R], out1[R];     // in1 / out1: D z in, y1 + y2 out (row-spectral D only)
#pragma unroll
            for (int j = 0; j < R; ++j) in[j] = in1[j] = out1[j] = (T)0;
            if (a.c.do_consume) to_pixels<T, R>(a, 0, row, m, twd, in);
            if (a.cp_spec && a.c.do_consume && p.step == 0) to_pixels<T, R>(a, 1, row, m, twd, in1);
#pragma unroll
            for (int j = 0; j < R; ++j) {
                const int n = M * j + m;
                const size_t o = ro + n;
                if (p.step == 0) {
                    T dz;                                                                     // cp:61
                    if (a.cp_spec) dz = in1[j];
                    else dz = p.z[o] - p.z[ro + (n == 0 ? W - 1 : n - 1)];
                    const T b = p.b[o];
                    const T g0 = p.y0[o] + p.s * in[j];                                       // cp:66
                    const T g1 = p.y1[o] + p.s * dz;
                    const T g2 = p.y2[o] + p.s * dz;
                    T pr0;
                    if (p.objective == 0) pr0 = soft(g0 / p.s - b, p.is); else pr0 = (g0 / p.s - b) / p.l2div;   // cp:69-72
                    const T y0n = g0 - p.s * b - p.s * pr0;
                    T i1, i2;
                    iso(g1 / p.s, g2 / p.s, p.isg, i1, i2);                                   // cp:75
                    p.y0[o] = y0n;
                    const T y1n = g1 - p.s * i1, y2n = g2 - p.s * i2;
                    p.y1[o] = y1n;
                    p.y2[o] = y2n;
                    out[j] = y0n;
                    out1[j] = y1n + y2n;
                } else if (a.c.do_consume) {
                    T dty = (T)0;                                                             // D^T(y1 + y2), cp:79 (row-spectral D: already inside `in`)
                    if (!a.cp_spec) {
                        const size_t on = ro + (n + 1 == W ? 0 : n + 1);
                        dty = (p.y1[o] + p.y2[o]) - (p.y1[on] + p.y2[on]);
                    }
                    const T x = p.x[o];
                    const T xn = clip01(x - p.t * (in[j] + dty));                             // cp:83-86
                    const T zn = (T)2 * xn - x;                                               // cp:87
                    const T d = xn - p.b[o];                                                  // cp:96
                    acc[0] += (double)t_abs(d); acc[1] += (double)d * (double)d;
                    p.x[o] = xn;
                    p.z[o] = zn;
                    out[j] = zn;
                } else {
                    out[j] = p.z[o];
                }
            }
            if (a.c.do_produce) from_pixels<T, R>(a, 0, row, m, twd, out);
            if (a.cp_spec && a.c.do_produce && p.step == 0) from_pixels<T, R>(a, 1, row, m, twd, out1);
        }
        if (p.step == 1 && a.c.do_consume) {
            block_reduce<4>(acc, red, tid, nt);
            if (tid == 0) {
#pragma unroll
                for (int q = 0; q < 4; ++q) a.c.partials[row * 4 + q] = acc[q];
            }
        }
    }
}

// ---- plain transforms: src -> Y (forward) or Y -> dst * scale (inverse)
template <typename T, int R>
__global__ void __launch_bounds__(256) k_long_plain(LongArgs<T> a, const T* src, T* dst, T scale) {
    const int tid = threadIdx.x, nt = blockDim.x, M = a.M, W = a.c.W;
    const size_t nrows = (size_t)a.c.batch * a.c.H;
    for (size_t row = blockIdx.x; row < nrows; row += gridDim.x) {
        for (int m = tid; m < M; m += nt) {
            cx<T> twd[R / 2 + 1];
            twiddles<T, R>(a, m, twd);
            T v[R];
            if (src) {
#pragma unroll
                for (int j = 0; j < R; ++j) v[j] = src[row * (size_t)W + (size_t)M * j + m];
                from_pixels<T, R>(a, 0, row, m, twd, v);
            } else {
                to_pixels<T, R>(a, 0, row, m, twd, v);
#pragma unroll
                for (int j = 0; j < R; ++j) dst[row * (size_t)W + (size_t)M * j + m] = v[j] * scale;
            }
        }
    }
}

}  // namespace dlong

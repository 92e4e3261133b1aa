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
// v / d with d uniform: the fast units multiply by the hardware reciprocal (hoisted out of the pixel loops)
template <typename T>
__device__ __forceinline__ T udiv(T v, T d) {
#ifdef DDN_WAVE_FAST
    if constexpr (sizeof(T) == 4) return v * __builtin_amdgcn_rcpf(d);
#endif
    return v / d;
}
// data-term prox, l1 (objective 0) or squared l2: both sides are cheap, no branch
template <typename T>
__device__ __forceinline__ T data_prox(int objective, T v, T lam, T div);
__device__ __forceinline__ float t_sqrt(float v) { return sqrtf(v); }
__device__ __forceinline__ double t_sqrt(double v) { return sqrt(v); }
// prox_iso (core/proximal_operators.py:22-50): per-pixel 2-vector shrink, zero where norm <= lam
// Translation units that define DDN_WAVE_FAST (the wave-per-row float kernels) use the hardware
// reciprocal square root (1 ulp) and no branch; every other unit the correctly rounded sqrt and divide.
template <typename T>
__device__ __forceinline__ void iso(T a, T c, T lam, T& oa, T& oc) {
#ifdef DDN_WAVE_FAST
    if constexpr (sizeof(T) == 4) {
        const float n2 = a * a + c * c;
        const float f = n2 > lam * lam ? 1.f - lam * __builtin_amdgcn_rsqf(n2) : 0.f;
        oa = f * a;
        oc = f * c;
        return;
    }
#endif
    const T n = t_sqrt(a * a + c * c);
    const T f = n > lam ? (T)1 - lam / n : (T)0;
    oa = f * a;
    oc = f * c;
}

template <typename T>
__device__ __forceinline__ T data_prox(int objective, T v, T lam, T div) {
#ifdef DDN_WAVE_FAST
    const T a = soft(v, lam), q = udiv(v, div);
    return objective == 0 ? a : q;
#else
    if (objective == 0) return soft(v, lam);
    return v / div;
#endif
}

// DDN_NT_ITER (wave-per-row units): the iterate arrays are touched once per iteration, so their scalar
// accesses carry the non-temporal hint and leave the caches to the spectra and the multipliers.
template <typename T, int V>
__device__ __forceinline__ void vload(const T* p, T (&r)[V]) {
    if constexpr (V == 1) {
#ifdef DDN_NT_ITER
        r[0] = __builtin_nontemporal_load(p);
#else
        r[0] = p[0];
#endif
    } else if constexpr (sizeof(T) == 4) {
        const float4 t = *reinterpret_cast<const float4*>(p);
        r[0] = t.x; r[1] = t.y; r[2] = t.z; r[3] = t.w;
    } else {
        const double2 a = *reinterpret_cast<const double2*>(p);
        const double2 b = *(reinterpret_cast<const double2*>(p) + 1);
        r[0] = a.x; r[1] = a.y; r[2] = b.x; r[3] = b.y;
    }
}
// b, the blurred image: read-only and small enough to stay in the last-level cache beside the half-spectra, so its
// loads keep the default policy even where the iterate arrays are streamed
template <typename T, int V>
__device__ __forceinline__ void vload_keep(const T* p, T (&r)[V]) {
    if constexpr (V == 1) {
        r[0] = p[0];
    } else {
        vload<T, V>(p, r);
    }
}
template <typename T, int V>
__device__ __forceinline__ void vstore(T* p, const T (&r)[V]) {
    if constexpr (V == 1) {
#ifdef DDN_NT_ITER
        __builtin_nontemporal_store(r[0], p);
#else
        p[0] = r[0];
#endif
    } else if constexpr (sizeof(T) == 4) {
        *reinterpret_cast<float4*>(p) = make_float4(r[0], r[1], r[2], r[3]);
    } else {
        *reinterpret_cast<double2*>(p) = make_double2(r[0], r[1]);
        *(reinterpret_cast<double2*>(p) + 1) = make_double2(r[2], r[3]);
    }
}

// Sum NS values over the workgroup; result valid in thread 0.  red: LDS scratch of >= 16*NS doubles.
// One step of a wavefront sum on the data-parallel-primitive path (no LDS crossbar): v + the value the control word
// routes to this lane, 0 where there is none.  CTRL: row_shr:n = 0x110 + n, row_bcast:15 = 0x142, row_bcast:31 = 0x143.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_add(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROW_MASK, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROW_MASK, 0xf, true);
    return v + __hiloint2double(hi, lo);
}
// Sum over the wavefront, valid in lane 63: an inclusive scan within each row of 16 lanes, then the row totals carried
// into the later rows.  (Six dependent steps like a shuffle tree, at a fraction of its latency: the shuffles go through
// the LDS crossbar -- the four loss sums of a row took 3.3 k cycles of a 33 k-cycle row.)
__device__ __forceinline__ double wave_sum63(double v) {
    v = dpp_add<0x111, 0xf>(v);
    v = dpp_add<0x112, 0xf>(v);
    v = dpp_add<0x114, 0xf>(v);
    v = dpp_add<0x118, 0xf>(v);
    v = dpp_add<0x142, 0xa>(v);
    v = dpp_add<0x143, 0xc>(v);
    return v;
}

template <int NS>
__device__ __forceinline__ void block_reduce(double (&v)[NS], double* red, int tid, int nt) {
#pragma unroll
    for (int s = 0; s < NS; ++s) v[s] = wave_sum63(v[s]);
    const int wave = tid >> 6, lane = tid & 63, nw = (nt + 63) >> 6;
    if (nw == 1) {
        // one wavefront: lane 63 hands its sums to thread 0
        if (lane == 63) {
#pragma unroll
            for (int s = 0; s < NS; ++s) red[s] = v[s];
        }
        __syncthreads();
        if (tid == 0) {
#pragma unroll
            for (int s = 0; s < NS; ++s) v[s] = red[s];
        }
        __syncthreads();
        return;
    }
    if (lane == 63) {
#pragma unroll
        for (int s = 0; s < NS; ++s) red[wave * NS + s] = v[s];
    }
    __syncthreads();
    if (tid == 0) {
#pragma unroll
        for (int s = 0; s < NS; ++s) v[s] = red[s];
        for (int w = 1; w < nw; ++w) {
#pragma unroll
            for (int s = 0; s < NS; ++s) v[s] += red[w * NS + s];
        }
    }
    __syncthreads();
}

// ------------------------------------------------------------------ spectral array addressing
// Half-row-spectra and multipliers are stored column-tile-major: tile t holds columns
// [t*C, (t+1)*C) of every row contiguously, element (row, k2) of image `plane` at
//   (((plane*ntile + (k2 >> cs)) * H + row) << cs) + (k2 & (C-1)),   C = 1 << cs.
// A column workgroup therefore streams one contiguous H*C block; a row workgroup touches C-element
// pieces, one per tile.
struct SpecLayout {
    int H, cs, ntile;
};
struct SpecRow {
    size_t base;     // offset of (row, k2 = 0)
    size_t tstride;  // H << cs
    int cs;
    __device__ __forceinline__ size_t at(int k2) const { return base + (size_t)(k2 >> cs) * tstride + (k2 & ((1 << cs) - 1)); }
};
__device__ __forceinline__ SpecRow spec_row(const SpecLayout& l, int plane, int row) {
    SpecRow r;
    r.base = ((size_t)plane * l.ntile * l.H + row) << l.cs;
    r.tstride = (size_t)l.H << l.cs;
    r.cs = l.cs;
    return r;
}

// ------------------------------------------------------------------- row transform helpers
// Pack the half-spectra of two real rows a, b (either source may be null = zero; optional addends)
// into the full spectrum of a + i b, at digit-reversed LDS positions.  No barrier inside.
// All global loads of a thread are issued before the first use (KP = elements per thread, a
// compile-time bound): the pack phase is otherwise a chain of dependent HBM round trips.
template <typename T, int KP>
__device__ __forceinline__ void row_pack_load(cx<T> (&U)[KP], cx<T> (&V)[KP], int W, const cx<T>* srcA, const SpecRow& ra, const cx<T>* srcB,
                                              const SpecRow& rb, int tid, int nt, const cx<T>* addA = nullptr, const cx<T>* addB = nullptr) {
    const int half = W >> 1;
#pragma unroll
    for (int i = 0; i < KP; ++i) {
        const int k = tid + i * nt;
        U[i] = mk<T>(0, 0);
        V[i] = mk<T>(0, 0);
        if (k <= half) {
            if (srcA) U[i] = srcA[ra.at(k)];
            if (srcB) V[i] = srcB[rb.at(k)];
            if (addA) U[i] = U[i] + addA[ra.at(k)];
            if (addB) V[i] = V[i] + addB[rb.at(k)];
        }
    }
}
template <typename T, int KP, int NF = 0, int SCH = 16>
__device__ __forceinline__ void row_pack_store(cx<T>* lds, const FftDesc& fw, int W, const cx<T> (&U)[KP], const cx<T> (&V)[KP], int tid, int nt) {
    const int half = W >> 1;
#pragma unroll
    for (int i = 0; i < KP; ++i) {
        const int k = tid + i * nt;
        if (k <= half) {
            const int km = (k == 0) ? 0 : W - k;
            if (km == k) {
                lds[SW(pos_freq<NF, SCH>(fw, k))] = mk<T>(U[i].x, V[i].x);  // DC / Nyquist of a real row: imaginary part dropped
            } else {
                lds[SW(pos_freq<NF, SCH>(fw, k))] = mk<T>(U[i].x - V[i].y, U[i].y + V[i].x);
                lds[SW(pos_freq<NF, SCH>(fw, km))] = mk<T>(U[i].x + V[i].y, V[i].x - U[i].y);
            }
        }
    }
}
// KP * nt must cover W/2 + 1 entries (the callers' thread counts guarantee it).
template <typename T, int KP, int NF = 0, int SCH = 16>
__device__ __forceinline__ void row_pack(cx<T>* lds, const FftDesc& fw, int W, const cx<T>* srcA, const SpecRow& ra, const cx<T>* srcB,
                                         const SpecRow& rb, int tid, int nt, const cx<T>* addA = nullptr, const cx<T>* addB = nullptr) {
    cx<T> U[KP], V[KP];
    row_pack_load<T, KP>(U, V, W, srcA, ra, srcB, rb, tid, nt, addA, addB);
    row_pack_store<T, KP, NF, SCH>(lds, fw, W, U, V, tid, nt);
}
// As row_pack but a single source multiplied by a row table (row-spectral D: U[k] * eD[k]).
template <typename T, int NF = 0, int SCH = 16>
__device__ __forceinline__ void row_pack_mul(cx<T>* lds, const FftDesc& fw, int W, const cx<T>* srcA, const SpecRow& ra, const cx<T>* rowmul, int tid, int nt) {
    const int half = W >> 1;
    for (int k = tid; k <= half; k += nt) {
        cx<T> U = srcA[ra.at(k)] * rowmul[k];
        const int km = (k == 0) ? 0 : W - k;
        if (km == k) {
            lds[SW(pos_freq<NF, SCH>(fw, k))] = mk<T>(U.x, 0);
        } else {
            lds[SW(pos_freq<NF, SCH>(fw, k))] = U;
            lds[SW(pos_freq<NF, SCH>(fw, km))] = cconj(U);
        }
    }
}
// LDS holds the digit-reversed spectrum Z of a + i b.  Separate A = F(a), B = F(b) for k2 = 0..W/2
// and store them (null destination = skip).  mulA != null: A is multiplied by conj(mulA[k]) first
// (row-spectral D^T); accA adds to what dstA already holds.
template <typename T, int NF = 0, int SCH = 16>
__device__ __forceinline__ void row_unpack(const cx<T>* lds, const FftDesc& fw, int W, cx<T>* dstA, const SpecRow& ra, cx<T>* dstB, const SpecRow& rb,
                                           const cx<T>* mulA, bool accA, int tid, int nt) {
    const int half = W >> 1;
    for (int k = tid; k <= half; k += nt) {
        const int km = (k == 0) ? 0 : W - k;
        const cx<T> Z1 = lds[SW(pos_freq<NF, SCH>(fw, k))];
        const cx<T> Z2 = lds[SW(pos_freq<NF, SCH>(fw, km))];
        if (dstA) {
            cx<T> A = mk<T>((T)0.5 * (Z1.x + Z2.x), (T)0.5 * (Z1.y - Z2.y));
            if (mulA) A = mulc(A, mulA[k]);
            const size_t o = ra.at(k);
            if (accA) A = A + dstA[o];
            dstA[o] = A;
        }
        if (dstB) dstB[rb.at(k)] = mk<T>((T)0.5 * (Z1.y + Z2.y), (T)0.5 * (Z2.x - Z1.x));
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

static __global__ void k_loss_finalize(LossFin lf) {
    __shared__ double red[16 * 4];
    loss_finalize(lf, blockIdx.x, red, threadIdx.x, blockDim.x);
}

// ------------------------------------------------------------------------------ column kernel
#define COL_RES2 0  // (S0,S1) <- ( (S0 + conj(e) S1) g ,  e (S0 + conj(e) S1) g )
#define COL_MUL 1   // S0 <- scale * m * S0   (conj_mult: conj(m))
#define COL_MULR 2  // S0 <- g * S0            (real or complex g per gcplx)
#define COL_FWD 3   // S0 <- scale * transform only (dir 0 forward / 1 inverse), rows left in position order
#define COL_FWD2 4  // as COL_FWD on both S0 and S1

template <typename T>
struct ColArgs {
    cx<T>* S0;
    cx<T>* S1;
    const cx<T>* mult;   // complex multiplier, same tiled layout, rows in position order
    const T* greal;      // real resolvent multiplier
    const cx<T>* gcplx;  // complex resolvent multiplier (injected eigen arrays)
    const cx<T>* tw;     // twiddle tables of the transform done by this launch
    const cx<T>* twH;    // direct table for H (outer twiddles of the split transform)
    FftDesc fd;          // transform done by this launch
    FftDesc fa;          // first-stage factor of a split transform (outer != 0)
    SpecLayout lay;
    int batch;
    int nseg, row_stride, seg_stride;
    int outer;      // split transform, middle stage: bit0 = w_H^(b*ka) on load, bit1 = conj of it on store
    int dir;        // COL_FWD only: 0 forward, 1 inverse
    int conj_mult;
    T scale;
    LossFin lf;
    int mult_pitch;      // tile pitch (rows) of the multiplier arrays when it differs from lay.H (0: the same)
    int rb_shift;        // rows stored in blocks of 1 << rb_shift rows (one per source rank of a row-slab decomposition):
    long long rb_extra;  //   row r lives rb_extra * (r >> rb_shift) rows past its place in a plain tile; 0 = plain
};

// NF > 0: the transform length is NF, with the compile-time radix schedule of ddn_fft.hpp (the same digit order as the
// run-time one, so stored multipliers agree); NF = 0: any length, run-time schedule.
// SMALL: the launch has at most 512 threads and at most four elements per thread and array (small images); see below.
template <typename T, int MODE, int NF = 0, bool SMALL = false>
__global__ void __launch_bounds__(SMALL ? 512 : 1024) k_col(ColArgs<T> a) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int tid = threadIdx.x, nt = blockDim.x;
    const int n = a.fd.n, cs = a.lay.cs, C = 1 << cs;
    const int arr = lds_round(n << cs);
    constexpr int NARR = (MODE == COL_RES2 || MODE == COL_FWD2) ? 2 : 1;
    cx<T>* lds = reinterpret_cast<cx<T>*>(smem);
    cx<T>* twa = lds + NARR * arr;
    double* red = reinterpret_cast<double*>(twa + tw_lds_entries(n));

    int bid = blockIdx.x;
    const int tile = bid % a.lay.ntile;
    bid /= a.lay.ntile;
    const int seg = bid % a.nseg;
    const int plane = bid / a.nseg;

    if (a.lf.enabled && tile == 0 && seg == 0) loss_finalize(a.lf, plane, red, tid, nt);
    if (a.lf.stop) {
        // image frozen by an earlier launch: nothing to transform.  (An image frozen by this very
        // launch may still be transformed once; harmless, its row kernels are skipped afterwards.)
        const int s = a.lf.stop[plane];
        if (s >= 0 && s < a.lf.iter) return;
    }

    const size_t tbase = ((size_t)plane * a.lay.ntile + tile) * a.lay.H;  // in rows
    const size_t mbase = (size_t)tile * (a.mult_pitch ? a.mult_pitch : a.lay.H);
    const int ka = a.outer ? freq_of_pos(a.fa, seg) : 0;
    // row of the tile -> row of storage
    auto srow = [&](size_t r) -> size_t { return a.rb_extra ? r + (r >> a.rb_shift) * (size_t)a.rb_extra : r; };
    const int total = n << cs;
    constexpr int UN = SMALL ? 4 : 8;  // loads in flight per thread and array
    constexpr bool POINTWISE = MODE != COL_FWD && MODE != COL_FWD2;
    // SMALL: a tile of at most PRE elements per thread (small images, whose workgroups are given that many threads) is
    // one chunk: its loads are issued before the twiddle table's, and the multipliers of the pointwise stage before the
    // forward transform, so that the three global-memory latencies of the workgroup overlap instead of following one
    // another (the multipliers stay in registers across the transform: hence the lower thread bound).
    constexpr int PRE = 4;
    constexpr bool one_chunk = SMALL;
    cx<T> v0[UN], v1[UN], w[UN];
    auto load_tile = [&](int i0) {
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int i = i0 + u * nt;
            if (i < total) {
                const int c = i & (C - 1), e = i >> cs;
                const size_t g = ((tbase + srow((size_t)seg * a.seg_stride + (size_t)e * a.row_stride)) << cs) + c;
                v0[u] = a.S0[g];
                if (NARR == 2) v1[u] = a.S1[g];
                if (a.outer & 1) w[u] = a.twH[e * ka];
            }
        }
    };
    auto tile_to_lds = [&](int i0) {
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int i = i0 + u * nt;
            if (i < total) {
                if (a.outer & 1) {
                    v0[u] = v0[u] * w[u];
                    if (NARR == 2) v1[u] = v1[u] * w[u];
                }
                lds[SW(i)] = v0[u];
                if (NARR == 2) lds[arr + SW(i)] = v1[u];
            }
        }
    };
    auto load_mult = [&](auto& mm, auto& gc, auto& gr, int i0, auto count) {
#pragma unroll
        for (int u = 0; u < decltype(count)::value; ++u) {
            const int i = i0 + u * nt;
            if (i < total) {
                const int c = i & (C - 1), e = i >> cs;
                const size_t gi = ((mbase + (size_t)seg * a.seg_stride + (size_t)e * a.row_stride) << cs) + c;
                if (MODE != COL_MULR) mm[u] = a.mult[gi];
                if (MODE != COL_MUL) {
                    if (a.gcplx) gc[u] = a.gcplx[gi]; else gr[u] = a.greal[gi];
                }
            }
        }
    };
    auto pointwise = [&](auto& mm, auto& gc, auto& gr, int i0, auto count) {
#pragma unroll
        for (int u = 0; u < decltype(count)::value; ++u) {
            const int i = i0 + u * nt;
            if (i < total) {
                const int li = SW(i);
                if (MODE == COL_RES2) {
                    const cx<T> ek = mm[u];
                    cx<T> U = lds[li] + mulc(lds[arr + li], ek);
                    if (a.gcplx) U = U * gc[u]; else U = U * gr[u];
                    lds[li] = U;
                    lds[arr + li] = ek * U;
                } else if (MODE == COL_MUL) {
                    const cx<T> x = lds[li];
                    lds[li] = (a.conj_mult ? mulc(x, mm[u]) : x * mm[u]) * a.scale;
                } else {
                    if (a.gcplx) lds[li] = lds[li] * gc[u]; else lds[li] = lds[li] * gr[u];
                }
            }
        }
    };
    cx<T> pmm[SMALL ? PRE : 1], pgc[SMALL ? PRE : 1];
    T pgr[SMALL ? PRE : 1];
    if constexpr (one_chunk) load_tile(tid);
    const Tw<T> tw = tw_setup(twa, a.tw, n, tid, nt);
    if constexpr (one_chunk) {
        if (POINTWISE) load_mult(pmm, pgc, pgr, tid, std::integral_constant<int, PRE>());
        tile_to_lds(tid);
    } else {
        for (int i0 = tid; i0 < total; i0 += nt * UN) {
            load_tile(i0);
            tile_to_lds(i0);
        }
    }
    __syncthreads();
    if ((MODE == COL_FWD || MODE == COL_FWD2) && a.dir == 1) {
        fft_inv<T, NF, 16, kFuseMax<T>>(lds, NARR, arr, tw, a.fd, cs, tid, nt);
    } else {
        fft_fwd<T, NF, 16, kFuseMax<T>>(lds, NARR, arr, tw, a.fd, cs, tid, nt);
    }
    if (POINTWISE) {
        if constexpr (one_chunk) {
            pointwise(pmm, pgc, pgr, tid, std::integral_constant<int, PRE>());
        } else {
            for (int i0 = tid; i0 < total; i0 += nt * UN) {
                cx<T> mm[UN], gc[UN];
                T gr[UN];
                load_mult(mm, gc, gr, i0, std::integral_constant<int, UN>());
                pointwise(mm, gc, gr, i0, std::integral_constant<int, UN>());
            }
        }
        __syncthreads();
        fft_inv<T, NF, 16, kFuseMax<T>>(lds, NARR, arr, tw, a.fd, cs, tid, nt);
    }
    for (int i = tid; i < total; i += nt) {
        const int c = i & (C - 1), e = i >> cs;
        const size_t g = ((tbase + srow((size_t)seg * a.seg_stride + (size_t)e * a.row_stride)) << cs) + c;
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
    int H, W, batch;
    int V_groups;        // W / V
    FftDesc fw;
    SpecLayout lay;
    const cx<T>* twW;    // global twiddle tables for W
    const cx<T>* wtw;    // per-lane twiddles of the wave-per-row kernels ([64][16]) or null
    T* nyqd;             // dense home of spectral column W/2 (real parts, [batch][2 arrays][H]) or null = inside the tiles
    cx<T>* S0;
    cx<T>* S1;
    const cx<T>* eDrow;  // row-spectral D table (W/2+1 entries) or null = 2-tap stencils
    const int* stop;     // per image, may be null
    double* partials;    // [batch*H][4]
    int do_consume, do_produce;
    int ilv;             // S0 holds the (S0, S1) pairs interleaved element by element (wave kernels only); S1 unused
    long long* dbg;      // diagnostic builds only: per-workgroup phase stamps (null = off)
    int dbg_mode;        // diagnostic: bit0 skip the transforms, bit1 skip pack/unpack, bit2 skip iterate stores
    // Chambolle-Pock on the wave kernels: spectral column W/2 never enters the column stage.  Its multiplier is the
    // transform of nyq_kh real taps, so K (or K^T) acts on it as a short circular correlation along the rows, which the
    // consuming row launch evaluates itself from the values the previous launch left in the dense array:
    //   out[row] = nyq_scale * sum_u nyq_taps[u] * in[(row + nyq_dir * (u - nyq_ph)) mod H].
    // The dense array's two halves (per image) alternate as input and output: nyq_flip = 1 reads the second, writes the first.
    const T* nyq_taps;   // null = off (column W/2 goes through the column stage)
    int nyq_kh, nyq_ph, nyq_dir, nyq_flip;
    T nyq_scale;
};

// LDS carve-up shared by all row kernels: data | twiddles | reduction scratch | edge values
template <typename T>
struct RowLds {
    cx<T>* data;
    cx<T>* tw;
    T* edge;
    double* red;
    __device__ __forceinline__ RowLds(unsigned char* smem, int W) {
        data = reinterpret_cast<cx<T>*>(smem);
        tw = data + lds_round(W);
        red = reinterpret_cast<double*>(tw + tw_lds_entries(W, tw_direct_rows<T>()));
        edge = reinterpret_cast<T*>(red + 16 * 4);
    }
};
template <typename T>
__host__ __device__ inline size_t row_lds_bytes(int W, int V) {
    return (size_t)(lds_round(W) + tw_lds_entries(W, tw_direct_rows<T>())) * sizeof(cx<T>) + 16 * 4 * sizeof(double) + (size_t)(W / V + 8) * sizeof(T);
}

// ============================================================ solver policies (one pixel group)
// step(): o = offset of the group's first pixel in an image array.  When `consume`, (u, k, du) are the
// inverse-transformed outputs of the column stage at these pixels (resolvent output, K applied to it,
// D applied to it): finish the iteration in place.  When `produce`, return the inputs of the next
// column stage: re (identity path, without the D^T part), im (K^T path) and s, the array whose
// D^T image is added to re.  acc: sums of |d| and d^2 for the loss pairs.

// primal_dr/algorithm.py:94-129
template <typename T>
struct PdrP {
    struct Args {
        T* __restrict__ z1; T* __restrict__ z20; T* z21; T* z22;   // z21 and z22 may be ONE array (equal duals: ddn_pdr_run)
        const T* __restrict__ b;
        T t, rho, tg, l2div;  // tg = t*gamma, l2div = 2t + 1
        int objective;
    };
    static __device__ __forceinline__ void xy(const Args& a, T z1, T z20, T z21, T z22, T b, T& x, T& y0, T& y1, T& y2) {
        x = clip01(z1);                                                                           // pdr:96
        y0 = b + data_prox(a.objective, z20 - b, a.t, a.l2div);                                   // pdr:105 / 110
        iso(z21, z22, a.tg, y1, y2);                                                              // pdr:103 / 108
    }
    template <int V>
    struct St {
        T z1[V], z20[V], z21[V], z22[V], b[V];
    };
    template <int V>
    static __device__ __forceinline__ void load(const Args& a, size_t o, St<V>& q) {
        vload<T, V>(a.z1 + o, q.z1);
        vload<T, V>(a.z20 + o, q.z20);
        vload<T, V>(a.z21 + o, q.z21);
        vload<T, V>(a.z22 + o, q.z22);
        vload_keep<T, V>(a.b + o, q.b);
    }
    template <int V>
    static __device__ __forceinline__ void step(const Args& a, bool consume, bool produce, size_t o, const T (&u)[V], const T (&k)[V], const T (&du)[V],
                                                T (&re)[V], T (&im)[V], T (&s)[V], double (&acc)[4]) {
        St<V> q;
        load<V>(a, o, q);
        step_regs<V>(a, consume, produce, o, q, u, k, du, re, im, s, acc);
    }
    template <int V>
    static __device__ __forceinline__ void step_regs(const Args& a, bool consume, bool produce, size_t o, St<V>& q, const T (&u)[V], const T (&k)[V],
                                                     const T (&du)[V], T (&re)[V], T (&im)[V], T (&s)[V], double (&acc)[4]) {
        T (&z1)[V] = q.z1; T (&z20)[V] = q.z20; T (&z21)[V] = q.z21; T (&z22)[V] = q.z22; T (&b)[V] = q.b;
        if (consume) {
#pragma unroll
            for (int v = 0; v < V; ++v) {
                T x, y0, y1, y2;
                xy(a, z1[v], z20[v], z21[v], z22[v], b[v], x, y0, y1, y2);
                const T n1 = z1[v] + a.rho * (u[v] - x);                                          // pdr:128
                const T d1 = n1 - z1[v], d2 = n1 - b[v];
                acc[0] += (double)t_abs(d1); acc[1] += (double)d1 * (double)d1;
                acc[2] += (double)t_abs(d2); acc[3] += (double)d2 * (double)d2;
                z1[v] = n1;
                z20[v] = z20[v] + a.rho * (k[v] - y0);                                            // pdr:129
                z21[v] = z21[v] + a.rho * (du[v] - y1);
                z22[v] = z22[v] + a.rho * (du[v] - y2);
            }
            vstore<T, V>(a.z1 + o, z1);
            vstore<T, V>(a.z20 + o, z20);
            vstore<T, V>(a.z21 + o, z21);
            vstore<T, V>(a.z22 + o, z22);
        }
        if (produce) {
#pragma unroll
            for (int v = 0; v < V; ++v) {
                T x, y0, y1, y2;
                xy(a, z1[v], z20[v], z21[v], z22[v], b[v], x, y0, y1, y2);
                re[v] = (T)2 * x - z1[v];                                                         // pdr:119
                im[v] = (T)2 * y0 - z20[v];                                                       // pdr:115
                s[v] = ((T)2 * y1 - z21[v]) + ((T)2 * y2 - z22[v]);
            }
        }
    }
};

// primal_dual_dr/algorithm.py:65-101
template <typename T>
struct PddrP {
    struct Args {
        T* __restrict__ p; T* __restrict__ q0; T* __restrict__ q1; T* __restrict__ q2;
        const T* __restrict__ b;
        T t, it, rho, itg;  // it = 1/t, itg = gamma/t
    };
    static __device__ __forceinline__ void xz(const Args& a, T p, T q0, T q1, T q2, T b, T& x, T& z0, T& z1, T& z2) {
        x = clip01(p);                                                                            // pddr:66
        z0 = q0 - a.t * b - a.t * soft(a.it * q0 - b, a.it);                                      // pddr:69
        T i1, i2;
        iso(a.it * q1, a.it * q2, a.itg, i1, i2);                                                 // pddr:70
        z1 = q1 - a.t * i1;
        z2 = q2 - a.t * i2;
    }
    template <int V>
    struct St {
        T p[V], q0[V], q1[V], q2[V], b[V];
    };
    template <int V>
    static __device__ __forceinline__ void load(const Args& a, size_t o, St<V>& q) {
        vload<T, V>(a.p + o, q.p);
        vload<T, V>(a.q0 + o, q.q0);
        vload<T, V>(a.q1 + o, q.q1);
        vload<T, V>(a.q2 + o, q.q2);
        vload_keep<T, V>(a.b + o, q.b);
    }
    template <int V>
    static __device__ __forceinline__ void step(const Args& a, bool consume, bool produce, size_t o, const T (&w)[V], const T (&kw)[V], const T (&dw)[V],
                                                T (&re)[V], T (&im)[V], T (&s)[V], double (&acc)[4]) {
        St<V> q;
        load<V>(a, o, q);
        step_regs<V>(a, consume, produce, o, q, w, kw, dw, re, im, s, acc);
    }
    template <int V>
    static __device__ __forceinline__ void step_regs(const Args& a, bool consume, bool produce, size_t o, St<V>& st, const T (&w)[V], const T (&kw)[V],
                                                     const T (&dw)[V], T (&re)[V], T (&im)[V], T (&s)[V], double (&acc)[4]) {
        T (&p)[V] = st.p; T (&q0)[V] = st.q0; T (&q1)[V] = st.q1; T (&q2)[V] = st.q2; T (&b)[V] = st.b;
        if (consume) {
#pragma unroll
            for (int v = 0; v < V; ++v) {
                T x, z0, z1, z2;
                xz(a, p[v], q0[v], q1[v], q2[v], b[v], x, z0, z1, z2);
                const T o0 = (T)2 * z0 - q0[v], o1 = (T)2 * z1 - q1[v], o2 = (T)2 * z2 - q2[v];  // pddr:77-80
                p[v] = p[v] + a.rho * (w[v] - x);                                                 // pddr:100
                q0[v] = q0[v] + a.rho * ((a.t * kw[v] + o0) - z0);                                // pddr:90-101
                q1[v] = q1[v] + a.rho * ((a.t * dw[v] + o1) - z1);
                q2[v] = q2[v] + a.rho * ((a.t * dw[v] + o2) - z2);
                const T d = clip01(p[v]) - b[v];                                                  // pddr:104
                acc[0] += (double)t_abs(d); acc[1] += (double)d * (double)d;
            }
            vstore<T, V>(a.p + o, p);
            vstore<T, V>(a.q0 + o, q0);
            vstore<T, V>(a.q1 + o, q1);
            vstore<T, V>(a.q2 + o, q2);
        }
        if (produce) {
#pragma unroll
            for (int v = 0; v < V; ++v) {
                T x, z0, z1, z2;
                xz(a, p[v], q0[v], q1[v], q2[v], b[v], x, z0, z1, z2);
                re[v] = (T)2 * x - p[v];                                                          // pddr:76
                im[v] = -a.t * ((T)2 * z0 - q0[v]);                                               // pddr:83
                s[v] = -a.t * (((T)2 * z1 - q1[v]) + ((T)2 * z2 - q2[v]));
            }
        }
    }
};

// admm/algorithm.py:69-115 (and the solves of admm:118-132, 160-173)
template <typename T>
struct AdmmP {
    struct Args {
        T* __restrict__ u; T* __restrict__ w; T* __restrict__ y0; T* __restrict__ y1; T* __restrict__ y2; T* __restrict__ z0; T* __restrict__ z1; T* __restrict__ z2;
        const T* __restrict__ b;
        T* __restrict__ sol;  // when non-null the consume step stores x (the look-ahead solution of admm:118-132)
        T t, it, rho, omr, itg, l2div;  // it = 1/t, omr = 1 - rho, itg = gamma/t, l2div = 2/t + 1
        int objective;
        int sol_only;  // consume computes x and the loss only (final solve, admm:160-173)
    };
    template <int V>
    struct St {
        T b[V], u[V], w[V], y0[V], y1[V], y2[V], z0[V], z1[V], z2[V];
    };
    template <int V>
    static __device__ __forceinline__ void load(const Args& a, size_t o, St<V>& q) {
        vload_keep<T, V>(a.b + o, q.b);
        if (a.sol_only) return;
        vload<T, V>(a.u + o, q.u);
        vload<T, V>(a.w + o, q.w);
        vload<T, V>(a.y0 + o, q.y0);
        vload<T, V>(a.y1 + o, q.y1);
        vload<T, V>(a.y2 + o, q.y2);
        vload<T, V>(a.z0 + o, q.z0);
        vload<T, V>(a.z1 + o, q.z1);
        vload<T, V>(a.z2 + o, q.z2);
    }
    template <int V>
    static __device__ __forceinline__ void step(const Args& a, bool consume, bool produce, size_t o, const T (&x)[V], const T (&kx)[V], const T (&dx)[V],
                                                T (&re)[V], T (&im)[V], T (&s)[V], double (&acc)[4]) {
        St<V> q;
        load<V>(a, o, q);
        step_regs<V>(a, consume, produce, o, q, x, kx, dx, re, im, s, acc);
    }
    template <int V>
    static __device__ __forceinline__ void step_regs(const Args& a, bool consume, bool produce, size_t o, St<V>& st, const T (&x)[V], const T (&kx)[V],
                                                     const T (&dx)[V], T (&re)[V], T (&im)[V], T (&s)[V], double (&acc)[4]) {
        T (&b)[V] = st.b; T (&u)[V] = st.u; T (&w)[V] = st.w; T (&y0)[V] = st.y0; T (&y1)[V] = st.y1; T (&y2)[V] = st.y2;
        T (&z0)[V] = st.z0; T (&z1)[V] = st.z1; T (&z2)[V] = st.z2;
        if (consume) {
#pragma unroll
            for (int v = 0; v < V; ++v) {
                const T d = x[v] - b[v];
                acc[0] += (double)t_abs(d); acc[1] += (double)d * (double)d;
            }
            if (a.sol) vstore<T, V>(a.sol + o, x);
        }
        if (a.sol_only) return;
        if (consume) {
#pragma unroll
            for (int v = 0; v < V; ++v) {
                const T un = clip01(a.rho * x[v] + a.omr * u[v] + a.it * w[v]);                   // admm:87-88
                const T pp0 = a.rho * kx[v] + a.omr * y0[v] + a.it * z0[v];                       // admm:98
                const T pp1 = a.rho * dx[v] + a.omr * y1[v] + a.it * z1[v];
                const T pp2 = a.rho * dx[v] + a.omr * y2[v] + a.it * z2[v];
                T y0n, y1n, y2n;
                y0n = b[v] + data_prox(a.objective, pp0 - b[v], a.it, a.l2div);                   // admm:99-104
                iso(pp1, pp2, a.itg, y1n, y2n);                                                   // admm:105
                w[v] = w[v] + a.t * (x[v] - un);                                                  // admm:108
                z0[v] = z0[v] + a.t * (kx[v] - y0n);                                              // admm:109
                z1[v] = z1[v] + a.t * (dx[v] - y1n);
                z2[v] = z2[v] + a.t * (dx[v] - y2n);
                u[v] = un; y0[v] = y0n; y1[v] = y1n; y2[v] = y2n;
            }
            vstore<T, V>(a.u + o, u);
            vstore<T, V>(a.w + o, w);
            vstore<T, V>(a.y0 + o, y0);
            vstore<T, V>(a.y1 + o, y1);
            vstore<T, V>(a.y2 + o, y2);
            vstore<T, V>(a.z0 + o, z0);
            vstore<T, V>(a.z1 + o, z1);
            vstore<T, V>(a.z2 + o, z2);
        }
        if (produce) {
#pragma unroll
            for (int v = 0; v < V; ++v) {
                re[v] = u[v] - a.it * w[v];                                                       // admm:85
                im[v] = y0[v] - a.it * z0[v];
                s[v] = (y1[v] + y2[v]) - a.it * (z1[v] + z2[v]);
            }
        }
    }
};

// XCD-aware order of the rows a launch works through.  A row's piece of a spectral tile is (8 << cs) bytes (float), so
// 16 >> cs consecutive rows share a 128-byte line; workgroups go to the eight XCDs round-robin by block index, each XCD
// with an L2 of its own, so consecutive rows on consecutive workgroups make every L2 fetch the whole line for its piece
// of it.  Within each group of 8 x (rows per line) rows, the rows of a line are dealt to workgroups eight indices apart --
// the same XCD, resident together.  v: position in launch order (block index + k x grid), n: rows in all; identity when
// the grid is not a multiple of eight (block index and position then differ modulo 8) and for a last partial group.
__device__ __forceinline__ int xcd_row(int v, int n, int rpl_shift) {
    if (rpl_shift <= 0 || (gridDim.x & 7)) return v;
    const int G = 8 << rpl_shift;
    const int g0 = v & ~(G - 1);
    if (g0 + G > n) return v;
    const int j = v - g0;
    return g0 + ((j & 7) << rpl_shift) + (j >> 3);
}
template <typename T>
__device__ __forceinline__ int rows_per_line_shift(int cs) { return max(0, (sizeof(T) == 4 ? 4 : 3) - cs); }

// ---------------------------------------------------------------- fused solver row kernel
// One workgroup per image row (grid-stride over rows).  Per row:
//   consume: pack (S0, S1) -> inverse transform -> per pixel (u, K u) and D u (2-tap stencil through
//            LDS, or a second inverse transform of eD * S0 when SPEC);
//   per pixel group: policy step (iterate update, loss sums, next transform inputs) written back in
//            place into LDS as re + D^T s (stencil; the last pixel of a group is fixed up after a
//            barrier from the neighbour's edge value) + i im;
//   produce: forward transform -> separate the two half-spectra -> (S0, S1)
//            (SPEC: a second forward transform of s adds conj(eD) * F(s) to S0).
// PF: every global load of a row (half-spectra and iterate arrays) is issued at the top of the row
// iteration and held in registers, so that ~100 KB per workgroup are in flight while the previous
// phase drains and the iterate loads hide behind the pack and inverse-transform phases; without
// it each phase is a chain of exposed HBM round trips and a CU keeps far too few bytes in flight.
template <typename T, class P, int V, int NG, bool SPEC, int PF, int WF, int MAXT, int MINW>
__global__ void __launch_bounds__(MAXT, MINW) k_solver_row(RowCommon<T> c, typename P::Args a) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int tid = threadIdx.x, nt = blockDim.x, W = c.W;
    RowLds<T> L(smem, W);
    const Tw<T> tw = tw_setup(L.tw, c.twW, W, tid, nt, tw_direct_rows<T>());
    const int nrows = c.batch * c.H;
    constexpr int NH = SPEC ? NG : 1;
    constexpr int KP = (V * NG) / 2 + 1;  // half-spectrum entries per thread when nt = W / (V*NG)
    constexpr int SCH = (V * NG == 8) ? 8 : 16;  // eight pixels per thread: radix-8 passes, one butterfly per thread
    const int rpl = (c.dbg_mode & 64) ? 0 : rows_per_line_shift<T>(c.lay.cs);
    for (int vrow = blockIdx.x; vrow < nrows; vrow += gridDim.x) {
        const int row = xcd_row(vrow, nrows, rpl);
        const int plane = row / c.H;
        if (c.stop && c.stop[plane] >= 0) continue;
        const size_t ro = (size_t)row * W;
        const SpecRow sr = spec_row(c.lay, plane, row - plane * c.H);
        T uprev[NG];
        T duh[NH][V], sh[NH][V];
        // PF == 1: iterate loads at the row top (needs ~250 registers); PF == 2: iterate loads of all
        // pixel groups issued together at the start of the per-pixel phase (fits 128 registers).
        // PF == 3: rolling double buffer, the loads of group g+1 are issued before group g is processed.
        constexpr int NPF = PF == 3 ? 2 : (PF ? NG : 1);
        constexpr int KPF = PF == 1 ? KP : 1;
        typename P::template St<V> stq[NPF];
        cx<T> pU[KPF], pV[KPF];
        if (PF == 1) {
            if (c.do_consume) row_pack_load<T, KPF>(pU, pV, W, c.S0, sr, c.S1, sr, tid, nt);
#pragma unroll
            for (int g = 0; g < NPF; ++g) {
                const int j = (tid + g * nt) * V;
                if (j < W) P::template load<V>(a, ro + j, stq[g]);
            }
        }
        __syncthreads();
#define DDN_STAMP(i) if (c.dbg && tid == 0 && vrow == (int)blockIdx.x + (int)gridDim.x) c.dbg[blockIdx.x * 16 + (i)] = clock64();
        DDN_STAMP(0)
        if (c.do_consume) {
            if (SPEC) {
                row_pack_mul<T, WF, SCH>(L.data, c.fw, W, c.S0, sr, c.eDrow, tid, nt);
                __syncthreads();
                fft_inv<T, WF, SCH>(L.data, 1, 0, tw, c.fw, 0, tid, nt);
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    const int j = (tid + g * nt) * V;
                    if (j < W) {
#pragma unroll
                        for (int v = 0; v < V; ++v) duh[g % NH][v] = L.data[SW(j + v)].x;
                    }
                }
                __syncthreads();
            }
            if (!(c.dbg_mode & 2)) {
                if (PF == 1) row_pack_store<T, KPF, WF, SCH>(L.data, c.fw, W, pU, pV, tid, nt);
                else row_pack<T, KP, WF, SCH>(L.data, c.fw, W, c.S0, sr, c.S1, sr, tid, nt);
            } else if (PF == 1) {
                T sacc = 0;
#pragma unroll
                for (int i = 0; i < KPF; ++i) sacc += pU[i].x + pV[i].y;
                L.data[SW(tid)].x = sacc;
            }
            __syncthreads();
            DDN_STAMP(1)
            if (!(c.dbg_mode & 1)) fft_inv<T, WF, SCH>(L.data, 1, 0, tw, c.fw, 0, tid, nt);
            DDN_STAMP(2)
            if (!SPEC) {
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    const int j = (tid + g * nt) * V;
                    if (j < W) uprev[g] = L.data[SW(j == 0 ? W - 1 : j - 1)].x;
                }
                __syncthreads();
            }
        }
        double acc[4] = {0, 0, 0, 0};
        DDN_STAMP(3)
        if (PF == 2) {
#pragma unroll
            for (int g = 0; g < NPF; ++g) {
                const int j = (tid + g * nt) * V;
                if (j < W) P::template load<V>(a, ro + j, stq[g]);
            }
        }
        if (PF == 3) {
            if (tid * V < W) P::template load<V>(a, ro + tid * V, stq[0]);
        }
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const int j = (tid + g * nt) * V;
            if (PF == 3 && g + 1 < NG) {
                const int jn = (tid + (g + 1) * nt) * V;
                if (jn < W) P::template load<V>(a, ro + jn, stq[(g + 1) % NPF]);
            }
            if (j < W) {
                T u[V], k[V], du[V], re[V], im[V], s[V];
                if (c.do_consume) {
                    T prev = SPEC ? (T)0 : uprev[g];
#pragma unroll
                    for (int v = 0; v < V; ++v) {
                        const cx<T> e = L.data[SW(j + v)];
                        u[v] = e.x;
                        k[v] = e.y;
                        du[v] = SPEC ? duh[g % NH][v] : e.x - prev;
                        prev = e.x;
                    }
                }
                const bool cons = c.do_consume != 0 && !(c.dbg_mode & 4);
                if (c.dbg_mode & 8) {
#pragma unroll
                    for (int v = 0; v < V; ++v) { re[v] = u[v]; im[v] = k[v]; s[v] = du[v]; }
                } else if (PF) P::template step_regs<V>(a, cons, c.do_produce != 0, ro + j, stq[g % NPF], u, k, du, re, im, s, acc);
                else P::template step<V>(a, cons, c.do_produce != 0, ro + j, u, k, du, re, im, s, acc);
                if (c.do_produce) {
                    if (SPEC) {
#pragma unroll
                        for (int v = 0; v < V; ++v) {
                            sh[g % NH][v] = s[v];
                            L.data[SW(j + v)] = mk<T>(re[v], im[v]);
                        }
                    } else {
#pragma unroll
                        for (int v = 0; v < V; ++v) {
                            const T nx = (v + 1 < V) ? s[(v + 1) % V] : (T)0;
                            L.data[SW(j + v)] = mk<T>(re[v] + (s[v] - nx), im[v]);
                        }
                        L.edge[tid + g * nt] = s[0];
                    }
                }
            }
        }
        DDN_STAMP(4)
        if (c.do_consume) {
            block_reduce<4>(acc, L.red, tid, nt);
            if (tid == 0) {
#pragma unroll
                for (int q = 0; q < 4; ++q) c.partials[(size_t)row * 4 + q] = acc[q];
            }
        }
        __syncthreads();
        DDN_STAMP(5)
        if (c.do_produce) {
            if (!SPEC) {
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    const int j = (tid + g * nt) * V;
                    if (j < W) {
                        const int grp = tid + g * nt;
                        const int li = SW(j + V - 1);
                        L.data[li].x -= L.edge[grp + 1 == c.V_groups ? 0 : grp + 1];
                    }
                }
                __syncthreads();
            }
            DDN_STAMP(6)
            if (!(c.dbg_mode & 1)) fft_fwd<T, WF, SCH>(L.data, 1, 0, tw, c.fw, 0, tid, nt);
            DDN_STAMP(7)
            if (!(c.dbg_mode & 2)) row_unpack<T, WF, SCH>(L.data, c.fw, W, c.S0, sr, c.S1, sr, nullptr, false, tid, nt);
            DDN_STAMP(8)
            if (SPEC) {
                __syncthreads();
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    const int j = (tid + g * nt) * V;
                    if (j < W) {
#pragma unroll
                        for (int v = 0; v < V; ++v) L.data[SW(j + v)] = mk<T>(sh[g % NH][v], 0);
                    }
                }
                __syncthreads();
                fft_fwd<T, WF, SCH>(L.data, 1, 0, tw, c.fw, 0, tid, nt);
                row_unpack<T, WF, SCH>(L.data, c.fw, W, c.S0, sr, nullptr, sr, c.eDrow, true, tid, nt);
            }
        }
    }
}

// ------------------------------------------------------------------ Chambolle-Pock row kernel
// chambolle_pock/algorithm.py:58-92.  A workgroup owns two image rows (the two real slots of the
// complex transform).  step 0 ("dual"): consume K z, update y (cp:60-76), produce the transform
// input y0.  step 1 ("primal"): consume K^T y0, update x and z (cp:79-87), produce the transform
// input z.  D z and D^T (y1 + y2) are 2-tap stencils read straight from the (read-only in that
// step) arrays; with SPEC they are row-spectral: the dual step spends one extra transform pair on
// D z and leaves conj(eD) F_row(y1 + y2) / W in S1, which the primal step adds before its inverse.
template <typename T>
struct CpArgs {
    T* __restrict__ x; T* __restrict__ z; T* __restrict__ y0; T* __restrict__ y1; T* __restrict__ y2;
    const T* b;
    T t, s, is, isg, l2div;  // is = 1/s, isg = gamma/s, l2div = 2/s + 1
    int objective;
    int step;
};

template <typename T, int V, int NG, bool SPEC, int WF, int MAXT, int MINW>
__global__ void __launch_bounds__(MAXT, MINW) k_cp_row(RowCommon<T> c, CpArgs<T> a) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int tid = threadIdx.x, nt = blockDim.x, W = c.W, H = c.H;
    RowLds<T> L(smem, W);
    const Tw<T> tw = tw_setup(L.tw, c.twW, W, tid, nt, tw_direct_rows<T>());
    const int ppp = (H + 1) >> 1;  // row pairs per image
    const int npairs = c.batch * ppp;
    constexpr int NH = SPEC ? NG : 1;
    constexpr int KP = (V * NG) / 2 + 1;
    constexpr int SCH = (V * NG == 8) ? 8 : 16;
    const int ppl = (c.dbg_mode & 64) ? 0 : max(0, rows_per_line_shift<T>(c.lay.cs) - 1);   // (a pair is two adjacent rows)
    for (int vpr = blockIdx.x; vpr < npairs; vpr += gridDim.x) {
        const int pr = xcd_row(vpr, npairs, ppl);
        const int plane = pr / ppp;
        const int r0 = (pr - plane * ppp) * 2;
        const int nslot = (r0 + 1 < H) ? 2 : 1;
        const size_t row0 = (size_t)plane * H + r0;
        const SpecRow sa = spec_row(c.lay, plane, r0), sb = spec_row(c.lay, plane, r0 + (nslot - 1));
        T aux[2][NH][V];  // row-spectral D z (dual step, SPEC)
        __syncthreads();
        if (c.do_consume) {
            if (SPEC && a.step == 0) {
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
                fft_fwd<T, WF, SCH>(L.data, 1, 0, tw, c.fw, 0, tid, nt);
                {
                    // both slots are multiplied by the Hermitian-extended eD: Z'[k] = eD[k] Z[k], Z'[W-k] = conj(eD[k]) Z[W-k]
                    const int half = W >> 1;
                    for (int k = tid; k <= half; k += nt) {
                        const int km = (k == 0) ? 0 : W - k;
                        const int p1 = SW(pos_freq<WF, SCH>(c.fw, k)), p2 = SW(pos_freq<WF, SCH>(c.fw, km));
                        const cx<T> e = c.eDrow[k];
                        const cx<T> Z1 = L.data[p1], Z2 = L.data[p2];
                        L.data[p1] = Z1 * e;
                        if (km != k) L.data[p2] = Z2 * cconj(e);
                    }
                }
                __syncthreads();
                fft_inv<T, WF, SCH>(L.data, 1, 0, tw, c.fw, 0, tid, nt);
                const T sc = (T)1 / (T)W;
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    const int j = (tid + g * nt) * V;
                    if (j < W) {
#pragma unroll
                        for (int v = 0; v < V; ++v) {
                            const cx<T> e = L.data[SW(j + v)];
                            aux[0][g % NH][v] = e.x * sc;
                            aux[1][g % NH][v] = e.y * sc;
                        }
                    }
                }
                __syncthreads();
            }
            const bool addS1 = SPEC && a.step == 1;
            row_pack<T, KP, WF, SCH>(L.data, c.fw, W, c.S0, sa, nslot == 2 ? c.S0 : (const cx<T>*)nullptr, sb, tid, nt, addS1 ? c.S1 : (const cx<T>*)nullptr, (addS1 && nslot == 2) ? c.S1 : (const cx<T>*)nullptr);
            __syncthreads();
            fft_inv<T, WF, SCH>(L.data, 1, 0, tw, c.fw, 0, tid, nt);
        }
        double acc[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const int j = (tid + g * nt) * V;
            if (j < W) {
                T in[2][V], out[2][V];
                if (c.do_consume) {
#pragma unroll
                    for (int v = 0; v < V; ++v) {
                        const cx<T> e = L.data[SW(j + v)];
                        in[0][v] = e.x;
                        in[1][v] = e.y;
                    }
                }
#pragma unroll
                for (int sl = 0; sl < 2; ++sl) {
#pragma unroll
                    for (int v = 0; v < V; ++v) out[sl][v] = (T)0;
                    if (sl < nslot) {
                        const size_t ro = (row0 + sl) * W;
                        if (a.step == 0) {
                            T z[V], y0[V], y1[V], y2[V], b[V];
                            vload<T, V>(a.z + ro + j, z);
                            vload<T, V>(a.y0 + ro + j, y0);
                            vload<T, V>(a.y1 + ro + j, y1);
                            vload<T, V>(a.y2 + ro + j, y2);
                            vload<T, V>(a.b + ro + j, b);
                            T prev = SPEC ? (T)0 : a.z[ro + (j == 0 ? W - 1 : j - 1)];
#pragma unroll
                            for (int v = 0; v < V; ++v) {
                                const T dz = SPEC ? aux[sl][g % NH][v] : z[v] - prev;             // cp:61
                                prev = z[v];
                                const T g0 = y0[v] + a.s * in[sl][v];                             // cp:66
                                const T g1 = y1[v] + a.s * dz;
                                const T g2 = y2[v] + a.s * dz;
                                T pr0;
                                if (a.objective == 0) pr0 = soft(g0 / a.s - b[v], a.is); else pr0 = (g0 / a.s - b[v]) / a.l2div;  // cp:69-72
                                y0[v] = g0 - a.s * b[v] - a.s * pr0;
                                T i1, i2;
                                iso(g1 / a.s, g2 / a.s, a.isg, i1, i2);                           // cp:75
                                y1[v] = g1 - a.s * i1;
                                y2[v] = g2 - a.s * i2;
                                out[sl][v] = y0[v];
                            }
                            vstore<T, V>(a.y0 + ro + j, y0);
                            vstore<T, V>(a.y1 + ro + j, y1);
                            vstore<T, V>(a.y2 + ro + j, y2);
                        } else {
                            T z[V];
                            if (c.do_consume) {
                                T x[V], b[V], dty[V];
                                vload<T, V>(a.x + ro + j, x);
                                vload<T, V>(a.b + ro + j, b);
                                if (SPEC) {
#pragma unroll
                                    for (int v = 0; v < V; ++v) dty[v] = (T)0;  // already inside `in` (added in the row spectrum)
                                } else {
                                    T y1[V], y2[V];
                                    vload<T, V>(a.y1 + ro + j, y1);
                                    vload<T, V>(a.y2 + ro + j, y2);
                                    const int jn = (j + V == W) ? 0 : j + V;
                                    const T snext = a.y1[ro + jn] + a.y2[ro + jn];
#pragma unroll
                                    for (int v = 0; v < V; ++v) {
                                        const T sc = y1[v] + y2[v];
                                        const T sn = (v + 1 < V) ? (y1[(v + 1) % V] + y2[(v + 1) % V]) : snext;
                                        dty[v] = sc - sn;                                         // D^T(y1 + y2), cp:79
                                    }
                                }
#pragma unroll
                                for (int v = 0; v < V; ++v) {
                                    const T aty = in[sl][v] + dty[v];
                                    const T xn = clip01(x[v] - a.t * aty);                        // cp:83-86
                                    z[v] = (T)2 * xn - x[v];                                      // cp:87
                                    x[v] = xn;
                                    const T d = xn - b[v];                                        // cp:96
                                    acc[sl][0] += (double)t_abs(d); acc[sl][1] += (double)d * (double)d;
                                }
                                vstore<T, V>(a.x + ro + j, x);
                                vstore<T, V>(a.z + ro + j, z);
                            } else {
                                vload<T, V>(a.z + ro + j, z);
                            }
#pragma unroll
                            for (int v = 0; v < V; ++v) out[sl][v] = z[v];
                        }
                    }
                }
                if (c.do_produce) {
#pragma unroll
                    for (int v = 0; v < V; ++v) L.data[SW(j + v)] = mk<T>(out[0][v], out[1][v]);
                }
            }
        }
        if (a.step == 1 && c.do_consume) {
            double accs[4];
            for (int sl = 0; sl < 2; ++sl) {
#pragma unroll
                for (int q = 0; q < 4; ++q) accs[q] = acc[sl][q];
                block_reduce<4>(accs, L.red, tid, nt);
                if (tid == 0 && sl < nslot) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) c.partials[(row0 + sl) * 4 + q] = accs[q];
                }
            }
        }
        __syncthreads();
        if (c.do_produce) {
            fft_fwd<T, WF, SCH>(L.data, 1, 0, tw, c.fw, 0, tid, nt);
            row_unpack<T, WF, SCH>(L.data, c.fw, W, c.S0, sa, nslot == 2 ? c.S0 : nullptr, sb, nullptr, false, tid, nt);
            if (SPEC && a.step == 0) {
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
                fft_fwd<T, WF, SCH>(L.data, 1, 0, tw, c.fw, 0, tid, nt);
                {
                    const int half = W >> 1;
                    const T hw = (T)0.5 / (T)W;  // 1/W: this spectrum skips the normalised column stage
                    for (int k = tid; k <= half; k += nt) {
                        const int km = (k == 0) ? 0 : W - k;
                        const cx<T> Z1 = L.data[SW(pos_freq<WF, SCH>(c.fw, k))], Z2 = L.data[SW(pos_freq<WF, SCH>(c.fw, km))];
                        const cx<T> e = c.eDrow[k];
                        c.S1[sa.at(k)] = mulc(mk<T>(hw * (Z1.x + Z2.x), hw * (Z1.y - Z2.y)), e);
                        if (nslot == 2) c.S1[sb.at(k)] = mulc(mk<T>(hw * (Z1.y + Z2.y), hw * (Z2.x - Z1.x)), e);
                    }
                }
            }
        }
    }
}

// ------------------------------------------------------------ plain row transforms (row pairs)
template <typename T>
struct PlainRowArgs {
    int H, W, batch;
    FftDesc fw;
    SpecLayout lay;
    const cx<T>* twW;
    cx<T>* S;       // half-row-spectra, tiled layout
    const T* src;   // real -> spectrum
    T* dst;         // spectrum -> real
    T scale;        // applied to the real output
};

template <typename T>
__global__ void __launch_bounds__(1024) k_rows_r2c(PlainRowArgs<T> a) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int tid = threadIdx.x, nt = blockDim.x, W = a.W, H = a.H;
    RowLds<T> L(smem, W);
    const Tw<T> tw = tw_setup(L.tw, a.twW, W, tid, nt, tw_direct_rows<T>());
    const int ppp = (H + 1) >> 1, npairs = a.batch * ppp;
    for (int pr = blockIdx.x; pr < npairs; pr += gridDim.x) {
        const int plane = pr / ppp, r0 = (pr - plane * ppp) * 2;
        const bool two = r0 + 1 < H;
        const size_t row0 = (size_t)plane * H + r0;
        const SpecRow sa = spec_row(a.lay, plane, r0), sb = spec_row(a.lay, plane, two ? r0 + 1 : r0);
        __syncthreads();
        for (int j = tid; j < W; j += nt) L.data[SW(j)] = mk<T>(a.src[row0 * W + j], two ? a.src[(row0 + 1) * W + j] : (T)0);
        __syncthreads();
        fft_forward<T>(L.data, 1, 0, tw, a.fw, 0, tid, nt);
        row_unpack<T>(L.data, a.fw, W, a.S, sa, two ? a.S : nullptr, sb, nullptr, false, tid, nt);
    }
}

template <typename T>
__global__ void __launch_bounds__(1024) k_rows_c2r(PlainRowArgs<T> a) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int tid = threadIdx.x, nt = blockDim.x, W = a.W, H = a.H;
    RowLds<T> L(smem, W);
    const Tw<T> tw = tw_setup(L.tw, a.twW, W, tid, nt, tw_direct_rows<T>());
    const int ppp = (H + 1) >> 1, npairs = a.batch * ppp;
    for (int pr = blockIdx.x; pr < npairs; pr += gridDim.x) {
        const int plane = pr / ppp, r0 = (pr - plane * ppp) * 2;
        const bool two = r0 + 1 < H;
        const size_t row0 = (size_t)plane * H + r0;
        const SpecRow sa = spec_row(a.lay, plane, r0), sb = spec_row(a.lay, plane, two ? r0 + 1 : r0);
        __syncthreads();
        row_pack<T, 5>(L.data, a.fw, W, a.S, sa, two ? a.S : (const cx<T>*)nullptr, sb, tid, nt);
        __syncthreads();
        fft_inverse<T>(L.data, 1, 0, tw, a.fw, 0, tid, nt);
        for (int j = tid; j < W; j += nt) {
            const cx<T> e = L.data[SW(j)];
            a.dst[row0 * W + j] = e.x * a.scale;
            if (two) a.dst[(row0 + 1) * W + j] = e.y * a.scale;
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
static __global__ void k_reduce_final(const double* partial, int chunks, double npix, double* out4) {
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
__global__ void k_import_mult(const cx<TI>* in, cx<T>* out, T* out_real, int H, int W, int Wh, SpecLayout lay, ColMap cm, int recip, int conj_in, double scale) {
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
        const size_t o = spec_row(lay, 0, r).at(k2);
        if (out) out[o] = mk<T>((T)rx, (T)ry);
        if (out_real) out_real[o] = (T)rx;
    }
}

// G[r][k2] = scale / (1 + cK |eK|^2 + cD ed2[k2])  (mm:67-72 in real arithmetic)
template <typename T>
__global__ void k_make_G(const cx<T>* eK, const T* ed2, T* G, int H, int Wh, SpecLayout lay, T cK, T cD, T scale) {
    const size_t total = (size_t)H * Wh;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int r = (int)(i / Wh), k2 = (int)(i - (size_t)r * Wh);
        const size_t o = spec_row(lay, 0, r).at(k2);
        const cx<T> e = eK[o];
        G[o] = scale / ((T)1 + cK * (e.x * e.x + e.y * e.y) + cD * ed2[k2]);
    }
}
// mult[r][k2] = 1 + |eK|^2 + 2 ed2  (apply_Mat, mm:106-108)
template <typename T>
__global__ void k_make_mat(const cx<T>* eK, const T* ed2, cx<T>* out, int H, int Wh, SpecLayout lay) {
    const size_t total = (size_t)H * Wh;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int r = (int)(i / Wh), k2 = (int)(i - (size_t)r * Wh);
        const size_t o = spec_row(lay, 0, r).at(k2);
        const cx<T> e = eK[o];
        out[o] = mk<T>((T)1 + (e.x * e.x + e.y * e.y) + (T)2 * ed2[k2], (T)0);
    }
}
// broadcast a row table to every row: out[r][k2] = tab[k2]
template <typename T>
__global__ void k_bcast_row(const cx<T>* tab, cx<T>* out, int H, int Wh, SpecLayout lay) {
    const size_t total = (size_t)H * Wh;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int r = (int)(i / Wh), k2 = (int)(i - (size_t)r * Wh);
        out[spec_row(lay, 0, r).at(k2)] = tab[k2];
    }
}

// Export a half-plane, position-ordered multiplier to the reference's full natural layout.
template <typename T>
__global__ void k_export_full(const cx<T>* half, cx<T>* out, int H, int W, SpecLayout lay, ColMap cm) {
    const size_t total = (size_t)H * W;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int k1 = (int)(i / W), k2 = (int)(i - (size_t)k1 * W);
        if (k2 <= W / 2) {
            out[i] = half[spec_row(lay, 0, colpos_of_freq(cm, k1)).at(k2)];
        } else {
            const int m1 = k1 == 0 ? 0 : H - k1;
            out[i] = cconj(half[spec_row(lay, 0, colpos_of_freq(cm, m1)).at(W - k2)]);
        }
    }
}

static __global__ void k_iters_done(const int* stop, int* out, int n, int n_iters) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) out[i] = stop[i] >= 0 ? stop[i] : n_iters;
}

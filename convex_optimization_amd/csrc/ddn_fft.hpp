// Device-side FFT building blocks for gfx950: complex type, small-radix butterflies,
// in-place LDS passes (decimation in frequency forward, decimation in time inverse) and the
// digit-reversal maps between natural frequency index and LDS position.
//
// Layout convention.  A length-n transform over `C = 1 << cshift` interleaved sequences lives in
// LDS as element (pos, c) at index SW((pos << cshift) + c).  The forward transform takes natural
// order to digit-reversed order, the inverse takes digit-reversed order back to natural order,
// so a forward / pointwise multiply / inverse round trip needs no reordering pass: multiplier
// arrays are stored once in the same digit-reversed order.
#pragma once
#include <type_traits>
#include <hip/hip_runtime.h>

#define DDN_MAXPASS 8

template <typename T>
struct cx {
    T x, y;
};

template <typename T>
__device__ __forceinline__ cx<T> mk(T a, T b) {
    cx<T> r;
    r.x = a;
    r.y = b;
    return r;
}
template <typename T>
__device__ __forceinline__ cx<T> operator+(cx<T> a, cx<T> b) { return mk<T>(a.x + b.x, a.y + b.y); }
template <typename T>
__device__ __forceinline__ cx<T> operator-(cx<T> a, cx<T> b) { return mk<T>(a.x - b.x, a.y - b.y); }
template <typename T>
__device__ __forceinline__ cx<T> operator*(cx<T> a, cx<T> b) { return mk<T>(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
template <typename T>
__device__ __forceinline__ cx<T> operator*(cx<T> a, T s) { return mk<T>(a.x * s, a.y * s); }
template <typename T>
__device__ __forceinline__ cx<T> cconj(cx<T> a) { return mk<T>(a.x, -a.y); }
// a * conj(b)
template <typename T>
__device__ __forceinline__ cx<T> mulc(cx<T> a, cx<T> b) { return mk<T>(a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y); }
// multiply by -i (forward rotation) or +i (inverse rotation)
template <typename T, bool INV>
__device__ __forceinline__ cx<T> rot90(cx<T> a) {
    if (INV) return mk<T>(-a.y, a.x);
    return mk<T>(a.y, -a.x);
}

// Factorisation of one transform length.  Pass s works on blocks of length m[s]*radix[s];
// m[s] = n / (radix[0] * ... * radix[s]);  twmul[s] = n / (m[s]*radix[s]).
struct FftDesc {
    int n;
    int npass;
    int pow2;                 // every radix is a power of two
    int radix[DDN_MAXPASS];
    int rshift[DDN_MAXPASS];  // log2(radix) when pow2
    int m[DDN_MAXPASS];
    int mshift[DDN_MAXPASS];  // log2(m) when m is a power of two, else -1
    int twmul[DDN_MAXPASS];
    // ceil(2^32 / radix) and ceil(2^32 / m) (0 when the divisor is 1): the quotient of n by d is the high word of
    // n * ceil(2^32 / d), exactly, whenever n * d < 2^32 -- every index here is below 2^17 and every divisor below 2^13.
    // A run-time integer division costs some forty instructions, and the position of a frequency needs two per pass.
    unsigned rmagic[DDN_MAXPASS];
    unsigned mmagic[DDN_MAXPASS];
};
__device__ __forceinline__ int fast_div(int n, unsigned magic) { return magic ? (int)__umulhi((unsigned)n, magic) : n; }

// Twiddle source exp(-2 pi i e / n), e in [0, n).  Short transforms keep the whole table in LDS; long
// ones keep two small tables and form w^e = w^(64*(e>>6)) * w^(e&63) with one complex multiply, so
// that the table does not compete with the data for LDS (a 4096-entry table would halve occupancy).
// (Direct up to 2048 entries, 4096 in the row kernels, whose LDS holds one row: a two-level twiddle is two LDS reads and
// a complex multiply where a butterfly's own work is a read, a write and a few operations per element, and these
// kernels are bound by their vector instructions -- one image of 1920^2 28.4 -> 29.9 k Mpx-it/s with 2048.  The column
// kernels fill the LDS with columns: heights of 2049 ... 2490 rows only fit four at a time beside the small tables.)
#define DDN_TW_DIRECT_MAX 2048
#define DDN_TW_DIRECT_ROWS 4096
// (float rows; a 4096-entry table of doubles beside a 4096-pixel row of doubles leaves no room for what the split-row
// kernels keep there as well)
template <typename T>
__host__ __device__ constexpr int tw_direct_rows() { return sizeof(T) == 4 ? DDN_TW_DIRECT_ROWS : DDN_TW_DIRECT_MAX; }
template <typename T>
struct Tw {
    const cx<T>* t1;
    const cx<T>* t2;
    int two;
    __device__ __forceinline__ cx<T> operator[](int e) const {
        if (!two) return t1[e];
        return t1[e >> 6] * t2[e & 63];
    }
};
__host__ __device__ __forceinline__ int tw_hi_entries(int n) { return ((n - 1) >> 6) + 1; }
// entries of LDS needed for the twiddles of a length-n transform
__host__ __device__ __forceinline__ int tw_lds_entries(int n, int dmax = DDN_TW_DIRECT_MAX) { return n <= dmax ? n : tw_hi_entries(n) + 64; }
// Global layout produced by the host: [direct n][hi tw_hi_entries(n)][lo 64].
template <typename T>
__device__ __forceinline__ Tw<T> tw_setup(cx<T>* area, const cx<T>* glob, int n, int tid, int nt, int dmax = DDN_TW_DIRECT_MAX) {
    Tw<T> tw;
    if (n <= dmax) {
        for (int i = tid; i < n; i += nt) area[i] = glob[i];
        tw.t1 = area;
        tw.t2 = area;
        tw.two = 0;
    } else {
        const int nh = tw_hi_entries(n);
        for (int i = tid; i < nh + 64; i += nt) area[i] = glob[n + i];
        tw.t1 = area;
        tw.t2 = area + nh;
        tw.two = 1;
    }
    return tw;
}

// LDS index swizzle: XOR the low five index bits with two higher bit groups so that the strided
// accesses of the butterfly passes (strides 16, 32, 256, 1024 ... elements) and the
// digit-reversed gathers spread over the banks.  A bijection on every aligned block of 32.
#ifdef DDN_SW_ONE_TERM
// (units whose kernels hold ONE row in the LDS: no access there strides by multiples of 1024 elements, and the second
// term is two of the five instructions of every LDS index)
__device__ __forceinline__ int SW(int i) { return i ^ ((i >> 5) & 31); }
#else
__device__ __forceinline__ int SW(int i) { return i ^ ((i >> 5) & 31) ^ ((i >> 10) & 31); }
#endif
__host__ __device__ __forceinline__ int lds_round(int n) { return (n + 31) & ~31; }

__device__ __forceinline__ int pos_of_freq(const FftDesc& d, int k) {
    int pos = 0;
    if (d.pow2) {
        for (int s = 0; s < d.npass; ++s) {
            pos += (k & (d.radix[s] - 1)) << d.mshift[s];
            k >>= d.rshift[s];
        }
        return pos;
    }
    for (int s = 0; s < d.npass; ++s) {
        const int r = d.radix[s];
        const int kq = fast_div(k, d.rmagic[s]);
        pos += (k - kq * r) * d.m[s];
        k = kq;
    }
    return pos;
}
__device__ __forceinline__ int freq_of_pos(const FftDesc& d, int p) {
    int k = 0, mult = 1;
    if (d.pow2) {
        int sh = 0;
        for (int s = 0; s < d.npass; ++s) {
            k += ((p >> d.mshift[s]) & (d.radix[s] - 1)) << sh;
            sh += d.rshift[s];
        }
        return k;
    }
    for (int s = 0; s < d.npass; ++s) {
        const int r = d.radix[s];
        const int pm = fast_div(p, d.mmagic[s]);
        const int q = pm - fast_div(pm, d.rmagic[s]) * r;
        k += q * mult;
        mult *= r;
    }
    return k;
}

// ------------------------------------------------------------------ small DFTs in registers
template <typename T, int R, bool INV>
struct Dft;

template <typename T, bool INV>
struct Dft<T, 2, INV> {
    static __device__ __forceinline__ void run(cx<T>* v) {
        cx<T> a = v[0], b = v[1];
        v[0] = a + b;
        v[1] = a - b;
    }
};

template <typename T, bool INV>
struct Dft<T, 4, INV> {
    static __device__ __forceinline__ void run(cx<T>* v) {
        cx<T> t0 = v[0] + v[2], t1 = v[0] - v[2], t2 = v[1] + v[3], t3 = rot90<T, INV>(v[1] - v[3]);
        v[0] = t0 + t2;
        v[1] = t1 + t3;
        v[2] = t0 - t2;
        v[3] = t1 - t3;
    }
};

template <typename T, bool INV>
struct Dft<T, 8, INV> {
    static __device__ __forceinline__ void run(cx<T>* v) {
        cx<T> e[4] = {v[0], v[2], v[4], v[6]};
        cx<T> o[4] = {v[1], v[3], v[5], v[7]};
        Dft<T, 4, INV>::run(e);
        Dft<T, 4, INV>::run(o);
        const T h = (T)0.70710678118654752440;
        // w8^1 = (1 -+ i)/sqrt2, w8^2 = -+i, w8^3 = (-1 -+ i)/sqrt2
        cx<T> o1 = INV ? mk<T>((o[1].x - o[1].y) * h, (o[1].x + o[1].y) * h) : mk<T>((o[1].x + o[1].y) * h, (o[1].y - o[1].x) * h);
        cx<T> o2 = rot90<T, INV>(o[2]);
        cx<T> o3 = INV ? mk<T>((-o[3].x - o[3].y) * h, (o[3].x - o[3].y) * h) : mk<T>((o[3].y - o[3].x) * h, (-o[3].x - o[3].y) * h);
        v[0] = e[0] + o[0];
        v[4] = e[0] - o[0];
        v[1] = e[1] + o1;
        v[5] = e[1] - o1;
        v[2] = e[2] + o2;
        v[6] = e[2] - o2;
        v[3] = e[3] + o3;
        v[7] = e[3] - o3;
    }
};

template <typename T, bool INV>
struct Dft<T, 16, INV> {
    static __device__ __forceinline__ void run(cx<T>* v) {
        cx<T> e[8], o[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            e[i] = v[2 * i];
            o[i] = v[2 * i + 1];
        }
        Dft<T, 8, INV>::run(e);
        Dft<T, 8, INV>::run(o);
        // w16^k = cos(pi k/8) -+ i sin(pi k/8)
        const T c[8] = {(T)1.0, (T)0.92387953251128675613, (T)0.70710678118654752440, (T)0.38268343236508977173,
                        (T)0.0, (T)-0.38268343236508977173, (T)-0.70710678118654752440, (T)-0.92387953251128675613};
        const T s[8] = {(T)0.0, (T)0.38268343236508977173, (T)0.70710678118654752440, (T)0.92387953251128675613,
                        (T)1.0, (T)0.92387953251128675613, (T)0.70710678118654752440, (T)0.38268343236508977173};
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            cx<T> w = mk<T>(c[k], INV ? s[k] : -s[k]);
            cx<T> t = (k == 0) ? o[0] : o[k] * w;
            v[k] = e[k] + t;
            v[k + 8] = e[k] - t;
        }
    }
};

template <typename T, bool INV>
struct Dft<T, 3, INV> {
    static __device__ __forceinline__ void run(cx<T>* v) {
        const T hs = (T)0.86602540378443864676;
        cx<T> t1 = v[1] + v[2];
        cx<T> t2 = mk<T>(v[0].x - (T)0.5 * t1.x, v[0].y - (T)0.5 * t1.y);
        cx<T> t3 = rot90<T, INV>((v[1] - v[2]) * hs);
        v[0] = v[0] + t1;
        v[1] = t2 + t3;
        v[2] = t2 - t3;
    }
};

template <typename T, bool INV>
struct Dft<T, 5, INV> {
    static __device__ __forceinline__ void run(cx<T>* v) {
        const T c1 = (T)0.30901699437494742410, c2 = (T)-0.80901699437494742410;
        const T s1 = (T)0.95105651629515357212, s2 = (T)0.58778525229247312917;
        cx<T> a1 = v[1] + v[4], a2 = v[2] + v[3], b1 = v[1] - v[4], b2 = v[2] - v[3];
        cx<T> p1 = mk<T>(v[0].x + c1 * a1.x + c2 * a2.x, v[0].y + c1 * a1.y + c2 * a2.y);
        cx<T> p2 = mk<T>(v[0].x + c2 * a1.x + c1 * a2.x, v[0].y + c2 * a1.y + c1 * a2.y);
        cx<T> q1 = rot90<T, INV>(mk<T>(s1 * b1.x + s2 * b2.x, s1 * b1.y + s2 * b2.y));
        cx<T> q2 = rot90<T, INV>(mk<T>(s2 * b1.x - s1 * b2.x, s2 * b1.y - s1 * b2.y));
        v[0] = v[0] + a1 + a2;
        v[1] = p1 + q1;
        v[4] = p1 - q1;
        v[2] = p2 + q2;
        v[3] = p2 - q2;
    }
};

// Odd prime radix (7, 11, 13): direct evaluation with exact table twiddles w^k = c_k - i s_k, k = 1 .. (R - 1) / 2,
// in the paired form
//     out[q], out[R - q] = v0 + sum_a c_(a q) (v[a] + v[R - a])  -+  i sum_a s_(a q) (v[a] - v[R - a]),   a = 1 .. (R - 1) / 2
// (the inverse flips the sign of the second sum): real coefficients, half the products of the plain double loop, and --
// what matters more -- (R - 1) / 2 table reads and about 3 R live complex values.  The plain loop read a twiddle per
// (a, q); the compiler hoisted all of them, and every kernel that contains the run-time-radix transform then kept
// 1 - 3 KB per lane in scratch memory, whatever radices its launch actually ran.
// tw is the length-n table exp(-2 pi i e / n) of the enclosing transform, n % R == 0.
template <typename T, int R, bool INV>
__device__ __forceinline__ void dft_prime(cx<T>* v, const Tw<T>& tw, int n) {
    constexpr int HALF = (R - 1) / 2;
    const int step = n / R;
    T c[HALF + 1], sn[HALF + 1];
#pragma unroll
    for (int k = 1; k <= HALF; ++k) {
        const cx<T> w = tw[k * step];
        c[k] = w.x;
        sn[k] = -w.y;      // the table holds exp(-i ...): s_k = sin(2 pi k / R) > 0
    }
    cx<T> sa[HALF + 1], da[HALF + 1];
    cx<T> sum0 = v[0];
#pragma unroll
    for (int a = 1; a <= HALF; ++a) {
        sa[a] = v[a] + v[R - a];
        da[a] = v[a] - v[R - a];
        sum0 = sum0 + sa[a];
    }
    const cx<T> v0 = v[0];
    v[0] = sum0;
#pragma unroll
    for (int q = 1; q <= HALF; ++q) {
        cx<T> A = v0, B = mk<T>((T)0, (T)0);
#pragma unroll
        for (int a = 1; a <= HALF; ++a) {
            const int k = (a * q) % R;                       // compile-time after unrolling
            const int kk = k <= HALF ? k : R - k;            // c_(R - k) = c_k, s_(R - k) = -s_k
            A.x += c[kk] * sa[a].x;
            A.y += c[kk] * sa[a].y;
            if (k <= HALF) {
                B.x += sn[kk] * da[a].x;
                B.y += sn[kk] * da[a].y;
            } else {
                B.x -= sn[kk] * da[a].x;
                B.y -= sn[kk] * da[a].y;
            }
        }
        // forward: out[q] = A - i B, out[R - q] = A + i B; inverse: the other way round
        const cx<T> iB = mk<T>(-B.y, B.x);
        v[q] = INV ? A + iB : A - iB;
        v[R - q] = INV ? A - iB : A + iB;
    }
}

// One butterfly of pass s on column c: elements base + q*m, q = 0..R-1.
template <typename T, int R, bool INV>
__device__ __forceinline__ void butterfly(cx<T>* lds, const Tw<T>& tw, int n, int base, int m, int jj, int twmul, int cshift, int c) {
    cx<T> v[R];
    int idx[R];
#pragma unroll
    for (int q = 0; q < R; ++q) {
        idx[q] = SW(((base + q * m) << cshift) + c);
        v[q] = lds[idx[q]];
    }
    if (INV) {
        if (m > 1) {
            const int e = jj * twmul;
#pragma unroll
            for (int q = 1; q < R; ++q) v[q] = mulc(v[q], tw[q * e]);
        }
    }
    if constexpr (R == 7 || R == 11 || R == 13)
        dft_prime<T, R, INV>(v, tw, n);
    else
        Dft<T, R, INV>::run(v);
    if (!INV) {
        if (m > 1) {
            const int e = jj * twmul;
#pragma unroll
            for (int q = 1; q < R; ++q) v[q] = v[q] * tw[q * e];
        }
    }
#pragma unroll
    for (int q = 0; q < R; ++q) lds[idx[q]] = v[q];
}

// Pass s with a radix known only at run time (prime factors above 13): one thread per OUTPUT element, direct
// evaluation of its r-term sum from LDS.  A butterfly's outputs overwrite its own inputs only, so the pass goes in
// rounds over whole butterflies: compute up to KREG outputs per thread into registers, barrier, store, barrier.
// Every thread of the workgroup must call it (it contains barriers).  A round is at least one whole butterfly, so
// the launch must provide KREG * nt >= r threads-times-registers (the host's fft_min_threads() sizes workgroups for
// it): with fewer, outputs tid + i * nt >= KREG * nt of a butterfly would never be computed.  KREG = 2 serves radices up
// to 2 x 1024 threads; beyond that (up to 8 x 1024) eight outputs per thread, and float sums are accumulated in double
// (thousands of terms: a float accumulator alone would use up the 1e-5 the iterates are held to).
template <typename T, bool INV, int KREG>
__device__ __forceinline__ void prime_pass_k(cx<T>* lds, int narr, int arr_stride, const Tw<T>& tw, const FftDesc& d, int s, int cshift, int tid, int nt) {
    using A = typename std::conditional<KREG == 2, T, double>::type;
    const int r = d.radix[s], m = d.m[s], twmul = d.twmul[s], step = d.n / r;
    const int per = (d.n / r) << cshift;
    const int total = per * narr;
    const int per_round = max(1, (nt * KREG) / r);
    for (int b0 = 0; b0 < total; b0 += per_round) {
        const int nout = min(per_round, total - b0) * r;
        cx<T> acc[KREG];
        int where[KREG];
#pragma unroll
        for (int i = 0; i < KREG; ++i) {
            const int o = tid + i * nt;
            where[i] = -1;
            if (o < nout) {
                const int w = b0 + o / r, q = o % r;
                const int a = w / per, ww = w - a * per;
                const int c = ww & ((1 << cshift) - 1), bf = ww >> cshift;
                const int blk = bf / m, jj = bf - blk * m;
                const int base = blk * (m * r) + jj, e = jj * twmul;
                const cx<T>* L = lds + a * arr_stride;
                A sx = (A)0, sy = (A)0;
                int aq = 0;                              // (a2 * q) mod r
                for (int a2 = 0; a2 < r; ++a2) {
                    cx<T> v = L[SW(((base + a2 * m) << cshift) + c)];
                    if (INV && m > 1 && a2 > 0) v = mulc(v, tw[a2 * e]);
                    const cx<T> wq = tw[aq * step];
                    const cx<T> t = INV ? mulc(v, wq) : v * wq;
                    sx += (A)t.x;
                    sy += (A)t.y;
                    aq += q;
                    if (aq >= r) aq -= r;
                }
                cx<T> sum = mk<T>((T)sx, (T)sy);
                if (!INV && m > 1 && q > 0) sum = sum * tw[q * e];
                acc[i] = sum;
                where[i] = a * arr_stride + SW(((base + q * m) << cshift) + c);
            }
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < KREG; ++i)
            if (where[i] >= 0) lds[where[i]] = acc[i];
        __syncthreads();
    }
}
template <typename T, bool INV>
__device__ __forceinline__ void prime_pass(cx<T>* lds, int narr, int arr_stride, const Tw<T>& tw, const FftDesc& d, int s, int cshift, int tid, int nt) {
    if (d.radix[s] > 2 * nt) prime_pass_k<T, INV, 8>(lds, narr, arr_stride, tw, d, s, cshift, tid, nt);   // uniform over the workgroup
    else prime_pass_k<T, INV, 2>(lds, narr, arr_stride, tw, d, s, cshift, tid, nt);
}

// One full pass over `narr` arrays of n * C elements each (array a starts at lds + a*arr_stride).
template <typename T, bool INV>
__device__ __forceinline__ void fft_pass(cx<T>* lds, int narr, int arr_stride, const Tw<T>& tw, const FftDesc& d, int s, int cshift, int tid, int nt) {
    const int r = d.radix[s], m = d.m[s], ms = d.mshift[s], twmul = d.twmul[s];
    if (__builtin_expect(r > 16, 0)) {       // uniform over the workgroup; rare: keep it out of the hot path's layout
        prime_pass<T, INV>(lds, narr, arr_stride, tw, d, s, cshift, tid, nt);
        return;
    }
    const int per = (d.n / r) << cshift;
    const int total = per * narr;
    for (int w = tid; w < total; w += nt) {
        int a = 0, ww = w;
        if (narr == 2) {                 // (the two half-spectra of the column kernels: no division)
            a = w >= per ? 1 : 0;
            ww = w - (a ? per : 0);
        } else if (narr > 2) {
            a = w / per;
            ww = w - a * per;
        }
        const int c = ww & ((1 << cshift) - 1);
        const int bf = ww >> cshift;
        int blk, jj;
        if (ms >= 0) {
            blk = bf >> ms;
            jj = bf & (m - 1);
        } else {
            blk = fast_div(bf, d.mmagic[s]);
            jj = bf - blk * m;
        }
        const int base = blk * (m * r) + jj;
        cx<T>* L = lds + a * arr_stride;
        switch (r) {
            case 2: butterfly<T, 2, INV>(L, tw, d.n, base, m, jj, twmul, cshift, c); break;
            case 3: butterfly<T, 3, INV>(L, tw, d.n, base, m, jj, twmul, cshift, c); break;
            case 4: butterfly<T, 4, INV>(L, tw, d.n, base, m, jj, twmul, cshift, c); break;
            case 5: butterfly<T, 5, INV>(L, tw, d.n, base, m, jj, twmul, cshift, c); break;
            case 7: butterfly<T, 7, INV>(L, tw, d.n, base, m, jj, twmul, cshift, c); break;
            case 8: butterfly<T, 8, INV>(L, tw, d.n, base, m, jj, twmul, cshift, c); break;
            case 11: butterfly<T, 11, INV>(L, tw, d.n, base, m, jj, twmul, cshift, c); break;
            case 13: butterfly<T, 13, INV>(L, tw, d.n, base, m, jj, twmul, cshift, c); break;
            default: butterfly<T, 16, INV>(L, tw, d.n, base, m, jj, twmul, cshift, c); break;
        }
    }
}

// Forward: natural -> digit-reversed.  Caller guarantees the input is visible (barrier before).
// Ends with a barrier.
template <typename T>
__device__ __forceinline__ void fft_forward(cx<T>* lds, int narr, int arr_stride, const Tw<T>& tw, const FftDesc& d, int cshift, int tid, int nt) {
    for (int s = 0; s < d.npass; ++s) {
        fft_pass<T, false>(lds, narr, arr_stride, tw, d, s, cshift, tid, nt);
        __syncthreads();
    }
}
// Inverse (unnormalised): digit-reversed -> natural.  Ends with a barrier.
template <typename T>
__device__ __forceinline__ void fft_inverse(cx<T>* lds, int narr, int arr_stride, const Tw<T>& tw, const FftDesc& d, int cshift, int tid, int nt) {
    for (int s = d.npass - 1; s >= 0; --s) {
        fft_pass<T, true>(lds, narr, arr_stride, tw, d, s, cshift, tid, nt);
        __syncthreads();
    }
}

// ======================================================================================
// Compile-time specialised transforms: lengths 2^a 3^b 5^c 7^d 11^e 13^f.  For powers of two the radix schedule is
// the host's factorize() (so positions agree with the generic code and with stored multipliers); every radix, stride
// and shift is a constant: straight-line code, a fraction of the generic kernels' instruction footprint (which
// overflows the instruction cache) and none of their index arithmetic.
// ======================================================================================
constexpr int ilog2_c(int n) { return n <= 1 ? 0 : 1 + ilog2_c(n >> 1); }
constexpr int pow2_part(int n) {
    int p = 1;
    while (n > 1 && n % 2 == 0) {
        n /= 2;
        p *= 2;
    }
    return p;
}
constexpr int kOddRadices[5] = {3, 5, 7, 11, 13};   // passes after the power-of-two ones, in the order of the host's factorize()
constexpr int odd_npass(int n) {
    int c = 0;
    n /= pow2_part(n);
    for (int i = 0; i < 5; ++i)
        while (n % kOddRadices[i] == 0) {
            n /= kOddRadices[i];
            ++c;
        }
    return c;
}
constexpr int odd_radix(int n, int j) {
    n /= pow2_part(n);
    for (int i = 0; i < 5; ++i)
        while (n % kOddRadices[i] == 0) {
            if (j == 0) return kOddRadices[i];
            --j;
            n /= kOddRadices[i];
        }
    return 1;
}
// Two radix schedules of the power-of-two part 2^t: SCH = 16 (radix-16 first; identical to the host's factorize(),
// used wherever the digit-reversed order is shared with stored multipliers, i.e. the column transforms) and
// SCH = 8 (radix-8 first: one butterfly per thread per pass with N/8 threads; row transforms only,
// whose digit-reversed order never leaves the workgroup).
template <int SCH>
constexpr int p2_npass(int t) {
    if (t == 0) return 0;
    if (SCH == 16) {
        const int a = t / 4, r = t % 4;
        return r == 0 ? a : (r == 3 ? a + 1 : (a >= 1 ? a + 1 : 1));
    }
    const int a = t / 3, r = t % 3;
    return r == 0 ? a : (r == 2 ? a + 1 : (a >= 1 ? a + 1 : 1));
}
template <int SCH>
constexpr int p2_radix(int t, int s) {
    if (SCH == 16) {
        const int a = t / 4, r = t % 4;
        if (r == 0) return 16;
        if (r == 3) return s < a ? 16 : 8;
        if (r == 2) return a >= 1 ? (s < a - 1 ? 16 : 8) : 4;
        return a >= 1 ? (s < a - 1 ? 16 : (s == a - 1 ? 8 : 4)) : 2;
    }
    const int a = t / 3, r = t % 3;
    if (r == 0) return 8;
    if (r == 2) return s < a ? 8 : 4;          // 8 ... 8 4
    return a >= 1 ? (s < a - 1 ? 8 : 4) : 2;   // 8 ... 8 4 4   (2^1: radix 2)
}
template <int SCH>
constexpr int fixed_npass(int n) { return p2_npass<SCH>(ilog2_c(pow2_part(n))) + odd_npass(n); }
template <int SCH>
constexpr int fixed_radix(int n, int s) {
    const int t = ilog2_c(pow2_part(n)), np = p2_npass<SCH>(t);
    return s < np ? p2_radix<SCH>(t, s) : odd_radix(n, s - np);
}
template <int SCH>
constexpr int fixed_m(int n, int s) {  // n / (radix[0] * ... * radix[s])
    int m = n;
    for (int i = 0; i <= s; ++i) m /= fixed_radix<SCH>(n, i);
    return m;
}
static_assert(fixed_npass<8>(1920) == 5 && fixed_radix<8>(1920, 0) == 8 && fixed_radix<8>(1920, 3) == 3 && fixed_radix<8>(1920, 4) == 5 && fixed_m<8>(1920, 4) == 1,
              "1920 = 8 x 4 x 4 x 3 x 5");
static_assert(fixed_npass<16>(1920) == 4 && fixed_radix<16>(1920, 0) == 16 && fixed_radix<16>(1920, 1) == 8 && fixed_radix<16>(1920, 2) == 3 && fixed_radix<16>(1920, 3) == 5,
              "1920 = 16 x 8 x 3 x 5, the host's schedule");
static_assert(fixed_npass<16>(4096) == 3 && fixed_radix<16>(4096, 2) == 16 && fixed_npass<8>(512) == 3 && fixed_radix<16>(32768, 3) == 8, "powers of two as before");

template <int N, int SCH>
__device__ __forceinline__ int pos_of_freq_fixed(int k) {
    int pos = 0;
#pragma unroll
    for (int s = 0; s < fixed_npass<SCH>(N); ++s) {
        const int r = fixed_radix<SCH>(N, s);
        pos += (k % r) * fixed_m<SCH>(N, s);      // (constants: a mask and a shift for the power-of-two radices)
        k /= r;
    }
    return pos;
}

template <typename T, int N, int SCH, int S, bool INV>
__device__ __forceinline__ void fixed_pass(cx<T>* lds, int narr, int arr_stride, const Tw<T>& tw, int cshift, int tid, int nt) {
    constexpr int R = fixed_radix<SCH>(N, S), M = fixed_m<SCH>(N, S), TWMUL = N / (M * R);
    const int per = (N / R) << cshift;
    const int total = per * narr;
    for (int w = tid; w < total; w += nt) {
        int a = 0, ww = w;
        if (narr == 2) {                 // (the two half-spectra of the column kernels: no division)
            a = w >= per ? 1 : 0;
            ww = w - (a ? per : 0);
        } else if (narr > 2) {
            a = w / per;
            ww = w - a * per;
        }
        const int c = ww & ((1 << cshift) - 1);
        const int bf = ww >> cshift;
        const int blk = bf / M, jj = bf % M;  // M is a power of two: shifts
        butterfly<T, R, INV>(lds + a * arr_stride, tw, N, blk * (M * R) + jj, M, jj, TWMUL, cshift, c);
    }
}

// Two consecutive passes in one: the R1 x R2 elements base + a M1 + b M2 (M1 = R2 M2) go through pass S (R2 butterflies
// of radix R1 over a) and pass S + 1 (R1 butterflies of radix R2 over b) in registers -- the same butterflies, twiddles
// and positions as the two passes one after the other, without the trip through the LDS and the barrier between them.
template <typename T, int N, int SCH, int S, bool INV>
__device__ __forceinline__ void fixed_pass2(cx<T>* lds, int narr, int arr_stride, const Tw<T>& tw, int cshift, int tid, int nt) {
    constexpr int R1 = fixed_radix<SCH>(N, S), R2 = fixed_radix<SCH>(N, S + 1), M1 = fixed_m<SCH>(N, S), M2 = fixed_m<SCH>(N, S + 1);
    constexpr int TW1 = N / (M1 * R1), TW2 = N / (M2 * R2);
    static_assert(M1 == R2 * M2, "consecutive passes");
    const int per = (N / (R1 * R2)) << cshift;
    const int total = per * narr;
    for (int w = tid; w < total; w += nt) {
        int arrn = 0, ww = w;
        if (narr == 2) {
            arrn = w >= per ? 1 : 0;
            ww = w - (arrn ? per : 0);
        } else if (narr > 2) {
            arrn = w / per;
            ww = w - arrn * per;
        }
        const int c = ww & ((1 << cshift) - 1);
        const int bf = ww >> cshift;
        const int blk = bf / M2, jj2 = bf % M2;
        const int base = blk * (M1 * R1) + jj2;
        cx<T>* L = lds + arrn * arr_stride;
        cx<T> v[R1 * R2];
        int idx[R1 * R2];
#pragma unroll
        for (int a = 0; a < R1; ++a)
#pragma unroll
            for (int b = 0; b < R2; ++b) {
                idx[a * R2 + b] = SW(((base + a * M1 + b * M2) << cshift) + c);
                v[a * R2 + b] = L[idx[a * R2 + b]];
            }
        auto outer = [&]() {     // pass S: radix R1 over a, for every b
#pragma unroll
            for (int b = 0; b < R2; ++b) {
                cx<T> t[R1];
#pragma unroll
                for (int a = 0; a < R1; ++a) t[a] = v[a * R2 + b];
                const int e = (b * M2 + jj2) * TW1;
                if (INV) {
#pragma unroll
                    for (int q = 1; q < R1; ++q) t[q] = mulc(t[q], tw[q * e]);
                }
                Dft<T, R1, INV>::run(t);
                if (!INV) {
#pragma unroll
                    for (int q = 1; q < R1; ++q) t[q] = t[q] * tw[q * e];
                }
#pragma unroll
                for (int a = 0; a < R1; ++a) v[a * R2 + b] = t[a];
            }
        };
        auto inner = [&]() {     // pass S + 1: radix R2 over b, for every a
#pragma unroll
            for (int a = 0; a < R1; ++a) {
                cx<T> t[R2];
#pragma unroll
                for (int b = 0; b < R2; ++b) t[b] = v[a * R2 + b];
                const int e = jj2 * TW2;
                if (INV && M2 > 1) {
#pragma unroll
                    for (int q = 1; q < R2; ++q) t[q] = mulc(t[q], tw[q * e]);
                }
                Dft<T, R2, INV>::run(t);
                if (!INV && M2 > 1) {
#pragma unroll
                    for (int q = 1; q < R2; ++q) t[q] = t[q] * tw[q * e];
                }
#pragma unroll
                for (int b = 0; b < R2; ++b) v[a * R2 + b] = t[b];
            }
        };
        if (INV) {
            inner();
            outer();
        } else {
            outer();
            inner();
        }
#pragma unroll
        for (int i = 0; i < R1 * R2; ++i) L[idx[i]] = v[i];
    }
}

// Which passes go in pairs: left to right, a pass joins the next when the product of their radices is at most the
// caller's limit FUSE and neither is a table-driven prime radix (7, 11, 13).  The column kernels ask for kFuseMax<T>
// (float: 16 values in registers -- 24 or 25 cost the 128-register column kernels 0.3 KB of scratch; double: none): one 1080 x 1080 image, column launch 34 -> 30 us, sixteen 1920 x 1080
// images 0.39 -> 0.33 ms.  The row kernels ask for none: their transforms run one radix-8 butterfly per thread, a pair
// leaves half of the threads idle and, at 128 registers, 0.25 KB of scratch (1920^2 row launch 81 -> 108 us).
template <typename T>
constexpr int kFuseMax = sizeof(T) == 4 ? 16 : 0;
template <int SCH>
constexpr bool fuse_ok(int n, int s, int lim) {
    if (s + 1 >= fixed_npass<SCH>(n)) return false;
    const int r1 = fixed_radix<SCH>(n, s), r2 = fixed_radix<SCH>(n, s + 1);
    return r1 * r2 <= lim && r1 != 7 && r1 != 11 && r1 != 13 && r2 != 7 && r2 != 11 && r2 != 13;
}
// 0: pass s stands alone; 1: s is the first of a pair; 2: s is the second of a pair
template <int SCH>
constexpr int fuse_role(int n, int s, int lim) {
    int i = 0;
    while (i < fixed_npass<SCH>(n)) {
        const bool pair = fuse_ok<SCH>(n, i, lim);
        if (i == s) return pair ? 1 : 0;
        if (pair && i + 1 == s) return 2;
        i += pair ? 2 : 1;
    }
    return 0;
}
static_assert(fuse_role<8>(1920, 0, 16) == 0 && fuse_role<8>(1920, 1, 16) == 1 && fuse_role<8>(1920, 2, 16) == 2 && fuse_role<8>(1920, 3, 16) == 1 &&
              fuse_role<8>(1920, 4, 16) == 2, "1920 = 8 x (4 x 4) x (3 x 5)");
static_assert(fuse_role<16>(1080, 0, 16) == 0 && fuse_role<16>(1080, 1, 16) == 1 && fuse_role<16>(1080, 3, 16) == 1, "1080 = 8 x (3 x 3) x (3 x 5)");

template <typename T, int N, int SCH, int FUSE, int S = 0>
__device__ __forceinline__ void fft_forward_fixed(cx<T>* lds, int narr, int arr_stride, const Tw<T>& tw, int cshift, int tid, int nt) {
    if constexpr (S < fixed_npass<SCH>(N)) {
        constexpr int role = fuse_role<SCH>(N, S, FUSE);
        if constexpr (role == 1) fixed_pass2<T, N, SCH, S, false>(lds, narr, arr_stride, tw, cshift, tid, nt);
        else fixed_pass<T, N, SCH, S, false>(lds, narr, arr_stride, tw, cshift, tid, nt);
        __syncthreads();
        fft_forward_fixed<T, N, SCH, FUSE, S + (role == 1 ? 2 : 1)>(lds, narr, arr_stride, tw, cshift, tid, nt);
    }
}
template <typename T, int N, int SCH, int FUSE, int S = fixed_npass<SCH>(N) - 1>
__device__ __forceinline__ void fft_inverse_fixed(cx<T>* lds, int narr, int arr_stride, const Tw<T>& tw, int cshift, int tid, int nt) {
    if constexpr (S >= 0) {
        constexpr int role = fuse_role<SCH>(N, S, FUSE);
        if constexpr (role == 2) fixed_pass2<T, N, SCH, S - 1, true>(lds, narr, arr_stride, tw, cshift, tid, nt);
        else fixed_pass<T, N, SCH, S, true>(lds, narr, arr_stride, tw, cshift, tid, nt);
        __syncthreads();
        fft_inverse_fixed<T, N, SCH, FUSE, S - (role == 2 ? 2 : 1)>(lds, narr, arr_stride, tw, cshift, tid, nt);
    }
}

// Dispatchers: NF > 0 selects the specialised code (schedule SCH), NF == 0 the generic run-time schedule.
template <typename T, int NF, int SCH = 16, int FUSE = 0>
__device__ __forceinline__ void fft_fwd(cx<T>* lds, int narr, int arr_stride, const Tw<T>& tw, const FftDesc& d, int cshift, int tid, int nt) {
    if constexpr (NF > 0) fft_forward_fixed<T, NF, SCH, FUSE>(lds, narr, arr_stride, tw, cshift, tid, nt);
    else fft_forward<T>(lds, narr, arr_stride, tw, d, cshift, tid, nt);
}
template <typename T, int NF, int SCH = 16, int FUSE = 0>
__device__ __forceinline__ void fft_inv(cx<T>* lds, int narr, int arr_stride, const Tw<T>& tw, const FftDesc& d, int cshift, int tid, int nt) {
    if constexpr (NF > 0) fft_inverse_fixed<T, NF, SCH, FUSE>(lds, narr, arr_stride, tw, cshift, tid, nt);
    else fft_inverse<T>(lds, narr, arr_stride, tw, d, cshift, tid, nt);
}
template <int NF, int SCH = 16>
__device__ __forceinline__ int pos_freq(const FftDesc& d, int k) {
    if constexpr (NF > 0) return pos_of_freq_fixed<NF, SCH>(k);
    else return pos_of_freq(d, k);
}

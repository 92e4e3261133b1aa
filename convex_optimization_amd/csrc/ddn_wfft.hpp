// Length-4096 complex transform held entirely in the registers of ONE wavefront.
//
// Lane l, register r holds element 64*r + l, on input and on output (both directions), so a
// wave-wide load or store of register r touches 64 consecutive elements.  The transform is the
// four-step 64 x 64 decomposition: a 64-point transform per lane over its own registers (two
// radix-8 levels, every twiddle a compile-time constant), a per-lane twiddle w_4096^(lane*k1), ONE
// transposition through a wave-private LDS tile (real parts, then imaginary parts, through the
// same 64 x 65 float tile) and a second 64-point transform per lane.  No workgroup barrier is
// involved: the tile is private to the wave and LDS executes a wave's accesses in order.
#pragma once
#include "ddn_fft.hpp"

namespace wfft {

// cos(2 pi m / 64), m = 0..16, exact quarter wave; everything else by symmetry.
__host__ __device__ constexpr double qcos(int m) {
    constexpr double q[17] = {1.0,
                              0.99518472667219688624,
                              0.98078528040323044913,
                              0.95694033573220886494,
                              0.92387953251128675613,
                              0.88192126434835502971,
                              0.83146961230254523708,
                              0.77301045336273696081,
                              0.70710678118654752440,
                              0.63439328416364549822,
                              0.55557023301960222474,
                              0.47139673682599764856,
                              0.38268343236508977173,
                              0.29028467725446236764,
                              0.19509032201612826785,
                              0.09801714032956060199,
                              0.0};
    return q[m];
}
__host__ __device__ constexpr double cos64(int m) {
    m &= 63;
    if (m > 32) m = 64 - m;
    return m <= 16 ? qcos(m) : -qcos(32 - m);
}
__host__ __device__ constexpr double sin64(int m) { return cos64(m - 16); }

// v * exp(-+ 2 pi i m / 64) with m a compile-time constant (trivial cases cost no multiply)
template <bool INV, int M>
__device__ __forceinline__ cx<float> mul_w64(cx<float> v) {
    constexpr int m = M & 63;
    if constexpr (m == 0) return v;
    else if constexpr (m == 16) return rot90<float, INV>(v);
    else if constexpr (m == 32) return mk<float>(-v.x, -v.y);
    else if constexpr (m == 48) return rot90<float, !INV>(v);
    else {
        constexpr float c = (float)cos64(m);
        constexpr float s = (float)(INV ? sin64(m) : -sin64(m));
        return mk<float>(v.x * c - v.y * s, v.x * s + v.y * c);
    }
}

template <bool INV, int C, int P = 1>
__device__ __forceinline__ void twiddle_row(cx<float> (&t)[8]) {
    if constexpr (P < 8) {
        t[P] = mul_w64<INV, C * P>(t[P]);
        twiddle_row<INV, C, P + 1>(t);
    }
}

// stage A of the 64-point transform for residue C: elements C + 8a -> y[8C + p] (in place in slot
// order: slot 8a + C receives index p = a), followed by the w_64^(C p) twiddle.
template <bool INV, int C = 0>
__device__ __forceinline__ void fft64_stage_a(cx<float> (&d)[64]) {
    if constexpr (C < 8) {
        cx<float> t[8];
#pragma unroll
        for (int a = 0; a < 8; ++a) t[a] = d[C + 8 * a];
        Dft<float, 8, INV>::run(t);
        twiddle_row<INV, C>(t);
#pragma unroll
        for (int p = 0; p < 8; ++p) d[C + 8 * p] = t[p];
        fft64_stage_a<INV, C + 1>(d);
    }
}

// 64-point transform of d[0..63] (natural order in, natural order out).
template <bool INV>
__device__ __forceinline__ void fft64(cx<float> (&d)[64]) {
    fft64_stage_a<INV>(d);
    // now d[c + 8p] = y[c][p]; stage B: for each p transform over c, output q -> X[p + 8q]
    cx<float> o[64];
#pragma unroll
    for (int p = 0; p < 8; ++p) {
        cx<float> t[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) t[c] = d[c + 8 * p];
        Dft<float, 8, INV>::run(t);
#pragma unroll
        for (int q = 0; q < 8; ++q) o[p + 8 * q] = t[q];
    }
#pragma unroll
    for (int k = 0; k < 64; ++k) d[k] = o[k];
}

// Per-lane twiddle factors P[c] = w_4096^(lane*c), Q[a] = w_4096^(8*lane*a), c, a = 0..7 (forward sign;
// the inverse conjugates).  Global table layout: [lane][16] complex (P then Q), 8 KB, cache resident.
// They are fetched at the start of every transform and are dead after the twiddle step: holding
// them across a whole row would cost 32 registers the per-pixel phase needs.
struct LaneTw {
    cx<float> P[8], Q[8];
    __device__ __forceinline__ void load(const cx<float>* tab, int lane) {
        unsigned off = (unsigned)lane * 16u;
        asm volatile("" : "+v"(off));  // opaque: keeps the loads inside the row loop
        const cx<float>* t = tab + off;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            P[i] = t[i];
            Q[i] = t[8 + i];
        }
    }
};

#define WFFT_TILE_FLOATS (64 * 65)

__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Transposition through the wave's LDS tile: on entry lane l holds d[k] = T(k, l); on exit lane l
// holds d[n] = T(l, n).
__device__ __forceinline__ void transpose64(cx<float> (&d)[64], float* tile, int lane) {
    // the lane's tile addresses are rebuilt here on every call (opaque lane index): as loop invariants
    // they would be hoisted out of the row loop, sixteen registers (one per 1 KB of paired-write
    // offset range) held through the per-pixel phase that has none to spare
    unsigned lo = (unsigned)lane;
    asm volatile("" : "+v"(lo));
    float* wr = tile + lo;
    const float* rd = tile + lo * 65u;
#pragma unroll
    for (int k = 0; k < 64; ++k) wr[k * 65] = d[k].x;
    wave_sync();
#pragma unroll
    for (int k = 0; k < 64; ++k) d[k].x = rd[k];
    wave_sync();
#pragma unroll
    for (int k = 0; k < 64; ++k) wr[k * 65] = d[k].y;
    wave_sync();
#pragma unroll
    for (int k = 0; k < 64; ++k) d[k].y = rd[k];
    wave_sync();
}

// ---- packed-math form -----------------------------------------------------------------------------
// The same transform on two-float vectors: a complex add is one v_pk_add_f32, a complex multiply two
// packed instructions, about 28 % fewer vector instructions than the scalar form above and no more
// registers.  (The vectors are written out by hand: letting the SLP vectoriser pair scalars instead
// fragments the register file and spills.)
typedef float v2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2 mk2(float a, float b) { v2 r; r.x = a; r.y = b; return r; }
template <bool INV> __device__ __forceinline__ v2 rot90_v2(v2 a) { return INV ? mk2(-a.y, a.x) : mk2(a.y, -a.x); }
// a * (c + i s)
__device__ __forceinline__ v2 cmul(v2 a, v2 w) {
    const v2 wr = mk2(-w.y, w.x);
    return __builtin_elementwise_fma(a.yy, wr, a.xx * w);
}
__device__ __forceinline__ v2 cmulc(v2 a, v2 w) {  // a * conj(w)
    const v2 wr = mk2(w.y, w.x);
    const v2 wc = mk2(w.x, -w.y);
    return __builtin_elementwise_fma(a.yy, wr, a.xx * wc);
}
template <bool INV> __device__ __forceinline__ void dft4(v2 (&v)[4]) {
    v2 t0 = v[0] + v[2], t1 = v[0] - v[2], t2 = v[1] + v[3], t3 = rot90_v2<INV>(v[1] - v[3]);
    v[0] = t0 + t2; v[1] = t1 + t3; v[2] = t0 - t2; v[3] = t1 - t3;
}
template <bool INV> __device__ __forceinline__ void dft8(v2 (&v)[8]) {
    v2 e[4] = {v[0], v[2], v[4], v[6]};
    v2 o[4] = {v[1], v[3], v[5], v[7]};
    dft4<INV>(e); dft4<INV>(o);
    const float h = 0.70710678118654752440f;
    // w8^1 = (1 -+ i)/sqrt2 ; w8^3 = (-1 -+ i)/sqrt2
    v2 o1 = (o[1] + rot90_v2<INV>(o[1])) * h;
    v2 o2 = rot90_v2<INV>(o[2]);
    v2 o3 = (rot90_v2<INV>(o[3]) - o[3]) * h;
    v[0] = e[0] + o[0]; v[4] = e[0] - o[0];
    v[1] = e[1] + o1;   v[5] = e[1] - o1;
    v[2] = e[2] + o2;   v[6] = e[2] - o2;
    v[3] = e[3] + o3;   v[7] = e[3] - o3;
}
template <bool INV, int M> __device__ __forceinline__ v2 mul_w64_v2(v2 v) {
    constexpr int m = M & 63;
    if constexpr (m == 0) return v;
    else if constexpr (m == 16) return rot90_v2<INV>(v);
    else if constexpr (m == 32) return -v;
    else if constexpr (m == 48) return rot90_v2<!INV>(v);
    else {
        constexpr float c = (float)cos64(m);
        constexpr float s = (float)(INV ? sin64(m) : -sin64(m));
        return __builtin_elementwise_fma(v.yy, mk2(-s, c), v.xx * mk2(c, s));
    }
}
template <bool INV, int C, int P = 1> __device__ __forceinline__ void twiddle_row_v2(v2 (&t)[8]) {
    if constexpr (P < 8) { t[P] = mul_w64_v2<INV, C * P>(t[P]); twiddle_row_v2<INV, C, P + 1>(t); }
}
template <bool INV, int C = 0> __device__ __forceinline__ void stage_a(v2 (&d)[64]) {
    if constexpr (C < 8) {
        v2 t[8];
#pragma unroll
        for (int a = 0; a < 8; ++a) t[a] = d[C + 8 * a];
        dft8<INV>(t);
        twiddle_row_v2<INV, C>(t);
#pragma unroll
        for (int p = 0; p < 8; ++p) d[C + 8 * p] = t[p];
        stage_a<INV, C + 1>(d);
    }
}
template <bool INV> __device__ __forceinline__ void fft64_v2(v2 (&d)[64]) {
    stage_a<INV>(d);
    v2 o[64];
#pragma unroll
    for (int p = 0; p < 8; ++p) {
        v2 t[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) t[c] = d[c + 8 * p];
        dft8<INV>(t);
#pragma unroll
        for (int q = 0; q < 8; ++q) o[p + 8 * q] = t[q];
    }
#pragma unroll
    for (int k = 0; k < 64; ++k) d[k] = o[k];
}
__device__ __forceinline__ void transpose64_v2(v2 (&d)[64], float* tile, int lane) {
    float* wr = tile + lane;
    const float* rd = tile + lane * 65;
#pragma unroll
    for (int k = 0; k < 64; ++k) wr[k * 65] = d[k].x;
    wave_sync();
#pragma unroll
    for (int k = 0; k < 64; ++k) d[k].x = rd[k];
    wave_sync();
#pragma unroll
    for (int k = 0; k < 64; ++k) wr[k * 65] = d[k].y;
    wave_sync();
#pragma unroll
    for (int k = 0; k < 64; ++k) d[k].y = rd[k];
    wave_sync();
}
template <bool INV> __device__ __forceinline__ void fft4096_v2(v2 (&d)[64], const cx<float>* twtab, float* tile, int lane) {
    v2 P[8], Q[8];
    {
        unsigned off = (unsigned)lane * 16u;
        asm volatile("" : "+v"(off));
        const v2* t = reinterpret_cast<const v2*>(twtab) + off;
#pragma unroll
        for (int i = 0; i < 8; ++i) { P[i] = t[i]; Q[i] = t[8 + i]; }
    }
    __builtin_amdgcn_sched_barrier(0);
    fft64_v2<INV>(d);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int a = 0; a < 8; ++a) {
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            if (a == 0 && c == 0) continue;
            const v2 w = (a == 0) ? P[c] : ((c == 0) ? Q[a] : cmul(P[c], Q[a]));
            d[8 * a + c] = INV ? cmulc(d[8 * a + c], w) : cmul(d[8 * a + c], w);
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    transpose64_v2(d, tile, lane);
    __builtin_amdgcn_sched_barrier(0);
    fft64_v2<INV>(d);
    __builtin_amdgcn_sched_barrier(0);
}

// Unnormalised length-4096 transform, exp(-2 pi i nk/4096) forward, exp(+...) inverse.
#ifdef DDN_WFFT_PACKED  // off: in the fused kernels the pair alignment costs more registers than the packed ops save time
template <bool INV>
__device__ __forceinline__ void fft4096(cx<float> (&d)[64], const cx<float>* twtab, float* tile, int lane) {
    v2 q[64];
#pragma unroll
    for (int r = 0; r < 64; ++r) q[r] = mk2(d[r].x, d[r].y);
    fft4096_v2<INV>(q, twtab, tile, lane);
#pragma unroll
    for (int r = 0; r < 64; ++r) d[r] = mk<float>(q[r].x, q[r].y);
}
#else
template <bool INV>
__device__ __forceinline__ void fft4096(cx<float> (&d)[64], const cx<float>* twtab, float* tile, int lane) {
    LaneTw tw;
    tw.load(twtab, lane);
    __builtin_amdgcn_sched_barrier(0);
    fft64<INV>(d);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int a = 0; a < 8; ++a) {
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            if (a == 0 && c == 0) continue;
            const cx<float> w = (a == 0) ? tw.P[c] : ((c == 0) ? tw.Q[a] : tw.P[c] * tw.Q[a]);
            d[8 * a + c] = INV ? mulc(d[8 * a + c], w) : d[8 * a + c] * w;
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    transpose64(d, tile, lane);
    __builtin_amdgcn_sched_barrier(0);
    fft64<INV>(d);
    __builtin_amdgcn_sched_barrier(0);
}
#endif


// ---- shorter sequences: N = 64 R, R in {8, 16, 32}, M = 64 / R sequences per wavefront ------------------
// Pixel-side layout ("A"): register q R + j of lane l holds element 64 j + l of sequence q -- for rows
// stored back to back that is element 64 (register) + lane of one contiguous 4096-element segment,
// exactly the layout of the 4096-point case.  Frequency-side layout ("B"): lane q R + k1, register k2
// holds frequency k1 + R k2 of sequence q (R consecutive frequencies of a sequence per register).
// forward: A -> B  (R-point transforms in registers, twiddle w_N^(lane k1), the 64 x 64 transposition,
//                   64-point transforms);   inverse: B -> A, the same steps backwards.
// Per-lane twiddles: table [lane][R] = w_N^(lane k1), forward sign.
template <int R, bool INV>
__device__ __forceinline__ void small_dft(cx<float>* v) {
    if constexpr (R == 32) {
        cx<float> e[16], o[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            e[i] = v[2 * i];
            o[i] = v[2 * i + 1];
        }
        Dft<float, 16, INV>::run(e);
        Dft<float, 16, INV>::run(o);
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            // w32^k = w64^(2k)
            const cx<float> w = mk<float>((float)cos64(2 * k), (float)(INV ? sin64(2 * k) : -sin64(2 * k)));
            const cx<float> t = (k == 0) ? o[0] : o[k] * w;
            v[k] = e[k] + t;
            v[k + 16] = e[k] - t;
        }
    } else {
        Dft<float, R, INV>::run(v);
    }
}

template <int R, bool INV>
__device__ __forceinline__ void fftN(cx<float> (&d)[64], const cx<float>* twtab, float* tile, int lane) {
    constexpr int M = 64 / R;
    cx<float> tw[R];
    {
        unsigned off = (unsigned)lane * (unsigned)R;
        asm volatile("" : "+v"(off));
        const cx<float>* t = twtab + off;
#pragma unroll
        for (int i = 0; i < R; ++i) tw[i] = t[i];
    }
    __builtin_amdgcn_sched_barrier(0);
    if (!INV) {
#pragma unroll
        for (int q = 0; q < M; ++q) {
            cx<float> v[R];
#pragma unroll
            for (int j = 0; j < R; ++j) v[j] = d[q * R + j];
            small_dft<R, false>(v);
#pragma unroll
            for (int k = 0; k < R; ++k) d[q * R + k] = (k == 0) ? v[0] : v[k] * tw[k];
        }
        __builtin_amdgcn_sched_barrier(0);
        transpose64(d, tile, lane);
        __builtin_amdgcn_sched_barrier(0);
        fft64<false>(d);
    } else {
        fft64<true>(d);
        __builtin_amdgcn_sched_barrier(0);
        transpose64(d, tile, lane);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < M; ++q) {
            cx<float> v[R];
#pragma unroll
            for (int k = 0; k < R; ++k) v[k] = (k == 0) ? d[q * R] : mulc(d[q * R + k], tw[k]);
            small_dft<R, true>(v);
#pragma unroll
            for (int j = 0; j < R; ++j) d[q * R + j] = v[j];
        }
    }
    __builtin_amdgcn_sched_barrier(0);
}

}  // namespace wfft

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

// Unnormalised length-4096 transform, exp(-2 pi i nk/4096) forward, exp(+...) inverse.
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

}  // namespace wfft

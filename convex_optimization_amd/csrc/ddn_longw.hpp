// Frequency-side kernels of the split row transform (ddn_long.hpp) for M = 4096, float: the length-M transforms
// run in the registers of one wavefront each (ddn_wfft.hpp) instead of through workgroup-wide LDS passes.
//
// One workgroup = the R/2 + 1 residues q of one image row, one wavefront per residue.  Register r of lane l holds
// term kappa = 64 r + l of its residue, i.e. half-spectrum element R kappa + q: the terms of the residues
// interleave in memory, so the workgroup moves the row's half-spectrum in contiguous pieces through an LDS
// staging buffer (Geo below) and every wave takes its own terms from there.  The Y side ([array][row][q][m]) is
// contiguous per wave.
#pragma once
#include "ddn_kernels.hpp"
#include "ddn_long_args.hpp"
#include "ddn_wfft.hpp"

namespace dlongw {

__device__ __forceinline__ void sched_fence() { __builtin_amdgcn_sched_barrier(0); }
// materialise the sequence here: keeps the optimiser from sinking a transform's last butterfly level into the code
// that follows (and its inputs alive, spilled, there)
__device__ __forceinline__ void pin(cx<float> (&d)[64]) {
#pragma unroll
    for (int r = 0; r < 64; ++r) asm volatile("" : "+v"(d[r].x), "+v"(d[r].y));
}
// element at a wave-uniform base plus a per-lane 32-bit BYTE offset: the form that maps onto the scalar-base
// addressing mode, so that the 64 accesses of a lane need no per-access address registers
// The base is made opaque (an empty asm on a scalar register, global address space kept): it blocks the
// re-association base + lane offset that would be hoisted out of the row loop as 64 vector addresses and spilled.
template <typename P>
__device__ __forceinline__ P& at_b(P* base, unsigned byte_off) {
    auto g = (__attribute__((address_space(1))) P*)base;
    asm volatile("" : "+s"(g));
    return *reinterpret_cast<P*>(reinterpret_cast<char*>(const_cast<typename std::remove_const<P>::type*>((P*)g)) + byte_off);
}

__device__ __forceinline__ size_t y_index(const LongArgs<float>& a, int arr, size_t row, int q) {
    return ((((size_t)arr * ((size_t)a.c.batch * a.c.H) + row) * (a.R / 2 + 1)) + q) * (size_t)4096;
}

// Geometry shared by both kernels.  With e = R lane + q, register r < 32 of wave q holds element 64 R r + e of the
// row's half-spectrum; register r >= 32 holds the conjugate of element W - (64 R r + e) = 64 R (63 - r) + (64 R - e).
// The half-spectrum is moved in four rounds of E = 512 R consecutive elements (plus element E of the round, the
// first of the next) through a staging buffer: global accesses are whole 128-byte runs of the spectral tiles, and
// a wave picks or drops its 8 direct and 8 mirrored registers of the round from LDS.  Staging slot of element i of
// a round: i + (i >> 5), which spreads a wave's stride-R accesses over all banks.
template <int R>
struct Geo {
    static constexpr int NQ = R / 2 + 1, NT = 64 * NQ;
    static constexpr int E = 512 * R;                        // elements per round (8 blocks of 64 R)
    static constexpr int STG = E + 1 + ((E + 1) >> 5) + 1;   // staging slots
    static constexpr int NLD = (E + 1 + NT - 1) / NT;        // elements per thread per round
    static constexpr size_t lds_bytes() { return (size_t)NQ * WFFT_TILE_FLOATS * sizeof(float) + (size_t)STG * sizeof(cx<float>); }
};
__device__ __forceinline__ int slot(int i) { return i + (i >> 5); }
// slot(c + x) for c a multiple of 32: a compile-time displacement on top of slot(x)
constexpr int slot_of_base(int c) { return c + (c >> 5); }

// Per-thread pieces of the cooperative accesses (thread `tid` handles elements tid + n NT of a round; NT and E are
// multiples of the spectral tile width, so the tile-layout offset of element E t + n NT + tid is a uniform part
// plus a per-thread 32-bit part).
template <int R>
struct Coop {
    unsigned g;      // byte offset of element `tid` within the row's half-spectrum (tile layout)
    unsigned tb;     // byte offset of element `tid` of a row table
    int st;          // staging slot of element `tid`
    size_t ts;       // array elements per spectral tile
    int cs;
    __device__ __forceinline__ Coop(const SpecLayout& lay, int tid) {
        cs = lay.cs;
        ts = (size_t)lay.H << cs;
        g = (unsigned)(((size_t)(tid >> cs) * ts + (size_t)(tid & ((1 << cs) - 1))) * sizeof(cx<float>));
        tb = (unsigned)tid * (unsigned)sizeof(cx<float>);
        st = slot(tid);
    }
    // uniform element offset of element c (a multiple of the tile width) of the row
    __device__ __forceinline__ size_t ub(int c) const { return (size_t)(c >> cs) * ts; }
};

// half-spectra -> Y.  array 0: S0 (u); array 1: S1 (K u); array 2: eD * S0 (D u).  narr = 1: S0 only.
template <int R>
__global__ void __launch_bounds__(64 * (R / 2 + 1), 2) k_long_inv_w(LongArgs<float> a) {
    using G = Geo<R>;
    extern __shared__ __align__(16) float wtiles[];
    const int tid0 = threadIdx.x, q = __builtin_amdgcn_readfirstlane(tid0 >> 6);   // q: wave-uniform, per-wave bases stay scalar
    float* const tile = wtiles + (size_t)q * WFFT_TILE_FLOATS;
    cx<float>* const stg = reinterpret_cast<cx<float>*>(wtiles + (size_t)G::NQ * WFFT_TILE_FLOATS);
    const size_t nrows = (size_t)a.c.batch * a.c.H;
    const int W = a.c.W;
    for (size_t row = blockIdx.x; row < nrows; row += gridDim.x) {
        const int plane = (int)(row / a.c.H);
        if (a.c.stop && a.c.stop[plane] >= 0) continue;
        const size_t rbase = spec_row(a.c.lay, plane, (int)(row - (size_t)plane * a.c.H)).base;
        for (int ai = 0; ai < a.narr; ++ai) {
            const int arr = ai == 0 ? 0 : 3 - ai;           // 0, 2, 1: the two readers of S0 back to back
            const cx<float>* const src = (arr == 1 ? a.c.S1 : a.c.S0) + rbase;
            // the thread index is made opaque here: everything derived from it is then recomputed per array instead
            // of being hoisted out of the row loop as a hundred address registers
            int tid = tid0;
            asm volatile("" : "+v"(tid));
            const int lane = tid & 63, e = R * lane + q;
            const Coop<R> co(a.c.lay, tid);
            const cx<float>* const sd = stg + slot(e);          // direct terms of this lane
            const cx<float>* const sm = stg + slot(64 * R - e); // mirrored terms
            cx<float> d[64];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                cx<float> v[G::NLD];
#pragma unroll
                for (int n = 0; n < G::NLD; ++n)
                    if (n + 1 < G::NLD || tid + n * G::NT <= G::E) v[n] = at_b(src + co.ub(G::E * t + n * G::NT), co.g);
                if (arr == 2) {
#pragma unroll
                    for (int n = 0; n < G::NLD; ++n)
                        if (n + 1 < G::NLD || tid + n * G::NT <= G::E) v[n] = v[n] * at_b(a.ed + (G::E * t + n * G::NT), co.tb);
                }
                if (t == 0 && tid == 0) v[0].y = 0.0f;                                     // mean of a real row
                if (t == 3 && tid + (G::NLD - 1) * G::NT == G::E) v[G::NLD - 1].y = 0.0f;  // its term W/2
#pragma unroll
                for (int n = 0; n < G::NLD; ++n)
                    if (n + 1 < G::NLD || tid + n * G::NT <= G::E) stg[co.st + slot_of_base(n * G::NT)] = v[n];
                __syncthreads();
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    d[8 * t + j] = sd[slot_of_base(64 * R * j)];
                    const cx<float> m = sm[slot_of_base(64 * R * j)];
                    d[63 - 8 * t - j] = mk<float>(m.x, -m.y);
                }
                __syncthreads();
            }
            pin(d);
            wfft::fft4096<true>(d, a.wtw, tile, lane);
            cx<float>* const y = a.Y + y_index(a, arr, row, q);
            const unsigned ln = (unsigned)lane * (unsigned)sizeof(cx<float>);
#pragma unroll
            for (int r = 0; r < 64; ++r) at_b(y + 64 * r, ln) = d[r];
            sched_fence();
        }
    }
    (void)W;
}

// One array of one row: Y -> its length-4096 transform -> the row's half-spectrum `dst` (already at the row).
// MODE 0 store, 1 store times conj(eD), 2 add to what an earlier pass stored.  A residue 0 < q < R/2 also provides
// the terms of residue R - q (the conjugates of its upper half); residues 0 and R/2 own both halves of theirs.
template <int R, int MODE>
__device__ __forceinline__ void fwd_pass(const LongArgs<float>& a, const cx<float>* y, cx<float>* dst, float* tile, cx<float>* stg, int tid0, int q) {
    using G = Geo<R>;
    int tid = tid0;                                              // opaque: see k_long_inv_w
    asm volatile("" : "+v"(tid));
    const int lane = tid & 63, e = R * lane + q;
    const Coop<R> co(a.c.lay, tid);
    const unsigned ln = (unsigned)lane * (unsigned)sizeof(cx<float>);
    const bool own_both = q == 0 || 2 * q == R;
    cx<float>* const sd = stg + slot(e);
    cx<float>* const sm = stg + slot(64 * R - e);
    cx<float> d[64];
#pragma unroll
    for (int r = 0; r < 64; ++r) d[r] = at_b(y + 64 * r, ln);
    pin(d);
    wfft::fft4096<false>(d, a.wtw, tile, lane);
    pin(d);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            sd[slot_of_base(64 * R * j)] = d[8 * t + j];
            if (!own_both) sm[slot_of_base(64 * R * j)] = mk<float>(d[63 - 8 * t - j].x, -d[63 - 8 * t - j].y);
        }
        if (t == 3 && e == 0) stg[slot_of_base(G::E)] = d[32];   // element W/2 itself
        __syncthreads();
        // element E of a round is element 0 of the next, except in the last round
        const bool tail = tid + (G::NLD - 1) * G::NT < G::E || (t == 3 && tid + (G::NLD - 1) * G::NT == G::E);
        cx<float> v[G::NLD];
#pragma unroll
        for (int n = 0; n < G::NLD; ++n)
            if (n + 1 < G::NLD || tail) v[n] = stg[co.st + slot_of_base(n * G::NT)];
        if (MODE == 1) {
#pragma unroll
            for (int n = 0; n < G::NLD; ++n)
                if (n + 1 < G::NLD || tail) v[n] = mulc(v[n], at_b(a.ed + (G::E * t + n * G::NT), co.tb));
        }
        if (MODE == 2) {
#pragma unroll
            for (int n = 0; n < G::NLD; ++n)
                if (n + 1 < G::NLD || tail) v[n] = v[n] + at_b(dst + co.ub(G::E * t + n * G::NT), co.g);
        }
#pragma unroll
        for (int n = 0; n < G::NLD; ++n)
            if (n + 1 < G::NLD || tail) at_b(dst + co.ub(G::E * t + n * G::NT), co.g) = v[n];
        __syncthreads();
    }
}

// Y -> half-spectra.  narr = 3: S0 <- conj(eD) F(array 2) + F(array 0), S1 <- F(array 1).  narr = 1: S0 <- F(array 0).
template <int R>
__global__ void __launch_bounds__(64 * (R / 2 + 1), 2) k_long_fwd_w(LongArgs<float> a) {
    using G = Geo<R>;
    extern __shared__ __align__(16) float wtiles[];
    const int tid = threadIdx.x, q = __builtin_amdgcn_readfirstlane(tid >> 6);   // q: wave-uniform, per-wave bases stay scalar
    float* const tile = wtiles + (size_t)q * WFFT_TILE_FLOATS;
    cx<float>* const stg = reinterpret_cast<cx<float>*>(wtiles + (size_t)G::NQ * WFFT_TILE_FLOATS);
    const size_t nrows = (size_t)a.c.batch * a.c.H;
    for (size_t row = blockIdx.x; row < nrows; row += gridDim.x) {
        const int plane = (int)(row / a.c.H);
        if (a.c.stop && a.c.stop[plane] >= 0) continue;
        const size_t rbase = spec_row(a.c.lay, plane, (int)(row - (size_t)plane * a.c.H)).base;
        if (a.narr == 3) {
            fwd_pass<R, 1>(a, a.Y + y_index(a, 2, row, q), a.c.S0 + rbase, tile, stg, tid, q);
            fwd_pass<R, 2>(a, a.Y + y_index(a, 0, row, q), a.c.S0 + rbase, tile, stg, tid, q);
            fwd_pass<R, 0>(a, a.Y + y_index(a, 1, row, q), a.c.S1 + rbase, tile, stg, tid, q);
        } else {
            fwd_pass<R, 0>(a, a.Y + y_index(a, 0, row, q), a.c.S0 + rbase, tile, stg, tid, q);
        }
    }
}

}  // namespace dlongw

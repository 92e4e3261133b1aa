// Column stage for H = 64 R, R in {8, 16, 32} (512, 1024, 2048), float32, in ONE pass: one wavefront
// per M = 64 / R ADJACENT spectral columns, all of them held in registers (wfft::fftN<R>).
//
// The M columns are 8 M contiguous bytes of every row of a tile (tiles are 8 or 16 columns wide), so a lane loads and
// stores its rows directly (16-byte accesses, no staging through LDS): register q R + j of lane l
// holds row 64 j + l of column q.  Resolvent mode: a workgroup is the wave pair (S0, S1) of one
// column group, which swaps register halves through the two transposition tiles for the pointwise
// stage (as in ddn_wcol.hpp); multiplier mode: a workgroup is one wave.  With 33 KB of LDS per pair
// four pairs share a CU and progress independently: one pair's loads overlap another's transforms.
// The column groups of a tile get block indices 8 apart: same XCD, same L2, dispatched
// together, so a 128-byte line is fetched from HBM once.
// Multipliers: the column-major natural-order copy of eK (ddn_wcol_args.hpp); G is recomputed.
#pragma once
#include "ddn_wcol_args.hpp"
#include "ddn_wrow.hpp"

namespace wcol {

using wrow::opaque;
using wrow::pin;
using wrow::sched_fence;

typedef float f4 __attribute__((ext_vector_type(4)));

// Frequency-side layout: lane q R + k1, register k2 holds frequency k1 + R k2 of column q; its
// multiplier sits at (eKc + R k2)[eoff], eoff = column * pitch + k1 (per lane).
// U = (P + conj(e) Q) g, V = e U; wave HH finishes registers 32 HH .. 32 HH + 31 (see res2_pair).
template <int HH, int R>
__device__ __forceinline__ void res2_pairN(cx<float> (&d)[64], const cx<float>* eKc, unsigned eoff, float cK, float cDd, float gscale, cx<float>* mine, cx<float>* theirs, int lane) {
    constexpr int RB = 32 * HH, OB = 32 * (1 - HH);
    cx<float> e[32];
#pragma unroll
    for (int rr = 0; rr < 32; ++rr) e[rr] = (eKc + R * (RB + rr))[eoff];
#pragma unroll
    for (int rr = 0; rr < 32; ++rr) mine[rr * 64 + lane] = d[OB + rr];
    __syncthreads();
#pragma unroll
    for (int rr = 0; rr < 32; ++rr) {
        const int r = RB + rr;
        const cx<float> X = theirs[rr * 64 + lane];
        const cx<float> P = HH == 0 ? d[r] : X, Q = HH == 0 ? X : d[r];
        const float g = gscale / (1.f + cK * (e[rr].x * e[rr].x + e[rr].y * e[rr].y) + cDd);
        const cx<float> U = (P + mulc(Q, e[rr])) * g;
        const cx<float> V = e[rr] * U;
        d[r] = HH == 0 ? U : V;
        theirs[rr * 64 + lane] = HH == 0 ? V : U;
        asm volatile("" : "+v"(d[r].x), "+v"(d[r].y));
    }
    __syncthreads();
#pragma unroll
    for (int rr = 0; rr < 32; ++rr) d[OB + rr] = mine[rr * 64 + lane];
}

// NP: column groups (wave pairs in resolvent mode) per workgroup, adjacent groups of one tile: the
// more of a tile's 128-byte lines one CU uses, the fewer times they cross from the L2.
// ILV (resolvent mode): S0 holds the (S0, S1) pairs interleaved element by element, so the M columns of
// both spectra are 16 M contiguous bytes of a row.  Each wave of a pair then loads whole pairs for
// half of the rows (wave 0 the first R / 2 slabs of 64 rows, wave 1 the rest) and the two swap the
// halves they loaded for each other through their tiles; the store mirrors it.
template <int MODE, int R, int NP, bool ILV = false, bool RB = false>
__global__ void __launch_bounds__((MODE == COL_RES2 ? 128 : 64) * NP, 2) k_wave_colN(Args a) {
    static_assert(!ILV || MODE == COL_RES2, "interleaved spectra: resolvent mode only");
    static_assert(!RB || !ILV, "row blocks: separate spectra");
    constexpr int M = 64 / R;       // columns per wave
    constexpr int LM = M == 2 ? 1 : (M == 4 ? 2 : 3);
    constexpr int NT = (MODE == COL_RES2 ? 128 : 64) * NP;
    constexpr int LP = NP == 1 ? 0 : (NP == 2 ? 1 : 2);
    const int cs = a.lay.cs, gs = cs - LM - LP;  // 2^gs workgroups per tile (tiles are at least NP M columns wide)
    extern __shared__ __align__(16) float tiles[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = wv >> LP, pr = wv & (NP - 1);  // h: 0 = S0, 1 = S1; pr: the workgroup's column group
    // block -> (plane, tile, group of the tile): the G groups of a tile are 8 block indices apart
    const int per_plane = (8 << gs) * ((a.lay.ntile + 7) / 8);
    const int plane = blockIdx.x / per_plane;
    const int b = blockIdx.x - plane * per_plane;
    const int x = b & 7, aa = b >> 3;
    const int tile = 8 * (aa >> gs) + x, sub = aa & ((1 << gs) - 1);
    if (a.lf.enabled && b == 0) loss_finalize(a.lf, plane, reinterpret_cast<double*>(tiles), tid, NT);
    if (tile >= a.lay.ntile) return;
    const int k2w = (tile << cs) + sub * (M * NP);  // first column of the workgroup
    if (k2w >= a.Wh) return;                        // nothing but padding columns
    const int k2g = k2w + pr * M;                   // first column of the wave's group
    if (a.lf.stop) {
        const int s = a.lf.stop[plane];
        if (s >= 0 && s < a.lf.iter) return;
    }
    __syncthreads();
    cx<float>* const Sarr = (MODE == COL_RES2 && h == 1) ? a.S1 : a.S0;
    const size_t gbase = opaque(((((size_t)plane * a.lay.ntile + tile) * a.lay.H) << cs) + (size_t)((sub * NP + pr) * M));
    float* const mytile = tiles + (size_t)wv * WFFT_TILE_FLOATS;
    const unsigned loff = (unsigned)lane << cs;  // row `lane` of a 64-row slab, in elements

    // columns past the half-spectrum are tile padding: loaded, transformed and stored like the others
    // (their values are never read by a row kernel), with the multipliers of the last real column
    cx<float> d[64];
    if (ILV) {
        cx<float>* const mine = reinterpret_cast<cx<float>*>(mytile);
        cx<float>* const theirs = reinterpret_cast<cx<float>*>(tiles + (size_t)(wv ^ NP) * WFFT_TILE_FLOATS);
        const f4* const pairs = reinterpret_cast<const f4*>(a.S0) + gbase;   // in pairs: same element offset
        constexpr int RH = R / 2;
        // own part into its final register, the partner's part into the register that will receive
        // the partner's contribution: wave 0 keeps S0, wave 1 keeps S1
#pragma unroll
        for (int jj = 0; jj < RH; ++jj) {
            const int j = h * RH + jj;                 // slab loaded by this wave (h is wave-uniform)
            const f4* const p = pairs + ((size_t)(64 * j) << cs);
#pragma unroll
            for (int q = 0; q < M; ++q) {
                const f4 v = (p + q)[loff];
                const cx<float> P = mk<float>(v.x, v.y), Q = mk<float>(v.z, v.w);
                // registers are indexed statically: jj and jj + RH; which of the two is the own slot depends on h
                d[q * R + jj] = P;                     // wave 0: own S0 at slab jj; wave 1: S0 of slab jj + RH, for the partner
                d[q * R + jj + RH] = Q;                // wave 0: S1 of slab jj, for the partner; wave 1: own S1 at slab jj + RH
            }
        }
        // send: wave 0 its S1 values (upper registers), wave 1 its S0 values (lower registers)
#pragma unroll
        for (int q = 0; q < M; ++q) {
#pragma unroll
            for (int jj = 0; jj < RH; ++jj) {
                const cx<float> lo = d[q * R + jj], hi = d[q * R + jj + RH];
                mine[(q * RH + jj) * 64 + lane] = h == 0 ? hi : lo;
            }
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < M; ++q) {
#pragma unroll
            for (int jj = 0; jj < RH; ++jj) {
                const cx<float> x = theirs[(q * RH + jj) * 64 + lane];
                const cx<float> lo = d[q * R + jj], hi = d[q * R + jj + RH];
                // wave 0 receives S0 of slab jj + RH (upper register); wave 1 receives S1 of slab jj (lower register)
                d[q * R + jj] = h == 0 ? lo : x;
                d[q * R + jj + RH] = h == 0 ? x : hi;
            }
        }
        __syncthreads();  // both tiles are transposition tiles again
    } else {
#pragma unroll
        for (int j = 0; j < R; ++j) {
            const f4* const p = reinterpret_cast<const f4*>(Sarr + gbase + ((size_t)(64 * j) << cs) + (RB ? (size_t)(j >> a.rb_sh6) * (size_t)a.rb_extra : (size_t)0));
#pragma unroll
            for (int q2 = 0; q2 < M / 2; ++q2) {
                const f4 v = (p + q2)[loff >> 1];
                d[(2 * q2) * R + j] = mk<float>(v.x, v.y);
                d[(2 * q2 + 1) * R + j] = mk<float>(v.z, v.w);
            }
        }
    }
    pin(d);
    wfft::fftN<R, false>(d, a.wtw, mytile, lane);
    pin(d);
    sched_fence();

    const int col = min(k2g + lane / R, a.Wh - 1);
    const unsigned eoff = (unsigned)col * (unsigned)kColPitch + (unsigned)(lane & (R - 1));
    if (MODE == COL_RES2) {
        const float cDd = a.cD * a.ed2[col];
        cx<float>* const mine = reinterpret_cast<cx<float>*>(mytile);
        cx<float>* const theirs = reinterpret_cast<cx<float>*>(tiles + (size_t)(wv ^ NP) * WFFT_TILE_FLOATS);
        if (h == 0) res2_pairN<0, R>(d, a.eKc, eoff, a.cK, cDd, a.gscale, mine, theirs, lane);
        else res2_pairN<1, R>(d, a.eKc, eoff, a.cK, cDd, a.gscale, mine, theirs, lane);
    } else {
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) {
            cx<float> e[32];
#pragma unroll
            for (int rr = 0; rr < 32; ++rr) e[rr] = (a.eKc + R * (32 * blk + rr))[eoff];
#pragma unroll
            for (int rr = 0; rr < 32; ++rr) {
                const int r = 32 * blk + rr;
                d[r] = (a.conj_mult ? mulc(d[r], e[rr]) : d[r] * e[rr]) * a.scale;
                asm volatile("" : "+v"(d[r].x), "+v"(d[r].y));
            }
        }
    }
    pin(d);
    sched_fence();
    // (after the pair's second barrier each wave touches only its own tile: no barrier needed here)
    wfft::fftN<R, true>(d, a.wtw, mytile, lane);
    pin(d);
    if (ILV) {
        cx<float>* const mine = reinterpret_cast<cx<float>*>(mytile);
        cx<float>* const theirs = reinterpret_cast<cx<float>*>(tiles + (size_t)(wv ^ NP) * WFFT_TILE_FLOATS);
        constexpr int RH = R / 2;
        // wave 0 (holding S0 of every slab) stores the pairs of slabs 0 .. RH - 1 and needs S1 there; it
        // hands S0 of the upper slabs to wave 1, which stores the pairs of slabs RH .. R - 1
#pragma unroll
        for (int q = 0; q < M; ++q) {
#pragma unroll
            for (int jj = 0; jj < RH; ++jj) {
                const cx<float> lo = d[q * R + jj], hi = d[q * R + jj + RH];
                mine[(q * RH + jj) * 64 + lane] = h == 0 ? hi : lo;
            }
        }
        __syncthreads();
        f4* const pairs = reinterpret_cast<f4*>(a.S0) + opaque(gbase);
#pragma unroll
        for (int jj = 0; jj < RH; ++jj) {
            const int j = h * RH + jj;
            f4* const p = pairs + ((size_t)(64 * j) << cs);
#pragma unroll
            for (int q = 0; q < M; ++q) {
                const cx<float> x = theirs[(q * RH + jj) * 64 + lane];
                const cx<float> lo = d[q * R + jj], hi = d[q * R + jj + RH];
                // wave 0: (own S0 of slab jj, received S1 of slab jj); wave 1: (received S0 of slab jj + RH, own S1)
                const cx<float> P = h == 0 ? lo : x, Q = h == 0 ? x : hi;
                f4 v;
                v.x = P.x; v.y = P.y; v.z = Q.x; v.w = Q.y;
                (p + q)[loff] = v;
            }
        }
    } else {
        cx<float>* const Sout = Sarr + opaque(gbase);
#pragma unroll
        for (int j = 0; j < R; ++j) {
            f4* const p = reinterpret_cast<f4*>(Sout + ((size_t)(64 * j) << cs) + (RB ? (size_t)(j >> a.rb_sh6) * (size_t)a.rb_extra : (size_t)0));
#pragma unroll
            for (int q2 = 0; q2 < M / 2; ++q2) {
                f4 v;
                v.x = d[(2 * q2) * R + j].x; v.y = d[(2 * q2) * R + j].y;
                v.z = d[(2 * q2 + 1) * R + j].x; v.w = d[(2 * q2 + 1) * R + j].y;
                (p + q2)[loff >> 1] = v;
            }
        }
    }
}

}  // namespace wcol

// Column stage for H = 64 R, R in {8, 16, 32} (512, 1024, 2048), float32, in ONE pass: one wavefront
// per M = 64 / R ADJACENT spectral columns, all of them held in registers (wfft::fftN<R>).
//
// The M columns are 8 M contiguous bytes of every row of a tile (tiles are 8 or 16 columns wide).  Register q R + j of
// lane l holds row 64 j + l of column q, but global memory is accessed row-major: a wave instruction covers 64 / NV
// rows x NV 16-byte pieces (NV = pieces of a row the wave owns) and the wave's transposition tile hands the pieces to
// the row owners (rows_to_owner / owner_to_rows).  With one row per lane -- 64 rows x 16 bytes per instruction, NV
// instructions per 16 NV-byte piece -- the same kernels moved 3.0 TB/s at most (a copy in that pattern: 3.0 TB/s;
// 16 rows x 64 bytes: 4.9 TB/s; tools/ubench_pattern.hip).  Resolvent mode: a workgroup is the wave pair (S0, S1) of one
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


// Row-major global access <-> row-owner registers, within one wavefront, through its transposition tile.
// rows_to_owner: on entry v[t] of lane (rs = lane / NV, q = lane % NV) holds piece q of row (64 / NV) t + rs of a slab of
// 64 rows; on exit v[q] of lane l holds piece q of row l.  owner_to_rows: the way back.  Staged rows are NV + 1 pieces
// apart (an odd pitch in 16-byte units: the owners' reads spread over all banks).
template <int NV>
__device__ __forceinline__ void rows_to_owner(f4 (&v)[NV], float* tile, int lane) {
    if constexpr (NV > 1) {
        constexpr int RPI = 64 / NV, PITCH = NV + 1;
        f4* const t4 = reinterpret_cast<f4*>(tile);
        const int rs = lane / NV, q = lane % NV;
#pragma unroll
        for (int t = 0; t < NV; ++t) t4[(RPI * t + rs) * PITCH + q] = v[t];
        wfft::wave_sync();
#pragma unroll
        for (int qq = 0; qq < NV; ++qq) v[qq] = t4[lane * PITCH + qq];
        wfft::wave_sync();
    }
}
template <int NV>
__device__ __forceinline__ void owner_to_rows(f4 (&v)[NV], float* tile, int lane) {
    if constexpr (NV > 1) {
        constexpr int RPI = 64 / NV, PITCH = NV + 1;
        f4* const t4 = reinterpret_cast<f4*>(tile);
        const int rs = lane / NV, q = lane % NV;
#pragma unroll
        for (int qq = 0; qq < NV; ++qq) t4[lane * PITCH + qq] = v[qq];
        wfft::wave_sync();
#pragma unroll
        for (int t = 0; t < NV; ++t) v[t] = t4[(RPI * t + rs) * PITCH + q];
        wfft::wave_sync();
    }
}
__device__ __forceinline__ f4 pack4(cx<float> a, cx<float> b) {
    f4 v;
    v.x = a.x; v.y = a.y; v.z = b.x; v.w = b.y;
    return v;
}

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
        const float g = gscale * __builtin_amdgcn_rcpf(1.f + cK * (e[rr].x * e[rr].x + e[rr].y * e[rr].y) + cDd);   // hardware reciprocal (1 ulp): a correctly rounded divide is ten instructions
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
        // every global load of the wave is issued here, row-major (a wave instruction: 64 / M rows x 16 M bytes); piece t of
        // slab jj waits in the registers of column t until its slab goes through the tile
        constexpr int RPI = 64 / M;
        const unsigned roff = ((unsigned)(lane / M) << cs) + (unsigned)(lane % M);
#pragma unroll
        for (int jj = 0; jj < RH; ++jj) {
            const int j = h * RH + jj;                 // slab loaded by this wave (h is wave-uniform)
            const f4* const p = pairs + ((size_t)(64 * j) << cs);
#pragma unroll
            for (int t = 0; t < M; ++t) {
                const f4 v = (p + ((size_t)(RPI * t) << cs))[roff];
                d[t * R + jj] = mk<float>(v.x, v.y);
                d[t * R + jj + RH] = mk<float>(v.z, v.w);
            }
        }
        sched_fence();
#pragma unroll
        for (int jj = 0; jj < RH; ++jj) {
            f4 v[M];
#pragma unroll
            for (int t = 0; t < M; ++t) v[t] = pack4(d[t * R + jj], d[t * R + jj + RH]);
            rows_to_owner<M>(v, mytile, lane);
#pragma unroll
            for (int q = 0; q < M; ++q) {
                // registers are indexed statically: jj and jj + RH; which of the two is the own slot depends on h
                d[q * R + jj] = mk<float>(v[q].x, v[q].y);        // wave 0: own S0 at slab jj; wave 1: S0 of slab jj + RH, for the partner
                d[q * R + jj + RH] = mk<float>(v[q].z, v[q].w);   // wave 0: S1 of slab jj, for the partner; wave 1: own S1 at slab jj + RH
            }
        }
        // (a wave stages in its own tile and sends through its own tile: no barrier between the two)
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
        constexpr int NV = M / 2, RPI = 64 / NV;        // 16-byte pieces (two columns) of a row
        const unsigned roff = ((unsigned)(lane / NV) << (cs - 1)) + (unsigned)(lane % NV);
#pragma unroll
        for (int j = 0; j < R; ++j) {
            const f4* const p = reinterpret_cast<const f4*>(Sarr + gbase + ((size_t)(64 * j) << cs) + (RB ? (size_t)(j >> a.rb_sh6) * (size_t)a.rb_extra : (size_t)0));
#pragma unroll
            for (int t = 0; t < NV; ++t) {
                const f4 v = (p + ((size_t)(RPI * t) << (cs - 1)))[roff];
                d[(2 * t) * R + j] = mk<float>(v.x, v.y);
                d[(2 * t + 1) * R + j] = mk<float>(v.z, v.w);
            }
        }
        if constexpr (NV > 1) {
            sched_fence();
#pragma unroll
            for (int j = 0; j < R; ++j) {
                f4 v[NV];
#pragma unroll
                for (int t = 0; t < NV; ++t) v[t] = pack4(d[(2 * t) * R + j], d[(2 * t + 1) * R + j]);
                rows_to_owner<NV>(v, mytile, lane);
#pragma unroll
                for (int q2 = 0; q2 < NV; ++q2) {
                    d[(2 * q2) * R + j] = mk<float>(v[q2].x, v[q2].y);
                    d[(2 * q2 + 1) * R + j] = mk<float>(v[q2].z, v[q2].w);
                }
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
        // the received halves go into the registers that held what was sent: every slab's pairs are then complete in
        // (d[q R + jj], d[q R + jj + RH]) and the partner's tile is free again
#pragma unroll
        for (int q = 0; q < M; ++q) {
#pragma unroll
            for (int jj = 0; jj < RH; ++jj) {
                const cx<float> x = theirs[(q * RH + jj) * 64 + lane];
                const cx<float> lo = d[q * R + jj], hi = d[q * R + jj + RH];
                // wave 0: (own S0 of slab jj, received S1 of slab jj); wave 1: (received S0 of slab jj + RH, own S1)
                d[q * R + jj] = h == 0 ? lo : x;
                d[q * R + jj + RH] = h == 0 ? x : hi;
            }
        }
        __syncthreads();  // both waves have read their partner's tile: the tiles are staging space from here on
        f4* const pairs = reinterpret_cast<f4*>(a.S0) + opaque(gbase);
        constexpr int RPI = 64 / M;
        int lane_s = lane;
        asm volatile("" : "+v"(lane_s));
        const unsigned roff = ((unsigned)(lane_s / M) << cs) + (unsigned)(lane_s % M);
#pragma unroll
        for (int jj = 0; jj < RH; ++jj) {
            const int j = h * RH + jj;
            f4* const p = pairs + ((size_t)(64 * j) << cs);
            f4 v[M];
#pragma unroll
            for (int q = 0; q < M; ++q) v[q] = pack4(d[q * R + jj], d[q * R + jj + RH]);
            owner_to_rows<M>(v, mytile, lane_s);
#pragma unroll
            for (int t = 0; t < M; ++t) (p + ((size_t)(RPI * t) << cs))[roff] = v[t];
        }
    } else {
        cx<float>* const Sout = Sarr + opaque(gbase);
        constexpr int NV = M / 2, RPI = 64 / NV;
        int lane_s = lane;
        asm volatile("" : "+v"(lane_s));
        const unsigned roff = ((unsigned)(lane_s / NV) << (cs - 1)) + (unsigned)(lane_s % NV);
#pragma unroll
        for (int j = 0; j < R; ++j) {
            f4* const p = reinterpret_cast<f4*>(Sout + ((size_t)(64 * j) << cs) + (RB ? (size_t)(j >> a.rb_sh6) * (size_t)a.rb_extra : (size_t)0));
            f4 v[NV];
#pragma unroll
            for (int q2 = 0; q2 < NV; ++q2) v[q2] = pack4(d[(2 * q2) * R + j], d[(2 * q2 + 1) * R + j]);
            owner_to_rows<NV>(v, mytile, lane_s);
#pragma unroll
            for (int t = 0; t < NV; ++t) (p + ((size_t)(RPI * t) << (cs - 1)))[roff] = v[t];
        }
    }
}

}  // namespace wcol

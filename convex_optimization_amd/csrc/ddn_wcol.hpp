// Column stage for H = 4096, float32, in ONE pass: one wavefront per spectral column, the column
// held in registers (wfft::fft4096), eight waves per workgroup.
//
// A workgroup owns a quarter tile -- four adjacent columns, 32 contiguous bytes of every row -- of
// both half-spectra (resolvent mode: waves 0-3 take the columns of S0, waves 4-7 the same columns
// of S1) or two quarter tiles of S0 (multiplier mode).  Global accesses are made row-major (a wave
// instruction covers 16 rows x 32 bytes) and redistributed to the column owners through LDS, 16
// row slabs per round; the transforms need no workgroup barrier.  The pointwise stage of the
// resolvent mode needs both spectra of a frequency in one lane: the S1 wave hands its spectrum to
// its S0 partner through its own (idle) transposition tile and gets e*U back the same way.
// The four quarter-tile workgroups of a tile are given block indices that differ by 8, so that
// they share an XCD (and its L2) and are dispatched together: each 128-byte line is then fetched
// from HBM once and written back whole.
// Multipliers are read from column-major natural-order copies (contiguous along the column).
#pragma once
#include "ddn_wcol_args.hpp"
#include "ddn_wrow.hpp"

namespace wcol {

using wrow::opaque;
using wrow::pin;
using wrow::sched_fence;


constexpr int kSlabs = 16;                 // row slabs (of 64 rows) staged per round
constexpr int kSlabPitch = 4 * 64 + 16;    // complex elements per staged slab (row pitch padded every 4 rows)
constexpr size_t kLdsBytes = 8 * WFFT_TILE_FLOATS * sizeof(float);

// Row-major global <-> column-owner registers for one quarter tile (4 columns), executed by the
// four waves wq = 0..3 of a half; wave wq ends up with column wq: lane l, register r = row 64 r + l.
// base: the array offset to (row 0, first column of the quarter); rpitch = elements per row (tile width).
__device__ __forceinline__ void load_quarter(cx<float> (&d)[64], const cx<float>* base, int rpitch, cx<float>* stg, int wq, int lane, bool active) {
    const int rl = 16 * wq + (lane >> 2), cl = lane & 3;
    const unsigned goff = (unsigned)(rl * rpitch + cl);
    const int wr = 4 * rl + cl + (rl >> 2);
    const int rd = 4 * lane + wq + (lane >> 2);
    // every global load of the column is issued before the first use: 64 x 512 bytes in flight per wave
#pragma unroll
    for (int r = 0; r < 64; ++r) d[r] = active ? (base + (size_t)(64 * r) * rpitch)[goff] : mk<float>(0.f, 0.f);
    sched_fence();
#pragma unroll
    for (int round = 0; round < 64 / kSlabs; ++round) {
#pragma unroll
        for (int s = 0; s < kSlabs; ++s) stg[s * kSlabPitch + wr] = d[round * kSlabs + s];
        __syncthreads();
#pragma unroll
        for (int s = 0; s < kSlabs; ++s) d[round * kSlabs + s] = stg[s * kSlabPitch + rd];
        __syncthreads();
    }
}
__device__ __forceinline__ void store_quarter(const cx<float> (&d)[64], cx<float>* base, int rpitch, cx<float>* stg, int wq, int lane, bool active, int ncols) {
    const int rl = 16 * wq + (lane >> 2), cl = lane & 3;
    const unsigned goff = (unsigned)(rl * rpitch + cl);
    const int wr = 4 * rl + cl + (rl >> 2);
    const int rd = 4 * lane + wq + (lane >> 2);
#pragma unroll
    for (int round = 0; round < 64 / kSlabs; ++round) {
#pragma unroll
        for (int s = 0; s < kSlabs; ++s) stg[s * kSlabPitch + rd] = d[round * kSlabs + s];
        __syncthreads();
        cx<float> tmp[kSlabs];
#pragma unroll
        for (int s = 0; s < kSlabs; ++s) tmp[s] = stg[s * kSlabPitch + wr];
        if (active && cl < ncols) {
#pragma unroll
            for (int s = 0; s < kSlabs; ++s) {
                const size_t row0 = (size_t)(64 * (round * kSlabs + s));
                (base + row0 * rpitch)[goff] = tmp[s];
            }
        }
        __syncthreads();
    }
}

// MODE: COL_RES2 or COL_MUL (ddn_kernels.hpp)
template <int MODE>
__global__ void __launch_bounds__(512, 2) k_wave_col(Args a) {
    extern __shared__ __align__(16) float tiles[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6), h = wv >> 2, wq = wv & 3;  // wave-uniform, in scalar registers
    constexpr int QPG = (MODE == COL_RES2) ? 4 : 2;  // workgroups per tile
    const int cs = a.lay.cs, C = 1 << cs;
    const int nq = (a.Wh + 3) >> 2;                  // quarter tiles with at least one valid column (tiles are 16 wide: 4 per tile)
    // block -> (plane, tile, part): the workgroups of a tile get indices 8 apart
    const int per_plane = 8 * QPG * ((a.lay.ntile + 7) / 8);
    const int plane = blockIdx.x / per_plane;
    const int b = blockIdx.x - plane * per_plane;
    const int x = b & 7, aa = b >> 3;
    const int tile = 8 * (aa / QPG) + x, part = aa % QPG;
    // quarter handled by this half of the workgroup: a part is one quarter in resolvent mode, two in multiplier mode
    const int quarter = 4 * tile + (MODE == COL_RES2 ? part : 2 * part + h);
    const bool grp_valid = tile < a.lay.ntile && 4 * tile + (MODE == COL_RES2 ? part : 2 * part) < nq;
    if (a.lf.enabled && tile == 0 && part == 0) loss_finalize(a.lf, plane, reinterpret_cast<double*>(tiles), tid, 512);
    if (!grp_valid) return;
    if (a.lf.stop) {
        const int s = a.lf.stop[plane];
        if (s >= 0 && s < a.lf.iter) return;
    }
    __syncthreads();
    const bool qvalid = quarter < nq;
    const int k2q = quarter << 2;                    // first column of the quarter
    const int ncols = qvalid ? min(4, a.Wh - k2q) : 0;
    const int k2 = k2q + wq;                         // this wave's column
    const bool cvalid = qvalid && k2 < a.Wh;
    cx<float>* const Sarr = (MODE == COL_RES2 && h == 1) ? a.S1 : a.S0;
    const size_t qbase = opaque(((((size_t)plane * a.lay.ntile + (k2q >> cs)) * a.lay.H) << cs) + (k2q & (C - 1)));
    cx<float>* const stg = reinterpret_cast<cx<float>*>(tiles + (size_t)(4 * h) * WFFT_TILE_FLOATS);
    float* const mytile = tiles + (size_t)wv * WFFT_TILE_FLOATS;

    cx<float> d[64];
    load_quarter(d, Sarr + qbase, C, stg, wq, lane, qvalid);
    pin(d);
    wfft::fft4096<false>(d, a.wtw, mytile, lane);
    pin(d);
    sched_fence();

    const size_t mbase = opaque((size_t)(cvalid ? k2 : 0) * 4096);
    const cx<float>* const ecol = a.eKc + mbase;
    const unsigned ul = (unsigned)lane;
    if (MODE == COL_RES2) {
        const float* const gcol = a.Gc + mbase;
        // the S1 wave's tile doubles as the exchange buffer of the pair (wq, wq + 4)
        cx<float>* const ex = reinterpret_cast<cx<float>*>(tiles + (size_t)(4 + wq) * WFFT_TILE_FLOATS);
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            // multipliers stream through a two-deep pipeline of 8-register chunks
            cx<float> e[2][8];
            float g[2][8];
            if (h == 0) {
#pragma unroll
                for (int rr = 0; rr < 8; ++rr) {
                    e[0][rr] = (ecol + 64 * (32 * half + rr))[ul];
                    g[0][rr] = (gcol + 64 * (32 * half + rr))[ul];
                }
            } else {
#pragma unroll
                for (int rr = 0; rr < 32; ++rr) ex[rr * 64 + lane] = d[32 * half + rr];
            }
            __syncthreads();
            if (h == 0) {
#pragma unroll
                for (int ch = 0; ch < 4; ++ch) {
                    if (ch + 1 < 4) {
#pragma unroll
                        for (int rr = 0; rr < 8; ++rr) {
                            e[(ch + 1) & 1][rr] = (ecol + 64 * (32 * half + 8 * (ch + 1) + rr))[ul];
                            g[(ch + 1) & 1][rr] = (gcol + 64 * (32 * half + 8 * (ch + 1) + rr))[ul];
                        }
                    }
#pragma unroll
                    for (int rr = 0; rr < 8; ++rr) {
                        const int r = 32 * half + 8 * ch + rr;
                        const cx<float> Q = ex[(8 * ch + rr) * 64 + lane];
                        const cx<float> U = (d[r] + mulc(Q, e[ch & 1][rr])) * g[ch & 1][rr];
                        d[r] = U;
                        ex[(8 * ch + rr) * 64 + lane] = e[ch & 1][rr] * U;
                        asm volatile("" : "+v"(d[r].x), "+v"(d[r].y));
                    }
                    sched_fence();
                }
            }
            __syncthreads();
            if (h == 1) {
#pragma unroll
                for (int rr = 0; rr < 32; ++rr) d[32 * half + rr] = ex[rr * 64 + lane];
            }
        }
    } else {
#pragma unroll
        for (int blk = 0; blk < 4; ++blk) {
            cx<float> e[16];
#pragma unroll
            for (int rr = 0; rr < 16; ++rr) e[rr] = (ecol + 64 * (16 * blk + rr))[ul];
#pragma unroll
            for (int rr = 0; rr < 16; ++rr) {
                const int r = 16 * blk + rr;
                d[r] = (a.conj_mult ? mulc(d[r], e[rr]) : d[r] * e[rr]) * a.scale;
            }
        }
    }
    pin(d);
    sched_fence();
    __syncthreads();  // exchange buffers are transposition tiles again from here on
    wfft::fft4096<true>(d, a.wtw, mytile, lane);
    pin(d);
    __syncthreads();  // every wave is done with its tile before the tiles become staging space
    // offsets recomputed here: keeping the 64 load addresses alive across the kernel would spill them
    store_quarter(d, Sarr + opaque(qbase), C, stg, wq, lane, qvalid, ncols);
}

}  // namespace wcol

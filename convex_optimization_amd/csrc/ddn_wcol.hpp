// Column stage for H = 4096, float32, in ONE pass: one wavefront per spectral column, the column
// held in registers (wfft::fft4096), eight waves per workgroup.
//
// A workgroup owns a quarter tile -- four adjacent columns, 32 contiguous bytes of every row -- of
// both half-spectra (resolvent mode, eight waves: waves 0-3 take the columns of S0, waves 4-7 the
// same columns of S1) or of S0 alone (multiplier mode, four waves, two workgroups per CU).  Global accesses are made row-major (a wave
// instruction covers 16 rows x 32 bytes) and redistributed to the column owners through LDS, 16
// row slabs per round; the transforms need no workgroup barrier.  The pointwise stage of the
// resolvent mode needs both spectra of a frequency in one lane: the S0 and S1 waves of a column
// swap half of their registers through their (idle) transposition tiles and each finishes one half.
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


constexpr int kSlabs = 16;                 // row slabs (of 64 rows) staged per round (32 with a thinner padding measured no faster)
constexpr int kSlabPitch = 4 * 64 + 16;    // complex elements per staged slab (row pitch padded every 4 rows)
constexpr size_t kLdsBytes = WFFT_TILE_FLOATS * sizeof(float);  // per wave
static_assert(kSlabs * kSlabPitch * 2 <= 4 * WFFT_TILE_FLOATS, "a round of staged slabs must fit the four tiles of a half");
constexpr size_t kLdsExtra = 2 * 64 * sizeof(cx<float>);          // lane-0 tables of the packed pair

// Row-major global <-> column-owner registers for one quarter tile (4 columns), executed by the
// four waves wq = 0..3 of a half; wave wq ends up with column wq: lane l, register r = row 64 r + l.
// base: the array offset to (row 0, first column of the quarter); rpitch = elements per row (tile width).
template <bool RB = false>
__device__ __forceinline__ void load_quarter(cx<float> (&d)[64], const cx<float>* base, int rpitch, cx<float>* stg, int wq, int lane, bool active, int dbg_mode = 0,
                                             int rbsh = 0, size_t rbx = 0) {
    const int rl = 16 * wq + (lane >> 2), cl = lane & 3;
    const unsigned goff = (unsigned)(rl * rpitch + cl);
    const int wr = 4 * rl + cl + (rl >> 2);
    const int rd = 4 * lane + wq + (lane >> 2);
    // every global load of the column is issued before the first use: 64 x 512 bytes in flight per wave
#pragma unroll
    for (int r = 0; r < 64; ++r)
        d[r] = active ? (base + (size_t)((dbg_mode & 4) ? 0 : ((dbg_mode & 8) ? 16 * r : 64 * r)) * rpitch + (RB ? (size_t)(r >> rbsh) * rbx : (size_t)0))[goff] : mk<float>(0.f, 0.f);
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
template <bool RB = false>
__device__ __forceinline__ void store_quarter(const cx<float> (&d)[64], cx<float>* base, int rpitch, cx<float>* stg, int wq, int lane, bool active, int ncols,
                                              int rbsh = 0, size_t rbx = 0) {
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
                (base + row0 * rpitch + (RB ? (size_t)((round * kSlabs + s) >> rbsh) * rbx : (size_t)0))[goff] = tmp[s];
            }
        }
        __syncthreads();
    }
}

// The same for interleaved half-spectra (pairs (S0, S1) of 16 bytes): the eight waves of the
// workgroup load the quarter's rows together -- a wave instruction covers 16 rows x 64 contiguous
// bytes, waves 0-3 the even slabs and waves 4-7 the odd ones, 32 loads per lane -- and hand the S0
// parts to the column owners of half 0 (through stg0), the S1 parts to those of half 1 (stg1).
// A loaded pair occupies the registers of d[2 i], d[2 i + 1] until its round.
// base: the pair array offset to (row 0, first column of the quarter), in pairs.
__device__ __forceinline__ void load_quarter_ilv(cx<float> (&d)[64], const wrow::f4* base, int rpitch, cx<float>* stg0, cx<float>* stg1, int wv, int lane) {
    const int wq = wv & 3, hh = wv >> 2;
    const int rl = 16 * wq + (lane >> 2), cl = lane & 3;
    const unsigned goff = (unsigned)(rl * rpitch + cl);
    const int wr = 4 * rl + cl + (rl >> 2);
    const int rd = 4 * lane + wq + (lane >> 2);
    const cx<float>* const mystg = hh ? stg1 : stg0;
#pragma unroll
    for (int i = 0; i < 32; ++i) {
        const wrow::f4 v = (base + (size_t)(64 * (2 * i + hh)) * rpitch)[goff];
        d[2 * i] = mk<float>(v.x, v.y);
        d[2 * i + 1] = mk<float>(v.z, v.w);
    }
    sched_fence();
#pragma unroll
    for (int round = 0; round < 64 / kSlabs; ++round) {
#pragma unroll
        for (int ii = 0; ii < kSlabs / 2; ++ii) {
            stg0[(2 * ii + hh) * kSlabPitch + wr] = d[round * kSlabs + 2 * ii];
            stg1[(2 * ii + hh) * kSlabPitch + wr] = d[round * kSlabs + 2 * ii + 1];
        }
        __syncthreads();
#pragma unroll
        for (int s = 0; s < kSlabs; ++s) d[round * kSlabs + s] = mystg[s * kSlabPitch + rd];
        __syncthreads();
    }
}
__device__ __forceinline__ void store_quarter_ilv(const cx<float> (&d)[64], wrow::f4* base, int rpitch, cx<float>* stg0, cx<float>* stg1, int wv, int lane, int ncols) {
    const int wq = wv & 3, hh = wv >> 2;
    const int rl = 16 * wq + (lane >> 2), cl = lane & 3;
    const unsigned goff = (unsigned)(rl * rpitch + cl);
    const int wr = 4 * rl + cl + (rl >> 2);
    const int rd = 4 * lane + wq + (lane >> 2);
    cx<float>* const mystg = hh ? stg1 : stg0;
#pragma unroll
    for (int round = 0; round < 64 / kSlabs; ++round) {
#pragma unroll
        for (int s = 0; s < kSlabs; ++s) mystg[s * kSlabPitch + rd] = d[round * kSlabs + s];
        __syncthreads();
        cx<float> t0[kSlabs / 2], t1[kSlabs / 2];
#pragma unroll
        for (int ii = 0; ii < kSlabs / 2; ++ii) {
            t0[ii] = stg0[(2 * ii + hh) * kSlabPitch + wr];
            t1[ii] = stg1[(2 * ii + hh) * kSlabPitch + wr];
        }
        if (cl < ncols) {
#pragma unroll
            for (int ii = 0; ii < kSlabs / 2; ++ii) {
                wrow::f4 v;
                v.x = t0[ii].x; v.y = t0[ii].y; v.z = t1[ii].x; v.w = t1[ii].y;
                (base + (size_t)(64 * (round * kSlabs + 2 * ii + hh)) * rpitch)[goff] = v;
            }
        }
        __syncthreads();
    }
}

// Pointwise stage of the resolvent mode for the wave pair (A = S0 column, B = S1 column):
//   U = (P + conj(e) Q) g,  V = e U,   A must end with U, B with V.
// The work is split by register halves: wave HH finishes registers 32 HH .. 32 HH + 31 (it loads
// only those multipliers, all at once), after receiving the partner's spectrum there through the
// partner's transposition tile; results the partner must keep go back through the same slots.
// g is recomputed from e (same expression as k_make_G) instead of being read.
template <int HH>
__device__ __forceinline__ void res2_pair(cx<float> (&d)[64], const cx<float>* ecol, float cK, float cDd, float gscale, cx<float>* mine, cx<float>* theirs, int lane, int dbg_mode) {
    constexpr int RB = 32 * HH, OB = 32 * (1 - HH);
    const unsigned ul = (unsigned)lane;
    cx<float> e[32];
#pragma unroll
    for (int rr = 0; rr < 32; ++rr) e[rr] = (dbg_mode & 1) ? mk<float>(1.f, 0.f) : (ecol + 64 * (RB + rr))[ul];
    if (!(dbg_mode & 2)) {
#pragma unroll
        for (int rr = 0; rr < 32; ++rr) mine[rr * 64 + lane] = d[OB + rr];
    }
    __syncthreads();
#pragma unroll
    for (int rr = 0; rr < 32; ++rr) {
        const int r = RB + rr;
        const cx<float> X = (dbg_mode & 2) ? d[r] : theirs[rr * 64 + lane];
        const cx<float> P = HH == 0 ? d[r] : X, Q = HH == 0 ? X : d[r];
        const float g = gscale / (1.f + cK * (e[rr].x * e[rr].x + e[rr].y * e[rr].y) + cDd);
        const cx<float> U = (P + mulc(Q, e[rr])) * g;
        const cx<float> V = e[rr] * U;
        d[r] = HH == 0 ? U : V;
        if (!(dbg_mode & 2)) theirs[rr * 64 + lane] = HH == 0 ? V : U;
        asm volatile("" : "+v"(d[r].x), "+v"(d[r].y));
    }
    __syncthreads();
    if (!(dbg_mode & 2)) {
#pragma unroll
        for (int rr = 0; rr < 32; ++rr) d[OB + rr] = mine[rr * 64 + lane];
    }
}

// ---- the packed pair of real columns -------------------------------------------------------------
// Columns 0 and W/2 of a half-spectrum hold real sequences.  With 2049 columns the grid would be one
// workgroup larger than two full rounds of the chip, and that one workgroup costs a third round; so
// the wave of column 0 carries column W/2 in its imaginary part, z = c0 + i cN, and the pointwise
// stage takes the two spectra apart again: Z0 = (Z + Zm) / 2, ZN = -i (Z - Zm) / 2 with the mirror
// Zm[k] = conj(Z[-k]).  Element (lane l, register r) mirrors to (lane 64 - l, register 63 - r), for
// lane 0 to (lane 0, register 64 - r): lane 0's registers are kept in a small LDS table for that.
__device__ __forceinline__ cx<float> mirror_of(const cx<float>& far, const cx<float>* lane0tab, int idx0, int lane, int ml) {
    cx<float> t = mk<float>(wrow::lane_get(far.x, ml), wrow::lane_get(far.y, ml));
    const cx<float> o = lane0tab[idx0];
    t.x = lane == 0 ? o.x : t.x;
    t.y = lane == 0 ? o.y : t.y;
    return cconj(t);
}
__device__ __forceinline__ cx<float> half_sum(cx<float> z, cx<float> zm) { return mk<float>(0.5f * (z.x + zm.x), 0.5f * (z.y + zm.y)); }
__device__ __forceinline__ cx<float> half_dif_mi(cx<float> z, cx<float> zm) { return mk<float>(0.5f * (z.y - zm.y), -0.5f * (z.x - zm.x)); }  // -i (z - zm) / 2
__device__ __forceinline__ cx<float> join(cx<float> a0, cx<float> aN) { return mk<float>(a0.x - aN.y, a0.y + aN.x); }                           // a0 + i aN
__device__ __forceinline__ void lane0_table(const cx<float> (&d)[64], cx<float>* tab, int lane) {
    if (lane == 0) {
#pragma unroll
        for (int r = 0; r < 64; ++r) tab[r] = d[r];
    }
    wfft::wave_sync();
}

// Resolvent-mode pointwise stage of the packed pair (A: S0, B: S1): two sub-phases of 16 registers;
// each wave sends its spectrum AND its mirror at the registers the partner is about to finish.
// The register sets (A: 0..15 then 16..31, B: 48..63 then 32..47) are chosen so that every mirror
// source is still unmodified when it is read.  Four workgroup barriers.
template <int HH>
__device__ __forceinline__ void res2_pair_packed(cx<float> (&d)[64], const cx<float>* e0col, const cx<float>* eNcol, float cK, float cD0, float cDN, float gscale,
                                                 cx<float>* mine, cx<float>* theirs, const cx<float>* tab, int lane) {
    const unsigned ul = (unsigned)lane;
    const int ml = (64 - lane) & 63;
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {
        constexpr int dummy = 0;
        (void)dummy;
        const int MY = HH == 0 ? 16 * sub : (sub == 0 ? 48 : 32);
        const int PT = HH == 0 ? (sub == 0 ? 48 : 32) : 16 * sub;
        cx<float> e0[16], eN[16];
#pragma unroll
        for (int rr = 0; rr < 16; ++rr) {
            e0[rr] = (e0col + 64 * (MY + rr))[ul];
            eN[rr] = (eNcol + 64 * (MY + rr))[ul];
        }
#pragma unroll
        for (int rr = 0; rr < 16; ++rr) {
            const int r = PT + rr;
            mine[rr * 64 + lane] = d[r];
            mine[(16 + rr) * 64 + lane] = mirror_of(d[63 - r], tab, (64 - r) & 63, lane, ml);
        }
        __syncthreads();
#pragma unroll
        for (int rr = 0; rr < 16; ++rr) {
            const int r = MY + rr;
            const cx<float> X = theirs[rr * 64 + lane], Xm = theirs[(16 + rr) * 64 + lane];
            const cx<float> Y = d[r], Ym = mirror_of(d[63 - r], tab, (64 - r) & 63, lane, ml);
            const cx<float> P = HH == 0 ? Y : X, Pm = HH == 0 ? Ym : Xm, Q = HH == 0 ? X : Y, Qm = HH == 0 ? Xm : Ym;
            const cx<float> P0 = half_sum(P, Pm), PN = half_dif_mi(P, Pm), Q0 = half_sum(Q, Qm), QN = half_dif_mi(Q, Qm);
            const float g0 = gscale / (1.f + cK * (e0[rr].x * e0[rr].x + e0[rr].y * e0[rr].y) + cD0);
            const float gN = gscale / (1.f + cK * (eN[rr].x * eN[rr].x + eN[rr].y * eN[rr].y) + cDN);
            const cx<float> U0 = (P0 + mulc(Q0, e0[rr])) * g0, UN = (PN + mulc(QN, eN[rr])) * gN;
            const cx<float> U = join(U0, UN), V = join(e0[rr] * U0, eN[rr] * UN);
            d[r] = HH == 0 ? U : V;
            theirs[rr * 64 + lane] = HH == 0 ? V : U;
            asm volatile("" : "+v"(d[r].x), "+v"(d[r].y));
        }
        __syncthreads();
#pragma unroll
        for (int rr = 0; rr < 16; ++rr) d[PT + rr] = mine[rr * 64 + lane];
    }
}

// Multiplier-mode pointwise stage of the packed pair: Z' = scale (m0 Z0 + i mN ZN), register pairs (r, 63 - r).
__device__ __forceinline__ void mul_packed(cx<float> (&d)[64], const cx<float>* e0col, const cx<float>* eNcol, int conj_mult, float scale, const cx<float>* tab, int lane) {
    const unsigned ul = (unsigned)lane;
    const int ml = (64 - lane) & 63;
#pragma unroll
    for (int r = 0; r < 32; ++r) {
        const int q = 63 - r;
        const cx<float> a0 = (e0col + 64 * r)[ul], aN = (eNcol + 64 * r)[ul], b0 = (e0col + 64 * q)[ul], bN = (eNcol + 64 * q)[ul];
        const cx<float> Zr = d[r], Zq = d[q];
        const cx<float> Zrm = mirror_of(Zq, tab, (64 - r) & 63, lane, ml), Zqm = mirror_of(Zr, tab, (64 - q) & 63, lane, ml);
        const cx<float> r0 = half_sum(Zr, Zrm), rN = half_dif_mi(Zr, Zrm), q0 = half_sum(Zq, Zqm), qN = half_dif_mi(Zq, Zqm);
        d[r] = join(conj_mult ? mulc(r0, a0) : r0 * a0, conj_mult ? mulc(rN, aN) : rN * aN) * scale;
        d[q] = join(conj_mult ? mulc(q0, b0) : q0 * b0, conj_mult ? mulc(qN, bN) : qN * bN) * scale;
        asm volatile("" : "+v"(d[r].x), "+v"(d[r].y), "+v"(d[q].x), "+v"(d[q].y));
    }
}

// MODE: COL_RES2 (eight waves: four columns of S0 and of S1) or COL_MUL (four waves: four columns of S0)
template <int MODE, bool PACKED, bool ILV = false, bool RB = false>
__global__ void __launch_bounds__(MODE == COL_RES2 ? 512 : 256, 2) k_wave_col(Args a) {
    static_assert(!ILV || (MODE == COL_RES2 && !PACKED), "interleaved spectra: resolvent mode, plain columns");
    static_assert(!RB || (!ILV && !PACKED), "row blocks: plain columns, separate spectra");
    extern __shared__ __align__(16) float tiles[];
    constexpr int NT = MODE == COL_RES2 ? 512 : 256;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6), h = wv >> 2, wq = wv & 3;  // wave-uniform, in scalar registers
    const int cs = a.lay.cs, C = 1 << cs;
    const int ncol = a.Wh - (PACKED ? 1 : 0);        // columns with a wave of their own
    const int nq = (ncol + 3) >> 2;                  // quarter tiles with at least one such column (tiles are 16 wide: 4 per tile)
    // block -> (plane, quarter tile).  16-column tiles: the four workgroups of a tile get indices 8 apart;
    // 4-column tiles: a quarter is a tile of its own, nothing is shared.
    const int per_plane = cs == 2 ? nq : 32 * ((a.lay.ntile + 7) / 8);
    const int plane = blockIdx.x / per_plane;
    const int b = blockIdx.x - plane * per_plane;
    int quarter = b;
    if (cs != 2) {
        // the last group of eight tiles goes first: it holds the lone column W/2, whose workgroup then
        // runs beside the first round instead of after the second one
        const int ngrp4 = 4 * ((a.lay.ntile + 7) / 8);
        const int x = b & 7;
        int aa = (b >> 3) + ngrp4 - 4;
        aa = aa >= ngrp4 ? aa - ngrp4 : aa;
        quarter = 4 * (8 * (aa >> 2) + x) + (aa & 3);
    }
    if (a.lf.enabled && quarter == 0) loss_finalize(a.lf, plane, reinterpret_cast<double*>(tiles), tid, NT);
    if (quarter >= nq) return;
    if (a.lf.stop) {
        const int s = a.lf.stop[plane];
        if (s >= 0 && s < a.lf.iter) return;
    }
    __syncthreads();
    const int k2q = quarter << 2;                    // first column of the quarter
    const int ncols = min(4, ncol - k2q);
    const int k2 = k2q + wq;                         // this wave's column
    const bool cvalid = k2 < ncol;
    const bool packed = PACKED && k2 == 0;           // this wave carries column W/2 in its imaginary part
    const int kN = a.Wh - 1;
    const size_t nbase = opaque(((((size_t)plane * a.lay.ntile + (kN >> cs)) * a.lay.H) << cs) + (kN & (C - 1)));
    cx<float>* const l0tab = reinterpret_cast<cx<float>*>(tiles + (size_t)(NT / 64) * WFFT_TILE_FLOATS) + 64 * h;
    cx<float>* const Sarr = (MODE == COL_RES2 && h == 1) ? a.S1 : a.S0;
    const size_t qbase = opaque(((((size_t)plane * a.lay.ntile + (k2q >> cs)) * a.lay.H) << cs) + (k2q & (C - 1)));
    cx<float>* const stg = reinterpret_cast<cx<float>*>(tiles + (size_t)(4 * h) * WFFT_TILE_FLOATS);
    cx<float>* const stg0 = reinterpret_cast<cx<float>*>(tiles);
    cx<float>* const stg1 = reinterpret_cast<cx<float>*>(tiles + (size_t)4 * WFFT_TILE_FLOATS);
    float* const mytile = tiles + (size_t)wv * WFFT_TILE_FLOATS;

    // column W/2 with a dense home: its quarter holds nothing else, so the workgroup skips the tile traffic
    // (the column would use 8 bytes of every 128-byte line) and wave 0 of each half streams the dense array
    const bool dense = !PACKED && a.nyqd && (a.Wh & 3) == 1 && k2q == a.Wh - 1;
    float* const dcol = a.nyqd + ((size_t)plane * 2 + h) * 4096;
#define DDN_CSTAMP(i) if (a.dbg && tid == 0) a.dbg[(size_t)blockIdx.x * 16 + (i)] = clock64();
    DDN_CSTAMP(0)
    if (a.dbg && tid == 0) a.dbg[(size_t)blockIdx.x * 16 + 6] = wall_clock64();
    cx<float> d[64];
    if (dense) {
#pragma unroll
        for (int r = 0; r < 64; ++r) d[r] = mk<float>(wq == 0 ? dcol[64 * r + lane] : 0.f, 0.f);
    } else if (ILV) {
        load_quarter_ilv(d, reinterpret_cast<const wrow::f4*>(a.S0) + qbase, C, stg0, stg1, wv, lane);
    } else if ((cs == 2 && !(a.dbg_mode & 16)) || (cs != 2 && (a.dbg_mode & 64))) {
        // 4-column tiles: the quarter is one contiguous block of rows x 4 columns; each wave reads its
        // own column directly (32-byte stride; the four waves of a half use every byte of every line)
        const cx<float>* const colp = Sarr + qbase + wq;
#pragma unroll
        for (int r = 0; r < 64; ++r) d[r] = (colp + ((size_t)(64 * r) << cs))[(unsigned)lane << cs];
    } else load_quarter<RB>(d, Sarr + qbase, C, stg, wq, lane, true, a.dbg_mode, a.rb_sh6, (size_t)a.rb_extra);
    if (packed) {
        const unsigned noff = (unsigned)(lane << cs);
#pragma unroll
        for (int r = 0; r < 64; ++r) d[r].y = (Sarr + nbase + ((size_t)(64 * r) << cs))[noff].x;
    }
    pin(d);
    DDN_CSTAMP(1)
    wfft::fft4096<false>(d, a.wtw, mytile, lane);
    pin(d);
    sched_fence();
    DDN_CSTAMP(2)

    const size_t mbase = opaque((size_t)(cvalid ? k2 : 0) * kColPitch);
    const cx<float>* const ecol = a.eKc + mbase;
    if (MODE == COL_RES2) {
        const float cDd = a.cD * a.ed2[cvalid ? k2 : 0];
        cx<float>* const mine = reinterpret_cast<cx<float>*>(mytile);
        cx<float>* const theirs = reinterpret_cast<cx<float>*>(tiles + (size_t)(wv ^ 4) * WFFT_TILE_FLOATS);
        if (PACKED && quarter == 0) {
            // the workgroup of column 0: four barriers for everybody
            if (packed) {
                const cx<float>* const eNcol = a.eKc + (size_t)kN * kColPitch;
                const float cDN = a.cD * a.ed2[kN];
                lane0_table(d, l0tab, lane);
                if (h == 0) res2_pair_packed<0>(d, ecol, eNcol, a.cK, cDd, cDN, a.gscale, mine, theirs, l0tab, lane);
                else res2_pair_packed<1>(d, ecol, eNcol, a.cK, cDd, cDN, a.gscale, mine, theirs, l0tab, lane);
            } else {
                if (h == 0) res2_pair<0>(d, ecol, a.cK, cDd, a.gscale, mine, theirs, lane, 0);
                else res2_pair<1>(d, ecol, a.cK, cDd, a.gscale, mine, theirs, lane, 0);
                __syncthreads();
                __syncthreads();
            }
        } else if (h == 0) res2_pair<0>(d, ecol, a.cK, cDd, a.gscale, mine, theirs, lane, a.dbg_mode);
        else res2_pair<1>(d, ecol, a.cK, cDd, a.gscale, mine, theirs, lane, a.dbg_mode);
    } else {
        const unsigned ul = (unsigned)lane;
        if (packed) {
            lane0_table(d, l0tab, lane);
            mul_packed(d, ecol, a.eKc + (size_t)kN * kColPitch, a.conj_mult, a.scale, l0tab, lane);
        } else
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) {
            cx<float> e[32];
#pragma unroll
            for (int rr = 0; rr < 32; ++rr) e[rr] = (ecol + 64 * (32 * blk + rr))[ul];
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
    DDN_CSTAMP(3)
    __syncthreads();  // exchange buffers are transposition tiles again from here on
    wfft::fft4096<true>(d, a.wtw, mytile, lane);
    pin(d);
    DDN_CSTAMP(4)
    __syncthreads();  // every wave is done with its tile before the tiles become staging space
    float ny[64];
    if (packed) {
#pragma unroll
        for (int r = 0; r < 64; ++r) {
            ny[r] = d[r].y;
            d[r].y = 0.f;
        }
    }
    // offsets recomputed here: keeping the 64 load addresses alive across the kernel would spill them
    if (dense) {
        if (wq == 0) {
#pragma unroll
            for (int r = 0; r < 64; ++r) dcol[64 * r + lane] = d[r].x;
        }
    } else if (ILV) {
        store_quarter_ilv(d, reinterpret_cast<wrow::f4*>(a.S0) + opaque(qbase), C, stg0, stg1, wv, lane, ncols);
    } else if (cs == 2 && (a.dbg_mode & 32)) {  // direct stores measured slower than the staged ones: diagnostic only
        if (wq < ncols) {
            cx<float>* const colp = Sarr + opaque(qbase) + wq;
#pragma unroll
            for (int r = 0; r < 64; ++r) (colp + (size_t)(64 * r) * 4)[(unsigned)lane * 4u] = d[r];
        }
    } else store_quarter<RB>(d, Sarr + opaque(qbase), C, stg, wq, lane, true, ncols, a.rb_sh6, (size_t)a.rb_extra);
    if (packed) {
        const unsigned noff = (unsigned)(lane << cs);
        cx<float>* const ncolp = Sarr + opaque(nbase);
#pragma unroll
        for (int r = 0; r < 64; ++r) (ncolp + ((size_t)(64 * r) << cs))[noff] = mk<float>(ny[r], 0.f);
    }
    DDN_CSTAMP(5)
    if (a.dbg && tid == 0) a.dbg[(size_t)blockIdx.x * 16 + 7] = wall_clock64();
#undef DDN_CSTAMP
}

}  // namespace wcol

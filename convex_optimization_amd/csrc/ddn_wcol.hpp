// Column stage for H = 4096, float32, in ONE pass: one wavefront per spectral column, the column
// held in registers (wfft::fft4096), eight waves per workgroup.
//
// A workgroup owns a quarter tile -- four adjacent columns, 32 contiguous bytes of every row -- of
// both half-spectra (resolvent mode, eight waves: waves 0-3 take the columns of S0, waves 4-7 the
// same columns of S1) or of S0 alone (multiplier mode, four waves, two workgroups per CU).  Global accesses are made row-major (a wave
// instruction covers 16 rows x 32 bytes) and redistributed to the column owners through LDS, 16
// row slabs per round; the transforms need no workgroup barrier.  The pointwise stage of the
// resolvent mode needs both spectra of a frequency in one lane: the S0 and S1 waves of a column
// swap half of their registers (S0 keeps the even ones, S1 the odd ones) through their (idle)
// transposition tiles and each finishes one half.
// The four quarter-tile workgroups of a tile are given block indices that differ by 8, so that
// they share an XCD (and its L2) and are dispatched together: each 128-byte line is then fetched
// once and written back whole.
//
// Multipliers (the eigenvalues of the blur, multiplying_matrix.py:51): for blur kernels of at most 31
// rows they are not read from memory at all but rebuilt from the taps (taps_mult below); otherwise
// they are read from a column-major natural-order copy (contiguous along the column).
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

// Row-major global <-> column-owner registers for one quarter tile (4 columns), executed by the
// four waves wq = 0..3 of a half; wave wq ends up with column wq: lane l, register r = row 64 r + l.
// base: the array offset to (row 0, first column of the quarter); rpitch = elements per row (tile width).
template <bool RB = false>
__device__ __forceinline__ void load_quarter(cx<float> (&d)[64], const cx<float>* base, int rpitch, cx<float>* stg, int wq, int lane, int rbsh = 0, size_t rbx = 0) {
    const int rl = 16 * wq + (lane >> 2), cl = lane & 3;
    const unsigned goff = (unsigned)(rl * rpitch + cl);
    const int wr = 4 * rl + cl + (rl >> 2);
    const int rd = 4 * lane + wq + (lane >> 2);
    // every global load of the column is issued before the first use: 64 x 512 bytes in flight per wave
#pragma unroll
    for (int r = 0; r < 64; ++r) d[r] = (base + (size_t)(64 * r) * rpitch + (RB ? (size_t)(r >> rbsh) * rbx : (size_t)0))[goff];
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
__device__ __forceinline__ void store_quarter(const cx<float> (&d)[64], cx<float>* base, int rpitch, cx<float>* stg, int wq, int lane, int ncols, int rbsh = 0, size_t rbx = 0) {
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
        if (cl < ncols) {
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

// ---- multipliers rebuilt from the blur taps --------------------------------------------------------
// The eigenvalue of the periodic correlation with a kh x kw kernel centred at (ph, pw) = (kh/2, kw/2)
// (eig_vals_for_periodic_conv_op.py:41-59) at frequency (k1, k2) is
//     e[k1, k2] = sum_u c_u(k2) w^(k1 (ph - u)),   c_u(k2) = sum_v tap[u, v] exp(-2 pi i k2 (pw - v) / W),  w = exp(-2 pi i / 4096).
// A lane holds the frequencies k1 = 64 r + l of a column, so with n = ph - u
//     e[64 r + l] = sum_n (c_n w^(l n)) w64^(r n):
// a 64-point transform over r of at most kh non-zero terms.  A wave needs the multipliers of the registers of one parity,
// r = 2 m + PAR, which are a 32-point transform of y[j], j = n mod 32 (kh <= 31: n in [-15, 15], no two n share a j):
//     y[j] = c_n w^(l n) w64^(PAR j) (-1)^(PAR [n < 0]).
// The table ctab[column][PAR][j] holds c_n with the two constant factors folded in (computed in double at set-up); the
// per-lane factor w^(l n) is a product of two entries of the transform's own twiddle table.  32 uniform loads and about
// 550 vector instructions replace 32 per-lane loads of 8 bytes -- a fifth of the stage's memory traffic.
template <int PAR>
__device__ __forceinline__ void taps_mult(cx<float> (&e)[32], const cx<float>* ctab_col, const cx<float>* wtw, int lane) {
    const cx<float>* const c = ctab_col + 32 * PAR;
    // w^(l n) = P[n] for n < 8, Q[1] = w^(8 l), Q[2] = w^(16 l), products for the rest (table layout: wfft::LaneTw)
    cx<float> P[8], Q1, Q2;
    {
        unsigned off = (unsigned)lane * 16u;
        asm volatile("" : "+v"(off));
        const cx<float>* t = wtw + off;
#pragma unroll
        for (int n = 1; n < 8; ++n) P[n] = t[n];
        Q1 = t[9];
        Q2 = t[10];
    }
    e[0] = c[0];
    e[16] = mulc(c[16], Q2);                    // n = -16 (unused for kernels of at most 31 rows: the table holds zero)
    e[8] = c[8] * Q1;                           // n = 8
    e[24] = mulc(c[24], Q1);                    // n = -8: w^(l n) = conj(w^(8 l))
#pragma unroll
    for (int n = 1; n < 8; ++n) {
        e[n] = c[n] * P[n];                     // slot j = n
        e[32 - n] = mulc(c[32 - n], P[n]);      // slot j = 32 - n holds n' = -n
        const cx<float> t = P[n] * Q1;          // w^(l (n + 8))
        e[n + 8] = c[n + 8] * t;
        e[24 - n] = mulc(c[24 - n], t);         // slot 32 - (n + 8)
    }
    wfft::small_dft<32, false>(e);
}

// Pointwise stage of the resolvent mode for the wave pair (A = S0 column, B = S1 column):
//   U = (P + conj(e) Q) g,  V = e U,   A must end with U, B with V.
// The work is split by register parity: wave HH finishes registers 2 rr + HH (it has only those
// multipliers), after receiving the partner's spectrum there through the partner's transposition
// tile; results the partner must keep go back through the same slots.
// g is recomputed from e (same expression as k_make_G) instead of being read.
template <int HH>
__device__ __forceinline__ void res2_pair(cx<float> (&d)[64], const cx<float>* ecol, const cx<float>* ctab_col, const cx<float>* wtw, float cK, float cDd, float gscale,
                                          cx<float>* mine, cx<float>* theirs, int lane) {
    const unsigned ul = (unsigned)lane;
#pragma unroll
    for (int rr = 0; rr < 32; ++rr) mine[rr * 64 + lane] = d[2 * rr + (1 - HH)];
    sched_fence();  // the registers just sent are free from here on: the multipliers are built in them
    cx<float> e[32];
    if (ctab_col) {
        taps_mult<HH>(e, ctab_col, wtw, lane);
    } else {
#pragma unroll
        for (int rr = 0; rr < 32; ++rr) e[rr] = (ecol + 64 * (2 * rr + HH))[ul];
    }
    sched_fence();
    __syncthreads();
#pragma unroll
    for (int rr = 0; rr < 32; ++rr) {
        const int r = 2 * rr + HH;
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
    for (int rr = 0; rr < 32; ++rr) d[2 * rr + (1 - HH)] = mine[rr * 64 + lane];
}

// Multiplier mode: d <- scale * (e or conj(e)) * d, the registers of one parity at a time.
template <int PAR>
__device__ __forceinline__ void mul_half(cx<float> (&d)[64], const cx<float>* ecol, const cx<float>* ctab_col, const cx<float>* wtw, int conj_mult, float scale, int lane) {
    const unsigned ul = (unsigned)lane;
    cx<float> e[32];
    if (ctab_col) {
        taps_mult<PAR>(e, ctab_col, wtw, lane);
    } else {
#pragma unroll
        for (int rr = 0; rr < 32; ++rr) e[rr] = (ecol + 64 * (2 * rr + PAR))[ul];
    }
#pragma unroll
    for (int rr = 0; rr < 32; ++rr) {
        const int r = 2 * rr + PAR;
        d[r] = (conj_mult ? mulc(d[r], e[rr]) : d[r] * e[rr]) * scale;
        asm volatile("" : "+v"(d[r].x), "+v"(d[r].y));
    }
}

// Column W/2 with a dense home (one real number per row and spectrum): its workgroup holds no other column, so the
// whole workgroup transforms it together in LDS -- three radix-16 passes each way, one butterfly per thread and
// pass -- instead of leaving it to one wavefront per spectrum (the register transform of a lone wavefront takes as
// long as a whole quarter tile and the launch then ends that much later: 512 quarter tiles are exactly two rounds).
// The multipliers are read from the column-major copy (4096 entries, natural order); an LDS position p of the
// digit-reversed spectrum holds frequency ((p >> 8) & 15) + 16 ((p >> 4) & 15) + 256 (p & 15).
template <int MODE, int NT>
__device__ __forceinline__ void dense_column(const Args& a, int plane, float* tiles, int tid) {
    constexpr int NARR = MODE == COL_RES2 ? 2 : 1;
    constexpr int PER = 4096 / NT;                   // spectrum entries per thread
    cx<float>* const L = reinterpret_cast<cx<float>*>(tiles);
    float* const dc = a.nyqd + (size_t)plane * 2 * 4096;
    const int kc = a.Wh - 1;
    const cx<float>* const ecol = a.eKc + (size_t)kc * kColPitch;
    // every global read of the job is issued here, before the first wait: one memory latency instead of three in a row
    cx<float> ev[PER];
    float in[NARR * PER];
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const int p = tid + k * NT;
        ev[k] = ecol[((p >> 8) & 15) + (((p >> 4) & 15) << 4) + ((p & 15) << 8)];
    }
#pragma unroll
    for (int k = 0; k < NARR * PER; ++k) in[k] = dc[tid + k * NT];
    const float cDd = a.cD * a.ed2[kc];
    const Tw<float> tw = tw_setup<float>(L + NARR * 4096, a.twH, 4096, tid, NT);
#pragma unroll
    for (int k = 0; k < NARR * PER; ++k) {
        const int i = tid + k * NT;
        L[(i & ~4095) + SW(i & 4095)] = mk<float>(in[k], 0.f);
    }
    __syncthreads();
    fft_forward_fixed<float, 4096, 16, 0>(L, NARR, 4096, tw, 0, tid, NT);
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const cx<float> e = ev[k];
        const int i0 = SW(tid + k * NT);
        if (MODE == COL_RES2) {
            const cx<float> P = L[i0], Q = L[4096 + i0];
            const float g = a.gscale * __builtin_amdgcn_rcpf(1.f + a.cK * (e.x * e.x + e.y * e.y) + cDd);
            const cx<float> U = (P + mulc(Q, e)) * g;
            L[i0] = U;
            L[4096 + i0] = e * U;
        } else {
            const cx<float> d = L[i0];
            L[i0] = (a.conj_mult ? mulc(d, e) : d * e) * a.scale;
        }
    }
    __syncthreads();
    fft_inverse_fixed<float, 4096, 16, 0>(L, NARR, 4096, tw, 0, tid, NT);
    for (int i = tid; i < NARR * 4096; i += NT) dc[i] = L[(i & ~4095) + SW(i & 4095)].x;
}

// MODE: COL_RES2 (eight waves: four columns of S0 and of S1) or COL_MUL (four waves: four columns of S0)
template <int MODE, bool ILV = false, bool RB = false>
__global__ void __launch_bounds__(MODE == COL_RES2 ? 512 : 256, 2) k_wave_col(Args a) {
    static_assert(!ILV || MODE == COL_RES2, "interleaved spectra: resolvent mode");
    static_assert(!RB || !ILV, "row blocks: separate spectra");
    extern __shared__ __align__(16) float tiles[];
    constexpr int NT = MODE == COL_RES2 ? 512 : 256;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6), h = wv >> 2, wq = wv & 3;  // wave-uniform, in scalar registers
    const int cs = a.lay.cs, C = 1 << cs;
    const int ncol = a.Wh;
    const int nq = (ncol + 3) >> 2;                  // quarter tiles with at least one column (tiles are 16 wide: 4 per tile)
    // block -> (plane, quarter tile), nq blocks per plane and none without work.  Within a whole group of eight tiles
    // (32 blocks) the four workgroups of a tile get indices 8 apart: same XCD, dispatched together.  The quarters of
    // the last, partial group of tiles (column W/2) come first, one per block: a lone column's job then runs beside the
    // first round instead of after the last one.
    const int plane = blockIdx.x / nq;
    const int b = blockIdx.x - plane * nq;
    const int nfull = (a.lay.ntile >> 3) << 5, nrest = nq - nfull;
    int quarter, tgroup;
    {
        const int bb = b - nrest;
        if (bb >= 0) {
            quarter = 4 * (8 * (bb >> 5) + (bb & 7)) + ((bb >> 3) & 3);
            tgroup = bb >> 5;
        } else {
            quarter = nfull + b;
            tgroup = 0;
        }
    }
    if (a.lf.enabled && quarter == 0) loss_finalize(a.lf, plane, reinterpret_cast<double*>(tiles), tid, NT);
    if (quarter >= nq) return;
    if (a.lf.stop) {
        const int s = a.lf.stop[plane];
        if (s >= 0 && s < a.lf.iter) return;
    }
    __syncthreads();
    // Two phases across the chip.  Every workgroup of the first round would load, transform and store at the same time
    // as all the others: the memory system is then saturated during the load and store phases and idle during the
    // transforms (a workgroup that happens to run out of step finishes in 32 us instead of 42).  So half of the first
    // round -- every other group of eight tiles, the four quarters of a tile together, since they share cache lines --
    // starts about one uncontended load phase late; later rounds inherit the shift.
    if (a.stagger > 0 && (int)blockIdx.x < a.stagger_blocks && (tgroup & 1)) {
        const long long t0 = wall_clock64();
        while (wall_clock64() - t0 < (long long)a.stagger) __builtin_amdgcn_s_sleep(8);
    }
    const int k2q = quarter << 2;                    // first column of the quarter
    const int ncols = min(4, ncol - k2q);
    const int k2 = k2q + wq;                         // this wave's column
    const bool cvalid = k2 < ncol;
    cx<float>* const Sarr = (MODE == COL_RES2 && h == 1) ? a.S1 : a.S0;
    const size_t qbase = opaque(((((size_t)plane * a.lay.ntile + (k2q >> cs)) * a.lay.H) << cs) + (k2q & (C - 1)));
    cx<float>* const stg = reinterpret_cast<cx<float>*>(tiles + (size_t)(4 * h) * WFFT_TILE_FLOATS);
    cx<float>* const stg0 = reinterpret_cast<cx<float>*>(tiles);
    cx<float>* const stg1 = reinterpret_cast<cx<float>*>(tiles + (size_t)4 * WFFT_TILE_FLOATS);
    float* const mytile = tiles + (size_t)wv * WFFT_TILE_FLOATS;

    // column W/2 with a dense home: its quarter holds nothing else, so the workgroup skips the tile traffic
    // (the column would use 8 bytes of every 128-byte line) and transforms the dense array in LDS
    if (a.nyqd && (a.Wh & 3) == 1 && k2q == a.Wh - 1) {
        if (a.dbg && tid == 0) a.dbg[(size_t)blockIdx.x * 16 + 6] = wall_clock64();
        dense_column<MODE, NT>(a, plane, tiles, tid);
        if (a.dbg && tid == 0) a.dbg[(size_t)blockIdx.x * 16 + 7] = wall_clock64();
        return;
    }
#define DDN_CSTAMP(i) if (a.dbg && tid == 0) a.dbg[(size_t)blockIdx.x * 16 + (i)] = clock64();
    DDN_CSTAMP(0)
    if (a.dbg && tid == 0) a.dbg[(size_t)blockIdx.x * 16 + 6] = wall_clock64();
    cx<float> d[64];
    if (ILV) {
        load_quarter_ilv(d, reinterpret_cast<const wrow::f4*>(a.S0) + qbase, C, stg0, stg1, wv, lane);
    } else load_quarter<RB>(d, Sarr + qbase, C, stg, wq, lane, a.rb_sh6, (size_t)a.rb_extra);
    pin(d);
    DDN_CSTAMP(1)
    wfft::fft4096<false>(d, a.wtw, mytile, lane);
    pin(d);
    sched_fence();
    DDN_CSTAMP(2)

    const int kc = cvalid ? k2 : 0;
    const cx<float>* const ecol = a.eKc + opaque((size_t)kc * kColPitch);
    const cx<float>* const ctab_col = a.ctab ? a.ctab + opaque((size_t)kc * 64) : nullptr;
    if (MODE == COL_RES2) {
        const float cDd = a.cD * a.ed2[kc];
        cx<float>* const mine = reinterpret_cast<cx<float>*>(mytile);
        cx<float>* const theirs = reinterpret_cast<cx<float>*>(tiles + (size_t)(wv ^ 4) * WFFT_TILE_FLOATS);
        if (h == 0) res2_pair<0>(d, ecol, ctab_col, a.wtw, a.cK, cDd, a.gscale, mine, theirs, lane);
        else res2_pair<1>(d, ecol, ctab_col, a.wtw, a.cK, cDd, a.gscale, mine, theirs, lane);
    } else {
        mul_half<0>(d, ecol, ctab_col, a.wtw, a.conj_mult, a.scale, lane);
        mul_half<1>(d, ecol, ctab_col, a.wtw, a.conj_mult, a.scale, lane);
    }
    pin(d);
    sched_fence();
    DDN_CSTAMP(3)
    __syncthreads();  // exchange buffers are transposition tiles again from here on
    wfft::fft4096<true>(d, a.wtw, mytile, lane);
    pin(d);
    DDN_CSTAMP(4)
    __syncthreads();  // every wave is done with its tile before the tiles become staging space
    // offsets recomputed here (from an opaque copy of the lane index): the load phase's addresses and staging
    // offsets kept alive across the kernel would be spilled
    int lane_s = lane;
    asm volatile("" : "+v"(lane_s));
    if (ILV) {
        store_quarter_ilv(d, reinterpret_cast<wrow::f4*>(a.S0) + opaque(qbase), C, stg0, stg1, wv, lane_s, ncols);
    } else store_quarter<RB>(d, Sarr + opaque(qbase), C, stg, wq, lane_s, ncols, a.rb_sh6, (size_t)a.rb_extra);
    DDN_CSTAMP(5)
    if (a.dbg && tid == 0) a.dbg[(size_t)blockIdx.x * 16 + 7] = wall_clock64();
#undef DDN_CSTAMP
}

}  // namespace wcol

// Fused solver row kernels for W = 64 R, R in {8, 16, 32} (512, 1024, 2048), float32, 2-tap D:
// ONE WAVEFRONT PER M = 64 / R CONSECUTIVE IMAGE ROWS.
//
// The M rows are one contiguous 4096-pixel segment of every iterate array, so the pixel side is the
// same as for W = 4096 (ddn_wrow.hpp): lane l, register g holds pixel 64 g + l of the segment, which
// is pixel 64 (g % R) + l of row g / R, and every global access of a register is one contiguous
// 256-byte wave access.  The transforms are wfft::fftN<R> (M independent 64 R-point transforms in
// one wave); on the frequency side lane q R + k1, register k2 holds frequency k1 + R k2 of row q.
// D and D^T are lane rotations that wrap inside a row's R registers; the Hermitian split / merge
// is a lane reversal inside each group of R lanes.
#pragma once
#include "ddn_wrow.hpp"

namespace wrow {

// Spectral addressing on the frequency side: element k = k1 + R r of row (first + q) sits at
// rowbase(first) + lane_off + reg(r); rowbase and reg(r) are uniform, lane_off is per lane and the
// same for every item (consecutive rows of a tile are 2^cs elements apart, and the tile / column
// split of k1 + R r is the sum of the splits of k1 and of R r because R and the tile width are
// powers of two).
template <int R>
struct SpecAddrN {
    unsigned lane_off;
    size_t tstride;
    int cs;
    __device__ __forceinline__ SpecAddrN(const SpecLayout& l, int lane) {
        cs = l.cs;
        tstride = (size_t)l.H << l.cs;
        const int k1 = lane & (R - 1), q = lane / R;
        lane_off = (unsigned)(((size_t)q << l.cs) + (size_t)(k1 >> l.cs) * tstride + (k1 & ((1 << l.cs) - 1)));
    }
    __device__ __forceinline__ size_t reg(int r) const {
        const int k = R * r;
        return (size_t)(k >> cs) * tstride + (size_t)(k & ((1 << cs) - 1));
    }
};

// Merge: half-spectra A, B (k = 0 .. 32 R) of the two real fields of each row -> spectrum of a + i b.
// ILV: the half-spectra are interleaved element by element in one array of 16-byte pairs (see pack).
template <int R, bool ILV = false>
__device__ __forceinline__ void packN(cx<float> (&d)[64], const cx<float>* rowA, const cx<float>* rowB, const SpecAddrN<R>& sa, int lane) {
    const int k1 = lane & (R - 1);
    // column W/2 (real): register 32 of the lanes with k1 = 0; the other lanes would read past the row
    cx<float> Un = mk<float>(0.f, 0.f), Vn = mk<float>(0.f, 0.f);
    if (ILV) {
        const f4* const q = reinterpret_cast<const f4*>(rowA);
#pragma unroll
        for (int r = 0; r < 32; ++r) {
            const f4 v = (q + sa.reg(r))[sa.lane_off];
            d[r] = mk<float>(v.x, v.y);
            d[63 - r] = mk<float>(v.z, v.w);
        }
        if (k1 == 0) {
            const f4 v = (q + sa.reg(32))[sa.lane_off];
            Un = mk<float>(v.x, v.y);
            Vn = mk<float>(v.z, v.w);
        }
    } else {
#pragma unroll
        for (int r = 0; r < 32; ++r) {
            d[r] = (rowA + sa.reg(r))[sa.lane_off];       // A[k], k = k1 + R r
            d[63 - r] = (rowB + sa.reg(r))[sa.lane_off];  // B[k]
        }
        if (k1 == 0) {
            Un = (rowA + sa.reg(32))[sa.lane_off];
            Vn = (rowB + sa.reg(32))[sa.lane_off];
        }
    }
    sched_fence();
    const int ml = (lane & ~(R - 1)) | ((R - k1) & (R - 1));
    const float u0 = d[0].x, v0 = d[63].x;
#pragma unroll
    for (int r = 0; r < 32; ++r) {
        const cx<float> U = d[r], V = d[63 - r];
        d[r] = mk<float>(U.x - V.y, U.y + V.x);                     // Z[k]     = A[k] + i B[k]
        const float mx = U.x + V.y, my = V.x - U.y;                 // Z[W - k] = conj(A[k]) + i conj(B[k])
        d[63 - r] = mk<float>(lane_get(mx, ml), lane_get(my, ml));  // lands at lane group + (R - k1) % R
    }
    // k1 = 0 holds k = R r: its mirror R (64 - r) sits one register higher than for the other lanes
    // (values are copied out first: a conditional between two array elements is a conditional between
    // two addresses, which keeps the whole array in memory)
    const bool z = k1 == 0;
#pragma unroll
    for (int r = 0; r < 31; ++r) {
        const float lx = d[62 - r].x, ly = d[62 - r].y, hx = d[63 - r].x, hy = d[63 - r].y;
        d[63 - r].x = z ? lx : hx;
        d[63 - r].y = z ? ly : hy;
    }
    // the mean and the Nyquist term of a real row are real
    const float nx = d[32].x, ny = d[32].y, zx = d[0].x, zy = d[0].y;
    d[32].x = z ? Un.x : nx;
    d[32].y = z ? Vn.x : ny;
    d[0].x = z ? u0 : zx;
    d[0].y = z ? v0 : zy;
    sched_fence();
}

// Split: spectrum Z of a + i b -> half-spectra of a (to rowA) and b (to rowB).
template <int R, bool ILV = false>
__device__ __forceinline__ void unpackN(const cx<float> (&d)[64], cx<float>* rowA, cx<float>* rowB, const SpecAddrN<R>& sa, int lane) {
    const int k1 = lane & (R - 1);
    const int ml = (lane & ~(R - 1)) | ((R - k1) & (R - 1));
    f4* const q = reinterpret_cast<f4*>(rowA);
#pragma unroll
    for (int r = 0; r < 32; ++r) {
        cx<float> Zm = mk<float>(lane_get(d[63 - r].x, ml), lane_get(d[63 - r].y, ml));
        const cx<float> own = (r == 0) ? d[0] : d[64 - r];
        Zm.x = k1 == 0 ? own.x : Zm.x;
        Zm.y = k1 == 0 ? own.y : Zm.y;
        const cx<float> Z = d[r];
        if (ILV) {
            f4 v;
            v.x = 0.5f * (Z.x + Zm.x); v.y = 0.5f * (Z.y - Zm.y);
            v.z = 0.5f * (Z.y + Zm.y); v.w = 0.5f * (Zm.x - Z.x);
            (q + sa.reg(r))[sa.lane_off] = v;
        } else {
            (rowA + sa.reg(r))[sa.lane_off] = mk<float>(0.5f * (Z.x + Zm.x), 0.5f * (Z.y - Zm.y));
            (rowB + sa.reg(r))[sa.lane_off] = mk<float>(0.5f * (Z.y + Zm.y), 0.5f * (Zm.x - Z.x));
        }
    }
    if (k1 == 0) {
        if (ILV) {
            f4 v;
            v.x = d[32].x; v.y = 0.f; v.z = d[32].y; v.w = 0.f;
            (q + sa.reg(32))[sa.lane_off] = v;
        } else {
            (rowA + sa.reg(32))[sa.lane_off] = mk<float>(d[32].x, 0.f);
            (rowB + sa.reg(32))[sa.lane_off] = mk<float>(d[32].y, 0.f);
        }
    }
}

// PDR / PDDR / ADMM: the policies of ddn_kernels.hpp, one pixel per register (see k_wave_row).
template <class P, int PD, bool CONS, bool PROD, int FE, int R, bool ILV = false>
__global__ void __launch_bounds__(64, WROW_MINW) k_wave_rowN(RowCommon<float> c, typename P::Args a0) {
    constexpr int M = 64 / R;
    extern __shared__ __align__(16) float wtiles[];
    const int lane = threadIdx.x & 63;
    float* const tile = wtiles;
    const int nitems = c.batch * c.H / M;  // H is a multiple of M: an item never straddles two planes
    const SpecAddrN<R> sa(c.lay, lane);
    for (int it = blockIdx.x; it < nitems; it += gridDim.x) {
        const int row = it * M;
        const int plane = row / c.H;
        if (c.stop && c.stop[plane] >= 0) continue;
        typename P::Args a = a0;
        advance(a, opaque((size_t)it * 4096));
        const size_t sbase = opaque(spec_row_base(c.lay, plane, row - plane * c.H));
        cx<float>* const row0 = c.S0 + (ILV ? 2 : 1) * sbase;  // ILV: S0 holds the (S0, S1) pairs, 16 bytes each
        cx<float>* const row1 = c.S1 + sbase;
        const unsigned ul = (unsigned)lane;
        cx<float> d[64];
        if (CONS) {
            packN<R, ILV>(d, row0, row1, sa, lane);
            wfft::fftN<R, true>(d, c.wtw, tile, lane);
            pin(d);
        } else {
#pragma unroll
            for (int r = 0; r < 64; ++r) d[r] = mk<float>(0.f, 0.f);
        }
        sched_fence();
        typename P::template St<1> stq[PD];
#pragma unroll
        for (int g = 0; g < PD; ++g) P::template load<1>(at_reg(a, g), ul, stq[g]);
        double acc[4] = {0, 0, 0, 0};
        float rot_prev = 0.f, rs_first = 0.f, rs_prev = 0.f;
#pragma unroll
        for (int r = 0; r < 64; ++r) {
            const int j = r % R;  // register inside the row
            if (r % FE == 0) sched_fence();
            // pixel -1 of a row is its last pixel: lane 63 of the row's last register (not yet overwritten)
            if (j == 0) rot_prev = from_prev_lane(d[r + R - 1].x);
            float u[1] = {d[r].x}, k[1] = {d[r].y}, du[1], re[1], im[1], s[1];
            const float rot = from_prev_lane(u[0]);
            du[0] = u[0] - (lane == 0 ? rot_prev : rot);
            rot_prev = rot;
            P::template step_regs<1>(at_reg(a, r), CONS, PROD, ul, stq[r % PD], u, k, du, re, im, s, acc);
            if (r + PD < 64) P::template load<1>(at_reg(a, r + PD), ul, stq[r % PD]);
            if (PROD) {
                // transform input: re + D^T s = re + s[p] - s[p+1]; s[p+1] sits in lane l+1, for lane 63
                // in lane 0 of the next register of the row (the first one after the last)
                const float rs = from_next_lane(s[0]);
                d[r] = mk<float>(re[0] + s[0], im[0]);
                if (j == 0) rs_first = rs;
                else d[r - 1].x -= (lane == 63 ? rs : rs_prev);
                if (j == R - 1) d[r].x -= (lane == 63 ? rs_first : rs);
                rs_prev = rs;
                asm volatile("" : "+v"(d[r].x), "+v"(d[r].y), "+v"(rs_prev));
                if (j > 0) asm volatile("" : "+v"(d[r - 1].x));
            }
            if (CONS && j == R - 1) {
                wave_sum<4>(acc);
                if (lane == 0) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) c.partials[(size_t)(row + r / R) * 4 + q] = acc[q];
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[q] = 0.0;
            }
        }
        sched_fence();
        if (PROD) {
            pin(d);
            sched_fence();
            wfft::fftN<R, false>(d, c.wtw, tile, lane);
            pin(d);
            unpackN<R, ILV>(d, row0, row1, sa, lane);
        }
    }
}

template <class P, int PD, int FE, int R>
int wrown_go(hipStream_t st, int grid, const RowCommon<float>& c, const typename P::Args& a) {
    const size_t lds = (size_t)WFFT_TILE_FLOATS * sizeof(float);
    if (c.ilv) {
        if (c.do_consume && c.do_produce) return launch_row_fn(st, k_wave_rowN<P, PD, true, true, FE, R, true>, grid, 64, lds, c, a);
        if (c.do_consume) return launch_row_fn(st, k_wave_rowN<P, PD, true, false, 1, R, true>, grid, 64, lds, c, a);
        return launch_row_fn(st, k_wave_rowN<P, PD, false, true, 1, R, true>, grid, 64, lds, c, a);
    }
    if (c.do_consume && c.do_produce) return launch_row_fn(st, k_wave_rowN<P, PD, true, true, FE, R>, grid, 64, lds, c, a);
    if (c.do_consume) return launch_row_fn(st, k_wave_rowN<P, PD, true, false, 1, R>, grid, 64, lds, c, a);
    return launch_row_fn(st, k_wave_rowN<P, PD, false, true, 1, R>, grid, 64, lds, c, a);
}

template <class P, int PD, int FE>
int wrown_dispatch(hipStream_t st, int grid, const RowCommon<float>& c, const typename P::Args& a) {
    switch (c.W) {
        case 512: return wrown_go<P, PD, FE, 8>(st, grid, c, a);
        case 1024: return wrown_go<P, PD, FE, 16>(st, grid, c, a);
        case 2048: return wrown_go<P, PD, FE, 32>(st, grid, c, a);
        default: return DDN_E_UNSUPPORTED;
    }
}

// ----------------------------------------------------------------------------------------------
// Chambolle-Pock (chambolle_pock/algorithm.py:58-92; see k_wave_cp): a wave owns 2 M consecutive
// image rows, the first M in the real slot and the next M in the imaginary slot of the M complex
// transforms.  H is a multiple of 2 M.
template <int PD, int SL, int R>
__device__ __forceinline__ void cp_dualN(cx<float> (&d)[64], const CpArgs<float>& a, int lane) {
    struct Q { float z, y0, y1, y2, b; } q[PD];
    const unsigned lo = (unsigned)lane;
#define DDN_CP_LD(r, e) { (e).z = opaque_ptr(a.z + 64 * (r))[lo]; (e).y0 = ldi(opaque_ptr(a.y0 + 64 * (r)), lo); (e).y1 = ldi(opaque_ptr(a.y1 + 64 * (r)), lo); (e).y2 = ldi(opaque_ptr(a.y2 + 64 * (r)), lo); (e).b = opaque_ptr(a.b + 64 * (r))[lo]; }
    float zl = ldi(opaque_ptr(a.z + 64 * (R - 1)), lo);  // last register of the row: z[-1] of lane 0
#pragma unroll
    for (int g = 0; g < PD; ++g) DDN_CP_LD(g, q[g])
    float rot_prev = 0.f;
#pragma unroll
    for (int r = 0; r < 64; ++r) {
        const int j = r % R;
        sched_fence();
        if (j == 0) {
            rot_prev = from_prev_lane(zl);
            if (r + R < 64) zl = ldi(opaque_ptr(a.z + 64 * (r + 2 * R - 1)), lo);  // for the next row, a row ahead
        }
        const Q e = q[r % PD];
        if (r + PD < 64) DDN_CP_LD(r + PD, q[r % PD])
        const float in = slot<SL>(d[r]);
        const float rot = from_prev_lane(e.z);
        const float dz = e.z - (lane == 0 ? rot_prev : rot);                      // cp:61
        rot_prev = rot;
        const float g0 = e.y0 + a.s * in;                                         // cp:66
        const float g1 = e.y1 + a.s * dz;
        const float g2 = e.y2 + a.s * dz;
        const float pr0 = data_prox(a.objective, udiv(g0, a.s) - e.b, a.is, a.l2div);   // cp:69-72
        const float y0n = g0 - a.s * e.b - a.s * pr0;
        float i1, i2;
        iso(udiv(g1, a.s), udiv(g2, a.s), a.isg, i1, i2);                         // cp:75
        sti(opaque_ptr(a.y0 + 64 * r), lo, y0n);
        sti(opaque_ptr(a.y1 + 64 * r), lo, g1 - a.s * i1);
        sti(opaque_ptr(a.y2 + 64 * r), lo, g2 - a.s * i2);
        slot<SL>(d[r]) = y0n;
        asm volatile("" : "+v"(slot<SL>(d[r])));
    }
#undef DDN_CP_LD
    sched_fence();
}

// partials: the loss partials of the slot's first row (four doubles per row)
template <int PD, int SL, int R>
__device__ __forceinline__ void cp_primalN(cx<float> (&d)[64], const CpArgs<float>& a, int lane, double* partials) {
    struct Q { float x, b, y1, y2; } q[PD];
    const unsigned lo = (unsigned)lane;
#define DDN_CP_LD(r, e) { (e).x = ldi(opaque_ptr(a.x + 64 * (r)), lo); (e).b = opaque_ptr(a.b + 64 * (r))[lo]; (e).y1 = ldi(opaque_ptr(a.y1 + 64 * (r)), lo); (e).y2 = ldi(opaque_ptr(a.y2 + 64 * (r)), lo); }
#pragma unroll
    for (int g = 0; g < PD; ++g) DDN_CP_LD(g, q[g])
    double acc[4] = {0, 0, 0, 0};
    // D^T(y1 + y2)[p] = s[p] - s[p+1]: register r is finished when s of the row's next register is known
    float rs_first = 0.f, rs_prev = 0.f, in_prev = 0.f, sc_prev = 0.f, x_prev = 0.f, b_prev = 0.f;
#define DDN_CP_FIN(r, snext) {                                                                 \
        const float aty = in_prev + (sc_prev - (snext));                          /* cp:79 */  \
        const float xn = clip01(x_prev - a.t * aty);                              /* cp:83-86 */ \
        const float zn = 2.f * xn - x_prev;                                       /* cp:87 */  \
        const float dd = xn - b_prev;                                             /* cp:96 */  \
        acc[0] += (double)t_abs(dd); acc[1] += (double)dd * (double)dd;                         \
        sti(opaque_ptr(a.x + 64 * (r)), lo, xn);                                                    \
        opaque_ptr(a.z + 64 * (r))[lo] = zn;                                                    \
        slot<SL>(d[r]) = zn;                                                                    \
        asm volatile("" : "+v"(slot<SL>(d[r]))); }
#pragma unroll
    for (int r = 0; r < 64; ++r) {
        const int j = r % R;
        sched_fence();
        const Q e = q[r % PD];
        if (r + PD < 64) DDN_CP_LD(r + PD, q[r % PD])
        const float in = slot<SL>(d[r]);
        const float sc = e.y1 + e.y2;
        const float rs = from_next_lane(sc);
        if (j == 0) rs_first = rs;
        else DDN_CP_FIN(r - 1, lane == 63 ? rs : rs_prev)
        rs_prev = rs; in_prev = in; sc_prev = sc; x_prev = e.x; b_prev = e.b;
        if (j == R - 1) {
            DDN_CP_FIN(r, lane == 63 ? rs_first : rs)
            wave_sum<4>(acc);
            if (lane == 0) {
#pragma unroll
                for (int i = 0; i < 4; ++i) partials[(size_t)(r / R) * 4 + i] = acc[i];
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = 0.0;
        }
    }
#undef DDN_CP_FIN
#undef DDN_CP_LD
    sched_fence();
}

template <int PD, int STEP, bool CONS, bool PROD, int R>
__global__ void __launch_bounds__(64, 2) k_wave_cpN(RowCommon<float> c, CpArgs<float> a0) {
    constexpr int M = 64 / R;
    extern __shared__ __align__(16) float wtiles[];
    const int lane = threadIdx.x & 63, H = c.H;
    float* const tile = wtiles;
    const int nitems = c.batch * H / (2 * M);
    const SpecAddrN<R> sa(c.lay, lane);
    for (int it = blockIdx.x; it < nitems; it += gridDim.x) {
        const int row = it * 2 * M;
        const int plane = row / H;
        const int r0 = row - plane * H;
        CpArgs<float> a = a0, a1 = a0;
        advance(a, opaque((size_t)it * 8192));
        advance(a1, opaque((size_t)it * 8192 + 4096));
        cx<float>* const rw0 = c.S0 + opaque(spec_row_base(c.lay, plane, r0));
        cx<float>* const rw1 = c.S0 + opaque(spec_row_base(c.lay, plane, r0 + M));
        cx<float> d[64];
        if (CONS) {
            packN<R>(d, rw0, rw1, sa, lane);
            wfft::fftN<R, true>(d, c.wtw, tile, lane);
            pin(d);
        } else {
#pragma unroll
            for (int r = 0; r < 64; ++r) d[r] = mk<float>(0.f, 0.f);
        }
        sched_fence();
        if (STEP == 0) {
            cp_dualN<PD, 0, R>(d, a, lane);
            cp_dualN<PD, 1, R>(d, a1, lane);
        } else if (CONS) {
            cp_primalN<PD, 0, R>(d, a, lane, c.partials + (size_t)row * 4);
            cp_primalN<PD, 1, R>(d, a1, lane, c.partials + (size_t)(row + M) * 4);
        } else {
            cp_load_z<0>(d, a, lane);
            cp_load_z<1>(d, a1, lane);
        }
        if (PROD) {
            pin(d);
            sched_fence();
            wfft::fftN<R, false>(d, c.wtw, tile, lane);
            pin(d);
            unpackN<R>(d, rw0, rw1, sa, lane);
        }
    }
}

template <int PD, int R>
int wrown_cp_go(hipStream_t st, int grid, const RowCommon<float>& c, const CpArgs<float>& a) {
    const size_t lds = (size_t)WFFT_TILE_FLOATS * sizeof(float);
    if (a.step == 0) return launch_row_fn(st, k_wave_cpN<PD, 0, true, true, R>, grid, 64, lds, c, a);
    if (!c.do_consume) return launch_row_fn(st, k_wave_cpN<PD, 1, false, true, R>, grid, 64, lds, c, a);
    if (c.do_produce) return launch_row_fn(st, k_wave_cpN<PD, 1, true, true, R>, grid, 64, lds, c, a);
    return launch_row_fn(st, k_wave_cpN<PD, 1, true, false, R>, grid, 64, lds, c, a);
}

}  // namespace wrow

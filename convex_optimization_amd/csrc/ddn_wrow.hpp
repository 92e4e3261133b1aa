// Fused solver row kernels for W = 4096, float32, 2-tap D: ONE WAVEFRONT PER IMAGE ROW.
//
// The row lives in the wave's registers from the load of the two half-spectra to the store of the
// next two: lane l, register r holds pixel (or frequency) 64*r + l, so every global access of a
// register is one contiguous 256-byte (real) or 512-byte (complex) wave access.  The transforms are
// wfft::fft4096 (two 64-point register transforms around one LDS transposition); D and D^T are
// lane rotations; the Hermitian split / merge of the two real rows packed into one complex row is
// a lane reversal.  A workgroup is a single wave, so there is no workgroup barrier anywhere and
// eight rows per CU progress independently, which is what overlaps one row's HBM phases with
// another row's arithmetic.
//
// Addressing: everything that depends on the row is uniform (scalar registers); the only per-lane
// part of an address is a 32-bit offset computed once, so a load costs no vector arithmetic.
#pragma once
#include "ddn_kernels.hpp"
#include "ddn_wfft.hpp"

namespace wrow {

// value of lane (l - 1) & 63 / (l + 1) & 63: one DPP move, no LDS traffic
__device__ __forceinline__ float from_prev_lane(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x13C, 0xF, 0xF, false));  // wave_ror:1
}
__device__ __forceinline__ float from_next_lane(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x134, 0xF, 0xF, false));  // wave_rol:1
}
__device__ __forceinline__ float lane_get(float v, int src) { return __shfl(v, src, 64); }
// keeps the scheduler from moving code across (it would otherwise hoist every load of a row to
// the top of a 10k-instruction block and spill hundreds of registers)
__device__ __forceinline__ void sched_fence() { __builtin_amdgcn_sched_barrier(0); }
// materialise the row here: without it the optimiser sinks the last butterfly level of a transform
// into the per-pixel code that follows and keeps the transform's inputs alive (and spilled) there
__device__ __forceinline__ void pin(cx<float> (&d)[64]) {
#pragma unroll
    for (int r = 0; r < 64; ++r) asm volatile("" : "+v"(d[r].x), "+v"(d[r].y));
}

// A uniform offset made opaque to the optimiser: address arithmetic on top of it stays inside the
// row loop as scalar adds instead of being hoisted out of it as hundreds of per-register offsets.
__device__ __forceinline__ size_t opaque(size_t o) {
    asm volatile("" : "+s"(o));
    return o;
}

// A uniform global pointer made opaque (kept in the global address space, so accesses through it stay
// global_load / global_store with a scalar base): blocks the re-association base + (lane offset) that
// would otherwise be hoisted out of the row loop as one 64-bit vector address per array.
template <typename Q>
__device__ __forceinline__ Q* opaque_ptr(Q* p) {
    auto g = (__attribute__((address_space(1))) Q*)p;
    asm volatile("" : "+s"(g));
    return (Q*)g;
}

// Accesses of the iterate arrays (element o of an array already moved to the register's pixels).  Every
// iterate element is read once and written once per iteration, so under DDN_NT_ITER these carry the
// non-temporal hint: the streams then leave the 256 MB last-level cache to the half-spectra and the
// multipliers, which the column stage and the next row launch find there (measured on primal DR at
// 4096 x 4096: column stage 0.137 -> 0.102 ms, row kernel 0.175 -> 0.165 ms).
__device__ __forceinline__ float ldi(const float* p, unsigned o) {
#ifdef DDN_NT_ITER
    return __builtin_nontemporal_load(p + o);
#else
    return p[o];
#endif
}
__device__ __forceinline__ void sti(float* p, unsigned o, float v) {
#ifdef DDN_NT_ITER
    __builtin_nontemporal_store(v, p + o);
#else
    p[o] = v;
#endif
}

// Spectral row addressing: element k = 64 r + lane of a row sits at rowbase + lane_off + r * rstride
// (tiles are at most 64 columns wide), rowbase and rstride uniform.
struct SpecAddr {
    unsigned lane_off;  // elements
    size_t rstride;     // elements per register step
    size_t nyq;         // offset of k = 2048
    __device__ __forceinline__ SpecAddr(const SpecLayout& l, int lane) {
        const size_t tstride = (size_t)l.H << l.cs;
        lane_off = (unsigned)((size_t)(lane >> l.cs) * tstride + (lane & ((1 << l.cs) - 1)));
        rstride = tstride << (6 - l.cs);
        nyq = (size_t)(2048 >> l.cs) * tstride;
    }
};
__device__ __forceinline__ size_t spec_row_base(const SpecLayout& l, int plane, int row) { return ((size_t)plane * l.ntile * l.H + row) << l.cs; }

// Merge: half-spectra A = F(a), B = F(b), k = 0..2048, of two real rows -> spectrum of a + i b.
// rowA / rowB: the arrays already offset to the row (uniform); TWO = false: b = 0.
typedef float f4 __attribute__((ext_vector_type(4)));

// ILV: the two half-spectra are interleaved element by element in ONE array (A[k], B[k] adjacent, 16
// bytes): rowA points at the pair array offset to the row (in pairs), rowB is unused.  A lane then
// moves both spectra of a frequency with one 16-byte access, and the column stage, whose workgroup
// owns 4 columns of both spectra, finds them in 64 contiguous bytes of a row instead of 2 x 32.
template <bool TWO, bool ILV = false>
__device__ __forceinline__ void pack(cx<float> (&d)[64], const cx<float>* rowA, const cx<float>* rowB, const SpecAddr& sa, int lane, const float* nyqA = nullptr, const float* nyqB = nullptr,
                                     bool nyq_vals = false, float nyq_a = 0.f, float nyq_b = 0.f) {
    // nyq_vals: the two column-W/2 terms are handed in as values (computed by the caller) instead of being read
    cx<float> Un, Vn;
    if (ILV) {
        const f4* const q = reinterpret_cast<const f4*>(rowA);
#pragma unroll
        for (int r = 0; r < 32; ++r) {
            const f4 v = (q + r * sa.rstride)[sa.lane_off];
            d[r] = mk<float>(v.x, v.y);
            d[63 - r] = mk<float>(v.z, v.w);
        }
        if (nyqA) {
            Un = mk<float>(*nyqA, 0.f);
            Vn = mk<float>(*nyqB, 0.f);
        } else {
            const f4 v = q[sa.nyq];
            Un = mk<float>(v.x, v.y);
            Vn = mk<float>(v.z, v.w);
        }
    } else {
#pragma unroll
        for (int r = 0; r < 32; ++r) {
            d[r] = (rowA + r * sa.rstride)[sa.lane_off];                         // A[k], k = 64 r + lane
            d[63 - r] = TWO ? (rowB + r * sa.rstride)[sa.lane_off] : mk<float>(0.f, 0.f);  // B[k]
        }
        // column W/2 (real): from its dense home when there is one, else from the last tile
        if (nyq_vals) {
            Un = mk<float>(nyq_a, 0.f);
            Vn = mk<float>(TWO ? nyq_b : 0.f, 0.f);
        } else {
            Un = nyqA ? mk<float>(*nyqA, 0.f) : rowA[sa.nyq];
            Vn = TWO ? (nyqB ? mk<float>(*nyqB, 0.f) : rowB[sa.nyq]) : mk<float>(0.f, 0.f);
        }
    }
    sched_fence();
    const int ml = (64 - lane) & 63;
    const float u0 = d[0].x, v0 = d[63].x;
#pragma unroll
    for (int r = 0; r < 32; ++r) {
        const cx<float> U = d[r], V = d[63 - r];
        d[r] = mk<float>(U.x - V.y, U.y + V.x);                       // Z[k]     = A[k] + i B[k]
        const float mx = U.x + V.y, my = V.x - U.y;                   // Z[W - k] = conj(A[k]) + i conj(B[k])
        d[63 - r] = mk<float>(lane_get(mx, ml), lane_get(my, ml));    // lands at lane (64 - l) & 63
    }
    // lane 0 holds k = 64 r: its mirror 64 (64 - r) sits one register higher than for the other lanes
#pragma unroll
    for (int r = 0; r < 31; ++r) {
        d[63 - r].x = lane == 0 ? d[62 - r].x : d[63 - r].x;
        d[63 - r].y = lane == 0 ? d[62 - r].y : d[63 - r].y;
    }
    // the mean and the Nyquist term of a real row are real
    d[32].x = lane == 0 ? Un.x : d[32].x;
    d[32].y = lane == 0 ? Vn.x : d[32].y;
    d[0].x = lane == 0 ? u0 : d[0].x;
    d[0].y = lane == 0 ? v0 : d[0].y;
    sched_fence();
}

// Split: spectrum Z of a + i b -> half-spectra of a (to rowA) and b (to rowB when TWO).
template <bool TWO, bool ILV = false>
__device__ __forceinline__ void unpack(const cx<float> (&d)[64], cx<float>* rowA, cx<float>* rowB, const SpecAddr& sa, int lane, float* nyqA = nullptr, float* nyqB = nullptr) {
    const int ml = (64 - lane) & 63;
    f4* const q = reinterpret_cast<f4*>(rowA);
#pragma unroll
    for (int r = 0; r < 32; ++r) {
        cx<float> Zm = mk<float>(lane_get(d[63 - r].x, ml), lane_get(d[63 - r].y, ml));
        const cx<float> own = (r == 0) ? d[0] : d[64 - r];
        Zm.x = lane == 0 ? own.x : Zm.x;
        Zm.y = lane == 0 ? own.y : Zm.y;
        const cx<float> Z = d[r];
        if (ILV) {
            f4 v;
            v.x = 0.5f * (Z.x + Zm.x); v.y = 0.5f * (Z.y - Zm.y);
            v.z = 0.5f * (Z.y + Zm.y); v.w = 0.5f * (Zm.x - Z.x);
            (q + r * sa.rstride)[sa.lane_off] = v;
        } else {
            (rowA + r * sa.rstride)[sa.lane_off] = mk<float>(0.5f * (Z.x + Zm.x), 0.5f * (Z.y - Zm.y));
            if (TWO) (rowB + r * sa.rstride)[sa.lane_off] = mk<float>(0.5f * (Z.y + Zm.y), 0.5f * (Zm.x - Z.x));
        }
    }
    if (lane == 0) {
        if (ILV) {
            if (nyqA) {
                *nyqA = d[32].x;
                *nyqB = d[32].y;
            } else {
                f4 v;
                v.x = d[32].x; v.y = 0.f; v.z = d[32].y; v.w = 0.f;
                q[sa.nyq] = v;
            }
        } else {
            if (nyqA) *nyqA = d[32].x; else rowA[sa.nyq] = mk<float>(d[32].x, 0.f);
            if (TWO) {
                if (nyqB) *nyqB = d[32].y; else rowB[sa.nyq] = mk<float>(d[32].y, 0.f);
            }
        }
    }
}

template <int NS>
__device__ __forceinline__ void wave_sum(double (&v)[NS]) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
#pragma unroll
        for (int s = 0; s < NS; ++s) v[s] += __shfl_xor(v[s], off, 64);
    }
}

// the policies' array pointers moved to the start of a row (uniform arithmetic)
__device__ __forceinline__ void advance(PdrP<float>::Args& a, size_t o) { a.z1 += o; a.z20 += o; a.z21 += o; a.z22 += o; a.b += o; }
__device__ __forceinline__ void advance(PddrP<float>::Args& a, size_t o) { a.p += o; a.q0 += o; a.q1 += o; a.q2 += o; a.b += o; }
__device__ __forceinline__ void advance(AdmmP<float>::Args& a, size_t o) {
    a.u += o; a.w += o; a.y0 += o; a.y1 += o; a.y2 += o; a.z0 += o; a.z1 += o; a.z2 += o; a.b += o;
    if (a.sol) a.sol += o;
}
__device__ __forceinline__ void hide(PdrP<float>::Args& a) { a.z1 = opaque_ptr(a.z1); a.z20 = opaque_ptr(a.z20); a.z21 = opaque_ptr(a.z21); a.z22 = opaque_ptr(a.z22); a.b = opaque_ptr(a.b); }
__device__ __forceinline__ void hide(PddrP<float>::Args& a) { a.p = opaque_ptr(a.p); a.q0 = opaque_ptr(a.q0); a.q1 = opaque_ptr(a.q1); a.q2 = opaque_ptr(a.q2); a.b = opaque_ptr(a.b); }
__device__ __forceinline__ void hide(AdmmP<float>::Args& a) {
    a.u = opaque_ptr(a.u); a.w = opaque_ptr(a.w); a.y0 = opaque_ptr(a.y0); a.y1 = opaque_ptr(a.y1); a.y2 = opaque_ptr(a.y2);
    a.z0 = opaque_ptr(a.z0); a.z1 = opaque_ptr(a.z1); a.z2 = opaque_ptr(a.z2); a.b = opaque_ptr(a.b);
    if (a.sol) a.sol = opaque_ptr(a.sol);
}
// the arguments with every array moved to register r of the row (pixel 64 r + lane is then element `lane`)
template <class A>
__device__ __forceinline__ A at_reg(const A& a, int r) {
    A q = a;
    // the advanced pointers are opaque: each access then takes the scalar-base form (one shared
    // lane-offset register) instead of one hoisted 64-bit vector address per array
    advance(q, (size_t)(64 * r));
    hide(q);
    return q;
}

// ----------------------------------------------------------------------------------------------
// PDR / PDDR / ADMM: the policies of ddn_kernels.hpp, one pixel per register.
// PD = depth of the iterate-array load pipeline (registers r .. r+PD-1 are in flight).
template <class P, int PD, bool CONS, bool PROD, int FE = 1, int WPW = 1, int SP = 0, bool ILV = false>
#ifndef WROW_MINW
#define WROW_MINW 2
#endif
__global__ void __launch_bounds__(64 * WPW, WROW_MINW) k_wave_row(RowCommon<float> c, typename P::Args a0) {
    // WPW waves per workgroup, one row each.  WPW = 4 goes with 4-column spectral tiles, where a
    // 128-byte line holds four consecutive rows of a tile: the four waves of a group then touch the
    // same lines at the same time (one L1 fill serves all four, the L2 merges their stores).
    extern __shared__ __align__(16) float wtiles[];
    const int lane = threadIdx.x & 63;
    const int wave = WPW == 1 ? 0 : __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float* const tile = wtiles + (size_t)wave * WFFT_TILE_FLOATS;
    const int nrows = c.batch * c.H;
    const int ngroups = (nrows + WPW - 1) / WPW;
    const SpecAddr sa(c.lay, lane);
    for (int grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
        if (WPW > 1) __syncthreads();
        const int row = grp * WPW + wave;
        if (row >= nrows) continue;
        const int plane = row / c.H;
        if (c.stop && c.stop[plane] >= 0) continue;
        typename P::Args a = a0;
        advance(a, opaque((size_t)row * 4096));
        const size_t sbase = opaque(spec_row_base(c.lay, plane, row - plane * c.H));
        cx<float>* const row0 = c.S0 + (ILV ? 2 : 1) * sbase;  // ILV: S0 holds the (S0, S1) pairs, 16 bytes each
        cx<float>* const row1 = c.S1 + sbase;
        const unsigned ul = (unsigned)lane;
        float* const nq0 = c.nyqd ? c.nyqd + ((size_t)plane * 2 * c.H + (row - plane * c.H)) : nullptr;
        float* const nq1 = c.nyqd ? nq0 + c.H : nullptr;
        cx<float> d[64];
        if (CONS) {
            pack<true, ILV>(d, row0, row1, sa, lane, nq0, nq1);
            wfft::fft4096<true>(d, c.wtw, tile, lane);
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
        float rot_prev = from_prev_lane(d[63].x);  // pixel -1 of lane 0 is pixel 4095: lane 63, register 63
        float rs_first = 0.f, rs_prev = 0.f;
#pragma unroll
        for (int r = 0; r < 64; ++r) {
            if (r % FE == 0) sched_fence();  // FE: pixel registers the scheduler may interleave
            float u[1] = {d[r].x}, k[1] = {d[r].y}, du[1], re[1], im[1], s[1];
            const float rot = from_prev_lane(u[0]);
            du[0] = u[0] - (lane == 0 ? rot_prev : rot);
            rot_prev = rot;
            P::template step_regs<1>(at_reg(a, r), CONS, PROD, ul, stq[r % PD], u, k, du, re, im, s, acc);
            if (r + PD < 64) P::template load<1>(at_reg(a, r + PD), ul, stq[r % PD]);
            // SP > 0: a never-taken use of an accumulator every SP registers.  Measured, not derived:
            // with it the register allocator keeps admm's whole row in registers (without: 67
            // registers per lane spilled, 9 % more memory traffic); pdr / pddr fit without and spill with.
            if (SP > 0 && CONS && r % (SP > 0 ? SP : 1) == (SP > 0 ? SP : 1) - 1 && r != 63 && c.batch < 0) c.partials[0] = acc[0];
            if (PROD) {
                // transform input: re + D^T s = re + s[j] - s[j+1]; s[j+1] sits in lane l+1, or for lane 63 in lane 0 of the next register
                const float rs = from_next_lane(s[0]);
                d[r] = mk<float>(re[0] + s[0], im[0]);
                if (r == 0) rs_first = rs;
                else d[r - 1].x -= (lane == 63 ? rs : rs_prev);
                rs_prev = rs;
                // computed now, not when the transform first needs it (the inputs would stay live instead)
                asm volatile("" : "+v"(d[r].x), "+v"(d[r].y), "+v"(rs_prev));
                if (r > 0) asm volatile("" : "+v"(d[r - 1].x));
            }
        }
        sched_fence();
        if (PROD) d[63].x -= (lane == 63 ? rs_first : rs_prev);
        if (CONS) {
            wave_sum<4>(acc);
            if (lane == 0) {
#pragma unroll
                for (int q = 0; q < 4; ++q) c.partials[(size_t)row * 4 + q] = acc[q];
            }
        }
        if (PROD) {
            pin(d);
            sched_fence();
            wfft::fft4096<false>(d, c.wtw, tile, lane);
            pin(d);
            unpack<true, ILV>(d, row0, row1, sa, lane, nq0, nq1);
        }
    }
}

// ----------------------------------------------------------------------------------------------
// Chambolle-Pock (chambolle_pock/algorithm.py:58-92): a wave owns two image rows, the two real
// slots of the complex transform.  STEP 0 (dual): consume K z, update y (cp:60-76), produce y0.
// STEP 1 (primal): consume K^T y0, update x and z (cp:79-87), produce z.  D z and D^T (y1 + y2) are
// lane rotations of the registers just loaded.
template <int SL>
__device__ __forceinline__ float& slot(cx<float>& e) { return SL == 0 ? e.x : e.y; }

template <int PD, int SL>
__device__ __forceinline__ void cp_dual(cx<float> (&d)[64], const CpArgs<float>& a, int lane) {
    struct Q { float z, y0, y1, y2, b; } q[PD];
    const unsigned lo = (unsigned)lane;
#define DDN_CP_LD(r, e) { (e).z = opaque_ptr(a.z + 64 * (r))[lo]; (e).y0 = ldi(opaque_ptr(a.y0 + 64 * (r)), lo); (e).y1 = ldi(opaque_ptr(a.y1 + 64 * (r)), lo); (e).y2 = ldi(opaque_ptr(a.y2 + 64 * (r)), lo); (e).b = opaque_ptr(a.b + 64 * (r))[lo]; }
    float zl = opaque_ptr(a.z + 64 * 63)[lo];
#pragma unroll
    for (int g = 0; g < PD; ++g) DDN_CP_LD(g, q[g])
    float rot_prev = from_prev_lane(zl);  // z[-1] of lane 0 = z[4095]
#pragma unroll
    for (int r = 0; r < 64; ++r) {
        sched_fence();
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

// partials: the row's four loss partials (summed over the wave here, written by lane 0)
template <int PD, int SL>
__device__ __forceinline__ void cp_primal(cx<float> (&d)[64], const CpArgs<float>& a, int lane, double* partials) {
    double acc[4] = {0, 0, 0, 0};
    struct Q { float x, b, y1, y2; } q[PD];
    const unsigned lo = (unsigned)lane;
#define DDN_CP_LD(r, e) { (e).x = ldi(opaque_ptr(a.x + 64 * (r)), lo); (e).b = opaque_ptr(a.b + 64 * (r))[lo]; (e).y1 = ldi(opaque_ptr(a.y1 + 64 * (r)), lo); (e).y2 = ldi(opaque_ptr(a.y2 + 64 * (r)), lo); }
#pragma unroll
    for (int g = 0; g < PD; ++g) DDN_CP_LD(g, q[g])
    // D^T(y1 + y2)[j] = s[j] - s[j+1]: register r is finished when s of register r+1 is known
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
        sched_fence();
        const Q e = q[r % PD];
        if (r + PD < 64) DDN_CP_LD(r + PD, q[r % PD])
        const float in = slot<SL>(d[r]);
        const float sc = e.y1 + e.y2;
        const float rs = from_next_lane(sc);
        if (r == 0) rs_first = rs;
        else DDN_CP_FIN(r - 1, lane == 63 ? rs : rs_prev)
        rs_prev = rs; in_prev = in; sc_prev = sc; x_prev = e.x; b_prev = e.b;
    }
    DDN_CP_FIN(63, lane == 63 ? rs_first : rs_prev)
#undef DDN_CP_FIN
#undef DDN_CP_LD
    sched_fence();
    // reduced and stored here, per slot: accumulators kept live across the other slot's pass
    // (and one reduction for both) cost 70 spilled registers per lane
    wave_sum<4>(acc);
    if (lane == 0) {
#pragma unroll
        for (int q = 0; q < 4; ++q) partials[q] = acc[q];
    }
}

template <int SL>
__device__ __forceinline__ void cp_load_z(cx<float> (&d)[64], const CpArgs<float>& a, int lane) {
#pragma unroll
    for (int r = 0; r < 64; ++r) slot<SL>(d[r]) = ldi(opaque_ptr(a.z + 64 * r), (unsigned)lane);
}

__device__ __forceinline__ void advance(CpArgs<float>& a, size_t o) { a.x += o; a.z += o; a.y0 += o; a.y1 += o; a.y2 += o; a.b += o; }


template <int PD, int STEP, bool CONS, bool PROD, int WPW = 1>
__global__ void __launch_bounds__(64 * WPW, 2) k_wave_cp(RowCommon<float> c, CpArgs<float> a0) {
    extern __shared__ __align__(16) float wtiles[];
    const int lane = threadIdx.x & 63, H = c.H;
    const int wave = WPW == 1 ? 0 : __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float* const tile = wtiles + (size_t)wave * WFFT_TILE_FLOATS;
    const int ppp = (H + 1) >> 1;
    const int npairs = c.batch * ppp;
    const SpecAddr sa(c.lay, lane);
    const int ngroups = (npairs + WPW - 1) / WPW;
    for (int grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
        if (WPW > 1) __syncthreads();
        const int pr = grp * WPW + wave;
        if (pr >= npairs) continue;
        const int plane = pr / ppp;
        const int r0 = (pr - plane * ppp) * 2;
        const bool two = r0 + 1 < H;
        const size_t row0 = (size_t)plane * H + r0;
        const size_t sb0 = spec_row_base(c.lay, plane, r0), sb1 = spec_row_base(c.lay, plane, r0 + 1);
        CpArgs<float> a = a0, a1 = a0;
        advance(a, opaque(row0 * 4096));
        advance(a1, opaque((row0 + 1) * 4096));
        cx<float>* const rw0 = c.S0 + opaque(sb0);
        cx<float>* const rw1 = c.S0 + opaque(sb1);
        // column W/2: the dense array [image][2][H]; with nyq_taps its halves alternate as input and output
        const bool fir = c.nyq_taps != nullptr;
        float* const nqa = c.nyqd ? c.nyqd + ((size_t)plane * 2 * H + ((fir && !c.nyq_flip) ? H : 0) + r0) : nullptr;
        float* const nqb = c.nyqd ? nqa + 1 : nullptr;
        cx<float> d[64];
        if (CONS) {
            float un = 0.f, vn = 0.f;
            if (fir) {
                // K or K^T on column W/2 as a correlation along the rows (uniform loads, a few dozen multiply-adds)
                const float* const in = c.nyqd + ((size_t)plane * 2 * H + (c.nyq_flip ? H : 0));
                for (int u = 0; u < c.nyq_kh; ++u) {
                    const int s = c.nyq_dir * (u - c.nyq_ph);
                    const int ra = (r0 + s + H) % H, rb = (r0 + 1 + s + H) % H;
                    const float t = c.nyq_taps[u];
                    un += t * in[ra];
                    vn += t * in[rb];
                }
                un *= c.nyq_scale;
                vn *= c.nyq_scale;
            }
            if (two) pack<true>(d, rw0, rw1, sa, lane, nqa, nqb, fir, un, vn);
            else pack<false>(d, rw0, nullptr, sa, lane, nqa, nullptr, fir, un, vn);
            wfft::fft4096<true>(d, c.wtw, tile, lane);
            pin(d);
        } else {
#pragma unroll
            for (int r = 0; r < 64; ++r) d[r] = mk<float>(0.f, 0.f);
        }
        sched_fence();
        if (STEP == 0) {
            cp_dual<PD, 0>(d, a, lane);
            if (two) cp_dual<PD, 1>(d, a1, lane);
        } else if (CONS) {
            cp_primal<PD, 0>(d, a, lane, c.partials + row0 * 4);
            if (two) cp_primal<PD, 1>(d, a1, lane, c.partials + (row0 + 1) * 4);
        } else {
            cp_load_z<0>(d, a, lane);
            if (two) cp_load_z<1>(d, a1, lane);
        }
        if (!two) {
#pragma unroll
            for (int r = 0; r < 64; ++r) d[r].y = 0.f;
        }
        if (PROD) {
            pin(d);
            sched_fence();
            wfft::fft4096<false>(d, c.wtw, tile, lane);
            pin(d);
            if (two) unpack<true>(d, rw0, rw1, sa, lane, nqa, nqb);
            else unpack<false>(d, rw0, nullptr, sa, lane, nqa, nullptr);
        }
    }
}

}  // namespace wrow

// Fused solver row kernel for rows of 16384 and 32768 pixels (float): ONE WORKGROUP PER IMAGE ROW, the row in LDS.
//
// A row of W complex values does not fit a compute unit's LDS, but every array that crosses the row transform is real:
// a real row of W pixels is the length-W/2 complex sequence z[m] = x[2m] + i x[2m+1] (128 KB for W = 32768), and its
// half-spectrum X[0 .. W/2] follows from Z = F(z) by one pass over the pairs (k, W/2 - k):
//     E[k] = (Z[k] + conj(Z[W/2-k])) / 2,  O[k] = (Z[k] - conj(Z[W/2-k])) / (2i),  X[k] = E[k] + w_W^k O[k],
// and back: Z'[k] = (X[k] + conj(X[W/2-k])) + i conj(w_W^k) (X[k] - conj(X[W/2-k])), whose unnormalised inverse
// transform holds the pixels of the unnormalised length-W inverse.  So the iteration's row stage (the reference's
// ifft2 / fft2 along the rows, apply_periodic_conv_2d.py:28, with the per-pixel updates of the solver loops between
// them) runs per row as
//     u  = inverse(S0)  -> registers (64 pixels per thread at 32768, four consecutive ones per 16-byte access)
//     Ku = inverse(S1)  -> read from LDS by the pixel step; D u is the two-tap difference of u (multiplying_matrix.py:55-59)
//     the solver policy's step (the same P::step as every other row kernel): prox / dual / relaxation updates, loss sums
//     S1 = forward(im), S0 = forward(re + D^T s)
// with 512 threads (256 at 8192 pixels) and nothing written to memory between the transforms
// -- the three-kernel split transform of ddn_long.hpp moves every transformed array through memory twice more.
#pragma once
#include "ddn_kernels.hpp"
#include "ddn_long_args.hpp"

namespace wholerow {

#ifndef DDN_WHOLE_G
#define DDN_WHOLE_G 8
#endif
// wavefronts per SIMD the 256-thread kernels (rows of 8192 pixels, 35 KB of LDS) are compiled for: 2 = 256 registers, two rows per CU
constexpr int kMinWaves256 = 2;

// Radix schedule of the LDS transforms (ddn_fft.hpp): 16 = radix-16 passes first (four passes at 16384), 8 = radix-8 passes
// (fewer live registers).  Measured: the policy kernels are 16-19 % faster with 8 at 4096 and 8192 points (rows of 8192 and
// 16384 pixels, where the radix-16 butterflies spill beside the row's registers) and 1-3 % faster with 16 at 16384 points; the
// Chambolle-Pock kernels, which hold no row in registers, take 16 everywhere.
template <int NH, bool POLICY>
constexpr int sch_for() { return (POLICY && NH <= 8192) ? 8 : 16; }

template <int NH>
__host__ __device__ constexpr size_t lds_bytes() {
    return (size_t)(NH + tw_lds_entries(NH)) * sizeof(cx<float>) + 16 * 4 * sizeof(double) + 16 * 17 * sizeof(float);
}

// i * a
__device__ __forceinline__ cx<float> times_i(cx<float> a) { return mk<float>(-a.y, a.x); }

// Half-spectrum X[0 .. NH] of a real row (global, tiled) -> Z' at the digit-reversed LDS positions the inverse
// transform starts from.  No barrier inside.  A thread owns the pairs (k, NH - k), k = tid + i kThreads < NH / 2, and
// issues all of its global loads before the first use (a loop of load-use-load would be a chain of memory round
// trips: 16 per pass; here groups of 24 loads); thread 0 also takes the self-paired term k = NH / 2.
template <int NH, int kThreads, int kSch>
__device__ __forceinline__ void spectrum_to_lds(cx<float>* L, const cx<float>* S, const SpecRow& sr, const cx<float>* wW, int tid) {
    // (an opaque copy of the thread index: the unrolled addresses below would otherwise be hoisted out of the row loop
    // as a hundred loop invariants and spilled)
    asm volatile("" : "+v"(tid));
    constexpr int PER = NH / 2 / kThreads, G = PER < DDN_WHOLE_G ? PER : DDN_WHOLE_G;    // loads in flight per thread: 3 G
    cx<float> mid = mk<float>(0.f, 0.f);
    if (tid == 0) mid = S[sr.at(NH / 2)];
#pragma unroll
    for (int i0 = 0; i0 < PER; i0 += G) {
        cx<float> a[G], b[G], w[G];
#pragma unroll
        for (int i = 0; i < G; ++i) {
            const int k = tid + (i0 + i) * kThreads;
            a[i] = S[sr.at(k)];
            b[i] = S[sr.at(NH - k)];
            w[i] = wW[k];
        }
#pragma unroll
        for (int i = 0; i < G; ++i) {
            const int k = tid + (i0 + i) * kThreads;
            if (k == 0) {
                // the mean and the Nyquist term of a real row are real
                L[SW(pos_of_freq_fixed<NH, kSch>(0))] = mk<float>(a[i].x + b[i].x, a[i].x - b[i].x);
            } else {
                const cx<float> sum = a[i] + cconj(b[i]), dif = a[i] - cconj(b[i]);
                L[SW(pos_of_freq_fixed<NH, kSch>(k))] = sum + times_i(mulc(dif, w[i]));                  // + i conj(w^k) (a - conj(b))
                L[SW(pos_of_freq_fixed<NH, kSch>(NH - k))] = cconj(sum) + times_i(cconj(dif) * w[i]);    // (b + conj(a)) - i w^k (b - conj(a))
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    if (tid == 0) {
        // k = NH / 2 pairs with itself; w^(NH/2) = -i, so i conj(w) = i * i = -1
        const cx<float> sum = mid + cconj(mid), dif = mid - cconj(mid);
        L[SW(pos_of_freq_fixed<NH, kSch>(NH / 2))] = sum - dif;
    }
}

// Z (digit-reversed LDS positions, after the forward transform) -> half-spectrum X[0 .. NH] in global memory.
template <int NH, int kThreads, int kSch>
__device__ __forceinline__ void lds_to_spectrum(const cx<float>* L, cx<float>* S, const SpecRow& sr, const cx<float>* wW, int tid) {
    asm volatile("" : "+v"(tid));
    constexpr int PER = NH / 2 / kThreads, G = PER < DDN_WHOLE_G ? PER : DDN_WHOLE_G;
#pragma unroll
    for (int i0 = 0; i0 < PER; i0 += G) {
        cx<float> w[G];
#pragma unroll
        for (int i = 0; i < G; ++i) w[i] = wW[tid + (i0 + i) * kThreads];
#pragma unroll
        for (int i = 0; i < G; ++i) {
            const int k = tid + (i0 + i) * kThreads;
            const cx<float> zk = L[SW(pos_of_freq_fixed<NH, kSch>(k))];
            if (k == 0) {
                S[sr.at(0)] = mk<float>(zk.x + zk.y, 0.f);
                S[sr.at(NH)] = mk<float>(zk.x - zk.y, 0.f);
            } else {
                const cx<float> zm = L[SW(pos_of_freq_fixed<NH, kSch>(NH - k))];
                const cx<float> E = (zk + cconj(zm)) * 0.5f;
                const cx<float> dif = (zk - cconj(zm)) * 0.5f;
                const cx<float> wO = w[i] * mk<float>(dif.y, -dif.x);      // w^k O, O = dif / i
                S[sr.at(k)] = E + wO;
                S[sr.at(NH - k)] = cconj(E - wO);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    if (tid == 0) {
        // k = NH / 2: E = Re Z, O = Im Z, w = -i: X = Re Z - i Im Z
        const cx<float> z = L[SW(pos_of_freq_fixed<NH, kSch>(NH / 2))];
        S[sr.at(NH / 2)] = cconj(z);
    }
}

template <class P, int NH, int kThreads>
__global__ void __launch_bounds__(kThreads, kThreads == 256 ? kMinWaves256 : 1) k_row_whole(WholeRowArgs a, typename P::Args pa) {
    extern __shared__ __align__(16) unsigned char smem[];
    constexpr int kSch = sch_for<NH, true>();
    constexpr int W = 2 * NH, NC = W / (4 * kThreads);       // NC chunks of four consecutive pixels per thread
    cx<float>* const L = reinterpret_cast<cx<float>*>(smem);
    cx<float>* const twa = L + NH;
    double* const red = reinterpret_cast<double*>(twa + tw_lds_entries(NH));
    float* const sfw = reinterpret_cast<float*>(red + 16 * 4);      // [chunk][wave]: the first s of every wave's lane 0
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const Tw<float> tw = tw_setup<float>(twa, a.twHalf, NH, tid, kThreads);
    const cx<float>* const wW = a.c.twW;                     // exp(-2 pi i k / W), k < W
    const bool consume = a.c.do_consume != 0, produce = a.c.do_produce != 0;
    const size_t nrows = (size_t)a.c.batch * a.c.H;
    for (size_t row = blockIdx.x; row < nrows; row += gridDim.x) {
        const int plane = (int)(row / a.c.H);
        if (a.c.stop && a.c.stop[plane] >= 0) continue;
        const SpecRow sr = spec_row(a.c.lay, plane, (int)(row - (size_t)plane * a.c.H));
        float u[NC][4], left[NC], sfirst[NC];
        int tq = tid;                                        // opaque per row: the chunk addresses stay inside the row loop
        asm volatile("" : "+v"(tq));
        __syncthreads();                                     // the previous row's transforms are done with the LDS
        if (consume) {
            spectrum_to_lds<NH, kThreads, kSch>(L, a.c.S0, sr, wW, tid);
            __syncthreads();
            fft_inverse_fixed<float, NH, kSch, 0>(L, 1, 0, tw, 0, tid, kThreads);
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const int p0 = 2 * (tq + kThreads * c);
                const cx<float> z0 = L[SW(p0)], z1 = L[SW(p0 + 1)];
                u[c][0] = z0.x; u[c][1] = z0.y; u[c][2] = z1.x; u[c][3] = z1.y;
                left[c] = L[SW((p0 + NH - 1) & (NH - 1))].y;         // pixel 2 p0 - 1 (circular)
            }
            __syncthreads();
            spectrum_to_lds<NH, kThreads, kSch>(L, a.c.S1, sr, wW, tid);
            __syncthreads();
            fft_inverse_fixed<float, NH, kSch, 0>(L, 1, 0, tw, 0, tid, kThreads);
        } else {
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                u[c][0] = u[c][1] = u[c][2] = u[c][3] = 0.f;
                left[c] = 0.f;
            }
        }
        double acc[4] = {0, 0, 0, 0};
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            // keeps the scheduler from hoisting every chunk's iterate loads to the top of the loop (hundreds of live registers)
            __builtin_amdgcn_sched_barrier(0);
            const int p0 = 2 * (tq + kThreads * c);
            float ku[4] = {0.f, 0.f, 0.f, 0.f};
            if (consume) {
                const cx<float> z0 = L[SW(p0)], z1 = L[SW(p0 + 1)];
                ku[0] = z0.x; ku[1] = z0.y; ku[2] = z1.x; ku[3] = z1.y;
            }
            const float du[4] = {u[c][0] - left[c], u[c][1] - u[c][0], u[c][2] - u[c][1], u[c][3] - u[c][2]};
            float re[4], im[4], s[4];
            P::template step<4>(pa, consume, produce, row * (size_t)W + (size_t)(2 * p0), u[c], ku, du, re, im, s, acc);
            if (produce) {
                // the other transform input goes straight to the LDS slots this thread has just read
                L[SW(p0)] = mk<float>(im[0], im[1]);
                L[SW(p0 + 1)] = mk<float>(im[2], im[3]);
                // re + D^T s = re + s[j] - s[j+1]; the last pixel's right neighbour belongs to the next thread
                u[c][0] = re[0] + s[0] - s[1];
                u[c][1] = re[1] + s[1] - s[2];
                u[c][2] = re[2] + s[2] - s[3];
                u[c][3] = re[3] + s[3];
                sfirst[c] = s[0];
            }
        }
        if (consume) {
            block_reduce<4>(acc, red, tid, kThreads);
            if (tid == 0) {
#pragma unroll
                for (int q = 0; q < 4; ++q) a.c.partials[row * 4 + q] = acc[q];
            }
        }
        if (produce) {
            if (lane == 0) {
#pragma unroll
                for (int c = 0; c < NC; ++c) sfw[c * 17 + wave] = sfirst[c];
            }
            __syncthreads();                                 // also: every thread's `im` is in the LDS
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                float r = __shfl_down(sfirst[c], 1, 64);
                // lane 63: the next wave's lane 0; the last thread's neighbour is thread 0's next chunk (circular)
                if (lane == 63) r = wave < kThreads / 64 - 1 ? sfw[c * 17 + wave + 1] : sfw[((c + 1) % NC) * 17];
                u[c][3] -= r;
            }
            fft_forward_fixed<float, NH, kSch, 0>(L, 1, 0, tw, 0, tid, kThreads);
            lds_to_spectrum<NH, kThreads, kSch>(L, a.c.S1, sr, wW, tid);
            __syncthreads();
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const int p0 = 2 * (tq + kThreads * c);
                L[SW(p0)] = mk<float>(u[c][0], u[c][1]);
                L[SW(p0 + 1)] = mk<float>(u[c][2], u[c][3]);
            }
            __syncthreads();
            fft_forward_fixed<float, NH, kSch, 0>(L, 1, 0, tw, 0, tid, kThreads);
            lds_to_spectrum<NH, kThreads, kSch>(L, a.c.S0, sr, wW, tid);
        }
    }
}

// Chambolle-Pock (chambolle_pock/algorithm.py:58-92) on the same row layout: one array each way.  STEP 0 (dual):
// consume K z, update y (cp:60-76), produce y0.  STEP 1 (primal): consume K^T y0, update x and z (cp:79-87), produce z.
// D z and D^T(y1 + y2) are differences of arrays the step only reads, so the neighbours of a chunk's first / last
// pixel come from memory (circular along the row).
template <int NH, int kThreads>
__global__ void __launch_bounds__(kThreads, kThreads == 256 ? kMinWaves256 : 1) k_row_whole_cp(WholeRowArgs a, CpArgs<float> p) {
    extern __shared__ __align__(16) unsigned char smem[];
    constexpr int kSch = sch_for<NH, false>();
    constexpr int W = 2 * NH, NC = W / (4 * kThreads);
    cx<float>* const L = reinterpret_cast<cx<float>*>(smem);
    cx<float>* const twa = L + NH;
    double* const red = reinterpret_cast<double*>(twa + tw_lds_entries(NH));
    const int tid = threadIdx.x;
    const Tw<float> tw = tw_setup<float>(twa, a.twHalf, NH, tid, kThreads);
    const cx<float>* const wW = a.c.twW;
    const bool consume = a.c.do_consume != 0, produce = a.c.do_produce != 0;
    const size_t nrows = (size_t)a.c.batch * a.c.H;
    for (size_t row = blockIdx.x; row < nrows; row += gridDim.x) {
        const int plane = (int)(row / a.c.H);
        const SpecRow sr = spec_row(a.c.lay, plane, (int)(row - (size_t)plane * a.c.H));
        const size_t ro = row * (size_t)W;
        int tq = tid;
        asm volatile("" : "+v"(tq));
        __syncthreads();
        if (consume) {
            spectrum_to_lds<NH, kThreads, kSch>(L, a.c.S0, sr, wW, tid);
            __syncthreads();
            fft_inverse_fixed<float, NH, kSch, 0>(L, 1, 0, tw, 0, tid, kThreads);
        }
        double acc[4] = {0, 0, 0, 0};
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            __builtin_amdgcn_sched_barrier(0);
            const int p0 = 2 * (tq + kThreads * c), n0 = 2 * p0;
            const size_t o = ro + n0;
            float in[4] = {0.f, 0.f, 0.f, 0.f}, out[4];
            if (consume) {
                const cx<float> z0 = L[SW(p0)], z1 = L[SW(p0 + 1)];
                in[0] = z0.x; in[1] = z0.y; in[2] = z1.x; in[3] = z1.y;
            }
            if (p.step == 0) {
                float z[4], b[4], y0[4], y1[4], y2[4];
                vload_keep<float, 4>(p.z + o, z);
                const float zprev = p.z[ro + (n0 == 0 ? W - 1 : n0 - 1)];
                vload_keep<float, 4>(p.b + o, b);
                vload<float, 4>(p.y0 + o, y0);
                vload<float, 4>(p.y1 + o, y1);
                vload<float, 4>(p.y2 + o, y2);
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const float dz = z[v] - (v == 0 ? zprev : z[v - 1]);                         // cp:61
                    const float g0 = y0[v] + p.s * in[v];                                        // cp:66
                    const float g1 = y1[v] + p.s * dz;
                    const float g2 = y2[v] + p.s * dz;
                    float pr0;
                    if (p.objective == 0) pr0 = soft(g0 / p.s - b[v], p.is); else pr0 = (g0 / p.s - b[v]) / p.l2div;   // cp:69-72
                    const float y0n = g0 - p.s * b[v] - p.s * pr0;
                    float i1, i2;
                    iso(g1 / p.s, g2 / p.s, p.isg, i1, i2);                                      // cp:75
                    y0[v] = y0n;
                    y1[v] = g1 - p.s * i1;
                    y2[v] = g2 - p.s * i2;
                    out[v] = y0n;
                }
                vstore<float, 4>(p.y0 + o, y0);
                vstore<float, 4>(p.y1 + o, y1);
                vstore<float, 4>(p.y2 + o, y2);
            } else if (consume) {
                float x[4], b[4], y1[4], y2[4], zn[4];
                vload<float, 4>(p.y1 + o, y1);
                vload<float, 4>(p.y2 + o, y2);
                const size_t on = ro + (n0 + 4 == W ? 0 : n0 + 4);
                const float scnext = p.y1[on] + p.y2[on];
                vload<float, 4>(p.x + o, x);
                vload_keep<float, 4>(p.b + o, b);
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const float sc = y1[v] + y2[v];
                    const float nx = v == 3 ? scnext : y1[v + 1] + y2[v + 1];
                    const float dty = sc - nx;                                                   // D^T(y1 + y2), cp:79
                    const float xn = clip01(x[v] - p.t * (in[v] + dty));                         // cp:83-86
                    zn[v] = 2.f * xn - x[v];                                                     // cp:87
                    const float d = xn - b[v];                                                   // cp:96
                    acc[0] += (double)t_abs(d); acc[1] += (double)d * (double)d;
                    x[v] = xn;
                    out[v] = zn[v];
                }
                vstore<float, 4>(p.x + o, x);
                vstore<float, 4>(p.z + o, zn);
            } else {
                vload_keep<float, 4>(p.z + o, out);
            }
            if (produce) {
                L[SW(p0)] = mk<float>(out[0], out[1]);
                L[SW(p0 + 1)] = mk<float>(out[2], out[3]);
            }
        }
        if (p.step == 1 && consume) {
            block_reduce<4>(acc, red, tid, kThreads);
            if (tid == 0) {
#pragma unroll
                for (int q = 0; q < 4; ++q) a.c.partials[row * 4 + q] = acc[q];
            }
        }
        if (produce) {
            __syncthreads();
            fft_forward_fixed<float, NH, kSch, 0>(L, 1, 0, tw, 0, tid, kThreads);
            lds_to_spectrum<NH, kThreads, kSch>(L, a.c.S0, sr, wW, tid);
        }
    }
}

}  // namespace wholerow

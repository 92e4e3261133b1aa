// Argument block of the wave-per-column stage (ddn_wcol.hpp) and the set-up kernel that makes the
// column-major multiplier copies it reads; kept apart from the kernel itself so that the host
// translation unit does not depend on the kernel's source.
#pragma once
#include "ddn_kernels.hpp"

namespace wcol {

// Column pitch of the column-major multiplier copies: 4096 + 64 elements, so that the columns the
// resident waves stream in lockstep do not all sit at the same offset modulo the DRAM channel interleave.
constexpr int kColPitch = 4096 + 64;

struct Args {
    cx<float>* S0;
    cx<float>* S1;
    const cx<float>* eKc;  // [Wh][kColPitch], natural frequency order along the column
    const cx<float>* ctab; // [Wh][2][32]: the blur taps transformed along the rows (wcol::taps_mult), or null = read eKc
    const float* ed2;      // |eD|^2 per column (Wh entries): the resolvent multiplier is recomputed per frequency,
    float cK, cD, gscale;  //   g = gscale / (1 + cK |e|^2 + cD ed2[k2])  (mm:67-72), which saves reading a third array
    const cx<float>* wtw;  // per-lane twiddles of the 4096-point transform
    const cx<float>* twH;  // twiddle table of the length-4096 LDS transform (dense column W/2), layout of tw_setup
    SpecLayout lay;
    int Wh, batch;
    float* nyqd;           // dense home of column W/2 (real parts, [batch][2 arrays][H]) or null = inside the last tile
    int conj_mult;
    float scale;
    LossFin lf;
    long long* dbg;        // diagnostic: per-workgroup phase stamps (16 per workgroup) or null
    int ilv;               // resolvent mode: S0 holds the (S0, S1) pairs interleaved element by element
    int Hfft;              // transform length when lay.H is only a pitch (0: lay.H is both)
    int rb_sh6;            // rows stored in blocks of 64 << rb_sh6 rows (one block per source rank of a row-slab
    long long rb_extra;    //   decomposition): block i starts i * rb_extra elements past its place in a plain tile; 0 = plain
    int stagger;           // > 0: the workgroups of every other group of eight tiles among the first stagger_blocks blocks start
    int stagger_blocks;    //   `stagger` ticks of the 100 MHz wall clock late (see k_wave_col)
};

// Column-major natural-order copies of the multipliers: outc[k2][k1] = tiled[pos(k1)][k2].
template <typename T>
__global__ void k_mult_colmajor(const cx<T>* tiled, const T* tiled_real, cx<T>* outc, T* outr, int H, int Wh, SpecLayout lay, ColMap cm) {
    const size_t total = (size_t)H * Wh;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int k2 = (int)(i / H), k1 = (int)(i - (size_t)k2 * H);
        const size_t o = spec_row(lay, 0, colpos_of_freq(cm, k1)).at(k2);
        const size_t j = (size_t)k2 * kColPitch + k1;
        if (outc) outc[j] = tiled[o];
        if (outr) outr[j] = tiled_real[o];
    }
}

}  // namespace wcol

// Argument block of the wave-per-column stage (ddn_wcol.hpp) and the set-up kernel that makes the
// column-major multiplier copies it reads; kept apart from the kernel itself so that the host
// translation unit does not depend on the kernel's source.
#pragma once
#include "ddn_kernels.hpp"

namespace wcol {

struct Args {
    cx<float>* S0;
    cx<float>* S1;
    const cx<float>* eKc;  // [Wh][4096], natural frequency order along the column
    const float* Gc;       // [Wh][4096]
    const cx<float>* wtw;  // per-lane twiddles of the 4096-point transform
    SpecLayout lay;
    int Wh, batch;
    int conj_mult;
    float scale;
    LossFin lf;
};

// Column-major natural-order copies of the multipliers: outc[k2][k1] = tiled[pos(k1)][k2].
template <typename T>
__global__ void k_mult_colmajor(const cx<T>* tiled, const T* tiled_real, cx<T>* outc, T* outr, int H, int Wh, SpecLayout lay, ColMap cm) {
    const size_t total = (size_t)H * Wh;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int k2 = (int)(i / H), k1 = (int)(i - (size_t)k2 * H);
        const size_t o = spec_row(lay, 0, colpos_of_freq(cm, k1)).at(k2);
        if (outc) outc[i] = tiled[o];
        if (outr) outr[i] = tiled_real[o];
    }
}

}  // namespace wcol

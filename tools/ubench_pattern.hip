// Access-pattern microbenchmark for the column stage: what does the memory system deliver when a workgroup of eight
// wavefronts (one per CU: 133 KB of LDS requested) reads 256 KB and writes it back, with no arithmetic in between, in
// the patterns the column kernels use or could use?  Array: [plane][tile 129][row 4096][16 columns] of 16-byte pairs
// (the interleaved half-spectra of a 4096 x 4096 image); a workgroup owns a quarter tile (4 columns = 64 bytes a row).
//   pattern 0  staged (k_wave_col): a wave instruction covers 16 rows x 64 bytes
//   pattern 1  per-lane rows (k_wave_colN): a wave instruction covers 64 rows x 16 bytes, four instructions per 64-byte piece
//   pattern 2  contiguous: the workgroup's 256 KB as one block (what 4-column tiles would give), 1 KB per wave instruction
//   pattern 3  half tile per workgroup pair emulation: 8 rows x 128 bytes per wave instruction (whole lines)
// and for one spectrum alone (8-byte elements, 128 bytes a row; the multiplier mode of Chambolle-Pock):
//   pattern 4  quarter tile, four waves, two workgroups per CU: a wave instruction covers 16 rows x 32 bytes
//   pattern 5  half tile, eight waves: a wave instruction covers 8 rows x 64 bytes
// build: hipcc --offload-arch=gfx950 -O3 -o build/ubench_pattern tools/ubench_pattern.hip ; run: build/ubench_pattern
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f4 __attribute__((ext_vector_type(4)));

#define CHECK(x)                                                                       \
    do {                                                                               \
        hipError_t e_ = (x);                                                           \
        if (e_ != hipSuccess) {                                                        \
            printf("%s: %s\n", #x, hipGetErrorString(e_));                             \
            exit(1);                                                                   \
        }                                                                              \
    } while (0)

constexpr int kTiles = 128, kRows = 4096, kPitch = 16;   // 128 whole tiles per plane (the partial tile is left out)

template <int PAT, bool WRITE>
__global__ void __launch_bounds__(512, 2) k_pat(f4* __restrict__ buf, int nplanes) {
    extern __shared__ float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int per_plane = kTiles * 4;
    const int plane = blockIdx.x / per_plane, b = blockIdx.x % per_plane;
    // the four quarters of a tile are 8 block indices apart (same XCD)
    const int quarter = 4 * (8 * (b >> 5) + (b & 7)) + ((b >> 3) & 3);
    const int tile = quarter >> 2, sub = quarter & 3;
    f4* const tbase = buf + ((size_t)plane * kTiles + tile) * kRows * kPitch;
    f4 v[32];
    size_t off[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) {
        if (PAT == 0) {
            const int wq = wv & 3, hh = wv >> 2;
            const int row = 64 * (2 * i + hh) + 16 * wq + (lane >> 2);
            off[i] = (size_t)row * kPitch + sub * 4 + (lane & 3);
        } else if (PAT == 1) {
            // wave wv: columns pair (wv & 3) as 16-byte pair index, rows 64 j + lane for j = 2 i + (wv >> 2)
            const int row = 64 * (2 * i + (wv >> 2)) + lane;
            off[i] = (size_t)row * kPitch + sub * 4 + (wv & 3);
        } else if (PAT == 2) {
            // contiguous 256 KB per workgroup: quarter q of the tile = rows [1024 q, 1024 q + 1024) x 16 columns
            off[i] = (size_t)sub * 1024 * kPitch + (size_t)(i * 8 + wv) * 64 + lane;
        } else {
            // whole 128-byte lines: 8 rows x 8 pairs per wave instruction; the workgroup takes the half tile (sub >> 1)
            // for half of the rows (sub & 1)
            const int row = (sub & 1) * 2048 + (i * 8 + wv) * 8 + (lane >> 3);
            off[i] = (size_t)row * kPitch + (sub >> 1) * 8 + (lane & 7);
        }
    }
#pragma unroll
    for (int i = 0; i < 32; ++i) v[i] = tbase[off[i]];
    if (WRITE) {
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            v[i].x += 1.f;
            tbase[off[i]] = v[i];
        }
    } else {
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 32; ++i) s += v[i].x + v[i].w;
        if (s == 12345.678f) lds[tid] = s;   // keeps the loads alive
    }
    if (nplanes < 0) lds[tid] = 0.f;
}

typedef float f2 __attribute__((ext_vector_type(2)));
template <int PAT, bool WRITE>
__global__ void __launch_bounds__(PAT == 4 ? 256 : 512, 2) k_pat8(f2* __restrict__ buf, int nplanes) {
    extern __shared__ float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    constexpr int PER_TILE = PAT == 4 ? 4 : 2;
    const int per_plane = kTiles * PER_TILE;
    const int plane = blockIdx.x / per_plane, b = blockIdx.x % per_plane;
    int tile, sub;
    if (PAT == 4) {
        const int quarter = 4 * (8 * (b >> 5) + (b & 7)) + ((b >> 3) & 3);
        tile = quarter >> 2; sub = quarter & 3;
    } else {
        const int oct = 2 * (8 * (b >> 4) + (b & 7)) + ((b >> 3) & 1);
        tile = oct >> 1; sub = oct & 1;
    }
    f2* const tbase = buf + ((size_t)plane * kTiles + tile) * kRows * kPitch;
    f2 v[64];
    unsigned off;
    if (PAT == 4) off = (unsigned)((16 * wv + (lane >> 2)) * kPitch + sub * 4 + (lane & 3));
    else off = (unsigned)((8 * wv + (lane >> 3)) * kPitch + sub * 8 + (lane & 7));
#pragma unroll
    for (int i = 0; i < 64; ++i) v[i] = (tbase + (size_t)(64 * i) * kPitch)[off];
    if (WRITE) {
#pragma unroll
        for (int i = 0; i < 64; ++i) {
            v[i].x += 1.f;
            (tbase + (size_t)(64 * i) * kPitch)[off] = v[i];
        }
    } else {
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 64; ++i) s += v[i].x + v[i].y;
        if (s == 12345.678f) lds[tid] = s;
    }
    if (nplanes < 0) lds[tid] = 0.f;
}

template <int PAT, bool WRITE>
double run8(f2* buf, int nplanes, int reps) {
    const int grid = nplanes * kTiles * (PAT == 4 ? 4 : 2);
    const size_t lds = PAT == 4 ? 66 * 1024 : 133 * 1024;
    const int nt = PAT == 4 ? 256 : 512;
    CHECK(hipFuncSetAttribute((const void*)k_pat8<PAT, WRITE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k_pat8<PAT, WRITE>), dim3(grid), dim3(nt), lds, 0, buf, nplanes);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((k_pat8<PAT, WRITE>), dim3(grid), dim3(nt), lds, 0, buf, nplanes);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    return ms / reps;
}

template <int PAT, bool WRITE>
double run(f4* buf, int nplanes, int reps) {
    const int grid = nplanes * kTiles * 4;
    const size_t lds = 133 * 1024;
    CHECK(hipFuncSetAttribute((const void*)k_pat<PAT, WRITE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k_pat<PAT, WRITE>), dim3(grid), dim3(512), lds, 0, buf, nplanes);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((k_pat<PAT, WRITE>), dim3(grid), dim3(512), lds, 0, buf, nplanes);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    return ms / reps;
}

int main() {
    for (int nplanes : {1, 8}) {
        const size_t n = (size_t)nplanes * kTiles * kRows * kPitch;
        f4* buf;
        CHECK(hipMalloc(&buf, n * sizeof(f4)));
        CHECK(hipMemset(buf, 0, n * sizeof(f4)));
        const double bytes = (double)n * sizeof(f4);
        const int reps = nplanes == 1 ? 50 : 10;
        double t;
        printf("planes %d: %.1f MB per pass\n", nplanes, bytes / 1e6);
#define ROW(P)                                                                                                      \
    t = run<P, false>(buf, nplanes, reps);                                                                          \
    printf("  pattern %d read        %8.1f us  %7.1f GB/s\n", P, t * 1e3, bytes / t / 1e6);                        \
    t = run<P, true>(buf, nplanes, reps);                                                                           \
    printf("  pattern %d read+write  %8.1f us  %7.1f GB/s (both directions)\n", P, t * 1e3, 2 * bytes / t / 1e6);
        ROW(0)
        ROW(1)
        ROW(2)
        ROW(3)
        {
            f2* b8 = reinterpret_cast<f2*>(buf);
            const double bytes8 = bytes / 2;     // the same tiles with 8-byte elements: half of the buffer
#define ROW8(P)                                                                                                     \
    t = run8<P, false>(b8, nplanes, reps);                                                                          \
    printf("  pattern %d read        %8.1f us  %7.1f GB/s\n", P, t * 1e3, bytes8 / t / 1e6);                       \
    t = run8<P, true>(b8, nplanes, reps);                                                                           \
    printf("  pattern %d read+write  %8.1f us  %7.1f GB/s (both directions)\n", P, t * 1e3, 2 * bytes8 / t / 1e6);
            ROW8(4)
            ROW8(5)
        }
        CHECK(hipFree(buf));
    }
    return 0;
}

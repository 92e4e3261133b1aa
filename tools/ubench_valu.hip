// Vector-instruction issue rate on one SIMD of gfx950, by instruction kind and by waves per SIMD.
// build: hipcc --offload-arch=gfx950 -O3 -o build/ubench_valu tools/ubench_valu.hip ; run on the GPU box.
// Every wave runs REPS x 64 independent instructions on 16 accumulators; cycles per instruction per SIMD =
// elapsed shader clocks / (instructions per wave x waves per SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float v2 __attribute__((ext_vector_type(2)));

#define REPS 512

template <int KIND>
__global__ void __launch_bounds__(1024) k(float* out, long long* cyc, float s) {
    float a[16];
    v2 p[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        a[i] = (float)(threadIdx.x + i);
        p[i].x = a[i];
        p[i].y = a[i] + 1.f;
    }
    v2 sv;
    sv.x = s;
    sv.y = s;
    __syncthreads();
    const long long t0 = clock64();
    for (int r = 0; r < REPS; ++r) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(a[i]) : "v"(s));
                if (KIND == 1) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(s));
                if (KIND == 2) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(p[i]) : "v"(sv));
                if (KIND == 3) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(sv));
                if (KIND == 4) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(sv));
                if (KIND == 5) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(s));
                if (KIND == 6) asm volatile("v_fmac_f32 %0, %1, %1" : "+v"(a[i]) : "v"(s));
                if (KIND == 7) asm volatile("v_add_f32_dpp %0, %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(s));
                if (KIND == 8) asm volatile("v_pk_fma_f32 %0, %0, %1, %0 op_sel:[0,1,0] op_sel_hi:[1,0,1]" : "+v"(p[i]) : "v"(sv));
                if (KIND == 9) asm volatile("v_mov_b32 %0, %1" : "+v"(a[i]) : "v"(s));
            }
        }
    }
    const long long t1 = clock64();
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc += a[i] + p[i].x + p[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int KIND>
void run(const char* name, float* out, long long* cyc) {
    const int wps_list[] = {1, 2, 4};
    printf("%-28s", name);
    for (int wps : wps_list) {
        const int threads = 64 * 4 * wps;  // one workgroup on one CU: wps waves on each of the four SIMDs
        hipLaunchKernelGGL(k<KIND>, dim3(1), dim3(threads), 0, 0, out, cyc, 1.0001f);
        hipLaunchKernelGGL(k<KIND>, dim3(1), dim3(threads), 0, 0, out, cyc, 1.0001f);
        hipDeviceSynchronize();
        std::vector<long long> h(threads / 64);
        hipMemcpy(h.data(), cyc, sizeof(long long) * h.size(), hipMemcpyDeviceToHost);
        long long mx = 0;
        for (long long v : h) mx = v > mx ? v : mx;
        // clock64() ticks at a fixed 100 MHz on this part; report both raw ticks per instruction and the kernel-level figure below
        printf("  %d waves/SIMD: %.3f ticks/instr/wave, %.3f per SIMD-instr", wps, (double)mx / (REPS * 64.0), (double)mx / (REPS * 64.0 * wps));
    }
    printf("\n");
}

// the same with s_memtime-free timing: wall-clock of a full-chip launch, instructions per second per SIMD
template <int KIND>
void run_chip(const char* name, float* out, long long* cyc) {
    printf("%-28s", name);
    for (int wps : {1, 2, 4}) {
        const int threads = 64 * 4 * wps;
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(threads), 0, 0, out, cyc, 1.0001f);
        hipEventRecord(e0);
        for (int it = 0; it < 10; ++it) hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(threads), 0, 0, out, cyc, 1.0001f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double per_simd_instr = (double)REPS * 64.0 * wps * 10;  // wave-instructions issued on one SIMD
        printf("  %d w/SIMD: %.2f ns per wave-instr per SIMD (%.2f cyc @2.4GHz)", wps, ms * 1e6 / per_simd_instr, ms * 1e6 / per_simd_instr * 2.4);
    }
    printf("\n");
}

int main() {
    float* out;
    long long* cyc;
    hipMalloc(&out, 256 * 1024 * sizeof(float));
    hipMalloc(&cyc, 256 * 16 * sizeof(long long));
    printf("one workgroup per CU, all 256 CUs, 10 launches timed with events:\n");
    run_chip<0>("v_fma_f32", out, cyc);
    run_chip<1>("v_add_f32", out, cyc);
    run_chip<5>("v_mul_f32", out, cyc);
    run_chip<6>("v_fmac_f32", out, cyc);
    run_chip<9>("v_mov_b32", out, cyc);
    run_chip<7>("v_add_f32 dpp quad_perm", out, cyc);
    run_chip<2>("v_pk_fma_f32", out, cyc);
    run_chip<8>("v_pk_fma_f32 op_sel", out, cyc);
    run_chip<3>("v_pk_add_f32", out, cyc);
    run_chip<4>("v_pk_mul_f32", out, cyc);
    return 0;
}

// Second vector-issue microbenchmark for gfx950: operand forms (VOP2 fmac / fmamk / literals), register-bank
// placement of the three sources of v_fma_f32, dependent chains, and vector instructions interleaved with LDS traffic.
// build: hipcc --offload-arch=gfx950 -O3 -o build/ubench_valu2 tools/ubench_valu2.hip
#include <hip/hip_runtime.h>
#include <cstdio>

#define REPS 512
#define STR2(x) #x
#define STR(x) STR2(x)

// sixteen instructions with explicit registers; dst v64..v79
#define I16(fmt_a, fmt_b)                                                                                              \
    fmt_a "64" fmt_b "\n" fmt_a "65" fmt_b "\n" fmt_a "66" fmt_b "\n" fmt_a "67" fmt_b "\n" fmt_a "68" fmt_b "\n" fmt_a "69" fmt_b "\n" \
    fmt_a "70" fmt_b "\n" fmt_a "71" fmt_b "\n" fmt_a "72" fmt_b "\n" fmt_a "73" fmt_b "\n" fmt_a "74" fmt_b "\n" fmt_a "75" fmt_b "\n" \
    fmt_a "76" fmt_b "\n" fmt_a "77" fmt_b "\n" fmt_a "78" fmt_b "\n" fmt_a "79" fmt_b "\n"

#define CLOB "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79", \
             "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87", "v88", "v89", "v90", "v91", "v92", "v93", "v94", "v95", "v96", "v97"

template <int KIND>
__global__ void __launch_bounds__(1024) k(float* out, float s) {
    __shared__ float lds[4096];
    lds[threadIdx.x] = s;
    // initialise the explicit registers
    asm volatile(I16("v_mov_b32 v", ", 1.0") ::: CLOB);
    asm volatile("v_mov_b32 v80, 1.0\nv_mov_b32 v81, 0.5\nv_mov_b32 v82, 0.25\nv_mov_b32 v83, 2.0\nv_mov_b32 v84, 1.0\nv_mov_b32 v85, 0.5\nv_mov_b32 v86, 0.25\nv_mov_b32 v87, 2.0\n"
                 "v_mov_b32 v88, 1.0\nv_mov_b32 v89, 0.5\nv_mov_b32 v92, 0.5\nv_mov_b32 v96, 0.5\n" ::: CLOB);
    unsigned la = (threadIdx.x & 63) * 4;
    __syncthreads();
    for (int r = 0; r < REPS; ++r) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (KIND == 0) asm volatile(I16("v_fma_f32 v", ", v80, v81, v82") ::: CLOB);            // three sources, three banks (if bank = index mod 4)
            if (KIND == 1) asm volatile(I16("v_fma_f32 v", ", v80, v84, v88") ::: CLOB);            // three sources, one bank
            if (KIND == 2) asm volatile(I16("v_fma_f32 v", ", v80, v84, v81") ::: CLOB);            // two in one bank
            if (KIND == 3) asm volatile(I16("v_fmac_f32 v", ", v80, v81") ::: CLOB);                // VOP2 fmac, distinct sources
            if (KIND == 4) asm volatile(I16("v_fmamk_f32 v", ", v80, 0x3f000000, v81") ::: CLOB);   // literal multiplier
            if (KIND == 5) asm volatile(I16("v_mul_f32 v", ", 0x3f0ccccd, v80") ::: CLOB);          // VOP2 with a 32-bit literal
            if (KIND == 6) asm volatile(I16("v_sub_f32 v", ", v80, v81") ::: CLOB);
            if (KIND == 7) asm volatile(I16("v_add_f32 v", ", v80, v84") ::: CLOB);                 // two sources in one bank
            if (KIND == 8) asm volatile(I16("v_fmaak_f32 v", ", v80, v81, 0x3f000000") ::: CLOB);
            if (KIND == 9) asm volatile(I16("v_rcp_f32 v", ", v80") ::: CLOB);
            if (KIND == 10)  // dependent chain of length 16 on one register
                asm volatile("v_fma_f32 v64, v64, v80, v81\nv_fma_f32 v64, v64, v80, v81\nv_fma_f32 v64, v64, v80, v81\nv_fma_f32 v64, v64, v80, v81\n"
                             "v_fma_f32 v64, v64, v80, v81\nv_fma_f32 v64, v64, v80, v81\nv_fma_f32 v64, v64, v80, v81\nv_fma_f32 v64, v64, v80, v81\n"
                             "v_fma_f32 v64, v64, v80, v81\nv_fma_f32 v64, v64, v80, v81\nv_fma_f32 v64, v64, v80, v81\nv_fma_f32 v64, v64, v80, v81\n"
                             "v_fma_f32 v64, v64, v80, v81\nv_fma_f32 v64, v64, v80, v81\nv_fma_f32 v64, v64, v80, v81\nv_fma_f32 v64, v64, v80, v81\n" ::: CLOB);
            if (KIND == 11)  // two interleaved dependent chains
                asm volatile("v_fma_f32 v64, v64, v80, v81\nv_fma_f32 v65, v65, v80, v81\nv_fma_f32 v64, v64, v80, v81\nv_fma_f32 v65, v65, v80, v81\n"
                             "v_fma_f32 v64, v64, v80, v81\nv_fma_f32 v65, v65, v80, v81\nv_fma_f32 v64, v64, v80, v81\nv_fma_f32 v65, v65, v80, v81\n"
                             "v_fma_f32 v64, v64, v80, v81\nv_fma_f32 v65, v65, v80, v81\nv_fma_f32 v64, v64, v80, v81\nv_fma_f32 v65, v65, v80, v81\n"
                             "v_fma_f32 v64, v64, v80, v81\nv_fma_f32 v65, v65, v80, v81\nv_fma_f32 v64, v64, v80, v81\nv_fma_f32 v65, v65, v80, v81\n" ::: CLOB);
            if (KIND == 12) {  // twelve vector instructions + four LDS reads of 4 bytes
                asm volatile("v_add_f32 v64, v80, v81\nv_add_f32 v65, v80, v81\nv_add_f32 v66, v80, v81\nds_read_b32 v76, %0\n"
                             "v_add_f32 v67, v80, v81\nv_add_f32 v68, v80, v81\nv_add_f32 v69, v80, v81\nds_read_b32 v77, %0 offset:256\n"
                             "v_add_f32 v70, v80, v81\nv_add_f32 v71, v80, v81\nv_add_f32 v72, v80, v81\nds_read_b32 v78, %0 offset:512\n"
                             "v_add_f32 v73, v80, v81\nv_add_f32 v74, v80, v81\nv_add_f32 v75, v80, v81\nds_read_b32 v79, %0 offset:768\ns_waitcnt lgkmcnt(8)\n" :: "v"(la) : CLOB);
            }
            if (KIND == 13) {  // twelve vector instructions + four LDS writes of 4 bytes
                asm volatile("v_add_f32 v64, v80, v81\nv_add_f32 v65, v80, v81\nv_add_f32 v66, v80, v81\nds_write_b32 %0, v80\n"
                             "v_add_f32 v67, v80, v81\nv_add_f32 v68, v80, v81\nv_add_f32 v69, v80, v81\nds_write_b32 %0, v81 offset:256\n"
                             "v_add_f32 v70, v80, v81\nv_add_f32 v71, v80, v81\nv_add_f32 v72, v80, v81\nds_write_b32 %0, v82 offset:512\n"
                             "v_add_f32 v73, v80, v81\nv_add_f32 v74, v80, v81\nv_add_f32 v75, v80, v81\nds_write_b32 %0, v83 offset:768\ns_waitcnt lgkmcnt(8)\n" :: "v"(la) : CLOB);
            }
            if (KIND == 14) {  // sixteen LDS reads only
                asm volatile("ds_read_b32 v64, %0\nds_read_b32 v65, %0 offset:256\nds_read_b32 v66, %0 offset:512\nds_read_b32 v67, %0 offset:768\n"
                             "ds_read_b32 v68, %0 offset:1024\nds_read_b32 v69, %0 offset:1280\nds_read_b32 v70, %0 offset:1536\nds_read_b32 v71, %0 offset:1792\n"
                             "ds_read_b32 v72, %0\nds_read_b32 v73, %0 offset:256\nds_read_b32 v74, %0 offset:512\nds_read_b32 v75, %0 offset:768\n"
                             "ds_read_b32 v76, %0 offset:1024\nds_read_b32 v77, %0 offset:1280\nds_read_b32 v78, %0 offset:1536\nds_read_b32 v79, %0 offset:1792\ns_waitcnt lgkmcnt(8)\n" :: "v"(la) : CLOB);
            }
            if (KIND == 15) {  // sixteen LDS writes only
                asm volatile("ds_write_b32 %0, v80\nds_write_b32 %0, v81 offset:256\nds_write_b32 %0, v82 offset:512\nds_write_b32 %0, v83 offset:768\n"
                             "ds_write_b32 %0, v80 offset:1024\nds_write_b32 %0, v81 offset:1280\nds_write_b32 %0, v82 offset:1536\nds_write_b32 %0, v83 offset:1792\n"
                             "ds_write_b32 %0, v80\nds_write_b32 %0, v81 offset:256\nds_write_b32 %0, v82 offset:512\nds_write_b32 %0, v83 offset:768\n"
                             "ds_write_b32 %0, v80 offset:1024\nds_write_b32 %0, v81 offset:1280\nds_write_b32 %0, v82 offset:1536\nds_write_b32 %0, v83 offset:1792\ns_waitcnt lgkmcnt(8)\n" :: "v"(la) : CLOB);
            }
            if (KIND == 16) {  // sixteen 8-byte LDS reads
                asm volatile("ds_read_b64 v[64:65], %0\nds_read_b64 v[66:67], %0 offset:512\nds_read_b64 v[68:69], %0 offset:1024\nds_read_b64 v[70:71], %0 offset:1536\n"
                             "ds_read_b64 v[72:73], %0 offset:2048\nds_read_b64 v[74:75], %0 offset:2560\nds_read_b64 v[76:77], %0 offset:3072\nds_read_b64 v[78:79], %0 offset:3584\n"
                             "ds_read_b64 v[64:65], %0\nds_read_b64 v[66:67], %0 offset:512\nds_read_b64 v[68:69], %0 offset:1024\nds_read_b64 v[70:71], %0 offset:1536\n"
                             "ds_read_b64 v[72:73], %0 offset:2048\nds_read_b64 v[74:75], %0 offset:2560\nds_read_b64 v[76:77], %0 offset:3072\nds_read_b64 v[78:79], %0 offset:3584\ns_waitcnt lgkmcnt(8)\n" :: "v"(la * 2) : CLOB);
            }
            if (KIND == 17) {  // sixteen 8-byte LDS writes
                asm volatile("ds_write_b64 %0, v[80:81]\nds_write_b64 %0, v[82:83] offset:512\nds_write_b64 %0, v[84:85] offset:1024\nds_write_b64 %0, v[86:87] offset:1536\n"
                             "ds_write_b64 %0, v[80:81] offset:2048\nds_write_b64 %0, v[82:83] offset:2560\nds_write_b64 %0, v[84:85] offset:3072\nds_write_b64 %0, v[86:87] offset:3584\n"
                             "ds_write_b64 %0, v[80:81]\nds_write_b64 %0, v[82:83] offset:512\nds_write_b64 %0, v[84:85] offset:1024\nds_write_b64 %0, v[86:87] offset:1536\n"
                             "ds_write_b64 %0, v[80:81] offset:2048\nds_write_b64 %0, v[82:83] offset:2560\nds_write_b64 %0, v[84:85] offset:3072\nds_write_b64 %0, v[86:87] offset:3584\ns_waitcnt lgkmcnt(8)\n" :: "v"(la * 2) : CLOB);
            }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    float acc;
    asm volatile("v_add_f32 %0, v64, v65\nv_add_f32 %0, %0, v66\nv_add_f32 %0, %0, v79" : "=v"(acc)::CLOB);
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc + lds[(threadIdx.x * 7) & 4095];
}

template <int KIND>
void run_chip(const char* name, float* out) {
    printf("%-44s", name);
    for (int wps : {1, 2, 4}) {
        const int threads = 64 * 4 * wps;
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0);
        (void)hipEventCreate(&e1);
        hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(threads), 0, 0, out, 1.0001f);
        (void)hipEventRecord(e0);
        for (int it = 0; it < 10; ++it) hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(threads), 0, 0, out, 1.0001f);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        const double per_simd_instr = (double)REPS * 64.0 * wps * 10;
        printf("  %dw: %.2f ns/instr/SIMD", wps, ms * 1e6 / per_simd_instr);
    }
    printf("\n");
}

int main() {
    float* out;
    (void)hipMalloc(&out, 256 * 1024 * sizeof(float));
    printf("ns per wave-instruction per SIMD (256 CUs x 4 SIMDs, 1 / 2 / 4 waves per SIMD); 0.42 ns = 1 clock at 2.4 GHz\n");
    run_chip<0>("v_fma_f32, sources in three banks", out);
    run_chip<1>("v_fma_f32, sources in one bank", out);
    run_chip<2>("v_fma_f32, two sources in one bank", out);
    run_chip<3>("v_fmac_f32 (VOP2)", out);
    run_chip<4>("v_fmamk_f32 (literal)", out);
    run_chip<8>("v_fmaak_f32 (literal)", out);
    run_chip<5>("v_mul_f32 with literal", out);
    run_chip<6>("v_sub_f32", out);
    run_chip<7>("v_add_f32, sources in one bank", out);
    run_chip<9>("v_rcp_f32", out);
    run_chip<10>("v_fma_f32 dependent chain", out);
    run_chip<11>("v_fma_f32 two interleaved chains", out);
    run_chip<12>("12 v_add + 4 ds_read_b32 (per 16)", out);
    run_chip<13>("12 v_add + 4 ds_write_b32 (per 16)", out);
    run_chip<14>("ds_read_b32 only", out);
    run_chip<15>("ds_write_b32 only", out);
    run_chip<16>("ds_read_b64 only", out);
    run_chip<17>("ds_write_b64 only", out);
    return 0;
}

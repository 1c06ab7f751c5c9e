// Does a flash-attention-SHAPED instruction stream (per 64-key tile and wave: 8 Sᵀ MFMAs in two dependent
// chains -> VALU block that reads them -> 8 PV MFMAs that read the VALU results) reach the throughput of the
// same instructions freely interleaved?  Register-only, every SIMD busy, 1..3 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>

#define MF(acc, a, b) "v_mfma_f32_32x32x16_bf16 v[" acc "], v[" a "], v[" b "], v[" acc "]\n"
#define CLOB "v0","v1","v2","v3","v4","v5","v6","v7","v8","v9","v10","v11","v12","v13","v14","v15", \
 "v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31", \
 "v32","v33","v34","v35","v36","v37","v38","v39","v40","v41","v42","v43","v44","v45","v46","v47", \
 "v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","v60","v61","v62","v63", \
 "v64","v65","v66","v67","v68","v69","v70","v71","v72","v73","v74","v75","v76","v77","v78","v79", \
 "v80","v81","v82","v83","v84","v85","v86","v87","v88","v89","v90","v91","v92","v93","v94","v95"

// S accumulators: v[0:15], v[16:31]; O accumulators v[32:47], v[48:63]; Q/K operands v[64:71]; P words v[72:79]
// VALU on S element i: fma -> exp -> (pairs) cvt_pk into P words
#define SM1(i, j, pw) "v_fma_f32 v" #i ", v" #i ", v88, v89\n v_exp_f32 v" #i ", v" #i "\n" \
                      "v_fma_f32 v" #j ", v" #j ", v88, v89\n v_exp_f32 v" #j ", v" #j "\n" \
                      "v_add_f32 v90, v90, v" #i "\n v_add_f32 v91, v91, v" #j "\n" \
                      "v_cvt_pk_bf16_f32 v" #pw ", v" #i ", v" #j "\n"
#define MAXS(i, j) "v_max3_f32 v92, v92, v" #i ", v" #j "\n"
#define SMAX_A MAXS(0,1) MAXS(2,3) MAXS(4,5) MAXS(6,7) MAXS(8,9) MAXS(10,11) MAXS(12,13) MAXS(14,15)
#define SMAX_B MAXS(16,17) MAXS(18,19) MAXS(20,21) MAXS(22,23) MAXS(24,25) MAXS(26,27) MAXS(28,29) MAXS(30,31)
#define SOFT_A SM1(0,1,72) SM1(2,3,73) SM1(4,5,74) SM1(6,7,75) SM1(8,9,76) SM1(10,11,77) SM1(12,13,78) SM1(14,15,79)
#define SOFT_B SM1(16,17,72) SM1(18,19,73) SM1(20,21,74) SM1(22,23,75) SM1(24,25,76) SM1(26,27,77) SM1(28,29,78) SM1(30,31,79)
#define S_MFMA MF("0:15","64:67","68:71") MF("16:31","64:67","68:71") MF("0:15","64:67","68:71") MF("16:31","64:67","68:71") \
               MF("0:15","64:67","68:71") MF("16:31","64:67","68:71") MF("0:15","64:67","68:71") MF("16:31","64:67","68:71")
#define PV_A MF("32:47","64:67","72:75") MF("48:63","64:67","72:75") MF("32:47","64:67","76:79") MF("48:63","64:67","76:79")
#define PV_B PV_A

template <int MODE>
__global__ __launch_bounds__(256) void spin(int iters, float *out) {
    asm volatile("v_mov_b32 v64, 0x3c003c00\n v_mov_b32 v65, 0x3c003c00\n v_mov_b32 v66, 0x3c003c00\n v_mov_b32 v67, 0x3c003c00\n"
                 "v_mov_b32 v68, 0x3c003c00\n v_mov_b32 v69, 0x3c003c00\n v_mov_b32 v70, 0x3c003c00\n v_mov_b32 v71, 0x3c003c00\n"
                 "v_mov_b32 v88, 0x3a000000\n v_mov_b32 v89, 0xbf000000\n v_mov_b32 v90, 0\n v_mov_b32 v91, 0\n v_mov_b32 v92, 0\n" ::: CLOB);
    for (int i = 0; i < iters; ++i) {
        // MODE 0: flash order as hipcc emits it: S MFMAs, max, softmax A, PV A interleaved late, softmax B, PV B
        if (MODE == 0) asm volatile(S_MFMA "s_nop 9\n" SMAX_A SMAX_B SOFT_A PV_A SOFT_B PV_B ::: CLOB);
        // MODE 1: MFMA-only part of it
        if (MODE == 1) asm volatile(S_MFMA PV_A PV_B ::: CLOB);
        // MODE 2: VALU-only part of it
        if (MODE == 2) asm volatile(SMAX_A SMAX_B SOFT_A SOFT_B ::: CLOB);
        // MODE 3: software-pipelined: this tile's softmax interleaved with the NEXT tile's S MFMAs and this tile's PV
        if (MODE == 3) asm volatile(
            MF("0:15","64:67","68:71") MAXS(80,81) MAXS(82,83) MAXS(84,85) MAXS(86,87)
            MF("16:31","64:67","68:71") MAXS(80,81) MAXS(82,83) MAXS(84,85) MAXS(86,87)
            MF("0:15","64:67","68:71") MAXS(80,81) MAXS(82,83) MAXS(84,85) MAXS(86,87)
            MF("16:31","64:67","68:71") MAXS(80,81) MAXS(82,83) MAXS(84,85) MAXS(86,87)
            MF("0:15","64:67","68:71") SM1(80,81,72) SM1(82,83,73)
            MF("16:31","64:67","68:71") SM1(84,85,74) SM1(86,87,75)
            MF("0:15","64:67","68:71") SM1(80,81,76) SM1(82,83,77)
            MF("16:31","64:67","68:71") SM1(84,85,78) SM1(86,87,79)
            MF("32:47","64:67","72:75") SM1(80,81,93) SM1(82,83,93)
            MF("48:63","64:67","72:75") SM1(84,85,93) SM1(86,87,93)
            MF("32:47","64:67","76:79") SM1(80,81,93) SM1(82,83,93)
            MF("48:63","64:67","76:79") SM1(84,85,93) SM1(86,87,93)
            MF("32:47","64:67","72:75") MF("48:63","64:67","72:75") MF("32:47","64:67","76:79") MF("48:63","64:67","76:79")
            ::: CLOB);
    }
    float r;
    asm volatile("s_nop 15\n s_nop 15\n v_add_f32 %0, v0, v90\n v_add_f32 %0, %0, v32" : "=v"(r) :: CLOB);
    if (r == 12345.f) out[0] = r;
}

template <int MODE>
void run(const char *name, int wgs_per_cu) {
    float *d; hipMalloc(&d, 4);
    const int iters = 4000, grid = 256 * wgs_per_cu;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(spin<MODE>, dim3(grid), dim3(256), 0, 0, 10, d);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL(spin<MODE>, dim3(grid), dim3(256), 0, 0, iters, d);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    // one iteration = one 32q x 64k tile at d=64 = 524288 flop per wave
    const double tiles = (double)grid * 4 * iters;
    printf("%-60s %d waves/SIMD: %7.3f ms  %6.0f ns per tile-wave-slot  %7.1f TFLOP/s-equivalent\n", name, wgs_per_cu, ms,
           ms * 1e6 / iters / wgs_per_cu, tiles * 524288.0 / ms / 1e9);
    hipFree(d);
}

int main() {
    for (int w : {1, 2, 3}) {
        run<0>("flash order (S mfma | max | softmax | PV late)", w);
        run<1>("its 16 MFMAs only", w);
        run<2>("its VALU only (16 max3, 32 fma, 32 exp, 32 add, 16 cvt)", w);
        run<3>("software-pipelined interleave (same counts)", w);
    }
    return 0;
}

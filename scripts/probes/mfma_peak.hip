// Sustained MFMA rate of the whole chip (clock under load included): every wave runs a register-only
// stream of v_mfma_f32_32x32x16_bf16, optionally with plain VALU / exp instructions in between.
// build: hipcc --offload-arch=gfx950 -O2 scripts/probes/mfma_peak.hip -o scripts/probes/mfma_peak.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <string>

#define MF(a) "v_mfma_f32_32x32x16_bf16 v[" a "], v[64:67], v[68:71], v[" a "]\n"
#define EXP4 "v_exp_f32 v80, v88\n v_exp_f32 v81, v89\n v_exp_f32 v82, v90\n v_exp_f32 v83, v91\n"
#define FMA4 "v_fma_f32 v84, v88, v90, v84\n v_fma_f32 v85, v88, v90, v85\n v_fma_f32 v86, v88, v90, v86\n v_fma_f32 v87, v88, v90, v87\n"
#define LDSR "ds_read_b128 v[72:75], v92\n"
#define CLOB "v72","v73","v74","v75","v92","v0","v1","v2","v3","v4","v5","v6","v7","v8","v9","v10","v11","v12","v13","v14","v15", \
 "v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31", \
 "v32","v33","v34","v35","v36","v37","v38","v39","v40","v41","v42","v43","v44","v45","v46","v47", \
 "v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","v60","v61","v62","v63", \
 "v64","v65","v66","v67","v68","v69","v70","v71","v80","v81","v82","v83","v84","v85","v86","v87","v88","v89","v90","v91"

template <int MODE>
__global__ __launch_bounds__(256) void spin(int iters, float *out) {
    asm volatile("v_mov_b32 v64, 0x3f803f80\n v_mov_b32 v65, 0x3f803f80\n v_mov_b32 v66, 0x3f803f80\n v_mov_b32 v67, 0x3f803f80\n"
                 "v_mov_b32 v68, 0x3c003c00\n v_mov_b32 v69, 0x3c003c00\n v_mov_b32 v70, 0x3c003c00\n v_mov_b32 v71, 0x3c003c00\n"
                 "v_mov_b32 v88, 0xbf800000\n v_mov_b32 v89, 0xbf800000\n v_mov_b32 v90, 0x3f000000\n v_mov_b32 v91, 0x3f000000\n" ::: CLOB);
    if (MODE >= 3) {   // lane l reads 16 bytes at l * 16 of a 16 KB LDS block (conflict-free ds_read_b128)
        __shared__ char lds_block[16384];
        asm volatile("v_mov_b32 v92, %0" ::"v"((unsigned)(size_t)(__attribute__((address_space(3))) char *)lds_block + (threadIdx.x & 1023) * 16) : CLOB);
    }
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) asm volatile(".rept 4\n" MF("0:15") MF("16:31") MF("32:47") MF("48:63") ".endr\n" ::: CLOB);
        if (MODE == 1) asm volatile(".rept 4\n" MF("0:15") EXP4 MF("16:31") FMA4 MF("32:47") EXP4 MF("48:63") FMA4 ".endr\n" ::: CLOB);
        if (MODE == 2) asm volatile(".rept 4\n" MF("0:15") EXP4 FMA4 FMA4 MF("16:31") EXP4 FMA4 FMA4 MF("32:47") EXP4 FMA4 FMA4 MF("48:63") EXP4 FMA4 FMA4 ".endr\n" ::: CLOB);
        // round 6: the operand traffic of an attention tile -- 1 KB of LDS per MFMA (one ds_read_b128 per wave and MFMA; the data
        // lands in scratch registers, the MFMA operands stay put, so nothing waits) -- alone (3) and with a softmax-like VALU
        // load of 2 exp + 8 fma per MFMA (4)
        if (MODE == 3) asm volatile(".rept 4\n" MF("0:15") LDSR MF("16:31") LDSR MF("32:47") LDSR MF("48:63") LDSR ".endr\n s_waitcnt lgkmcnt(0)\n" ::: CLOB);
        if (MODE == 4) asm volatile(".rept 4\n" MF("0:15") LDSR EXP4 FMA4 MF("16:31") LDSR FMA4 FMA4 MF("32:47") LDSR EXP4 FMA4 MF("48:63") LDSR FMA4 FMA4 ".endr\n s_waitcnt lgkmcnt(0)\n" ::: CLOB);
    }
    float r;
    asm volatile("s_nop 15\n s_nop 15\n v_add_f32 %0, v0, v80" : "=v"(r) :: CLOB);
    if (r == 12345.f) out[0] = r;
}

template <int MODE>
void run(const char *name, int wgs_per_cu) {
    float *d; hipMalloc(&d, 4);
    const int iters = 2000, grid = 256 * wgs_per_cu;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(spin<MODE>, dim3(grid), dim3(256), 0, 0, 10, d);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL(spin<MODE>, dim3(grid), dim3(256), 0, 0, iters, d);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double mfma = (double)grid * 4 * iters * 16;
    const double tf = mfma * 32768.0 / ms / 1e9;
    printf("%-44s %d waves/SIMD: %8.3f ms  %8.1f TFLOP/s  -> %5.2f GHz-equivalent at 32 cyc/MFMA\n", name, wgs_per_cu, ms, tf,
           mfma / (1024.0 * wgs_per_cu) * wgs_per_cu * 32 / ms / 1e6);
    hipFree(d);
}

// `mfma_peak.bin sustain MODE SECONDS`: one stream back to back for SECONDS (scripts/mfma_stream_clock.py samples the shader
// clock and socket power meanwhile: the MFMA-only row of the power-limit evidence, round 6)
template <int MODE>
void sustain(double seconds) {
    float *d; hipMalloc(&d, 4);
    const int iters = 2000, grid = 256 * 4;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(spin<MODE>, dim3(grid), dim3(256), 0, 0, 10, d);
    hipDeviceSynchronize();
    double total_ms = 0; long n = 0;
    while (total_ms < seconds * 1e3) {
        hipEventRecord(a);
        for (int i = 0; i < 8; ++i) hipLaunchKernelGGL(spin<MODE>, dim3(grid), dim3(256), 0, 0, iters, d);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        total_ms += ms; n += 8;
    }
    const double mfma = (double)grid * 4 * iters * 16 * n;
    printf("{\"mode\": %d, \"launches\": %ld, \"ms\": %.3f, \"tflops\": %.1f}\n", MODE, n, total_ms, mfma * 32768.0 / total_ms / 1e9);
    hipFree(d);
}

int main(int argc, char **argv) {
    if (argc >= 4 && std::string(argv[1]) == "sustain") {
        const int mode = atoi(argv[2]); const double sec = atof(argv[3]);
        if (mode == 0) sustain<0>(sec); else if (mode == 1) sustain<1>(sec); else if (mode == 2) sustain<2>(sec);
        else if (mode == 3) sustain<3>(sec); else sustain<4>(sec);
        return 0;
    }
    for (int w : {1, 2, 3}) {
        run<0>("mfma only", w);
        run<1>("mfma + 2 exp + 2 fma per mfma", w);
        run<2>("mfma + 4 exp + 8 fma per mfma", w);
    }
    return 0;
}

// Issue-rate / overlap probe for gfx950: how long do blocks of VALU, transcendental and MFMA instructions
// take for one wave alone and for two waves on one SIMD, and do MFMA and VALU overlap?
// build: hipcc --offload-arch=gfx950 -O2 scripts/probes/pipe_probe.hip -o /tmp/pipe_probe ; run: /tmp/pipe_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define CLOB "v0","v1","v2","v3","v4","v5","v6","v7","v8","v9","v10","v11","v12","v13","v14","v15", \
 "v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31", \
 "v32","v33","v34","v35","v36","v37","v38","v39","v40","v41","v42","v43","v44","v45","v46","v47", \
 "v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","v60","v61","v62","v63", \
 "v64","v65","v66","v67","v68","v69","v70","v71","v72","v73","v74","v75","v76","v77","v78","v79", \
 "v80","v81","v82","v83","v84","v85","v86","v87","v88","v89","v90","v91","v93","v94","v95","memory"

#define T_BEGIN "s_waitcnt vmcnt(0) lgkmcnt(0)\n s_barrier\n s_memtime %0\n s_waitcnt lgkmcnt(0)\n"
#define T_END   "s_nop 15\n s_nop 15\n v_readfirstlane_b32 s20, v0\n v_readfirstlane_b32 s21, v80\n s_memtime %1\n s_waitcnt lgkmcnt(0)\n"

#define EXP4 "v_exp_f32 v80, v88\n v_exp_f32 v81, v89\n v_exp_f32 v82, v90\n v_exp_f32 v83, v91\n"
#define PKF4 "v_pk_fma_f32 v[80:81], v[88:89], v[90:91], v[80:81]\n v_pk_fma_f32 v[82:83], v[88:89], v[90:91], v[82:83]\n" \
             "v_pk_fma_f32 v[84:85], v[88:89], v[90:91], v[84:85]\n v_pk_fma_f32 v[86:87], v[88:89], v[90:91], v[86:87]\n"
#define FMA4 "v_fma_f32 v80, v88, v90, v80\n v_fma_f32 v81, v88, v90, v81\n v_fma_f32 v82, v88, v90, v82\n v_fma_f32 v83, v88, v90, v83\n"
#define CVT4 "v_cvt_pk_bf16_f32 v80, v88, v89\n v_cvt_pk_bf16_f32 v81, v90, v91\n v_cvt_pk_bf16_f32 v82, v88, v89\n v_cvt_pk_bf16_f32 v83, v90, v91\n"
#define MAX4 "v_max3_f32 v80, v88, v89, v80\n v_max3_f32 v81, v88, v89, v81\n v_max3_f32 v82, v88, v89, v82\n v_max3_f32 v83, v88, v89, v83\n"
#define MUL4 "v_mul_f32 v80, v88, v80\n v_mul_f32 v81, v88, v81\n v_mul_f32 v82, v88, v82\n v_mul_f32 v83, v88, v83\n"
#define ADD4 "v_add_f32 v80, v88, v80\n v_add_f32 v81, v88, v81\n v_add_f32 v82, v88, v82\n v_add_f32 v83, v88, v83\n"
#define PKM4 "v_pk_mul_f32 v[80:81], v[88:89], v[80:81]\n v_pk_mul_f32 v[82:83], v[88:89], v[82:83]\n" \
             "v_pk_mul_f32 v[84:85], v[88:89], v[84:85]\n v_pk_mul_f32 v[86:87], v[88:89], v[86:87]\n"
#define PKA4 "v_pk_add_f32 v[80:81], v[88:89], v[80:81]\n v_pk_add_f32 v[82:83], v[88:89], v[82:83]\n" \
             "v_pk_add_f32 v[84:85], v[88:89], v[84:85]\n v_pk_add_f32 v[86:87], v[88:89], v[86:87]\n"
#define DSR4 "ds_read_b128 v[72:75], v92\n ds_read_b128 v[76:79], v92 offset:1024\n ds_read_b64_tr_b16 v[84:85], v92 offset:2048\n ds_read_b64_tr_b16 v[86:87], v92 offset:4096\n"
#define MF(a) "v_mfma_f32_32x32x16_bf16 v[" a "], v[64:67], v[68:71], v[" a "]\n"
#define MF4 MF("0:15") MF("16:31") MF("32:47") MF("48:63")
#define MFDEP MF("0:15") MF("0:15") MF("0:15") MF("0:15")

#define CASE(id, body)                                                                   \
    if (which == id) {                                                                   \
        asm volatile(T_BEGIN ".rept 16\n" body ".endr\n" T_END : "=s"(t0), "=s"(t1) : : CLOB, "s20", "s21"); \
    }

__global__ void probe(int which, int mode, unsigned long long *out) {
    unsigned long long t0 = 0, t1 = 0;
    const int wave = threadIdx.x >> 6;
    __shared__ char lds[16384];
    asm volatile("v_mov_b32 v92, %0" :: "v"((unsigned)(threadIdx.x & 63) * 16u + (unsigned)(size_t)lds) : "v92");
    // mode 1: odd/even waves (waves w and w+4 share a SIMD) run different bodies: which = a | (b << 8)
    if (mode == 1) which = (wave >= 4) ? (which >> 8) : (which & 0xff);
    CASE(0, EXP4 EXP4 EXP4 EXP4)                     // 256 exp
    CASE(1, PKF4 PKF4 PKF4 PKF4)                     // 256 pk_fma
    CASE(2, FMA4 FMA4 FMA4 FMA4)                     // 256 fma
    CASE(3, CVT4 CVT4 CVT4 CVT4)                     // 256 cvt_pk
    CASE(4, MAX4 MAX4 MAX4 MAX4)                     // 256 max3
    CASE(5, MF4)                                     // 64 mfma independent
    CASE(6, MFDEP)                                   // 64 mfma dependent chain
    CASE(7, MF("0:15") EXP4 MF("16:31") EXP4 MF("32:47") EXP4 MF("48:63") EXP4)        // 64 mfma + 256 exp
    CASE(8, MF("0:15") PKF4 MF("16:31") PKF4 MF("32:47") PKF4 MF("48:63") PKF4)        // 64 mfma + 256 pk_fma
    CASE(9, MF("0:15") EXP4 PKF4 MF("16:31") EXP4 PKF4 MF("32:47") EXP4 PKF4 MF("48:63") EXP4 PKF4)  // + both
    CASE(10, MF("0:15") FMA4 FMA4 MF("16:31") FMA4 FMA4 MF("32:47") FMA4 FMA4 MF("48:63") FMA4 FMA4) // 64 mfma + 512 fma
    CASE(11, EXP4 PKF4 EXP4 PKF4 EXP4 PKF4 EXP4 PKF4)                                   // 256 exp + 256 pk_fma
    CASE(12, MF("0:15") EXP4 EXP4 MF("16:31") EXP4 EXP4 MF("32:47") EXP4 EXP4 MF("48:63") EXP4 EXP4) // 64 mfma + 512 exp
    CASE(13, EXP4 FMA4 EXP4 FMA4 EXP4 FMA4 EXP4 FMA4)                                   // 256 exp + 256 fma
    CASE(14, MF("0:15") CVT4 MF("16:31") CVT4 MF("32:47") CVT4 MF("48:63") CVT4)        // 64 mfma + 256 cvt
    CASE(15, MF("0:15") MAX4 MF("16:31") MAX4 MF("32:47") MAX4 MF("48:63") MAX4)        // 64 mfma + 256 max3
    CASE(16, MF("0:15") PKM4 MF("16:31") PKM4 MF("32:47") PKM4 MF("48:63") PKM4)        // 64 mfma + 256 pk_mul
    CASE(17, MF("0:15") PKA4 MF("16:31") PKA4 MF("32:47") PKA4 MF("48:63") PKA4)        // 64 mfma + 256 pk_add
    CASE(18, MF("0:15") MUL4 MF("16:31") MUL4 MF("32:47") MUL4 MF("48:63") MUL4)        // 64 mfma + 256 mul
    CASE(19, MF("0:15") ADD4 MF("16:31") ADD4 MF("32:47") ADD4 MF("48:63") ADD4)        // 64 mfma + 256 add
    CASE(20, MF("0:15") FMA4 MF("16:31") FMA4 MF("32:47") FMA4 MF("48:63") FMA4)        // 64 mfma + 256 fma
    CASE(21, MF("0:15") DSR4 MF("16:31") DSR4 MF("32:47") DSR4 MF("48:63") DSR4)        // 64 mfma + 256 ds_read
    CASE(22, DSR4 DSR4 DSR4 DSR4)                                                       // 256 ds_read
    CASE(23, MF("0:15") EXP4 FMA4 CVT4 MF("16:31") EXP4 FMA4 MAX4 MF("32:47") EXP4 FMA4 ADD4 MF("48:63") EXP4 FMA4 MUL4) // 64 mfma + 256 exp + 512 other
    CASE(24, PKM4 PKM4 PKM4 PKM4)                                                       // 256 pk_mul
    CASE(25, ADD4 ADD4 ADD4 ADD4)                                                       // 256 add
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + wave] = t1 - t0;
}

int main() {
    unsigned long long *d;
    hipMalloc(&d, 16 * sizeof(unsigned long long));
    const char *names[] = {"256 v_exp_f32", "256 v_pk_fma_f32", "256 v_fma_f32", "256 v_cvt_pk_bf16_f32", "256 v_max3_f32",
                           "64 mfma32x32x16 indep", "64 mfma dep chain", "64 mfma + 256 exp", "64 mfma + 256 pk_fma",
                           "64 mfma + 256 exp + 256 pk_fma", "64 mfma + 512 fma", "256 exp + 256 pk_fma", "64 mfma + 512 exp", "256 exp + 256 fma", "64 mfma + 256 cvt_pk",
                           "64 mfma + 256 max3", "64 mfma + 256 pk_mul", "64 mfma + 256 pk_add", "64 mfma + 256 mul", "64 mfma + 256 add",
                           "64 mfma + 256 fma", "64 mfma + 256 ds_read", "256 ds_read (128 b128 + 128 tr_b64)",
                           "64 mfma + 256 exp + 512 plain VALU", "256 v_pk_mul_f32", "256 v_add_f32"};
    for (int threads : {256, 512}) {
        printf("--- %d threads (%d wave(s) per SIMD), same body in every wave; s_memtime ticks of wave 0\n", threads,
               (threads + 255) / 256);
        for (int w = 0; w < 26; ++w) {
            hipMemset(d, 0, 128);
            hipLaunchKernelGGL(probe, dim3(1), dim3(threads), 0, 0, w, 0, d);
            unsigned long long h[16];
            hipMemcpy(h, d, 128, hipMemcpyDeviceToHost);
            printf("%-34s %8llu\n", names[w], h[0]);
        }
    }
    printf("--- 512 threads, waves 0-3 run A, waves 4-7 (same SIMDs) run B: ticks of wave 0 / wave 4\n");
    int pairs[][2] = {{5, 0}, {5, 1}, {5, 2}, {0, 2}, {0, 25}, {5, 3}, {5, 4}, {5, 24}, {5, 25}, {5, 22}, {23, 23}, {20, 20}};
    for (auto &pr : pairs) {
        hipMemset(d, 0, 128);
        hipLaunchKernelGGL(probe, dim3(1), dim3(512), 0, 0, pr[0] | (pr[1] << 8), 1, d);
        unsigned long long h[16];
        hipMemcpy(h, d, 128, hipMemcpyDeviceToHost);
        printf("A=%-30s B=%-30s %8llu %8llu\n", names[pr[0]], names[pr[1]], h[0], h[4]);
    }
    return 0;
}

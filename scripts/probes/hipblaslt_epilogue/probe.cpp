// Does hipBLASLt run the reference's fused-dense epilogues (GELU with auxiliary pre-activation output + bias;
// dGELU + bias gradient) at the speed of its plain GEMM on the trunk-MLP shapes?  (csrc/fused_dense_lib does this with
// cuBLASLt: fused_dense.cpp:195-197.)  Row-major torch layout mapped to column-major: D^T = op(A) op(B).
//   hipcc --offload-arch=gfx950 -O2 probe.cpp -lhipblaslt -o probe.bin && ./probe.bin [M]
#include <hip/hip_runtime.h>
#include <hipblaslt/hipblaslt.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { auto e_ = (x); if (e_ != 0) { printf("FAIL %s = %d (line %d)\n", #x, (int)e_, __LINE__); exit(1); } } while (0)

struct Problem { const char *name; int m, n, k; hipblasOperation_t ta, tb; int lda, ldb; hipblasLtEpilogue_t epi; bool aux, bias; };

int main(int argc, char **argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 32768, D = 768, H = 3072;
    hipblasLtHandle_t h; CK(hipblasLtCreate(&h));
    const size_t big = (size_t)M * H * 2 + 1024;
    void *x, *w1, *w2, *pre, *hid, *g, *bias; float *bgrad; void *ws; const size_t ws_bytes = 256u << 20;
    CK(hipMalloc(&x, (size_t)M * D * 2)); CK(hipMalloc(&w1, (size_t)H * D * 2)); CK(hipMalloc(&w2, (size_t)D * H * 2));
    CK(hipMalloc(&pre, big)); CK(hipMalloc(&hid, big)); CK(hipMalloc(&g, (size_t)M * D * 2)); CK(hipMalloc(&bias, H * 4));
    CK(hipMalloc(&bgrad, H * 4)); CK(hipMalloc(&ws, ws_bytes));
    // small pseudo-random bf16 fills (never zeros: zero-filled inputs clock higher)
    std::vector<uint16_t> host((size_t)M * H);
    unsigned s = 12345; for (auto &v : host) { s = s * 1664525u + 1013904223u; v = 0x3c00 + ((s >> 20) & 0xff) + ((s & 1) << 15); }
    CK(hipMemcpy(x, host.data(), (size_t)M * D * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(w1, host.data(), (size_t)H * D * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(w2, host.data(), (size_t)D * H * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(g, host.data(), (size_t)M * D * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(pre, host.data(), (size_t)M * H * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(bias, host.data(), H * 2, hipMemcpyHostToDevice));
    hipStream_t st; CK(hipStreamCreate(&st));

    // forward fc1: out (M x H) = x (M x D) W1^T (W1: H x D row-major)  ->  col-major: m = H, n = M, k = D, A = W1 (T), B = x (N)
    // backward dgrad: ghid (M x H) = g (M x D) W2 (W2: D x H row-major) ->  col-major: m = H, n = M, k = D, A = W2 (N), B = g (N)
    const Problem probs[] = {
        {"fc1 plain", H, M, D, HIPBLAS_OP_T, HIPBLAS_OP_N, D, D, HIPBLASLT_EPILOGUE_DEFAULT, false, false},
        {"fc1 +bias", H, M, D, HIPBLAS_OP_T, HIPBLAS_OP_N, D, D, HIPBLASLT_EPILOGUE_BIAS, false, true},
        {"fc1 +bias+gelu", H, M, D, HIPBLAS_OP_T, HIPBLAS_OP_N, D, D, HIPBLASLT_EPILOGUE_GELU_BIAS, false, true},
        {"fc1 +bias+gelu+aux", H, M, D, HIPBLAS_OP_T, HIPBLAS_OP_N, D, D, HIPBLASLT_EPILOGUE_GELU_AUX_BIAS, true, true},
        {"dgrad plain", H, M, D, HIPBLAS_OP_N, HIPBLAS_OP_N, H, D, HIPBLASLT_EPILOGUE_DEFAULT, false, false},
        {"dgrad +dgelu", H, M, D, HIPBLAS_OP_N, HIPBLAS_OP_N, H, D, HIPBLASLT_EPILOGUE_DGELU, true, false},
        {"dgrad +dgelu+bgrad", H, M, D, HIPBLAS_OP_N, HIPBLAS_OP_N, H, D, HIPBLASLT_EPILOGUE_DGELU_BGRAD, true, true},
    };
    for (const Problem &p : probs) {
        hipblasLtMatmulDesc_t desc; CK(hipblasLtMatmulDescCreate(&desc, HIPBLAS_COMPUTE_32F, HIP_R_32F));
        int32_t ta = p.ta, tb = p.tb; uint32_t epi = p.epi;
        CK(hipblasLtMatmulDescSetAttribute(desc, HIPBLASLT_MATMUL_DESC_TRANSA, &ta, sizeof ta));
        CK(hipblasLtMatmulDescSetAttribute(desc, HIPBLASLT_MATMUL_DESC_TRANSB, &tb, sizeof tb));
        CK(hipblasLtMatmulDescSetAttribute(desc, HIPBLASLT_MATMUL_DESC_EPILOGUE, &epi, sizeof epi));
        const bool fwd = p.ta == HIPBLAS_OP_T;
        if (p.bias) {
            void *bp = fwd ? bias : (void *)bgrad;
            CK(hipblasLtMatmulDescSetAttribute(desc, HIPBLASLT_MATMUL_DESC_BIAS_POINTER, &bp, sizeof bp));
            int32_t bt = fwd ? HIP_R_16BF : HIP_R_32F;
            CK(hipblasLtMatmulDescSetAttribute(desc, HIPBLASLT_MATMUL_DESC_BIAS_DATA_TYPE, &bt, sizeof bt));
        }
        if (p.aux) {
            void *ap = pre; int64_t ld = p.m;
            CK(hipblasLtMatmulDescSetAttribute(desc, HIPBLASLT_MATMUL_DESC_EPILOGUE_AUX_POINTER, &ap, sizeof ap));
            CK(hipblasLtMatmulDescSetAttribute(desc, HIPBLASLT_MATMUL_DESC_EPILOGUE_AUX_LD, &ld, sizeof ld));
        }
        hipblasLtMatrixLayout_t la, lb, ld_;
        const int a_rows = p.ta == HIPBLAS_OP_T ? p.k : p.m, a_cols = p.ta == HIPBLAS_OP_T ? p.m : p.k;
        CK(hipblasLtMatrixLayoutCreate(&la, HIP_R_16BF, a_rows, a_cols, p.lda));
        CK(hipblasLtMatrixLayoutCreate(&lb, HIP_R_16BF, p.k, p.n, p.ldb));
        CK(hipblasLtMatrixLayoutCreate(&ld_, HIP_R_16BF, p.m, p.n, p.m));
        hipblasLtMatmulPreference_t pref; CK(hipblasLtMatmulPreferenceCreate(&pref));
        uint64_t wsz = ws_bytes; CK(hipblasLtMatmulPreferenceSetAttribute(pref, HIPBLASLT_MATMUL_PREF_MAX_WORKSPACE_BYTES, &wsz, sizeof wsz));
        hipblasLtMatmulHeuristicResult_t res[8]; int found = 0;
        hipblasStatus_t hs = hipblasLtMatmulAlgoGetHeuristic(h, desc, la, lb, ld_, ld_, pref, 8, res, &found);
        if (hs != HIPBLAS_STATUS_SUCCESS || found == 0) { printf("%-22s no solution (status %d, found %d)\n", p.name, (int)hs, found); continue; }
        const void *A = fwd ? w1 : w2, *B = fwd ? x : g;
        float alpha = 1.f, beta = 0.f;
        double best = 1e9; int best_i = -1;
        for (int i = 0; i < found; ++i) {
            if (res[i].workspaceSize > ws_bytes) continue;
            bool ok = true;
            for (int w = 0; w < 3 && ok; ++w)
                ok = hipblasLtMatmul(h, desc, &alpha, A, la, B, lb, &beta, hid, ld_, hid, ld_, &res[i].algo, ws, ws_bytes, st) == HIPBLAS_STATUS_SUCCESS;
            if (!ok) continue;
            hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            CK(hipEventRecord(e0, st));
            for (int it = 0; it < 10; ++it)
                hipblasLtMatmul(h, desc, &alpha, A, la, B, lb, &beta, hid, ld_, hid, ld_, &res[i].algo, ws, ws_bytes, st);
            CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 10;
            if (ms < best) { best = ms; best_i = i; }
        }
        printf("%-22s m=%d n=%d k=%d: %d solutions, best #%d %.4f ms = %.0f TFLOP/s\n", p.name, p.m, p.n, p.k, found, best_i, best,
               2.0 * p.m * p.n * p.k / best / 1e9);
    }
    return 0;
}

// Fused softmax cross-entropy over vocabulary-sized rows for gfx950 (the
// loss side of the LM head).  Takes the place of the reference's xentropy_cuda_lib.forward / .backward
// (csrc/xentropy/interface.cpp + xentropy_kernel.cu, bound at flash_attn/losses/cross_entropy.py:9,37,54,103):
//
//   forward   lse_i  = log sum_j exp(x_ij)
//             loss_i = (1 - s) (lse_i - x_i[y_i]) + s (lse_i - sum_j x_ij / total_classes)      (s = smoothing)
//             (a label outside [0, cols) contributes no x_i[y_i] term: the vocabulary-parallel caller
//              passes shifted labels, cross_entropy.py:41-63; ignored rows are zeroed by the caller, :39)
//   backward  dx_ij  = g_i (exp(x_ij - lse_i) - (1 - s) [j == y_i] - s / total_classes)
//
// Both are one streaming pass over the logits (HBM-bound: 2 or 4 bytes per element read in forward, read +
// write in backward -- optionally in place, as upstream's inplace_backward).  One 256-thread workgroup per
// row; a thread walks the row in 16-byte chunks with an online (max, sum-of-exp) pair, so there is one
// exp per element and no second pass; the four waves combine through LDS.
#include "bp_common.h"
#include "bp_kernels.h"

namespace bp {

namespace {

template <class ET> struct XeLoad;   // 16 bytes -> floats
template <> struct XeLoad<float> {
    static constexpr int N = 4;
    static BP_DEV void load(const void *p, float (&v)[8]) {
        const u32x4 w = *reinterpret_cast<const u32x4 *>(p);
        const uint32_t a = w[0], b = w[1], c = w[2], d = w[3];
        v[0] = as_f32(a); v[1] = as_f32(b); v[2] = as_f32(c); v[3] = as_f32(d);
    }
    static BP_DEV float one(const void *p) { return *reinterpret_cast<const float *>(p); }
    static BP_DEV void store(void *p, const float (&v)[8]) {
        *reinterpret_cast<u32x4 *>(p) = u32x4{as_u32(v[0]), as_u32(v[1]), as_u32(v[2]), as_u32(v[3])};
    }
    static BP_DEV void store_one(void *p, float x) { *reinterpret_cast<float *>(p) = x; }
};
template <class H> struct XeLoad16 {
    static constexpr int N = 8;
    static BP_DEV void load(const void *p, float (&v)[8]) {
        const u32x4 w = *reinterpret_cast<const u32x4 *>(p);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint32_t x = w[i];
            v[2 * i] = Elem<H>::lo_f32(x);
            v[2 * i + 1] = Elem<H>::hi_f32(x);
        }
    }
    static BP_DEV float one(const void *p) { return Elem<H>::lo_f32(*reinterpret_cast<const uint16_t *>(p)); }
    static BP_DEV void store(void *p, const float (&v)[8]) {
        const u32x4 w = u32x4{Elem<H>::pack2(v[0], v[1]), Elem<H>::pack2(v[2], v[3]),
                              Elem<H>::pack2(v[4], v[5]), Elem<H>::pack2(v[6], v[7])};
        *reinterpret_cast<u32x4 *>(p) = w;   // (non-temporal here: +5 % time -- the line was just read, r02_p)
    }
    static BP_DEV void store_one(void *p, float x) { *reinterpret_cast<uint16_t *>(p) = Elem<H>::from_float(x); }
};
template <> struct XeLoad<BF16> : XeLoad16<BF16> {};
template <> struct XeLoad<F16> : XeLoad16<F16> {};
template <class ET> struct XeBytes { static constexpr int B = 2; };
template <> struct XeBytes<float> { static constexpr int B = 4; };

BP_DEV void online_merge(float &m, float &s, float m2, float s2) {
    const float mn = fmaxf(m, m2);
    // exp2(-inf - -inf) would be NaN: an empty partial has s = 0 and is skipped by the select
    const float a = (m == -INFINITY) ? 0.f : s * fast_exp2((m - mn) * kLog2e);
    const float b = (m2 == -INFINITY) ? 0.f : s2 * fast_exp2((m2 - mn) * kLog2e);
    m = mn;
    s = a + b;
}

}  // namespace

template <class ET>
__global__ __launch_bounds__(256) void xentropy_fwd_kernel(const XentParams p) {
    using L = XeLoad<ET>;
    constexpr int EB = XeBytes<ET>::B;
    __shared__ float red[3][4];
    const int64_t row = blockIdx.x;
    const char *x = static_cast<const char *>(p.logits) + row * p.row_stride * EB;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    float m = -INFINITY, s = 0.f, sx = 0.f;
    // head (elements before the first 16-byte boundary) and tail are walked element-wise by thread 0's wave
    const int head = (int)(((16 - (reinterpret_cast<uintptr_t>(x) & 15)) & 15) / EB);
    const int nhead = min(head, p.cols);
    const int nvec = (p.cols - nhead) / L::N;
    const int tail0 = nhead + nvec * L::N;
    for (int c = tid; c < nvec; c += 256) {
        float v[8];
        L::load(x + (int64_t)(nhead + c * L::N) * EB, v);
        float cm = v[0];
#pragma unroll
        for (int i = 1; i < L::N; ++i) cm = fmaxf(cm, v[i]);
        if (cm > m) {   // rescale the running sum only when the maximum moves
            s *= fast_exp2((m - cm) * kLog2e);   // m = -inf: s is 0, exp2(-inf) = 0
            m = cm;
        }
        const float mb = (m == -INFINITY) ? 0.f : m * kLog2e;   // a chunk of -inf logits before any finite one
#pragma unroll
        for (int i = 0; i < L::N; ++i) {
            s += fast_exp2(fmaf(v[i], kLog2e, -mb));
            sx += v[i];
        }
    }
    for (int c = tid; c < nhead + (p.cols - tail0); c += 256) {
        const int col = c < nhead ? c : tail0 + (c - nhead);
        const float v = L::one(x + (int64_t)col * EB);
        online_merge(m, s, v, 1.f);
        sx += v;
    }
    // wave reduction, then across the four waves through LDS
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        const float m2 = __shfl_xor(m, o), s2 = __shfl_xor(s, o);
        online_merge(m, s, m2, s2);
        sx += __shfl_xor(sx, o);
    }
    if (lane == 0) { red[0][wave] = m; red[1][wave] = s; red[2][wave] = sx; }
    __syncthreads();
    if (tid == 0) {
        m = red[0][0]; s = red[1][0]; sx = red[2][0];
        for (int w = 1; w < 4; ++w) {
            online_merge(m, s, red[0][w], red[1][w]);
            sx += red[2][w];
        }
        const float lse = m + fast_log2(s) * kLn2;
        const int64_t y = p.labels[row];
        float loss = p.smoothing * (lse - sx / (float)p.total_classes);
        if (y >= 0 && y < p.cols) loss += (1.f - p.smoothing) * (lse - L::one(x + y * EB));
        else if (p.smoothing == 0.f) loss = 0.f;
        p.losses[row] = loss;
        p.lse[row] = lse;
    }
}

template <class ET>
__global__ __launch_bounds__(256) void xentropy_bwd_kernel(const XentParams p) {
    using L = XeLoad<ET>;
    constexpr int EB = XeBytes<ET>::B;
    const int64_t row = blockIdx.x;
    const char *x = static_cast<const char *>(p.logits) + row * p.row_stride * EB;
    char *dx = static_cast<char *>(p.grad_logits) + row * p.grad_row_stride * EB;
    const int tid = threadIdx.x;
    const float g = p.grad_losses[row];
    const float lse2 = p.lse[row] * kLog2e;
    const int64_t y = p.labels[row];
    const float smooth = p.smoothing / (float)p.total_classes;
    const float hit = 1.f - p.smoothing;

    // x and dx share their alignment when they alias or have equal strides; otherwise go element-wise
    const bool same_phase = ((reinterpret_cast<uintptr_t>(x) ^ reinterpret_cast<uintptr_t>(dx)) & 15) == 0;
    const int head = (int)(((16 - (reinterpret_cast<uintptr_t>(x) & 15)) & 15) / EB);
    const int nhead = same_phase ? min(head, p.cols) : p.cols;
    const int nvec = (p.cols - nhead) / L::N;
    const int tail0 = nhead + nvec * L::N;
    for (int c = tid; c < nvec; c += 256) {
        const int col0 = nhead + c * L::N;
        float v[8];
        L::load(x + (int64_t)col0 * EB, v);
#pragma unroll
        for (int i = 0; i < L::N; ++i) {
            float d = fast_exp2(fmaf(v[i], kLog2e, -lse2)) - smooth;
            if (col0 + i == y) d -= hit;
            v[i] = g * d;
        }
        L::store(dx + (int64_t)col0 * EB, v);
    }
    for (int c = tid; c < nhead + (p.cols - tail0); c += 256) {
        const int col = c < nhead ? c : tail0 + (c - nhead);
        float d = fast_exp2(fmaf(L::one(x + (int64_t)col * EB), kLog2e, -lse2)) - smooth;
        if (col == y) d -= hit;
        L::store_one(dx + (int64_t)col * EB, g * d);
    }
}

// dtype: 0 fp16, 1 bf16, 2 fp32
hipError_t launch_xentropy_fwd(const XentParams &p, int dtype, hipStream_t stream) {
    dim3 g((unsigned)p.rows), t(256);
    if (dtype == 1) hipLaunchKernelGGL((xentropy_fwd_kernel<BF16>), g, t, 0, stream, p);
    else if (dtype == 0) hipLaunchKernelGGL((xentropy_fwd_kernel<F16>), g, t, 0, stream, p);
    else hipLaunchKernelGGL((xentropy_fwd_kernel<float>), g, t, 0, stream, p);
    return hipGetLastError();
}

hipError_t launch_xentropy_bwd(const XentParams &p, int dtype, hipStream_t stream) {
    dim3 g((unsigned)p.rows), t(256);
    if (dtype == 1) hipLaunchKernelGGL((xentropy_bwd_kernel<BF16>), g, t, 0, stream, p);
    else if (dtype == 0) hipLaunchKernelGGL((xentropy_bwd_kernel<F16>), g, t, 0, stream, p);
    else hipLaunchKernelGGL((xentropy_bwd_kernel<float>), g, t, 0, stream, p);
    return hipGetLastError();
}

}  // namespace bp

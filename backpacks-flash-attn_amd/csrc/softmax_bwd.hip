// Backward of the causal softmax that produces the Backpack sense weights alpha (training path of
// ContextSelfAttn, training/src/models/backpack.py:112-122, and of the fused sense contraction):
//
//     dS[t,s] = scale * alpha[t,s] * (dA[t,s] - sum_{s'<=t} alpha[t,s'] dA[t,s'])       s <= t,   0 above
//
// with dA the gradient w.r.t. alpha (for the fused contraction: dA[t,s] = dout[t,:] . C[s,:], a batched GEMM
// that hipBLASLt writes into the buffer this kernel then overwrites in place).  dS feeds the two small GEMMs
// dq = dS k, dk = dS^T q.  The reference reaches the same numbers through ATen's softmax backward plus the
// mask / scale ops of autograd: five elementwise passes over (B,k,S,S); this is one read of alpha and dA and
// one write, and rows stop at the diagonal (the part above it is only zero-filled).
// One wave per row, 16-byte accesses, the row (<= 4096 columns) stays in registers between the two phases.
#include "bp_common.h"
#include "bp_kernels.h"

namespace bp {

template <class ET, int CH>   // CH: 16-byte chunks per lane, row length <= CH * 512
__global__ __launch_bounds__(256) void softmax_bwd_causal_kernel(const SoftmaxBwdParams p) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= p.rows) return;
    const int t = (int)(row % p.s);                   // position inside its (S x S) matrix
    const uint16_t *a = reinterpret_cast<const uint16_t *>(p.alpha) + row * p.s;
    uint16_t *d = reinterpret_cast<uint16_t *>(p.dp) + row * p.s;
    const int live = t + 1;                           // columns 0 .. t carry probability mass

    u32x4 av[CH], dv[CH];
    float acc = 0.f;
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        const int col = (c * 64 + lane) * 8;
        av[c] = u32x4{0u, 0u, 0u, 0u};
        dv[c] = u32x4{0u, 0u, 0u, 0u};
        if (col < live) {                             // alpha is exactly 0 past the diagonal inside the chunk
            av[c] = *reinterpret_cast<const u32x4 *>(a + col);
            dv[c] = *reinterpret_cast<const u32x4 *>(d + col);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const uint32_t aw = av[c][i], dw = dv[c][i];
                acc = fmaf(Elem<ET>::lo_f32(aw), Elem<ET>::lo_f32(dw), acc);
                // the GEMM wrote finite garbage above the diagonal; alpha = 0 there kills it, but not a NaN/inf
                acc = fmaf(Elem<ET>::hi_f32(aw), Elem<ET>::hi_f32(dw), acc);
            }
        }
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) acc += __shfl_xor(acc, m);
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        const int col = (c * 64 + lane) * 8;
        if (col >= p.s) continue;
        u32x4 o = {0u, 0u, 0u, 0u};
        if (col < live) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const uint32_t aw = av[c][i], dw = dv[c][i];
                const float lo = p.scale * Elem<ET>::lo_f32(aw) * (Elem<ET>::lo_f32(dw) - acc);
                const float hi = p.scale * Elem<ET>::hi_f32(aw) * (Elem<ET>::hi_f32(dw) - acc);
                // 0 * garbage above the diagonal must be a clean 0
                o[i] = Elem<ET>::pack2(col + 2 * i <= t ? lo : 0.f, col + 2 * i + 1 <= t ? hi : 0.f);
            }
        }
        *reinterpret_cast<u32x4 *>(d + col) = o;
    }
}

template <class ET>
static hipError_t launch_sb(const SoftmaxBwdParams &p, hipStream_t stream) {
    const int ch = (p.s + 511) / 512;
    dim3 g((unsigned)((p.rows + 3) / 4)), t(256);
#define BP_SB_CASE(N) \
    if (ch <= N) { hipLaunchKernelGGL((softmax_bwd_causal_kernel<ET, N>), g, t, 0, stream, p); return hipGetLastError(); }
    BP_SB_CASE(1) BP_SB_CASE(2) BP_SB_CASE(4) BP_SB_CASE(8)
#undef BP_SB_CASE
    return hipErrorNotSupported;
}

// rows = n_matrices * s; s % 8 == 0 and <= 4096; both buffers contiguous (n, s, s), 16-byte aligned
hipError_t launch_softmax_bwd_causal(const SoftmaxBwdParams &p, int dtype, hipStream_t stream) {
    return dtype == 1 ? launch_sb<BF16>(p, stream) : launch_sb<F16>(p, stream);
}

}  // namespace bp

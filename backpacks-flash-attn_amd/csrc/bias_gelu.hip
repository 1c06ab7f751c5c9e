// Elementwise halves of the reference's fused dense layers for gfx950 (second half):
// what its cuBLASLt epilogues do around the GEMMs of `FusedDenseGeluDenseFunc`
// (flash_attn/ops/fused_dense.py:175-330; csrc/fused_dense_lib/fused_dense.cpp:195-197 `linear_gelu_forward`,
// `bias_gelu_linear_dgrad_bgrad`, `linear_bias_wgrad`), as stand-alone HBM-bound passes -- the GEMMs themselves stay
// on the BLAS library:
//
//   bias_gelu_fwd   y = gelu_tanh(x + bias)              and, when asked, pre = x + bias (what backward needs)
//   bias_gelu_bwd   dpre = g * gelu_tanh'(pre)           and  dbias[c] = sum_r dpre[r, c]   in the SAME pass
//   column_sum      dbias[c] = sum_r g[r, c]             (bias gradient of a plain dense layer, `linear_bias_wgrad`)
//
// gelu_tanh is GPT-2's `gelu_new`: 0.5 x (1 + tanh(sqrt(2/pi) (x + 0.044715 x^3))), the approximation the reference's
// epilogue and `F.gelu(approximate='tanh')` use.  tanh(u) = 1 - 2 / (1 + e^(2u)) through v_exp_f32 / v_rcp_f32 (gelu_r).
//
// Layout: rows x cols 16-bit, unit column stride, cols % 8 == 0; a thread owns 8 consecutive columns (one 16-byte
// access per row).  The column sums are deterministic, two stages, no atomics: workgroup (chunk of 512 columns, slice)
// walks its rows with the 8 sums per lane in registers and writes one partial row; the second kernel adds the partial
// rows in a fixed order.  Bytes per element: forward 2 read + 2 (or 4)
// written, backward 4 read + 2 written, column sum 2 read.
#include "bp_common.h"
#include "bp_kernels.h"

namespace bp {

namespace {

constexpr float kGeluA = 0.7978845608028654f;   // sqrt(2 / pi)
constexpr float kGeluB = 0.044715f;

// With r = 1 / (1 + e^(2u)), u = sqrt(2/pi) x (1 + 0.044715 x^2):  tanh(u) = 1 - 2r, so
//   gelu(x)  = 0.5 x (1 + tanh u)                        = x (1 - r)
//   gelu'(x) = 0.5 (1 + tanh u) + 0.5 x (1 - tanh^2 u) u' = (1 - r) (1 + 2 x r u'),   u' = sqrt(2/pi) (1 + 3 * 0.044715 x^2)
// (1 - tanh^2 = 4 r (1 - r)).  One v_exp_f32 and one v_rcp_f32 per element; exp2 overflow -> inf -> r = 0 -> gelu = x,
// gelu' = 1; underflow -> r = 1 -> both 0: no clamp needed.
BP_DEV float gelu_r(float x, float x2) {
    const float u2 = (x * fmaf(kGeluB, x2, 1.f)) * (2.f * kGeluA * kLog2e);
    return __builtin_amdgcn_rcpf(1.f + fast_exp2(u2));
}
BP_DEV float gelu_fwd(float x) { return x * (1.f - gelu_r(x, x * x)); }
BP_DEV float gelu_grad(float x) {
    const float x2 = x * x;
    const float r = gelu_r(x, x2);
    const float du2 = fmaf(6.f * kGeluA * kGeluB, x2, 2.f * kGeluA);   // 2 u'
    return (1.f - r) * fmaf(x * r, du2, 1.f);
}

template <class ET> BP_DEV void unpack8(const u32x4 w, float (&v)[8]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint32_t x = w[i];   // by-value copy (bp_common.h, as_f32)
        v[2 * i] = Elem<ET>::lo_f32(x);
        v[2 * i + 1] = Elem<ET>::hi_f32(x);
    }
}
template <class ET> BP_DEV u32x4 pack8(const float (&v)[8]) {
    return u32x4{Elem<ET>::pack2(v[0], v[1]), Elem<ET>::pack2(v[2], v[3]), Elem<ET>::pack2(v[4], v[5]),
                 Elem<ET>::pack2(v[6], v[7])};
}

}  // namespace

// ---- forward: flat over 16-byte chunks, grid-stride ------------------------------------------------------------
template <class ET, bool HAS_BIAS, bool SAVE_PRE>
__global__ __launch_bounds__(256) void bias_gelu_fwd_kernel(const BiasGeluParams p) {
    const int64_t nchunks = p.rows * (p.cols / 8);
    const int cpr = p.cols / 8;   // chunks per row
    const u32x4 *x = static_cast<const u32x4 *>(p.x);
    u32x4 *y = static_cast<u32x4 *>(p.y);
    u32x4 *pre = static_cast<u32x4 *>(p.pre);
    const u32x4 *bias = static_cast<const u32x4 *>(p.bias);
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nchunks; i += stride) {
        float v[8];
        unpack8<ET>(x[i], v);
        if (HAS_BIAS) {
            float b[8];
            unpack8<ET>(bias[i % cpr], b);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] += b[j];
            if (SAVE_PRE) {
                // the saved pre-activation is the ROUNDED sum and the GELU is taken of that rounded value, so that the
                // backward differentiates exactly the function the forward evaluated
                const u32x4 w = pack8<ET>(v);
                pre[i] = w;
                unpack8<ET>(w, v);
            }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = gelu_fwd(v[j]);
        y[i] = pack8<ET>(v);
    }
}

// ---- backward / column sums --------------------------------------------------------------------------------------
// grid (column chunks of 512, slices), 4 waves: every wave owns the SAME 512 columns (a lane: 8 consecutive ones) and
// every fourth row of the slice's rows, U rows per trip so that enough loads are in flight to cover the HBM latency
// (U x 1 KB per wave; two rows per trip reached 1.4 TB/s, r03_b); the four waves' sums meet in LDS in a fixed order.
// GELU = false: plain column sums of g (no pre, no dpre).
template <class ET, bool GELU, int U>
__global__ __launch_bounds__(256) void bias_gelu_bwd_kernel(const BiasGeluParams p) {
    __shared__ float red[3][64][9];   // waves 1..3 -> wave 0 (pitch 9: conflict-free column access)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = (blockIdx.x * 64 + lane) * 8;
    const bool active = col < p.cols;
    const int64_t cpr = p.cols / 8;
    const u32x4 *g = static_cast<const u32x4 *>(p.x) + col / 8;
    const u32x4 *pre = static_cast<const u32x4 *>(p.pre) + col / 8;
    u32x4 *dpre = static_cast<u32x4 *>(p.y) + col / 8;
    float sum[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) sum[j] = 0.f;
    const int64_t step = (int64_t)gridDim.y * 4;
    if (active) {
        int64_t r = (int64_t)blockIdx.y * 4 + wave;
        for (; r + (U - 1) * step < p.rows; r += U * step) {
            u32x4 gw[U], xw[GELU ? U : 1];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                gw[u] = g[(r + u * step) * cpr];
                if (GELU) xw[u] = pre[(r + u * step) * cpr];
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                float a[8];
                unpack8<ET>(gw[u], a);
                if (GELU) {
                    float xa[8];
                    unpack8<ET>(xw[u], xa);
#pragma unroll
                    for (int j = 0; j < 8; ++j) a[j] *= gelu_grad(xa[j]);
                    const u32x4 w = pack8<ET>(a);
                    dpre[(r + u * step) * cpr] = w;
                    unpack8<ET>(w, a);   // dbias sums the rounded values the weight-gradient GEMM will see
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) sum[j] += a[j];
            }
        }
        for (; r < p.rows; r += step) {
            float a[8];
            unpack8<ET>(g[r * cpr], a);
            if (GELU) {
                float xa[8];
                unpack8<ET>(pre[r * cpr], xa);
#pragma unroll
                for (int j = 0; j < 8; ++j) a[j] *= gelu_grad(xa[j]);
                const u32x4 w = pack8<ET>(a);
                dpre[r * cpr] = w;
                unpack8<ET>(w, a);
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) sum[j] += a[j];
        }
    }
    if (p.ws == nullptr) return;
    if (wave > 0) {
#pragma unroll
        for (int j = 0; j < 8; ++j) red[wave - 1][lane][j] = sum[j];
    }
    __syncthreads();
    if (wave == 0 && active) {
#pragma unroll
        for (int w = 0; w < 3; ++w)
#pragma unroll
            for (int j = 0; j < 8; ++j) sum[j] += red[w][lane][j];
        float *out = p.ws + (int64_t)blockIdx.y * p.cols + col;
        *reinterpret_cast<f32x4 *>(out) = f32x4{sum[0], sum[1], sum[2], sum[3]};
        *reinterpret_cast<f32x4 *>(out + 4) = f32x4{sum[4], sum[5], sum[6], sum[7]};
    }
}

// second stage: dbias[c] = sum over the partial rows, fixed order.  A workgroup owns 64 columns: thread = (column quad,
// one of 16 row groups), 16-byte loads, the 16 groups meet in LDS.
template <class ET>
__global__ __launch_bounds__(256) void colsum_finish_kernel(const BiasGeluParams p, int nsl) {
    __shared__ float red[16][16][5];
    const int cq = threadIdx.x & 15, grp = threadIdx.x >> 4;
    const int c = blockIdx.x * 64 + cq * 4;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (c < p.cols)
        for (int i = grp; i < nsl; i += 16) acc += *reinterpret_cast<const f32x4 *>(p.ws + (int64_t)i * p.cols + c);
#pragma unroll
    for (int j = 0; j < 4; ++j) red[grp][cq][j] = acc[j];
    __syncthreads();
    if (threadIdx.x < 64) {
        const int col = blockIdx.x * 64 + threadIdx.x;
        if (col < p.cols) {
            float s = 0.f;
#pragma unroll
            for (int gi = 0; gi < 16; ++gi) s += red[gi][threadIdx.x >> 2][threadIdx.x & 3];
            if (p.dbias_f32) static_cast<float *>(p.dbias)[col] = s;
            else static_cast<uint16_t *>(p.dbias)[col] = Elem<ET>::from_float(s);
        }
    }
}

template <class ET>
static hipError_t fwd_et(const BiasGeluParams &p, hipStream_t stream) {
    const int64_t nchunks = p.rows * (p.cols / 8);
    int64_t wgs = (nchunks + 255) / 256;
    if (wgs > 8192) wgs = 8192;   // 32 resident waves per CU x 256 CUs, then grid-stride
    dim3 g((unsigned)wgs), t(256);
    if (p.bias == nullptr) hipLaunchKernelGGL((bias_gelu_fwd_kernel<ET, false, false>), g, t, 0, stream, p);
    else if (p.pre != nullptr) hipLaunchKernelGGL((bias_gelu_fwd_kernel<ET, true, true>), g, t, 0, stream, p);
    else hipLaunchKernelGGL((bias_gelu_fwd_kernel<ET, true, false>), g, t, 0, stream, p);
    return hipGetLastError();
}

hipError_t launch_bias_gelu_fwd(const BiasGeluParams &p, int dtype, hipStream_t stream) {
    return dtype == 1 ? fwd_et<BF16>(p, stream) : fwd_et<F16>(p, stream);
}

int bias_gelu_bwd_slices(int64_t rows, int cols) {
    const int chunks = (cols + 511) / 512;
    int64_t nsl = (1024 + chunks - 1) / chunks;   // ~1024 workgroups of 4 waves: 16 waves per CU, 4-8 rows in flight each
    if (nsl > kBiasGeluMaxSlices) nsl = kBiasGeluMaxSlices;
    if (nsl > (rows + 3) / 4) nsl = (rows + 3) / 4;   // a row per wave at least, where there are that many
    return (int)(nsl < 1 ? 1 : nsl);
}

// rows per trip of the dGELU kernel (two 16-byte loads each)
#ifndef BP_GELU_BWD_ROWS
#define BP_GELU_BWD_ROWS 4
#endif

template <class ET>
static hipError_t bwd_et(const BiasGeluParams &p, bool gelu, hipStream_t stream) {
    const int chunks = (p.cols + 511) / 512;
    const int nsl = bias_gelu_bwd_slices(p.rows, p.cols);
    dim3 g(chunks, nsl), t(256);
    if (gelu) hipLaunchKernelGGL((bias_gelu_bwd_kernel<ET, true, BP_GELU_BWD_ROWS>), g, t, 0, stream, p);
    else hipLaunchKernelGGL((bias_gelu_bwd_kernel<ET, false, 8>), g, t, 0, stream, p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess || p.dbias == nullptr) return e;
    hipLaunchKernelGGL((colsum_finish_kernel<ET>), dim3((p.cols + 63) / 64), dim3(256), 0, stream, p, nsl);
    return hipGetLastError();
}

hipError_t launch_bias_gelu_bwd(const BiasGeluParams &p, int dtype, bool gelu, hipStream_t stream) {
    return dtype == 1 ? bwd_et<BF16>(p, gelu, stream) : bwd_et<F16>(p, gelu, stream);
}

}  // namespace bp

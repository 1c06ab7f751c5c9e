// Fused dropout + residual-add + LayerNorm forward for gfx950 (the reference's dropout_add_layer_norm:
// flash_attn/ops/layer_norm.py:102-230, csrc/layer_norm/ln_api.cpp:83-254, ln_fwd_kernels.cuh:20-162 -- no
// rowscale / colscale / subset):
//
//     x  = dropout(x0) / (1 - p) + x1     (x1 optional; x stored in the residual dtype, usually fp32)
//     z  = (x - mean(x)) * rsqrt(var(x) + eps) * gamma + beta      (fp32 math, stored in x0's dtype)
//
// x0 may be 16-bit or fp32 (under AMP the embedding output that enters the first LayerNorm is fp32 while the
// blocks' outputs are 16-bit; the reference's otype = itype rule, ln_api.cpp:104).  Dropout bits: bp_philox.h, one
// stream per call, counter (row, column / 4) -- the unit a lane owns; the backward regenerates them from the same
// two generator words instead of reading a mask back (the optional `dmask` output exists for callers that ask
// for it, layer_norm.py:207 return_dropout_mask).
//
// This is the memory-bound step on either side of every attention / MLP call:
// unfused it is three torch kernels (add -> cast -> LayerNorm) moving 15 KB per 768-wide row, fused it
// moves the algorithmic 9 KB once.  One wave per row, 4 rows per workgroup; a lane owns CH chunks of
// 4 consecutive columns (8-byte 16-bit loads, 16-byte fp32 loads, all coalesced), keeps them in
// registers for the two reductions (mean, then centred sum of squares as the reference does), and
// reduces across the wave with xor-shuffles.
#include "bp_common.h"
#include "bp_kernels.h"
#include "bp_philox.h"

namespace bp {

BP_DEV float wave_sum(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
    return v;
}

template <class ET> BP_DEV float to_f32(uint16_t h);
template <> BP_DEV float to_f32<BF16>(uint16_t h) { return as_f32((uint32_t)h << 16); }
template <> BP_DEV float to_f32<F16>(uint16_t h) { return (float)__builtin_bit_cast(_Float16, h); }

// Element widths are template parameters: one load / store form per instantiation.
template <class ET, bool F32> BP_DEV void load4(const void *base, int64_t idx, float (&v)[4]) {
    if constexpr (F32) {
        const u32x4 w = *reinterpret_cast<const u32x4 *>(static_cast<const char *>(base) + idx * 4);
        v[0] = as_f32(w[0]); v[1] = as_f32(w[1]); v[2] = as_f32(w[2]); v[3] = as_f32(w[3]);
    } else {
        const u32x2 w = *reinterpret_cast<const u32x2 *>(static_cast<const char *>(base) + idx * 2);
        v[0] = to_f32<ET>(w[0] & 0xffffu); v[1] = to_f32<ET>(w[0] >> 16);
        v[2] = to_f32<ET>(w[1] & 0xffffu); v[3] = to_f32<ET>(w[1] >> 16);
    }
}

template <class ET, bool F32> BP_DEV void store4(void *base, int64_t idx, const float (&v)[4]) {
    if constexpr (F32) {
        u32x4 w = {as_u32(v[0]), as_u32(v[1]), as_u32(v[2]), as_u32(v[3])};
        *reinterpret_cast<u32x4 *>(static_cast<char *>(base) + idx * 4) = w;
    } else {
        u32x2 w = {Elem<ET>::pack2(v[0], v[1]), Elem<ET>::pack2(v[2], v[3])};
        *reinterpret_cast<u32x2 *>(static_cast<char *>(base) + idx * 2) = w;
    }
}

// Streaming (non-temporal) forms for the forward's row data: every byte is touched once per launch.  BP_LN_NT selects
// which accesses carry the hint (development A/B: 0 none, 1 stores, 2 loads and stores, 3 loads, 4 loads + residual store); the launcher picks per
// call (see launch_flags).
template <class ET, bool F32, bool NT> BP_DEV void load4s(const void *base, int64_t idx, float (&v)[4]) {
    if constexpr (!NT) {
        load4<ET, F32>(base, idx, v);
    } else if constexpr (F32) {
        const u32x4 w = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(static_cast<const char *>(base) + idx * 4));
        v[0] = as_f32(w[0]); v[1] = as_f32(w[1]); v[2] = as_f32(w[2]); v[3] = as_f32(w[3]);
    } else {
        const u32x2 w = __builtin_nontemporal_load(reinterpret_cast<const u32x2 *>(static_cast<const char *>(base) + idx * 2));
        v[0] = to_f32<ET>(w[0] & 0xffffu); v[1] = to_f32<ET>(w[0] >> 16);
        v[2] = to_f32<ET>(w[1] & 0xffffu); v[3] = to_f32<ET>(w[1] >> 16);
    }
}
template <class ET, bool F32, bool NT> BP_DEV void store4s(void *base, int64_t idx, const float (&v)[4]) {
    if constexpr (!NT) {
        store4<ET, F32>(base, idx, v);
    } else if constexpr (F32) {
        u32x4 w = {as_u32(v[0]), as_u32(v[1]), as_u32(v[2]), as_u32(v[3])};
        __builtin_nontemporal_store(w, reinterpret_cast<u32x4 *>(static_cast<char *>(base) + idx * 4));
    } else {
        u32x2 w = {Elem<ET>::pack2(v[0], v[1]), Elem<ET>::pack2(v[2], v[3])};
        __builtin_nontemporal_store(w, reinterpret_cast<u32x2 *>(static_cast<char *>(base) + idx * 2));
    }
}

// RES_F32: dtype of the residual stream (x1 in, x_out) is fp32, else ET.  W_F32: gamma/beta are fp32.
// NTL / NTS / NTZ: non-temporal loads of x0 and the residual / store of the residual / store of z
// SCALED: rowscale / colscale present (reference ln_fwd_kernels.cuh:99,123-125: x0 * rowscale[row], dropout, * colscale[col])
template <class ET, int CH, bool RES_F32, bool W_F32, bool NTL = false, bool NTS = false, bool NTZ = NTS, bool SCALED = false>
__global__ __launch_bounds__(256) void add_layer_norm_kernel(const LnParams p) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= p.rows) return;
    const int64_t base = row * p.cols;
    const bool drop = p.drop_thr != 0u;
    DropoutStream rng = {0u, 0u};
    if (drop) rng = dropout_stream(p.rng_state, 0u);

    float x[CH][4];
    float sum = 0.f;
    float row_scale = 1.f;
    if (SCALED && p.rowscale != nullptr)
        row_scale = p.x0_f32 ? static_cast<const float *>(p.rowscale)[row]
                             : to_f32<ET>(static_cast<const uint16_t *>(p.rowscale)[row]);
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        const int col = (c * 64 + lane) * 4;
#pragma unroll
        for (int i = 0; i < 4; ++i) x[c][i] = 0.f;
        if (col < p.cols) {
            if (p.x0_f32) load4s<ET, true, NTL>(p.x0, base + col, x[c]);
            else load4s<ET, false, NTL>(p.x0, base + col, x[c]);
            if (SCALED) {
#pragma unroll
                for (int i = 0; i < 4; ++i) x[c][i] *= row_scale;
            }
            if (drop) {
                uint32_t lo, hi, m = 0u;
                dropout_bits4(rng, (uint32_t)row, (uint32_t)(col >> 2), lo, hi);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const bool keep = dropout_u16(lo, hi, i) < p.drop_thr;
                    x[c][i] = keep ? x[c][i] * p.drop_scale : 0.f;
                    m |= (keep ? 1u : 0u) << (8 * i);
                }
                if (p.dmask != nullptr) *reinterpret_cast<uint32_t *>(p.dmask + base + col) = m;
            }
            if (SCALED && p.colscale != nullptr) {
                float cs[4];
                load4<ET, W_F32>(p.colscale, col, cs);
#pragma unroll
                for (int i = 0; i < 4; ++i) x[c][i] *= cs[i];
            }
            if (p.x1 != nullptr) {
                float r[4];
                load4s<ET, RES_F32, NTL>(p.x1, base + col, r);
#pragma unroll
                for (int i = 0; i < 4; ++i) x[c][i] += r[i];
            }
            if (p.x_out != nullptr) {
                // stored in the residual dtype; z below is computed from the UNROUNDED fp32 sum, exactly as
                // the reference does (ln_fwd_kernels.cuh:131-133: x.data = x_ij; xf[...] = x_ij)
                store4s<ET, RES_F32, NTS>(p.x_out, base + col, x[c]);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) sum += x[c][i];
        }
    }
    const float inv_n = 1.f / (float)p.cols;
    const float mu = wave_sum(sum) * inv_n;
    float m2 = 0.f;
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        const int col = (c * 64 + lane) * 4;
        if (col < p.cols) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float d = x[c][i] - mu;
                m2 += d * d;
            }
        }
    }
    const float rs = rsqrtf(wave_sum(m2) * inv_n + p.eps);
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        const int col = (c * 64 + lane) * 4;
        if (col < p.cols) {
            float g[4], b[4], z[4];
            load4<ET, W_F32>(p.gamma, col, g);
            load4<ET, W_F32>(p.beta, col, b);
#pragma unroll
            for (int i = 0; i < 4; ++i) z[i] = (x[c][i] - mu) * rs * g[i] + b[i];
            if (p.x0_f32) store4s<ET, true, NTZ>(p.z, base + col, z);
            else store4s<ET, false, NTZ>(p.z, base + col, z);
        }
    }
}

template <class ET, bool RES_F32, bool W_F32>
static hipError_t launch_flags(const LnParams &p, hipStream_t stream) {
    const int ch = (p.cols + 255) / 256;
    dim3 g((unsigned)((p.rows + 3) / 4)), t(256);
    // shipped: 4 -- x0 and the incoming residual are read once and never again, the outgoing residual is next read a GEMM
    // and an attention launch later (long evicted at any batch that matters); z stays cacheable for the GEMM that follows.
    // r04_d / r04_e on one box each: kernel alone -6.7 % (B = 64) ... -4 % (B = 1536) with nt loads, in the model
    // 2.397 -> 2.336 ms per launch at B = 1536 (0.756 -> 0.776 of 8 TB/s), step +0.3 %
#ifndef BP_LN_NT
#define BP_LN_NT 4
#endif
    constexpr bool NTL = BP_LN_NT >= 2, NTS = BP_LN_NT == 1 || BP_LN_NT == 2 || BP_LN_NT == 4, NTZ = BP_LN_NT == 1 || BP_LN_NT == 2;
    const bool scaled = p.rowscale != nullptr || p.colscale != nullptr;
#define BP_LN_CASE(N) \
    if (ch <= N) { \
        if (scaled) hipLaunchKernelGGL((add_layer_norm_kernel<ET, N, RES_F32, W_F32, NTL, NTS, NTZ, true>), g, t, 0, stream, p); \
        else hipLaunchKernelGGL((add_layer_norm_kernel<ET, N, RES_F32, W_F32, NTL, NTS, NTZ, false>), g, t, 0, stream, p); \
        return hipGetLastError(); }
    BP_LN_CASE(1) BP_LN_CASE(2) BP_LN_CASE(3) BP_LN_CASE(4) BP_LN_CASE(6) BP_LN_CASE(8)
    BP_LN_CASE(12) BP_LN_CASE(16) BP_LN_CASE(24) BP_LN_CASE(32)
#undef BP_LN_CASE
    return hipErrorInvalidValue;
}

template <class ET>
static hipError_t launch_et(const LnParams &p, hipStream_t stream) {
    // residual dtype: x1's when given, else x_out's (the API guarantees they agree when both exist)
    const bool res_f32 = (p.x1 != nullptr) ? p.x1_f32 != 0 : p.xo_f32 != 0;
    const bool w_f32 = p.w_f32 != 0;
    if (res_f32) return w_f32 ? launch_flags<ET, true, true>(p, stream) : launch_flags<ET, true, false>(p, stream);
    return w_f32 ? launch_flags<ET, false, true>(p, stream) : launch_flags<ET, false, false>(p, stream);
}

hipError_t launch_add_layer_norm(const LnParams &p, int dtype, hipStream_t stream) {
    return dtype == 1 ? launch_et<BF16>(p, stream) : launch_et<F16>(p, stream);
}

// =====================================================================================================
// Backward (reference: DropoutAddLayerNormFn.backward, flash_attn/ops/layer_norm.py:131-152;
// csrc/layer_norm/ln_bwd_kernels.cuh, eval path: no dropout / rowscale / colscale)
//
//     xhat = (x - mu) rs            dy = dz * gamma
//     dx   = rs (dy - mean(dy) - xhat mean(dy xhat)) + dx_in         (dx0 = dx1 = dx: x = x0 + x1)
//     dgamma = sum_rows dz xhat     dbeta = sum_rows dz
//
// mu and rs are recomputed from the saved x (read anyway), so the forward stores no statistics.  One wave
// per row, each wave strides over rows and keeps its lanes' columns of dgamma / dbeta in registers; the
// four waves of a workgroup fold theirs through LDS into one partial row of the workspace and a second
// small kernel sums the partial rows (deterministic: no atomics).
// =====================================================================================================
BP_DEV void wave_sum2(float &a, float &b) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        a += __shfl_xor(a, m);
        b += __shfl_xor(b, m);
    }
}

// SCALED: the forward had a rowscale and / or a colscale: dx0 = dx * rowscale * mask / (1 - p) * colscale,
// dcolscale = sum over rows of dx * rowscale * mask / (1 - p) * x0 (reference ln_bwd_kernels.cuh:183-195)
template <class ET, int CH, bool RES_F32, bool W_F32, bool SCALED = false>
__global__ __launch_bounds__(256) void add_layer_norm_bwd_kernel(const LnBwdParams p) {
    __shared__ float fold[SCALED ? 3 : 2][CH * 256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float inv_n = 1.f / (float)p.cols;
    const bool drop = p.drop_thr != 0u;
    DropoutStream rng = {0u, 0u};
    if (drop) rng = dropout_stream(p.rng_state, 0u);

    float g[CH][4], dg[CH][4], db[CH][4];
    float cs[SCALED ? CH : 1][4], dcs[SCALED ? CH : 1][4];
    const bool has_cs = SCALED && p.colscale != nullptr;
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        const int col = (c * 64 + lane) * 4;
#pragma unroll
        for (int i = 0; i < 4; ++i) { g[c][i] = 0.f; dg[c][i] = 0.f; db[c][i] = 0.f; }
        if (SCALED) {
#pragma unroll
            for (int i = 0; i < 4; ++i) { cs[c][i] = 1.f; dcs[c][i] = 0.f; }
        }
        if (col < p.cols) {
            load4<ET, W_F32>(p.gamma, col, g[c]);
            if (has_cs) load4<ET, W_F32>(p.colscale, col, cs[c]);
        }
    }

    for (int64_t row = (int64_t)blockIdx.x * 4 + wave; row < p.rows; row += (int64_t)p.n_wg * 4) {
        const int64_t base = row * p.cols;
        float x[CH][4], dy[CH][4];
        float sum = 0.f;
        float row_scale = 1.f;
        if (SCALED && p.rowscale != nullptr)
            row_scale = p.x0_f32 ? static_cast<const float *>(p.rowscale)[row]
                                 : to_f32<ET>(static_cast<const uint16_t *>(p.rowscale)[row]);
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const int col = (c * 64 + lane) * 4;
#pragma unroll
            for (int i = 0; i < 4; ++i) { x[c][i] = 0.f; dy[c][i] = 0.f; }
            if (col < p.cols) {
                load4<ET, RES_F32>(p.x, base + col, x[c]);
                if (p.x0_f32) load4<ET, true>(p.dz, base + col, dy[c]);      // dz for now (z has x0's dtype)
                else load4<ET, false>(p.dz, base + col, dy[c]);
#pragma unroll
                for (int i = 0; i < 4; ++i) sum += x[c][i];
            }
        }
        const float mu = wave_sum(sum) * inv_n;
        float m2 = 0.f;
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const int col = (c * 64 + lane) * 4;
            if (col < p.cols) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float d = x[c][i] - mu;
                    m2 += d * d;
                }
            }
        }
        const float rs = rsqrtf(wave_sum(m2) * inv_n + p.eps);
        float s1 = 0.f, s2 = 0.f;   // sum dy, sum dy * xhat
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const int col = (c * 64 + lane) * 4;
            if (col < p.cols) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float xhat = (x[c][i] - mu) * rs;
                    const float dz = dy[c][i];
                    dg[c][i] += dz * xhat;
                    db[c][i] += dz;
                    const float d = dz * g[c][i];
                    s1 += d;
                    s2 += d * xhat;
                    x[c][i] = xhat;
                    dy[c][i] = d;
                }
            }
        }
        wave_sum2(s1, s2);
        const float c1 = s2 * inv_n, c2 = s1 * inv_n;
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const int col = (c * 64 + lane) * 4;
            if (col < p.cols) {
                float dx[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) dx[i] = rs * (dy[c][i] - c2 - x[c][i] * c1);
                if (p.dx_in != nullptr) {
                    float r[4];
                    load4<ET, RES_F32>(p.dx_in, base + col, r);
#pragma unroll
                    for (int i = 0; i < 4; ++i) dx[i] += r[i];
                }
                if (p.dx1 != nullptr) store4<ET, RES_F32>(p.dx1, base + col, dx);
                if (SCALED) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) dx[i] *= row_scale;
                }
                if (drop) {   // x0 entered through dropout: its gradient passes the same mask and scale
                    uint32_t lo, hi;
                    dropout_bits4(rng, (uint32_t)row, (uint32_t)(col >> 2), lo, hi);
#pragma unroll
                    for (int i = 0; i < 4; ++i) dx[i] = dropout_u16(lo, hi, i) < p.drop_thr ? dx[i] * p.drop_scale : 0.f;
                }
                if (has_cs) {
                    float x0v[4];
                    if (p.x0_f32) load4<ET, true>(p.x0, base + col, x0v);
                    else load4<ET, false>(p.x0, base + col, x0v);
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        dcs[c][i] += dx[i] * x0v[i];
                        dx[i] *= cs[c][i];
                    }
                }
                if (p.x0_f32) store4<ET, true>(p.dx0, base + col, dx);
                else store4<ET, false>(p.dx0, base + col, dx);
            }
        }
    }

    // fold the four waves' column sums (wave after wave: fixed order) and publish one partial row
    for (int w = 0; w < 4; ++w) {
        if (wave == w) {
#pragma unroll
            for (int c = 0; c < CH; ++c)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int idx = (c * 64 + lane) * 4 + i;
                    fold[0][idx] = (w == 0 ? 0.f : fold[0][idx]) + dg[c][i];
                    fold[1][idx] = (w == 0 ? 0.f : fold[1][idx]) + db[c][i];
                    if (SCALED) fold[2][idx] = (w == 0 ? 0.f : fold[2][idx]) + dcs[c][i];
                }
        }
        __syncthreads();
    }
    for (int col = threadIdx.x; col < p.cols; col += 256) {
        p.ws[(int64_t)blockIdx.x * p.cols + col] = fold[0][col];
        p.ws[((int64_t)kLnBwdMaxWg + blockIdx.x) * p.cols + col] = fold[1][col];
        if (SCALED && has_cs) p.ws[((int64_t)2 * kLnBwdMaxWg + blockIdx.x) * p.cols + col] = fold[2][col];
    }
}

// Sum of the partial rows: 64 columns per workgroup, 16 row slices per column (1024 threads) so that the
// serial part is n_wg / 16 coalesced loads per thread, then a fixed-order fold through LDS.
template <class ET, bool W_F32>
__global__ __launch_bounds__(1024) void ln_bwd_reduce_kernel(const LnBwdParams p) {
    __shared__ float part[3][16][64];
    const int c = threadIdx.x & 63, slice = threadIdx.x >> 6;
    const int col = blockIdx.x * 64 + c;
    const bool has_cs = p.dcolscale != nullptr;
    float a = 0.f, b = 0.f, e = 0.f;
    if (col < p.cols) {
        for (int r = slice; r < p.n_wg; r += 16) {
            a += p.ws[(int64_t)r * p.cols + col];
            b += p.ws[((int64_t)kLnBwdMaxWg + r) * p.cols + col];
            if (has_cs) e += p.ws[((int64_t)2 * kLnBwdMaxWg + r) * p.cols + col];
        }
    }
    part[0][slice][c] = a;
    part[1][slice][c] = b;
    part[2][slice][c] = e;
    __syncthreads();
    if (slice != 0 || col >= p.cols) return;
    for (int s = 1; s < 16; ++s) {
        a += part[0][s][c];
        b += part[1][s][c];
        e += part[2][s][c];
    }
    if (W_F32) {
        static_cast<float *>(p.dgamma)[col] = a;
        static_cast<float *>(p.dbeta)[col] = b;
        if (has_cs) static_cast<float *>(p.dcolscale)[col] = e;
    } else {
        static_cast<uint16_t *>(p.dgamma)[col] = Elem<ET>::from_float(a);
        static_cast<uint16_t *>(p.dbeta)[col] = Elem<ET>::from_float(b);
        if (has_cs) static_cast<uint16_t *>(p.dcolscale)[col] = Elem<ET>::from_float(e);
    }
}

template <class ET, bool RES_F32, bool W_F32>
static hipError_t launch_bwd_flags(const LnBwdParams &p, hipStream_t stream) {
    const int ch = (p.cols + 255) / 256;
    dim3 g((unsigned)p.n_wg), t(256);
    const bool scaled = p.rowscale != nullptr || p.colscale != nullptr;
#define BP_LNB_CASE(N) \
    if (ch <= N) { \
        if (scaled) hipLaunchKernelGGL((add_layer_norm_bwd_kernel<ET, N, RES_F32, W_F32, true>), g, t, 0, stream, p); \
        else hipLaunchKernelGGL((add_layer_norm_bwd_kernel<ET, N, RES_F32, W_F32, false>), g, t, 0, stream, p); \
    } else
    BP_LNB_CASE(1) BP_LNB_CASE(2) BP_LNB_CASE(3) BP_LNB_CASE(4) BP_LNB_CASE(6) BP_LNB_CASE(8)
    { return hipErrorNotSupported; }
#undef BP_LNB_CASE
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((ln_bwd_reduce_kernel<ET, W_F32>), dim3((p.cols + 63) / 64), dim3(1024), 0, stream, p);
    return hipGetLastError();
}

template <class ET>
static hipError_t launch_bwd_et(const LnBwdParams &p, hipStream_t stream) {
    if (p.res_f32) return p.w_f32 ? launch_bwd_flags<ET, true, true>(p, stream) : launch_bwd_flags<ET, true, false>(p, stream);
    return p.w_f32 ? launch_bwd_flags<ET, false, true>(p, stream) : launch_bwd_flags<ET, false, false>(p, stream);
}

// cols % 4 == 0 and <= 2048 (the model widths 384 / 640 / 768 and up); larger rows: hipErrorNotSupported
hipError_t launch_add_layer_norm_bwd(const LnBwdParams &p, int dtype, hipStream_t stream) {
    return dtype == 1 ? launch_bwd_et<BF16>(p, stream) : launch_bwd_et<F16>(p, stream);
}

}  // namespace bp

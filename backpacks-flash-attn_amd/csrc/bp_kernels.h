// Internal launch interface between the C ABI (bp_api.hip) and the kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

namespace bp {

// All strides are in ELEMENTS (16-bit elements for q/k/v/o/c, fp32 for lse).
struct FlashParams {
    const void *q, *k, *v;
    void *o;
    float *lse;
    const int *cu_q, *cu_k;   // NULL: fixed length, sequence b at rows [b*max_s, (b+1)*max_s)
    int64_t q_rs, q_hs, k_rs, k_hs, v_rs, v_hs, o_rs, o_hs;
    int64_t q_bs, k_bs, v_bs, o_bs;   // batch strides, used only when cu_q == NULL
    int64_t lse_stride;       // elements between consecutive (batch, head) rows of lse
    int b, h, d;
    int max_sq, max_sk;
    int n_qtiles;             // ceil(max_sq / 128)
    int causal;
    int pair;                 // flash_fwd_dma: one workgroup takes query tiles t and n-1-t (causal load balance)
    float scale_log2e;        // softmax_scale * log2(e)
    // dropout (training): see bp_philox.h.  drop_thr == 0: no dropout.
    const uint64_t *rng_state;   // device {seed, offset}
    uint32_t drop_thr;           // keep iff u16 < drop_thr  (round((1-p) * 65536))
    float drop_scale;            // 1 / (1 - p)
};

struct FlashBwdParams {
    const void *q, *k, *v, *dout, *out;
    const float *lse;         // (b, h, lse_stride) from the forward
    float *dsum;              // (b, h, lse_stride) workspace: D_i = sum_d dO_i[d] * O_i[d], written by the dQ kernel
    void *dq, *dk, *dv;
    const int *cu_q, *cu_k;   // NULL: fixed length
    int64_t q_rs, q_hs, k_rs, k_hs, v_rs, v_hs, do_rs, do_hs, o_rs, o_hs;
    int64_t dq_rs, dq_hs, dk_rs, dk_hs, dv_rs, dv_hs;
    int64_t lse_stride;
    int b, h, d, max_sq, max_sk, causal;
    float scale;
    const uint64_t *rng_state;   // dropout, as in FlashParams (must be the forward's state)
    uint32_t drop_thr;
    float drop_scale;
};

hipError_t launch_flash_bwd(const FlashBwdParams &p, int dtype, hipStream_t stream);

struct ProbsParams {
    const void *q, *k;
    const float *lse;
    void *p;
    int64_t q_bs, q_rs, q_hs, k_bs, k_rs, k_hs;
    int64_t lse_stride;
    int64_t p_bs, p_hs, p_rs;
    int b, h, d, sq, sk;
    int causal;
    int p_vec;                // 1: 8-byte stores into P are aligned
    int p_vec16;              // 1: 16-byte stores at multiples of 8 keys are aligned (full-line path)
    float scale_log2e;
    const uint64_t *rng_state;   // dropout: dropped entries are stored NEGATED (sign bit), see bp_attn_probs_dropout
    uint32_t drop_thr;
};

// job queues of the persistent sense-mix launches (sense_mix_dma.hip, sense_mix_bwd.hip): one ticket per XCD.
// 64 bytes, zeroed in front of every launch (arm_mix_queues)
struct MixQueues {
    unsigned int ticket[8];
    unsigned int pad[8];
};
hipError_t arm_mix_queues(MixQueues *&queues, hipStream_t stream);   // NULL -> a record of the library's ring

struct MixParams {
    const void *q, *k;        // q_l[t] = q + b*qk_bs + t*qk_rs + l*qk_ss ; k likewise
    const void *c;            // content[b, s, l, :] = c + b*c_bs + s*c_rs + l*c_ss
    void *o;                  // out[b, t, :] = o + b*o_bs + t*o_rs
    const float *lse;         // (b, k, lse_stride) natural-log LSE of every (sense, query)
    const float *kw;          // optional key weights w[b, l, s] (fp32, unit stride along s): alpha[b,l,:,s] *= w
    int64_t kw_bs, kw_ss;
    const int32_t *row_index; // optional (bp_sense_mix_gather): content[b, s, l, :] = c + row_index[b*idx_bs + s]*c_rs + l*c_ss
    int64_t idx_bs;
    uint32_t last_table_row;  // table_rows - 1: indices are clamped to it (unsigned, so a negative index also lands there)
    int64_t qk_bs, qk_rs, qk_ss;
    int64_t c_bs, c_rs, c_ss;
    int64_t o_bs, o_rs;
    int64_t lse_stride;
    int b, s, nsenses, dk, dout;
    int n_qtiles;             // ceil(s / 256)
    int n_chunks;             // ceil(dout / 256)
    float scale_log2e;
    MixQueues *queues;        // caller's record (queue_ws) or NULL; armed by launch_sense_mix_dma
};

// limits of the gathering sense mix: a job's row indices share the LDS with the ring as u16 (sense_mix_dma.hip)
constexpr int kMixGatherMaxKeys = 4096;
constexpr int64_t kMixGatherMaxRows = 65536;

// backward of the sense combination (sense_mix_bwd.hip)
struct MixBwdParams {
    const void *q, *k;        // as MixParams
    const void *dout;         // dout[b, t, :] = dout + b*do_bs + t*do_rs
    void *dc;                 // dC[b, s, l, :] = dc + b*c_bs + s*c_rs + l*c_ss
    const float *lse;         // (b, k, lse_stride)
    int64_t qk_bs, qk_rs, qk_ss;
    int64_t do_bs, do_rs;
    int64_t c_bs, c_rs, c_ss;
    int64_t lse_stride;
    int b, s, nsenses, dk, dout_cols;
    int n_ktiles;             // ceil(s / 256)
    int n_chunks;             // ceil(dout_cols / 256)
    float scale_log2e;
    MixQueues *queues;        // caller's record (queue_ws) or NULL; armed by launch_sense_mix_dc
};
hipError_t launch_sense_mix_dc(const MixBwdParams &p, int dtype, hipStream_t stream);

struct SenseGradParams {
    const void *q, *k;        // as MixParams
    const void *dpt;          // (b, N, 128) 16-bit: dP^T slab, row s*k + l, column = query t0 + j
    const float *lse;         // (b, k, lse_stride)
    float *dsum;              // (b, k, lse_stride): D, written by the dq kernel, read by the dk kernel
    void *dq;                 // dq_l[t] = dq + b*dq_bs + t*dq_rs + l*dq_ss   (16-bit)
    float *dk_acc;            // dk_l[s] += at dk_acc + b*dka_bs + s*dka_rs + l*dka_ss   (fp32)
    int64_t qk_bs, qk_rs, qk_ss;
    int64_t dpt_bs;
    int64_t dq_bs, dq_rs, dq_ss;
    int64_t dka_bs, dka_rs, dka_ss;
    int64_t lse_stride;
    int b, s, nsenses, dk;
    int t0;                   // first query of the slab (multiple of 128)
    float scale;
};
hipError_t launch_sense_dq_dk(const SenseGradParams &p, int dtype, hipStream_t stream);

struct SoftmaxBwdParams {
    const void *alpha;   // (n, s, s) 16-bit causal softmax output (zeros above the diagonal)
    void *dp;            // (n, s, s) 16-bit: gradient w.r.t. alpha in, gradient w.r.t. the scores out
    int64_t rows;        // n * s
    int s;
    float scale;
};
hipError_t launch_softmax_bwd_causal(const SoftmaxBwdParams &p, int dtype, hipStream_t stream);

struct XentParams {
    const void *logits;        // (rows, cols), row stride in elements, last stride 1
    const int64_t *labels;     // (rows)
    float *losses, *lse;       // (rows) fp32                                   [forward out / backward in: lse]
    const float *grad_losses;  // (rows) fp32                                   [backward]
    void *grad_logits;         // (rows, cols) logits' dtype, may alias logits   [backward]
    int64_t rows, row_stride, grad_row_stride;
    int cols, total_classes;
    float smoothing;
};

hipError_t launch_xentropy_fwd(const XentParams &p, int dtype, hipStream_t stream);
hipError_t launch_xentropy_bwd(const XentParams &p, int dtype, hipStream_t stream);

struct LnParams {
    const void *x0;           // (rows, cols) 16-bit, or fp32 when x0_f32 (then z is fp32 too)
    const void *x1;           // (rows, cols) residual in, 16-bit or fp32, may be NULL
    const void *gamma, *beta; // (cols) 16-bit or fp32
    void *z;                  // (rows, cols) in x0's dtype
    void *x_out;              // (rows, cols) residual out (x0 + x1), 16-bit or fp32, may be NULL
    int64_t rows;
    int cols;
    int x1_f32, xo_f32, w_f32;
    float eps;
    int x0_f32;
    uint8_t *dmask;              // optional (rows, cols) keep mask out (1 = kept), only written with dropout
    const uint64_t *rng_state;   // dropout on x0 (bp_philox.h); drop_thr == 0: none
    uint32_t drop_thr;
    float drop_scale;
    const void *rowscale;        // optional (rows) in x0's dtype: x0 row r is multiplied by rowscale[r] (DropPath)
    const void *colscale;        // optional (cols) in gamma's dtype: column c by colscale[c] (LayerScale)
};

hipError_t launch_add_layer_norm(const LnParams &p, int dtype, hipStream_t stream);

constexpr int kLnBwdMaxWg = 1024;   // row-parallel workgroups of the backward = rows of its partial-sum workspace
struct LnBwdParams {
    const void *dz;           // (rows, cols) 16-bit: gradient of the normalised output
    const void *dx_in;        // (rows, cols) residual dtype: gradient of the residual output (prenorm), may be NULL
    const void *x;            // (rows, cols) residual dtype: the summed stream x0 + x1 the forward normalised
    const void *gamma;        // (cols) 16-bit or fp32
    void *dx0;                // (rows, cols) 16-bit
    void *dx1;                // (rows, cols) residual dtype, may be NULL (same values as dx0)
    void *dgamma, *dbeta;     // (cols) gamma's dtype
    float *ws;                // (2, kLnBwdMaxWg, cols) fp32 partial sums; (3, ...) with a colscale
    int64_t rows;
    int cols, n_wg;
    int res_f32, w_f32;
    float eps;
    int x0_f32;                  // dz and dx0 are fp32 (the forward's x0 / z dtype)
    const uint64_t *rng_state;   // the forward's dropout state: dx0 = dropout-masked, rescaled dx
    uint32_t drop_thr;
    float drop_scale;
    const void *rowscale, *colscale;   // the forward's (optional)
    const void *x0;              // the forward's x0 (dz's dtype): needed for dcolscale only
    void *dcolscale;             // (cols) gamma's dtype, with colscale
};
hipError_t launch_add_layer_norm_bwd(const LnBwdParams &p, int dtype, hipStream_t stream);
// bias + tanh-GELU forward / backward and bias-gradient column sums (bias_gelu.hip)
constexpr int kBiasGeluMaxSlices = 1024;
struct BiasGeluParams {
    const void *x;       // fwd: (rows, cols) GEMM output;  bwd / column sum: the incoming gradient g
    const void *bias;    // fwd: (cols) 16-bit or NULL
    void *pre;           // fwd: optional (rows, cols) out = x + bias;  bwd: (rows, cols) in = saved pre-activation
    void *y;             // fwd: gelu out;  bwd: dpre out (may alias x)
    void *dbias;         // bwd: (cols) out, fp32 or 16-bit, may be NULL
    float *ws;           // bwd: (slices, cols) fp32 partial sums (required when dbias != NULL)
    int64_t rows;
    int cols;
    int dbias_f32;
};
hipError_t launch_bias_gelu_fwd(const BiasGeluParams &p, int dtype, hipStream_t stream);
hipError_t launch_bias_gelu_bwd(const BiasGeluParams &p, int dtype, bool gelu, hipStream_t stream);
int bias_gelu_bwd_slices(int64_t rows, int cols);
hipError_t launch_flash_fwd(const FlashParams &p, int dtype, bool vec, hipStream_t stream);
// LDS-DMA ring version; needs 16-byte friendly shapes (vec)
hipError_t launch_flash_fwd_dma(const FlashParams &p, int dtype, hipStream_t stream);
hipError_t launch_attn_probs(const ProbsParams &p, int dtype, bool vec, hipStream_t stream);
hipError_t launch_sense_mix(const MixParams &p, int dtype, bool vec_qk, bool vec_c, hipStream_t stream);
// LDS-DMA ring version; needs 16-byte friendly shapes (vec_qk && vec_c)
hipError_t launch_sense_mix_dma(const MixParams &p, int dtype, hipStream_t stream);
// wide senses, 128 < d_k <= kWideMaxDk (sense_wide.hip): the reference's few-sense ablations (vecs-4: 160, vecs-1: 640)
constexpr int kWideMaxDk = 640;
hipError_t launch_sense_lse_wide(const void *q, const void *k, float *lse, int64_t lse_stride, int64_t qk_bs,
                                 int64_t qk_rs, int64_t qk_ss, int b, int s, int nsenses, int dk, float scale_log2e,
                                 int dtype, bool vec, hipStream_t stream);
hipError_t launch_sense_alpha_wide(const void *q, const void *k, float *lse, int64_t lse_stride, void *alpha,
                                   int64_t qk_bs, int64_t qk_rs, int64_t qk_ss, int b, int s, int nsenses, int dk,
                                   float scale_log2e, int dtype, bool vec, hipStream_t stream);
hipError_t launch_sense_mix_wide(const MixParams &p, int dtype, bool vec_qk, bool vec_c, hipStream_t stream);
// LDS-DMA ring versions for the reference's two few-sense configurations exactly (d_k = 160 / 640, 16-byte friendly
// operands, s % 32 == 0): sense_wide_dma.hip
bool sense_wide_dma_takes(int s, int dk, int dout, bool vec_qk, bool vec_c, bool weighted);
hipError_t launch_sense_mix_wide_dma(const MixParams &p, int dtype, hipStream_t stream);
hipError_t launch_sense_lse_wide_dma(const void *q, const void *k, float *lse, int64_t lse_stride, int64_t qk_bs,
                                     int64_t qk_rs, int64_t qk_ss, int b, int s, int nsenses, int dk, float scale_log2e,
                                     int dtype, hipStream_t stream);

}  // namespace bp

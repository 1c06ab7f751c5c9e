// C ABI of libbackpack_hip.so (see include/bp_hip.h for the contract of every entry point).
// Host-side only: argument validation in the spirit of mha_fwd's TORCH_CHECKs
// (reference csrc/flash_attn/fmha_api.cpp:206-252), parameter packing, kernel dispatch.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/bp_hip.h"
#include "bp_common.h"
#include "bp_kernels.h"

namespace {

inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
inline bool mult8(int64_t x) { return (x & 7) == 0; }
inline int round_up(int x, int m) { return (x + m - 1) / m * m; }

inline bool scale_ok(float s) { return isfinite(s) && s > 0.f; }

// queue_ws == NULL means "a record of the library's own ring" (include/bp_hip.h).  A graph captured that way would
// replay on a record every other NULL launch also cycles through, so it is refused instead of documented as unsafe.
inline bool null_queue_ws_on_capturing_stream(const void *queue_ws, hipStream_t st) {
    if (queue_ws != nullptr) return false;
    hipStreamCaptureStatus status = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &status) != hipSuccess) return false;
    return status != hipStreamCaptureStatusNone;
}

// dropout argument check shared by the *_dropout entry points: p in [0, 1), a generator state when p > 0.
// thr: keep iff a 16-bit uniform < thr (bp_philox.h); 0 = dropout off.
inline bool dropout_args(float p, const uint64_t *rng_state, uint32_t &thr, float &rp_keep) {
    thr = 0u; rp_keep = 1.f;
    if (!(p >= 0.f && p < 1.f)) return false;
    if (p == 0.f) return true;
    if (rng_state == nullptr) return false;
    long t = lrintf((1.f - p) * 65536.f);
    thr = (uint32_t)(t < 1 ? 1 : t > 65535 ? 65535 : t);
    // rescale by the reference's 1 / (1 - p) (fmha_api.cpp:303 `rp_dropout`, ln_api.cpp:149), not by the keep rate the
    // 16-bit threshold realises (thr / 65536): the two differ by <= 2^-17 / (1 - p) relative, which the reference's own
    // fp32 LayerNorm test resolves (tests/ops/test_dropout_layer_norm.py compares with x0 * mask / (1 - p); 65536 / thr
    // was tried in round 3 and failed it by 8e-6 relative)
    rp_keep = 1.f / (1.f - p);
    return true;
}

// Measurement switches exist only in development builds (build_hip.py --variant NAME -- -DBP_DEV_BUILD):
// the shipped library reads no environment variable.  BP_FLASH_IMPL=staged / BP_MIX_IMPL=staged force the
// register-staged kernels for shapes the LDS-DMA ring kernels accept; BP_FLASH_PAIR=0 unpairs causal tiles.
#ifdef BP_DEV_BUILD
inline bool env_is(const char *name, const char *value) {
    const char *e = getenv(name);
    return e != nullptr && strcmp(e, value) == 0;
}
inline bool dev_force_staged_flash() { static const bool v = env_is("BP_FLASH_IMPL", "staged"); return v; }
inline bool dev_force_staged_mix() { static const bool v = env_is("BP_MIX_IMPL", "staged"); return v; }
inline bool dev_flash_pair() { static const bool v = !env_is("BP_FLASH_PAIR", "0"); return v; }
#else
inline bool dev_force_staged_flash() { return false; }
inline bool dev_force_staged_mix() { return false; }
inline bool dev_flash_pair() { return true; }
#endif

// 16-byte friendly shapes take the LDS-DMA ring kernel; anything else (odd head dims, unaligned
// views) the register-staged kernel with its element-wise loader.
inline hipError_t dispatch_flash(const bp::FlashParams &p, int dtype, bool vec, hipStream_t st) {
    if (vec && !dev_force_staged_flash()) return bp::launch_flash_fwd_dma(p, dtype, st);
    return bp::launch_flash_fwd(p, dtype, vec, st);
}

}  // namespace

extern "C" {

const char *bp_strerror(int code) {
    switch (code) {
        case BP_OK: return "ok";
        case BP_ERR_DTYPE: return "unsupported dtype (expected fp16 or bf16)";
        case BP_ERR_HEAD_DIM: return "head dimension must be in [1, 128] (sense width of bp_sense_lse / _alpha / _mix: [1, 640])";
        case BP_ERR_SHAPE: return "invalid shape or null pointer";
        case BP_ERR_SCALE: return "softmax_scale must be finite and > 0";
        case BP_ERR_LAUNCH: return "HIP kernel launch failed";
        case BP_ERR_DOUT: return "d_out must be >= 1";
        case BP_ERR_DROPOUT: return "dropout: p must be in [0, 1), rng_state non-NULL when p > 0, 16-byte friendly shapes only";
        case BP_ERR_QUEUE_WS: return "queue_ws must be caller-owned (non-NULL) while the stream is being captured";
        case BP_ERR_WORKSPACE: return "workspace smaller than the *_ws_floats() query of this entry point";
        default: return "unknown error";
    }
}

int bp_abi_version(void) { return BP_ABI_VERSION; }

int bp_build_flags(void) {
    int flags = 0;
#ifdef BP_FWD_WHATIF
    flags |= 1;
#endif
#if defined(BP_BWD_WHATIF) && BP_BWD_WHATIF != 0
    flags |= 2;
#endif
#ifdef BP_DEV_BUILD
    flags |= 4;
#endif
    return flags;
}

int bp_flash_fwd(const void *q, const void *k, const void *v, void *out, float *softmax_lse,
                 const int32_t *cu_seqlens_q, const int32_t *cu_seqlens_k,
                 int batch, int nheads, int head_dim, int max_seqlen_q, int max_seqlen_k,
                 int64_t q_row_stride, int64_t q_head_stride,
                 int64_t k_row_stride, int64_t k_head_stride,
                 int64_t v_row_stride, int64_t v_head_stride,
                 int64_t o_row_stride, int64_t o_head_stride,
                 int64_t lse_stride, float softmax_scale, int is_causal, int dtype,
                 bp_stream_t stream) {
    return bp_flash_fwd_dropout(q, k, v, out, softmax_lse, cu_seqlens_q, cu_seqlens_k, batch, nheads, head_dim,
                                max_seqlen_q, max_seqlen_k, q_row_stride, q_head_stride, k_row_stride,
                                k_head_stride, v_row_stride, v_head_stride, o_row_stride, o_head_stride,
                                lse_stride, softmax_scale, is_causal, dtype, 0.f, nullptr, stream);
}

int bp_flash_fwd_dropout(const void *q, const void *k, const void *v, void *out, float *softmax_lse,
                         const int32_t *cu_seqlens_q, const int32_t *cu_seqlens_k,
                         int batch, int nheads, int head_dim, int max_seqlen_q, int max_seqlen_k,
                         int64_t q_row_stride, int64_t q_head_stride,
                         int64_t k_row_stride, int64_t k_head_stride,
                         int64_t v_row_stride, int64_t v_head_stride,
                         int64_t o_row_stride, int64_t o_head_stride,
                         int64_t lse_stride, float softmax_scale, int is_causal, int dtype,
                         float p_dropout, const uint64_t *rng_state, bp_stream_t stream) {
    if (dtype != BP_DTYPE_F16 && dtype != BP_DTYPE_BF16) return BP_ERR_DTYPE;
    if (head_dim < 1 || head_dim > 128) return BP_ERR_HEAD_DIM;
    if (batch <= 0 || nheads <= 0 || max_seqlen_q <= 0 || max_seqlen_k < 0) return BP_ERR_SHAPE;
    if (q == nullptr || k == nullptr || softmax_lse == nullptr) return BP_ERR_SHAPE;
    if ((v == nullptr) != (out == nullptr)) return BP_ERR_SHAPE;
    if ((cu_seqlens_q == nullptr) != (cu_seqlens_k == nullptr)) return BP_ERR_SHAPE;
    if (!scale_ok(softmax_scale)) return BP_ERR_SCALE;

    bp::FlashParams p{};
    p.q = q; p.k = k; p.v = v; p.o = out; p.lse = softmax_lse;
    p.cu_q = cu_seqlens_q; p.cu_k = cu_seqlens_k;
    p.q_rs = q_row_stride; p.q_hs = q_head_stride;
    p.k_rs = k_row_stride; p.k_hs = k_head_stride;
    p.v_rs = v_row_stride; p.v_hs = v_head_stride;
    p.o_rs = o_row_stride; p.o_hs = o_head_stride;
    p.q_bs = (int64_t)max_seqlen_q * q_row_stride; p.o_bs = (int64_t)max_seqlen_q * o_row_stride;
    p.k_bs = (int64_t)max_seqlen_k * k_row_stride; p.v_bs = (int64_t)max_seqlen_k * v_row_stride;
    p.lse_stride = lse_stride;
    p.b = batch; p.h = nheads; p.d = head_dim;
    p.max_sq = max_seqlen_q; p.max_sk = max_seqlen_k;
    p.n_qtiles = (max_seqlen_q + 127) / 128;
    p.causal = is_causal ? 1 : 0;
    p.pair = (p.causal && p.n_qtiles > 1 && dev_flash_pair()) ? 1 : 0;
    p.scale_log2e = softmax_scale * bp::kLog2e;
    p.rng_state = rng_state;
    if (!dropout_args(p_dropout, rng_state, p.drop_thr, p.drop_scale)) return BP_ERR_DROPOUT;
    if (p.drop_thr != 0u && v == nullptr) return BP_ERR_DROPOUT;   // dropout acts on P V: nothing to drop in an LSE pass

    bool vec = (head_dim % 8 == 0) && aligned16(q) && aligned16(k) && mult8(q_row_stride) &&
               mult8(q_head_stride) && mult8(k_row_stride) && mult8(k_head_stride);
    if (v != nullptr)
        vec = vec && aligned16(v) && aligned16(out) && mult8(v_row_stride) && mult8(v_head_stride) &&
              mult8(o_row_stride) && mult8(o_head_stride);
    if (p.drop_thr != 0u && !vec) return BP_ERR_DROPOUT;   // the element-wise loader path has no dropout
    hipError_t e = dispatch_flash(p, dtype, vec, static_cast<hipStream_t>(stream));
    return e == hipSuccess ? BP_OK : BP_ERR_LAUNCH;
}

int bp_attn_probs(const void *q, const void *k, const float *softmax_lse, void *probs,
                  int batch, int nheads, int head_dim, int seqlen_q, int seqlen_k,
                  int64_t q_batch_stride, int64_t q_row_stride, int64_t q_head_stride,
                  int64_t k_batch_stride, int64_t k_row_stride, int64_t k_head_stride,
                  int64_t lse_stride,
                  int64_t p_batch_stride, int64_t p_head_stride, int64_t p_row_stride,
                  float softmax_scale, int is_causal, int dtype, bp_stream_t stream) {
    return bp_attn_probs_dropout(q, k, softmax_lse, probs, batch, nheads, head_dim, seqlen_q, seqlen_k,
                                 q_batch_stride, q_row_stride, q_head_stride, k_batch_stride, k_row_stride,
                                 k_head_stride, lse_stride, p_batch_stride, p_head_stride, p_row_stride,
                                 softmax_scale, is_causal, dtype, 0.f, nullptr, stream);
}

int bp_attn_probs_dropout(const void *q, const void *k, const float *softmax_lse, void *probs,
                          int batch, int nheads, int head_dim, int seqlen_q, int seqlen_k,
                          int64_t q_batch_stride, int64_t q_row_stride, int64_t q_head_stride,
                          int64_t k_batch_stride, int64_t k_row_stride, int64_t k_head_stride,
                          int64_t lse_stride,
                          int64_t p_batch_stride, int64_t p_head_stride, int64_t p_row_stride,
                          float softmax_scale, int is_causal, int dtype,
                          float p_dropout, const uint64_t *rng_state, bp_stream_t stream) {
    if (dtype != BP_DTYPE_F16 && dtype != BP_DTYPE_BF16) return BP_ERR_DTYPE;
    if (head_dim < 1 || head_dim > 128) return BP_ERR_HEAD_DIM;
    if (batch <= 0 || nheads <= 0 || seqlen_q <= 0 || seqlen_k <= 0) return BP_ERR_SHAPE;
    if (q == nullptr || k == nullptr || softmax_lse == nullptr || probs == nullptr) return BP_ERR_SHAPE;
    if (!scale_ok(softmax_scale)) return BP_ERR_SCALE;

    bp::ProbsParams p{};
    p.q = q; p.k = k; p.lse = softmax_lse; p.p = probs;
    p.q_bs = q_batch_stride; p.q_rs = q_row_stride; p.q_hs = q_head_stride;
    p.k_bs = k_batch_stride; p.k_rs = k_row_stride; p.k_hs = k_head_stride;
    p.lse_stride = lse_stride;
    p.p_bs = p_batch_stride; p.p_hs = p_head_stride; p.p_rs = p_row_stride;
    p.b = batch; p.h = nheads; p.d = head_dim; p.sq = seqlen_q; p.sk = seqlen_k;
    p.causal = is_causal ? 1 : 0;
    p.p_vec = ((reinterpret_cast<uintptr_t>(probs) & 7u) == 0 && (p_batch_stride & 3) == 0 &&
               (p_head_stride & 3) == 0 && (p_row_stride & 3) == 0) ? 1 : 0;
    p.p_vec16 = (aligned16(probs) && mult8(p_batch_stride) && mult8(p_head_stride) && mult8(p_row_stride)) ? 1 : 0;
    p.scale_log2e = softmax_scale * bp::kLog2e;
    p.rng_state = rng_state;
    float unused_scale;
    if (!dropout_args(p_dropout, rng_state, p.drop_thr, unused_scale)) return BP_ERR_DROPOUT;
    const bool vec = (head_dim % 8 == 0) && aligned16(q) && aligned16(k) && mult8(q_batch_stride) &&
                     mult8(q_row_stride) && mult8(q_head_stride) && mult8(k_batch_stride) &&
                     mult8(k_row_stride) && mult8(k_head_stride);
    hipError_t e = bp::launch_attn_probs(p, dtype, vec, static_cast<hipStream_t>(stream));
    return e == hipSuccess ? BP_OK : BP_ERR_LAUNCH;
}

// LSE of every (sense, query): the flash kernel in LSE-only mode with the k senses as heads.
static int sense_lse(const void *qk, float *lse_ws, int batch, int seqlen, int nsenses, int d_k,
                     int64_t qk_bs, int64_t qk_rs, int64_t qk_two, int64_t qk_ss,
                     float softmax_scale, int dtype, hipStream_t stream) {
    const uint16_t *qp = static_cast<const uint16_t *>(qk);
    const uint16_t *kp = qp + qk_two;
    bp::FlashParams p{};
    p.q = qp; p.k = kp; p.v = nullptr; p.o = nullptr; p.lse = lse_ws;
    p.cu_q = nullptr; p.cu_k = nullptr;
    p.q_rs = qk_rs; p.q_hs = qk_ss; p.k_rs = qk_rs; p.k_hs = qk_ss;
    p.q_bs = qk_bs; p.k_bs = qk_bs;
    p.lse_stride = round_up(seqlen, 16);
    p.b = batch; p.h = nsenses; p.d = d_k;
    p.max_sq = seqlen; p.max_sk = seqlen;
    p.n_qtiles = (seqlen + 127) / 128;
    p.causal = 1;
    p.pair = (p.n_qtiles > 1 && dev_flash_pair()) ? 1 : 0;
    p.scale_log2e = softmax_scale * bp::kLog2e;
    const bool vec = (d_k % 8 == 0) && aligned16(qp) && aligned16(kp) && mult8(qk_bs) && mult8(qk_rs) &&
                     mult8(qk_ss);
    hipError_t e;
    if (d_k > 128 && bp::sense_wide_dma_takes(seqlen, d_k, 8, vec, true, false))   // d_k = 160 / 640: sense_wide_dma.hip
        e = bp::launch_sense_lse_wide_dma(qp, kp, lse_ws, p.lse_stride, qk_bs, qk_rs, qk_ss, batch, seqlen, nsenses, d_k,
                                          p.scale_log2e, dtype, stream);
    else if (d_k > 128)   // wide senses (sense_wide.hip): the reference's vecs-4 / vecs-1 ablations
        e = bp::launch_sense_lse_wide(qp, kp, lse_ws, p.lse_stride, qk_bs, qk_rs, qk_ss, batch, seqlen, nsenses, d_k,
                                      p.scale_log2e, dtype, vec, stream);
    else e = dispatch_flash(p, dtype, vec, stream);
    return e == hipSuccess ? BP_OK : BP_ERR_LAUNCH;
}

int bp_sense_lse(const void *qk, float *lse, int batch, int seqlen, int nsenses, int d_k,
                 int64_t qk_batch_stride, int64_t qk_row_stride, int64_t qk_two_stride,
                 int64_t qk_sense_stride, float softmax_scale, int dtype, bp_stream_t stream) {
    if (dtype != BP_DTYPE_F16 && dtype != BP_DTYPE_BF16) return BP_ERR_DTYPE;
    if (d_k < 1 || d_k > bp::kWideMaxDk) return BP_ERR_HEAD_DIM;
    if (batch <= 0 || nsenses <= 0 || seqlen <= 0) return BP_ERR_SHAPE;
    if (qk == nullptr || lse == nullptr) return BP_ERR_SHAPE;
    if (!scale_ok(softmax_scale)) return BP_ERR_SCALE;
    return sense_lse(qk, lse, batch, seqlen, nsenses, d_k, qk_batch_stride, qk_row_stride,
                     qk_two_stride, qk_sense_stride, softmax_scale, dtype,
                     static_cast<hipStream_t>(stream));
}

int bp_sense_alpha(const void *qk, void *alpha, float *lse_ws, int lse_ready,
                   int batch, int seqlen, int nsenses, int d_k,
                   int64_t qk_batch_stride, int64_t qk_row_stride, int64_t qk_two_stride,
                   int64_t qk_sense_stride, float softmax_scale, int dtype, bp_stream_t stream) {
    if (dtype != BP_DTYPE_F16 && dtype != BP_DTYPE_BF16) return BP_ERR_DTYPE;
    if (d_k < 1 || d_k > bp::kWideMaxDk) return BP_ERR_HEAD_DIM;
    if (batch <= 0 || nsenses <= 0 || seqlen <= 0) return BP_ERR_SHAPE;
    if (qk == nullptr || alpha == nullptr || lse_ws == nullptr) return BP_ERR_SHAPE;
    if (!scale_ok(softmax_scale)) return BP_ERR_SCALE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (!lse_ready) {
        int rc = sense_lse(qk, lse_ws, batch, seqlen, nsenses, d_k, qk_batch_stride, qk_row_stride,
                           qk_two_stride, qk_sense_stride, softmax_scale, dtype, st);
        if (rc != BP_OK) return rc;
    }
    const uint16_t *qp = static_cast<const uint16_t *>(qk);
    const int64_t S = seqlen;
    if (d_k > 128) {
        const uint16_t *kp = qp + qk_two_stride;
        const bool vec = (d_k % 8 == 0) && aligned16(qp) && aligned16(kp) && mult8(qk_batch_stride) &&
                         mult8(qk_row_stride) && mult8(qk_sense_stride);
        const hipError_t e = bp::launch_sense_alpha_wide(qp, kp, lse_ws, round_up(seqlen, 16), alpha, qk_batch_stride,
                                                         qk_row_stride, qk_sense_stride, batch, seqlen, nsenses, d_k,
                                                         softmax_scale * bp::kLog2e, dtype, vec, st);
        return e == hipSuccess ? BP_OK : BP_ERR_LAUNCH;
    }
    return bp_attn_probs(qp, qp + qk_two_stride, lse_ws, alpha, batch, nsenses, d_k, seqlen, seqlen,
                         qk_batch_stride, qk_row_stride, qk_sense_stride,
                         qk_batch_stride, qk_row_stride, qk_sense_stride,
                         round_up(seqlen, 16),
                         (int64_t)nsenses * S * S, S * S, S,
                         softmax_scale, 1, dtype, stream);
}

int bp_sense_mix(const void *qk, const void *content, void *out, float *lse_ws, int lse_ready,
                 int batch, int seqlen, int nsenses, int d_k, int d_out,
                 int64_t qk_batch_stride, int64_t qk_row_stride, int64_t qk_two_stride,
                 int64_t qk_sense_stride,
                 int64_t c_batch_stride, int64_t c_row_stride, int64_t c_sense_stride,
                 int64_t o_batch_stride, int64_t o_row_stride,
                 float softmax_scale, int dtype, void *queue_ws, bp_stream_t stream) {
    return bp_sense_mix_weighted(qk, content, nullptr, out, lse_ws, lse_ready, batch, seqlen, nsenses, d_k, d_out,
                                 qk_batch_stride, qk_row_stride, qk_two_stride, qk_sense_stride, c_batch_stride,
                                 c_row_stride, c_sense_stride, 0, 0, o_batch_stride, o_row_stride, softmax_scale,
                                 dtype, queue_ws, stream);
}

int bp_sense_mix_weighted(const void *qk, const void *content, const float *key_weight, void *out,
                          float *lse_ws, int lse_ready,
                          int batch, int seqlen, int nsenses, int d_k, int d_out,
                          int64_t qk_batch_stride, int64_t qk_row_stride, int64_t qk_two_stride,
                          int64_t qk_sense_stride,
                          int64_t c_batch_stride, int64_t c_row_stride, int64_t c_sense_stride,
                          int64_t kw_batch_stride, int64_t kw_sense_stride,
                          int64_t o_batch_stride, int64_t o_row_stride,
                          float softmax_scale, int dtype, void *queue_ws, bp_stream_t stream) {
    if (dtype != BP_DTYPE_F16 && dtype != BP_DTYPE_BF16) return BP_ERR_DTYPE;
    if (d_k < 1 || d_k > bp::kWideMaxDk) return BP_ERR_HEAD_DIM;
    if (queue_ws != nullptr && !aligned16(queue_ws)) return BP_ERR_SHAPE;
    if (d_out < 1) return BP_ERR_DOUT;
    if (batch <= 0 || nsenses <= 0 || seqlen <= 0) return BP_ERR_SHAPE;
    if (qk == nullptr || content == nullptr || out == nullptr || lse_ws == nullptr) return BP_ERR_SHAPE;
    if (!scale_ok(softmax_scale)) return BP_ERR_SCALE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (null_queue_ws_on_capturing_stream(queue_ws, st)) return BP_ERR_QUEUE_WS;
    if (!lse_ready) {
        int rc = sense_lse(qk, lse_ws, batch, seqlen, nsenses, d_k, qk_batch_stride, qk_row_stride,
                           qk_two_stride, qk_sense_stride, softmax_scale, dtype, st);
        if (rc != BP_OK) return rc;
    }

    const uint16_t *qp = static_cast<const uint16_t *>(qk);
    bp::MixParams p{};
    p.q = qp; p.k = qp + qk_two_stride; p.c = content; p.o = out; p.lse = lse_ws;
    p.kw = key_weight; p.kw_bs = kw_batch_stride; p.kw_ss = kw_sense_stride;
    p.qk_bs = qk_batch_stride; p.qk_rs = qk_row_stride; p.qk_ss = qk_sense_stride;
    p.c_bs = c_batch_stride; p.c_rs = c_row_stride; p.c_ss = c_sense_stride;
    p.o_bs = o_batch_stride; p.o_rs = o_row_stride;
    p.lse_stride = round_up(seqlen, 16);
    p.b = batch; p.s = seqlen; p.nsenses = nsenses; p.dk = d_k; p.dout = d_out;
    p.n_qtiles = (seqlen + 255) / 256;
    p.n_chunks = (d_out + 255) / 256;
    p.scale_log2e = softmax_scale * bp::kLog2e;
    p.queues = static_cast<bp::MixQueues *>(queue_ws);
    const bool vec_qk = (d_k % 8 == 0) && aligned16(p.q) && aligned16(p.k) && mult8(qk_batch_stride) &&
                        mult8(qk_row_stride) && mult8(qk_sense_stride);
    const bool vec_c = (d_out % 8 == 0) && aligned16(content) && aligned16(out) && mult8(c_batch_stride) &&
                       mult8(c_row_stride) && mult8(c_sense_stride) && mult8(o_batch_stride) &&
                       mult8(o_row_stride);
    hipError_t e;
    if (d_k > 128 && bp::sense_wide_dma_takes(seqlen, d_k, d_out, vec_qk, vec_c, key_weight != nullptr))
        e = bp::launch_sense_mix_wide_dma(p, dtype, st);                         // d_k = 160 / 640: sense_wide_dma.hip
    else if (d_k > 128) e = bp::launch_sense_mix_wide(p, dtype, vec_qk, vec_c, st);   // few wide senses: sense_wide.hip
    else if (vec_qk && vec_c && p.n_qtiles <= 256 && !dev_force_staged_mix()) e = bp::launch_sense_mix_dma(p, dtype, st);
    else e = bp::launch_sense_mix(p, dtype, vec_qk, vec_c, st);
    return e == hipSuccess ? BP_OK : BP_ERR_LAUNCH;
}

int bp_sense_mix_gather(const void *qk, const void *table, const int32_t *row_index, void *out,
                        float *lse_ws, int lse_ready,
                        int batch, int seqlen, int nsenses, int d_k, int d_out, int64_t table_rows,
                        int64_t qk_batch_stride, int64_t qk_row_stride, int64_t qk_two_stride,
                        int64_t qk_sense_stride,
                        int64_t t_row_stride, int64_t t_sense_stride, int64_t idx_batch_stride,
                        int64_t o_batch_stride, int64_t o_row_stride,
                        float softmax_scale, int dtype, void *queue_ws, bp_stream_t stream) {
    if (dtype != BP_DTYPE_F16 && dtype != BP_DTYPE_BF16) return BP_ERR_DTYPE;
    // wide senses: the reference's two few-sense widths on the ring kernels of sense_wide_dma.hip (row indices as u32 in LDS:
    // no limit on the row count); everything else wider than 128 is gathered by the caller
    const bool wide = d_k > 128 && bp::sense_wide_dma_takes(seqlen, d_k, d_out, true, true, false);
    if (d_k < 8 || (d_k > 128 && !wide) || d_k % 8 != 0) return BP_ERR_HEAD_DIM;
    if (queue_ws != nullptr && !aligned16(queue_ws)) return BP_ERR_SHAPE;
    if (d_out < 8 || d_out % 8 != 0) return BP_ERR_DOUT;
    if (batch <= 0 || nsenses <= 0 || seqlen <= 0 || seqlen > bp::kMixGatherMaxKeys || table_rows <= 0 ||
        (!wide && table_rows > bp::kMixGatherMaxRows)) return BP_ERR_SHAPE;
    if (qk == nullptr || table == nullptr || row_index == nullptr || out == nullptr || lse_ws == nullptr) return BP_ERR_SHAPE;
    if (!scale_ok(softmax_scale)) return BP_ERR_SCALE;
    const uint16_t *qp = static_cast<const uint16_t *>(qk);
    if (!aligned16(qp) || !aligned16(qp + qk_two_stride) || !aligned16(table) || !aligned16(out)) return BP_ERR_SHAPE;
    const int64_t strides[] = {qk_batch_stride, qk_row_stride, qk_sense_stride, t_row_stride, t_sense_stride,
                               o_batch_stride, o_row_stride};
    for (int64_t v : strides) if (!mult8(v)) return BP_ERR_SHAPE;
    if (t_row_stride <= 0 || table_rows * t_row_stride * 2 >= (int64_t(1) << 32)) return BP_ERR_SHAPE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (null_queue_ws_on_capturing_stream(queue_ws, st)) return BP_ERR_QUEUE_WS;
    if (!lse_ready) {
        int rc = sense_lse(qk, lse_ws, batch, seqlen, nsenses, d_k, qk_batch_stride, qk_row_stride,
                           qk_two_stride, qk_sense_stride, softmax_scale, dtype, st);
        if (rc != BP_OK) return rc;
    }
    bp::MixParams p{};
    p.q = qp; p.k = qp + qk_two_stride; p.c = table; p.o = out; p.lse = lse_ws;
    p.row_index = row_index; p.idx_bs = idx_batch_stride; p.last_table_row = (uint32_t)(table_rows - 1);
    p.qk_bs = qk_batch_stride; p.qk_rs = qk_row_stride; p.qk_ss = qk_sense_stride;
    p.c_bs = 0; p.c_rs = t_row_stride; p.c_ss = t_sense_stride;
    p.o_bs = o_batch_stride; p.o_rs = o_row_stride;
    p.lse_stride = round_up(seqlen, 16);
    p.b = batch; p.s = seqlen; p.nsenses = nsenses; p.dk = d_k; p.dout = d_out;
    p.n_qtiles = (seqlen + 255) / 256;
    p.n_chunks = (d_out + 255) / 256;
    p.scale_log2e = softmax_scale * bp::kLog2e;
    p.queues = static_cast<bp::MixQueues *>(queue_ws);
    const hipError_t e = wide ? bp::launch_sense_mix_wide_dma(p, dtype, st) : bp::launch_sense_mix_dma(p, dtype, st);
    return e == hipSuccess ? BP_OK : BP_ERR_LAUNCH;
}

int bp_sense_mix_dc(const void *qk, const void *dout, const float *lse, void *dcontent,
                    int batch, int seqlen, int nsenses, int d_k, int d_out,
                    int64_t qk_batch_stride, int64_t qk_row_stride, int64_t qk_two_stride, int64_t qk_sense_stride,
                    int64_t do_batch_stride, int64_t do_row_stride,
                    int64_t c_batch_stride, int64_t c_row_stride, int64_t c_sense_stride,
                    float softmax_scale, int dtype, void *queue_ws, bp_stream_t stream) {
    if (dtype != BP_DTYPE_F16 && dtype != BP_DTYPE_BF16) return BP_ERR_DTYPE;
    if (d_k < 8 || d_k > 128 || d_k % 8 != 0) return BP_ERR_HEAD_DIM;
    if (queue_ws != nullptr && !aligned16(queue_ws)) return BP_ERR_SHAPE;
    if (d_out < 8 || d_out % 8 != 0) return BP_ERR_DOUT;
    if (batch <= 0 || nsenses <= 0 || seqlen <= 0 || seqlen > 65536) return BP_ERR_SHAPE;
    if (qk == nullptr || dout == nullptr || lse == nullptr || dcontent == nullptr) return BP_ERR_SHAPE;
    if (!scale_ok(softmax_scale)) return BP_ERR_SCALE;
    const uint16_t *qp = static_cast<const uint16_t *>(qk);
    if (!aligned16(qp) || !aligned16(qp + qk_two_stride) || !aligned16(dout) || !aligned16(dcontent)) return BP_ERR_SHAPE;
    const int64_t strides[] = {qk_batch_stride, qk_row_stride, qk_sense_stride, do_batch_stride, do_row_stride,
                               c_batch_stride, c_row_stride, c_sense_stride};
    for (int64_t st : strides) if (!mult8(st)) return BP_ERR_SHAPE;
    if (null_queue_ws_on_capturing_stream(queue_ws, static_cast<hipStream_t>(stream))) return BP_ERR_QUEUE_WS;
    bp::MixBwdParams p{};
    p.q = qp; p.k = qp + qk_two_stride; p.dout = dout; p.dc = dcontent; p.lse = lse;
    p.qk_bs = qk_batch_stride; p.qk_rs = qk_row_stride; p.qk_ss = qk_sense_stride;
    p.do_bs = do_batch_stride; p.do_rs = do_row_stride;
    p.c_bs = c_batch_stride; p.c_rs = c_row_stride; p.c_ss = c_sense_stride;
    p.lse_stride = round_up(seqlen, 16);
    p.b = batch; p.s = seqlen; p.nsenses = nsenses; p.dk = d_k; p.dout_cols = d_out;
    p.n_ktiles = (seqlen + 255) / 256;
    p.n_chunks = (d_out + 255) / 256;
    p.scale_log2e = softmax_scale * bp::kLog2e;
    p.queues = static_cast<bp::MixQueues *>(queue_ws);
    hipError_t e = bp::launch_sense_mix_dc(p, dtype, static_cast<hipStream_t>(stream));
    return e == hipSuccess ? BP_OK : BP_ERR_LAUNCH;
}

int bp_sense_dq_dk(const void *qk, const void *dpt, const float *lse, float *dsum_ws, void *dqk, float *dk_acc,
                   int batch, int seqlen, int nsenses, int d_k, int t0,
                   int64_t qk_batch_stride, int64_t qk_row_stride, int64_t qk_two_stride, int64_t qk_sense_stride,
                   int64_t dpt_batch_stride,
                   int64_t dqk_batch_stride, int64_t dqk_row_stride, int64_t dqk_sense_stride,
                   int64_t dka_batch_stride, int64_t dka_row_stride, int64_t dka_sense_stride,
                   float softmax_scale, int dtype, bp_stream_t stream) {
    if (dtype != BP_DTYPE_F16 && dtype != BP_DTYPE_BF16) return BP_ERR_DTYPE;
    if (d_k < 8 || d_k > 128 || d_k % 8 != 0) return BP_ERR_HEAD_DIM;
    if (batch <= 0 || nsenses <= 0 || seqlen <= 0 || t0 < 0 || t0 >= seqlen || t0 % 128 != 0) return BP_ERR_SHAPE;
    if (!qk || !dpt || !lse || !dsum_ws || !dqk || !dk_acc) return BP_ERR_SHAPE;
    if (!scale_ok(softmax_scale)) return BP_ERR_SCALE;
    const uint16_t *qp = static_cast<const uint16_t *>(qk);
    if (!aligned16(qp) || !aligned16(qp + qk_two_stride) || !aligned16(dpt) || !aligned16(dqk) || !aligned16(dk_acc))
        return BP_ERR_SHAPE;
    const int64_t strides[] = {qk_batch_stride, qk_row_stride, qk_sense_stride, dpt_batch_stride, dqk_batch_stride,
                               dqk_row_stride, dqk_sense_stride};
    for (int64_t st : strides) if (!mult8(st)) return BP_ERR_SHAPE;
    if ((dka_batch_stride | dka_row_stride | dka_sense_stride) & 3) return BP_ERR_SHAPE;
    bp::SenseGradParams p{};
    p.q = qp; p.k = qp + qk_two_stride; p.dpt = dpt; p.lse = lse; p.dsum = dsum_ws; p.dq = dqk; p.dk_acc = dk_acc;
    p.qk_bs = qk_batch_stride; p.qk_rs = qk_row_stride; p.qk_ss = qk_sense_stride;
    p.dpt_bs = dpt_batch_stride;
    p.dq_bs = dqk_batch_stride; p.dq_rs = dqk_row_stride; p.dq_ss = dqk_sense_stride;
    p.dka_bs = dka_batch_stride; p.dka_rs = dka_row_stride; p.dka_ss = dka_sense_stride;
    p.lse_stride = round_up(seqlen, 16);
    p.b = batch; p.s = seqlen; p.nsenses = nsenses; p.dk = d_k; p.t0 = t0;
    p.scale = softmax_scale;
    hipError_t e = bp::launch_sense_dq_dk(p, dtype, static_cast<hipStream_t>(stream));
    return e == hipSuccess ? BP_OK : BP_ERR_LAUNCH;
}

int64_t bp_flash_bwd_ws_floats(int batch, int nheads, int64_t lse_stride) {
    if (batch <= 0 || nheads <= 0 || lse_stride <= 0) return 0;
    return (int64_t)batch * nheads * 2 * lse_stride;
}

int bp_flash_bwd(const void *dout, const void *q, const void *k, const void *v, const void *out,
                 const float *softmax_lse, float *dsum_ws, int64_t dsum_ws_floats, void *dq, void *dk, void *dv,
                 const int32_t *cu_seqlens_q, const int32_t *cu_seqlens_k,
                 int batch, int nheads, int head_dim, int max_seqlen_q, int max_seqlen_k,
                 int64_t do_row_stride, int64_t do_head_stride,
                 int64_t q_row_stride, int64_t q_head_stride,
                 int64_t k_row_stride, int64_t k_head_stride,
                 int64_t v_row_stride, int64_t v_head_stride,
                 int64_t o_row_stride, int64_t o_head_stride,
                 int64_t dq_row_stride, int64_t dq_head_stride,
                 int64_t dk_row_stride, int64_t dk_head_stride,
                 int64_t dv_row_stride, int64_t dv_head_stride,
                 int64_t lse_stride, float softmax_scale, int is_causal, int dtype,
                 bp_stream_t stream) {
    return bp_flash_bwd_dropout(dout, q, k, v, out, softmax_lse, dsum_ws, dsum_ws_floats, dq, dk, dv, cu_seqlens_q, cu_seqlens_k, batch,
                                nheads, head_dim, max_seqlen_q, max_seqlen_k, do_row_stride, do_head_stride,
                                q_row_stride, q_head_stride, k_row_stride, k_head_stride, v_row_stride,
                                v_head_stride, o_row_stride, o_head_stride, dq_row_stride, dq_head_stride,
                                dk_row_stride, dk_head_stride, dv_row_stride, dv_head_stride, lse_stride,
                                softmax_scale, is_causal, dtype, 0.f, nullptr, stream);
}

int bp_flash_bwd_dropout(const void *dout, const void *q, const void *k, const void *v, const void *out,
                 const float *softmax_lse, float *dsum_ws, int64_t dsum_ws_floats, void *dq, void *dk, void *dv,
                 const int32_t *cu_seqlens_q, const int32_t *cu_seqlens_k,
                 int batch, int nheads, int head_dim, int max_seqlen_q, int max_seqlen_k,
                 int64_t do_row_stride, int64_t do_head_stride,
                 int64_t q_row_stride, int64_t q_head_stride,
                 int64_t k_row_stride, int64_t k_head_stride,
                 int64_t v_row_stride, int64_t v_head_stride,
                 int64_t o_row_stride, int64_t o_head_stride,
                 int64_t dq_row_stride, int64_t dq_head_stride,
                 int64_t dk_row_stride, int64_t dk_head_stride,
                 int64_t dv_row_stride, int64_t dv_head_stride,
                 int64_t lse_stride, float softmax_scale, int is_causal, int dtype,
                 float p_dropout, const uint64_t *rng_state, bp_stream_t stream) {
    if (dtype != BP_DTYPE_F16 && dtype != BP_DTYPE_BF16) return BP_ERR_DTYPE;
    if (head_dim < 8 || head_dim > 128 || head_dim % 8 != 0) return BP_ERR_HEAD_DIM;
    if (batch <= 0 || nheads <= 0 || max_seqlen_q <= 0 || max_seqlen_k <= 0) return BP_ERR_SHAPE;
    if (!dout || !q || !k || !v || !out || !softmax_lse || !dsum_ws || !dq || !dk || !dv) return BP_ERR_SHAPE;
    if ((cu_seqlens_q == nullptr) != (cu_seqlens_k == nullptr)) return BP_ERR_SHAPE;
    if (!scale_ok(softmax_scale)) return BP_ERR_SCALE;
    const void *ptrs[] = {dout, q, k, v, out, dq, dk, dv, softmax_lse, dsum_ws};
    for (const void *ptr : ptrs) if (!aligned16(ptr)) return BP_ERR_SHAPE;
    const int64_t strides[] = {do_row_stride, do_head_stride, q_row_stride, q_head_stride, k_row_stride,
                               k_head_stride, v_row_stride, v_head_stride, o_row_stride, o_head_stride, dq_row_stride, dq_head_stride,
                               dk_row_stride, dk_head_stride, dv_row_stride, dv_head_stride};
    for (int64_t st : strides) if (!mult8(st)) return BP_ERR_SHAPE;
    if (lse_stride % 16 != 0) return BP_ERR_SHAPE;
    // the dQ kernel writes both statistics rows of every (batch, head): an ABI-2-sized buffer would be overrun
    if (dsum_ws_floats < bp_flash_bwd_ws_floats(batch, nheads, lse_stride)) return BP_ERR_WORKSPACE;

    bp::FlashBwdParams p{};
    p.q = q; p.k = k; p.v = v; p.dout = dout; p.out = out; p.lse = softmax_lse; p.dsum = dsum_ws;
    p.dq = dq; p.dk = dk; p.dv = dv; p.cu_q = cu_seqlens_q; p.cu_k = cu_seqlens_k;
    p.q_rs = q_row_stride; p.q_hs = q_head_stride; p.k_rs = k_row_stride; p.k_hs = k_head_stride;
    p.v_rs = v_row_stride; p.v_hs = v_head_stride; p.do_rs = do_row_stride; p.do_hs = do_head_stride;
    p.o_rs = o_row_stride; p.o_hs = o_head_stride;
    p.dq_rs = dq_row_stride; p.dq_hs = dq_head_stride; p.dk_rs = dk_row_stride; p.dk_hs = dk_head_stride;
    p.dv_rs = dv_row_stride; p.dv_hs = dv_head_stride;
    p.lse_stride = lse_stride;
    p.b = batch; p.h = nheads; p.d = head_dim; p.max_sq = max_seqlen_q; p.max_sk = max_seqlen_k;
    p.causal = is_causal ? 1 : 0;
    p.scale = softmax_scale;
    p.rng_state = rng_state;
    if (!dropout_args(p_dropout, rng_state, p.drop_thr, p.drop_scale)) return BP_ERR_DROPOUT;
    hipError_t e = bp::launch_flash_bwd(p, dtype, static_cast<hipStream_t>(stream));
    if (e == hipErrorNotSupported) return BP_ERR_HEAD_DIM;
    return e == hipSuccess ? BP_OK : BP_ERR_LAUNCH;
}

int bp_add_layer_norm(const void *x0, const void *x1, const void *gamma, const void *beta, void *z,
                      void *x_out, int64_t rows, int cols, float epsilon, int dtype, int x1_is_f32,
                      int xout_is_f32, int w_is_f32, bp_stream_t stream) {
    return bp_dropout_add_layer_norm(x0, x1, gamma, beta, z, x_out, nullptr, rows, cols, epsilon, dtype, 0,
                                     x1_is_f32, xout_is_f32, w_is_f32, 0.f, nullptr, stream);
}

int bp_dropout_add_layer_norm(const void *x0, const void *x1, const void *gamma, const void *beta, void *z,
                              void *x_out, uint8_t *dmask, int64_t rows, int cols, float epsilon, int dtype,
                              int x0_is_f32, int x1_is_f32, int xout_is_f32, int w_is_f32,
                              float p_dropout, const uint64_t *rng_state, bp_stream_t stream) {
    return bp_dropout_add_layer_norm_scaled(x0, x1, gamma, beta, nullptr, nullptr, z, x_out, dmask, rows, cols, epsilon,
                                            dtype, x0_is_f32, x1_is_f32, xout_is_f32, w_is_f32, p_dropout, rng_state, stream);
}

int bp_dropout_add_layer_norm_scaled(const void *x0, const void *x1, const void *gamma, const void *beta,
                                     const void *rowscale, const void *colscale, void *z, void *x_out, uint8_t *dmask,
                                     int64_t rows, int cols, float epsilon, int dtype, int x0_is_f32, int x1_is_f32,
                                     int xout_is_f32, int w_is_f32, float p_dropout, const uint64_t *rng_state,
                                     bp_stream_t stream) {
    if (dtype != BP_DTYPE_F16 && dtype != BP_DTYPE_BF16) return BP_ERR_DTYPE;
    if (rows <= 0 || rows > 0xffffffffLL || cols <= 0 || cols % 4 != 0 || cols > 8192) return BP_ERR_SHAPE;
    if (x0 == nullptr || gamma == nullptr || beta == nullptr || z == nullptr) return BP_ERR_SHAPE;
    if (!aligned16(x0) || !aligned16(gamma) || !aligned16(beta) || !aligned16(z) ||
        (x1 != nullptr && !aligned16(x1)) || (x_out != nullptr && !aligned16(x_out)) ||
        (colscale != nullptr && !aligned16(colscale)) ||
        (rowscale != nullptr && (reinterpret_cast<uintptr_t>(rowscale) & (x0_is_f32 ? 3u : 1u)) != 0) ||
        (dmask != nullptr && (reinterpret_cast<uintptr_t>(dmask) & 3u) != 0))
        return BP_ERR_SHAPE;
    if (!(isfinite(epsilon) && epsilon >= 0.f)) return BP_ERR_SCALE;
    // one residual dtype: when both x1 and x_out exist they must agree (reference ln_api.cpp:99-102);
    // an fp32 x0 implies an fp32 residual stream
    if (x1 != nullptr && x_out != nullptr && (x1_is_f32 != 0) != (xout_is_f32 != 0)) return BP_ERR_DTYPE;
    if (x0_is_f32 && ((x1 != nullptr && !x1_is_f32) || (x_out != nullptr && !xout_is_f32))) return BP_ERR_DTYPE;
    bp::LnParams p{};
    p.x0 = x0; p.x1 = x1; p.gamma = gamma; p.beta = beta; p.z = z; p.x_out = x_out;
    p.rows = rows; p.cols = cols; p.eps = epsilon;
    p.x1_f32 = x1_is_f32 ? 1 : 0; p.xo_f32 = xout_is_f32 ? 1 : 0; p.w_f32 = w_is_f32 ? 1 : 0;
    p.x0_f32 = x0_is_f32 ? 1 : 0;
    p.dmask = dmask; p.rng_state = rng_state;
    p.rowscale = rowscale; p.colscale = colscale;
    if (!dropout_args(p_dropout, rng_state, p.drop_thr, p.drop_scale)) return BP_ERR_DROPOUT;
    hipError_t e = bp::launch_add_layer_norm(p, dtype, static_cast<hipStream_t>(stream));
    return e == hipSuccess ? BP_OK : BP_ERR_LAUNCH;
}

int bp_softmax_bwd_causal(const void *alpha, void *dalpha_inout, int64_t n_matrices, int seqlen,
                          float softmax_scale, int dtype, bp_stream_t stream) {
    if (dtype != BP_DTYPE_F16 && dtype != BP_DTYPE_BF16) return BP_ERR_DTYPE;
    if (n_matrices <= 0 || seqlen <= 0 || seqlen % 8 != 0 || seqlen > 4096) return BP_ERR_SHAPE;
    if (alpha == nullptr || dalpha_inout == nullptr || !aligned16(alpha) || !aligned16(dalpha_inout)) return BP_ERR_SHAPE;
    if (!scale_ok(softmax_scale)) return BP_ERR_SCALE;
    bp::SoftmaxBwdParams p{};
    p.alpha = alpha; p.dp = dalpha_inout; p.rows = n_matrices * seqlen; p.s = seqlen; p.scale = softmax_scale;
    hipError_t e = bp::launch_softmax_bwd_causal(p, dtype, static_cast<hipStream_t>(stream));
    if (e == hipErrorNotSupported) return BP_ERR_SHAPE;
    return e == hipSuccess ? BP_OK : BP_ERR_LAUNCH;
}

int bp_add_layer_norm_bwd(const void *dz, const void *dx_in, const void *x, const void *gamma,
                          void *dx0, void *dx1, void *dgamma, void *dbeta, float *ws,
                          int64_t rows, int cols, float epsilon, int dtype, int res_is_f32, int w_is_f32,
                          bp_stream_t stream) {
    return bp_dropout_add_layer_norm_bwd(dz, dx_in, x, gamma, dx0, dx1, dgamma, dbeta, ws, rows, cols, epsilon, dtype,
                                         0, res_is_f32, w_is_f32, 0.f, nullptr, stream);
}

int bp_dropout_add_layer_norm_bwd(const void *dz, const void *dx_in, const void *x, const void *gamma,
                                  void *dx0, void *dx1, void *dgamma, void *dbeta, float *ws,
                                  int64_t rows, int cols, float epsilon, int dtype, int x0_is_f32, int res_is_f32,
                                  int w_is_f32, float p_dropout, const uint64_t *rng_state, bp_stream_t stream) {
    // (the unscaled entry point keeps its documented workspace contract: 2 * BP_LN_BWD_WS_ROWS * cols floats)
    return bp_dropout_add_layer_norm_scaled_bwd(dz, dx_in, x, nullptr, gamma, nullptr, nullptr, dx0, dx1, dgamma, dbeta,
                                                nullptr, ws, bp_ln_bwd_ws_floats(cols, 0), rows, cols, epsilon, dtype,
                                                x0_is_f32, res_is_f32, w_is_f32, p_dropout, rng_state, stream);
}

int64_t bp_ln_bwd_ws_floats(int cols, int has_colscale) {
    if (cols <= 0) return 0;
    return (int64_t)(has_colscale ? 3 : 2) * bp::kLnBwdMaxWg * cols;
}

int bp_dropout_add_layer_norm_scaled_bwd(const void *dz, const void *dx_in, const void *x, const void *x0,
                                         const void *gamma, const void *rowscale, const void *colscale,
                                         void *dx0, void *dx1, void *dgamma, void *dbeta, void *dcolscale,
                                         float *ws, int64_t ws_floats, int64_t rows, int cols, float epsilon, int dtype,
                                         int x0_is_f32, int res_is_f32, int w_is_f32, float p_dropout,
                                         const uint64_t *rng_state, bp_stream_t stream) {
    static_assert(BP_LN_BWD_WS_ROWS == bp::kLnBwdMaxWg, "workspace rows");
    if (dtype != BP_DTYPE_F16 && dtype != BP_DTYPE_BF16) return BP_ERR_DTYPE;
    if (rows <= 0 || rows > 0xffffffffLL || cols <= 0 || cols % 4 != 0 || cols > 2048) return BP_ERR_SHAPE;
    if (!dz || !x || !gamma || !dx0 || !dgamma || !dbeta || !ws) return BP_ERR_SHAPE;
    if (colscale != nullptr && (x0 == nullptr || dcolscale == nullptr)) return BP_ERR_SHAPE;   // (layer_norm.py:36-37)
    const void *ptrs[] = {dz, dx_in, x, x0, gamma, colscale, dx0, dx1, dgamma, dbeta, dcolscale, ws};
    for (const void *ptr : ptrs) if (ptr != nullptr && !aligned16(ptr)) return BP_ERR_SHAPE;
    if (rowscale != nullptr && (reinterpret_cast<uintptr_t>(rowscale) & (x0_is_f32 ? 3u : 1u)) != 0) return BP_ERR_SHAPE;
    if (!(isfinite(epsilon) && epsilon >= 0.f)) return BP_ERR_SCALE;
    if (x0_is_f32 && !res_is_f32) return BP_ERR_DTYPE;
    if (ws_floats < bp_ln_bwd_ws_floats(cols, colscale != nullptr)) return BP_ERR_WORKSPACE;
    bp::LnBwdParams p{};
    p.dz = dz; p.dx_in = dx_in; p.x = x; p.gamma = gamma; p.dx0 = dx0; p.dx1 = dx1;
    p.dgamma = dgamma; p.dbeta = dbeta; p.ws = ws;
    p.rows = rows; p.cols = cols; p.eps = epsilon;
    p.n_wg = (int)((rows + 3) / 4 < bp::kLnBwdMaxWg ? (rows + 3) / 4 : bp::kLnBwdMaxWg);
    p.res_f32 = res_is_f32 ? 1 : 0; p.w_f32 = w_is_f32 ? 1 : 0; p.x0_f32 = x0_is_f32 ? 1 : 0;
    p.rng_state = rng_state;
    p.rowscale = rowscale; p.colscale = colscale; p.x0 = x0; p.dcolscale = colscale != nullptr ? dcolscale : nullptr;
    if (!dropout_args(p_dropout, rng_state, p.drop_thr, p.drop_scale)) return BP_ERR_DROPOUT;
    hipError_t e = bp::launch_add_layer_norm_bwd(p, dtype, static_cast<hipStream_t>(stream));
    if (e == hipErrorNotSupported) return BP_ERR_SHAPE;
    return e == hipSuccess ? BP_OK : BP_ERR_LAUNCH;
}

int64_t bp_bias_grad_ws_floats(int64_t rows, int cols) {
    if (rows <= 0 || cols <= 0) return 0;
    return (int64_t)bp::bias_gelu_bwd_slices(rows, cols) * cols;
}

int bp_bias_gelu_fwd(const void *x, const void *bias, void *pre_out, void *y, int64_t rows, int cols, int dtype,
                     bp_stream_t stream) {
    if (dtype != BP_DTYPE_F16 && dtype != BP_DTYPE_BF16) return BP_ERR_DTYPE;
    if (rows <= 0 || cols <= 0 || cols % 8 != 0) return BP_ERR_SHAPE;
    if (x == nullptr || y == nullptr || !aligned16(x) || !aligned16(y)) return BP_ERR_SHAPE;
    if ((bias != nullptr && !aligned16(bias)) || (pre_out != nullptr && !aligned16(pre_out))) return BP_ERR_SHAPE;
    if (pre_out != nullptr && bias == nullptr) return BP_ERR_SHAPE;   // without a bias the pre-activation IS x
    bp::BiasGeluParams p{};
    p.x = x; p.bias = bias; p.pre = pre_out; p.y = y; p.rows = rows; p.cols = cols;
    hipError_t e = bp::launch_bias_gelu_fwd(p, dtype, static_cast<hipStream_t>(stream));
    return e == hipSuccess ? BP_OK : BP_ERR_LAUNCH;
}

static int bias_grad_common(const void *grad, void *dbias, float *ws, int64_t rows, int cols, int dtype) {
    if (dtype != BP_DTYPE_F16 && dtype != BP_DTYPE_BF16) return BP_ERR_DTYPE;
    if (rows <= 0 || cols <= 0 || cols % 8 != 0) return BP_ERR_SHAPE;
    if (grad == nullptr || !aligned16(grad)) return BP_ERR_SHAPE;
    if (dbias != nullptr && (ws == nullptr || !aligned16(ws))) return BP_ERR_SHAPE;
    return BP_OK;
}

int bp_bias_gelu_bwd(const void *grad, const void *pre, void *dpre, void *dbias, float *ws, int64_t rows, int cols,
                     int dtype, int dbias_is_f32, bp_stream_t stream) {
    const int rc = bias_grad_common(grad, dbias, ws, rows, cols, dtype);
    if (rc != BP_OK) return rc;
    if (pre == nullptr || dpre == nullptr || !aligned16(pre) || !aligned16(dpre)) return BP_ERR_SHAPE;
    bp::BiasGeluParams p{};
    p.x = grad; p.pre = const_cast<void *>(pre); p.y = dpre; p.dbias = dbias; p.ws = dbias != nullptr ? ws : nullptr;
    p.rows = rows; p.cols = cols; p.dbias_f32 = dbias_is_f32 ? 1 : 0;
    hipError_t e = bp::launch_bias_gelu_bwd(p, dtype, true, static_cast<hipStream_t>(stream));
    return e == hipSuccess ? BP_OK : BP_ERR_LAUNCH;
}

int bp_column_sum(const void *grad, void *dbias, float *ws, int64_t rows, int cols, int dtype, int dbias_is_f32,
                  bp_stream_t stream) {
    const int rc = bias_grad_common(grad, dbias, ws, rows, cols, dtype);
    if (rc != BP_OK) return rc;
    if (dbias == nullptr) return BP_ERR_SHAPE;
    bp::BiasGeluParams p{};
    p.x = grad; p.dbias = dbias; p.ws = ws; p.rows = rows; p.cols = cols; p.dbias_f32 = dbias_is_f32 ? 1 : 0;
    hipError_t e = bp::launch_bias_gelu_bwd(p, dtype, false, static_cast<hipStream_t>(stream));
    return e == hipSuccess ? BP_OK : BP_ERR_LAUNCH;
}

static int xent_common(int64_t rows, int cols, int64_t row_stride, float smoothing, int dtype) {
    if (dtype != BP_DTYPE_F16 && dtype != BP_DTYPE_BF16 && dtype != BP_DTYPE_F32) return BP_ERR_DTYPE;
    if (rows <= 0 || rows > 0x7fffffffLL || cols <= 0 || row_stride < cols) return BP_ERR_SHAPE;
    if (!(smoothing >= 0.f && smoothing < 1.f)) return BP_ERR_SCALE;
    return BP_OK;
}

int bp_xentropy_fwd(const void *logits, const int64_t *labels, float *losses, float *lse,
                    int64_t rows, int cols, int64_t row_stride, float smoothing, int total_classes,
                    int dtype, bp_stream_t stream) {
    const int rc = xent_common(rows, cols, row_stride, smoothing, dtype);
    if (rc != BP_OK) return rc;
    if (logits == nullptr || labels == nullptr || losses == nullptr || lse == nullptr) return BP_ERR_SHAPE;
    bp::XentParams p{};
    p.logits = logits; p.labels = labels; p.losses = losses; p.lse = lse;
    p.rows = rows; p.cols = cols; p.row_stride = row_stride;
    p.total_classes = total_classes > 0 ? total_classes : cols;
    p.smoothing = smoothing;
    hipError_t e = bp::launch_xentropy_fwd(p, dtype, static_cast<hipStream_t>(stream));
    return e == hipSuccess ? BP_OK : BP_ERR_LAUNCH;
}

int bp_xentropy_bwd(const float *grad_losses, const void *logits, const float *lse, const int64_t *labels,
                    void *grad_logits, int64_t rows, int cols, int64_t row_stride, int64_t grad_row_stride,
                    float smoothing, int total_classes, int dtype, bp_stream_t stream) {
    const int rc = xent_common(rows, cols, row_stride, smoothing, dtype);
    if (rc != BP_OK) return rc;
    if (grad_row_stride < cols) return BP_ERR_SHAPE;
    if (grad_losses == nullptr || logits == nullptr || lse == nullptr || labels == nullptr ||
        grad_logits == nullptr)
        return BP_ERR_SHAPE;
    bp::XentParams p{};
    p.logits = logits; p.labels = labels; p.lse = const_cast<float *>(lse); p.grad_losses = grad_losses;
    p.grad_logits = grad_logits;
    p.rows = rows; p.cols = cols; p.row_stride = row_stride; p.grad_row_stride = grad_row_stride;
    p.total_classes = total_classes > 0 ? total_classes : cols;
    p.smoothing = smoothing;
    hipError_t e = bp::launch_xentropy_bwd(p, dtype, static_cast<hipStream_t>(stream));
    return e == hipSuccess ? BP_OK : BP_ERR_LAUNCH;
}

}  // extern "C"

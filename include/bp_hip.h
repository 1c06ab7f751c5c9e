/*
 * bp_hip.h -- C ABI of libbackpack_hip.so: the MI355X (gfx950) implementation of the
 * Backpack forward hot path.
 *
 * The reference (john-hewitt/backpacks-flash-attn) has no C ABI: its native boundary is the
 * pybind11 module `flash_attn_cuda` (csrc/flash_attn/fmha_api.cpp:776-782) taking at::Tensor,
 * and its Backpack-specific ops are eager ATen calls (training/src/models/backpack.py:107-122,313).
 * Every entry point below names the reference interface it replaces.  All of them
 *   - take raw DEVICE pointers, explicit sizes and int64 ELEMENT strides (no torch types),
 *   - allocate nothing and keep no caller-visible state (re-entrant; scratch is passed in by the caller).
 *     bp_sense_mix* / bp_sense_mix_dc launch persistent workgroups that pull jobs from ticket queues in a 64-byte
 *     record of device memory, `queue_ws` (BP_QUEUE_WS_BYTES, 16-byte aligned, contents undefined on entry: a
 *     memset node in front of the kernel zeroes it on the stream; it must belong to this launch alone until the
 *     launch has completed -- so a captured HIP graph owns the record it replays).  queue_ws == NULL takes the next
 *     record of a 64-entry ring owned by the library: fine for eager launches with fewer than 64 of them in flight;
 *     a NULL queue_ws on a stream that is being captured returns BP_ERR_QUEUE_WS (a replayed graph would share the
 *     library's record with every other launch),
 *   - enqueue on the given hipStream_t and return without synchronising,
 *   - return 0 on success or a negative BP_ERR_* (the Python layer raises RuntimeError, which
 *     is what TORCH_CHECK failures surface as in the reference: fmha_api.cpp:206-250).
 */
#ifndef BP_HIP_H
#define BP_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BP_ABI_VERSION 9   /* 2: *_dropout entry points added; 3: bias/GELU + column-sum entry points added, the
                              persistent sense-mix launches take a caller-owned `queue_ws`; 4: bp_flash_bwd* take the
                              size of `dsum_ws` (bp_flash_bwd_ws_floats) and check it, queue_ws == NULL is refused
                              while the stream is capturing (BP_ERR_QUEUE_WS); 5: bp_dropout_add_layer_norm_scaled{,_bwd}
                              (rowscale / colscale of the reference's dropout_add_ln) added; 6: bp_sense_mix_gather added;
                              7: bp_sense_mix_gather clamps row_index to the table and takes tables of at most 65 536 rows;
                              8: bp_sense_lse / _alpha / _mix / _mix_weighted take sense widths d_k up to 640 (wide senses:
                              the reference's vecs-4 / vecs-1 ablations), bp_build_flags() added;
                              9: bp_sense_mix_gather takes the two few-sense widths d_k = 160 / 640 (seqlen % 32 == 0; any
                              number of table rows) */

/* element type of q/k/v/out/content tensors */
#define BP_DTYPE_F16 0
#define BP_DTYPE_BF16 1
#define BP_DTYPE_F32 2   /* accepted by the cross-entropy entry points only */

#define BP_OK 0
#define BP_ERR_DTYPE -1       /* dtype is not BP_DTYPE_F16 / BP_DTYPE_BF16         (fmha_api.cpp:215-219) */
#define BP_ERR_HEAD_DIM -2    /* head_dim < 1 or > 128 (sense width d_k > 640)      (fmha_api.cpp:245)     */
#define BP_ERR_SHAPE -3       /* batch/nheads/seqlen <= 0, or a required pointer is NULL (fmha_api.cpp:244-252) */
#define BP_ERR_SCALE -4       /* softmax_scale is not finite or not > 0                                     */
#define BP_ERR_LAUNCH -5      /* hipLaunchKernel failed (FMHA_CHECK_CUDA, src/fmha_utils.h:39)              */
#define BP_ERR_DOUT -6        /* sense mix: d_out < 1                                                       */
#define BP_ERR_DROPOUT -7     /* p_dropout outside [0,1), rng_state NULL with p > 0, or a shape the dropout path lacks */
#define BP_ERR_QUEUE_WS -8    /* persistent launch with queue_ws == NULL on a stream that is being captured            */
#define BP_ERR_WORKSPACE -9   /* a caller-provided workspace is smaller than the entry point's *_ws_floats() query     */

#define BP_QUEUE_WS_BYTES 64   /* `queue_ws` of the persistent sense-mix launches */

typedef void *bp_stream_t; /* a hipStream_t */

/* Human-readable text for a BP_ERR_* code (static storage). */
const char *bp_strerror(int code);
int bp_abi_version(void);
/* 0 for a product build.  Bit 0: -DBP_FWD_WHATIF, bit 1: -DBP_BWD_WHATIF (timing builds that delete work on purpose:
 * results are garbage), bit 2: -DBP_DEV_BUILD (run-time experiment switches compiled in).  A binding should refuse to
 * load a library with bit 0 or 1 set as its default one. */
int bp_build_flags(void);

/*
 * bp_flash_fwd -- fused attention forward  O = softmax(scale * Q K^T [+ causal mask]) V  and the
 * row log-sum-exp.  Replaces flash_attn_cuda.fwd / mha_fwd (csrc/flash_attn/fmha_api.cpp:189-325),
 * no-dropout path; called by _flash_attn_forward (flash_attn/flash_attn_interface.py:13-28).
 *
 *   q            (total_q, nheads, head_dim) 16-bit, last stride 1; row/head strides free
 *   k, v         (total_k, nheads, head_dim) likewise.  v == NULL and out == NULL: LSE only.
 *   out          (total_q, nheads, head_dim), caller-allocated, written in place
 *   softmax_lse  (batch, nheads, lse_stride) fp32, natural log; -inf for a row with no key;
 *                entries >= that sequence's length are left untouched (fmha_api.cpp:276)
 *   cu_seqlens_* int32 (batch+1) device arrays of row offsets; NULL means fixed length:
 *                sequence b occupies rows [b*max_seqlen, (b+1)*max_seqlen)
 *   is_causal    mask is top-left aligned: key j visible to query i iff j <= i
 *                (csrc/flash_attn/src/fmha/mask.h:57-70)
 * Any head_dim in [1,128] is accepted; the 16-byte vector path needs head_dim % 8 == 0 with
 * 16-byte aligned rows (the reference's only mode), other shapes take an element-wise loader.
 */
int bp_flash_fwd(const void *q, const void *k, const void *v, void *out, float *softmax_lse,
                 const int32_t *cu_seqlens_q, const int32_t *cu_seqlens_k,
                 int batch, int nheads, int head_dim, int max_seqlen_q, int max_seqlen_k,
                 int64_t q_row_stride, int64_t q_head_stride,
                 int64_t k_row_stride, int64_t k_head_stride,
                 int64_t v_row_stride, int64_t v_head_stride,
                 int64_t o_row_stride, int64_t o_head_stride,
                 int64_t lse_stride, float softmax_scale, int is_causal, int dtype,
                 bp_stream_t stream);

/*
 * bp_flash_fwd_dropout -- bp_flash_fwd with in-kernel attention dropout (training): replaces the p_dropout > 0
 * path of flash_attn_cuda.fwd (csrc/flash_attn/fmha_api.cpp:199,306-320; kernel
 * src/fmha_fprop_kernel_1xN.h:494-506).  O = (dropout(P) / (1 - p)) V; the row log-sum-exp is that of the
 * UNdropped probabilities, as upstream.
 *   p_dropout  in [0, 1); 0 = identical to bp_flash_fwd (rng_state may then be NULL)
 *   rng_state  DEVICE pointer to two uint64 {seed, offset} (the role of the at::Generator's philox state,
 *              fmha_api.cpp:314-320).  Read by the kernel, never written; hand the SAME two words to
 *              bp_flash_bwd_dropout / bp_attn_probs_dropout to regenerate the same mask.  The mask is a pure
 *              function of (seed, offset, batch*nheads index, query index, key index): Philox2x32-10 keyed per
 *              (batch, head), one call per run of 4 keys, 16-bit uniforms against round((1-p) * 65536)
 *              (csrc/bp_philox.h; tests/philox_ref.py restates it on the host).
 * Dropout needs the 16-byte vector path (head_dim % 8 == 0, aligned rows); BP_ERR_DROPOUT otherwise.
 */
int bp_flash_fwd_dropout(const void *q, const void *k, const void *v, void *out, float *softmax_lse,
                         const int32_t *cu_seqlens_q, const int32_t *cu_seqlens_k,
                         int batch, int nheads, int head_dim, int max_seqlen_q, int max_seqlen_k,
                         int64_t q_row_stride, int64_t q_head_stride,
                         int64_t k_row_stride, int64_t k_head_stride,
                         int64_t v_row_stride, int64_t v_head_stride,
                         int64_t o_row_stride, int64_t o_head_stride,
                         int64_t lse_stride, float softmax_scale, int is_causal, int dtype,
                         float p_dropout, const uint64_t *rng_state, bp_stream_t stream);

/*
 * bp_attn_probs -- materialise normalised attention probabilities
 *   P[b,h,i,j] = exp(scale * q_i.k_j - lse[b,h,i])  for visible (i,j), exactly 0 elsewhere.
 * Serves `return_attn_probs=True` of the flash interface (flash_attn_interface.py:242-267; the
 * reference returns S_dmask from the same launch, fmha_api.cpp:279,322-324) and is the second pass
 * of bp_sense_alpha.  Fixed-length batches only.
 *   probs (batch, nheads, seqlen_q, seqlen_k) 16-bit, strides p_batch/p_head/p_row, last stride 1.
 */
int bp_attn_probs(const void *q, const void *k, const float *softmax_lse, void *probs,
                  int batch, int nheads, int head_dim, int seqlen_q, int seqlen_k,
                  int64_t q_batch_stride, int64_t q_row_stride, int64_t q_head_stride,
                  int64_t k_batch_stride, int64_t k_row_stride, int64_t k_head_stride,
                  int64_t lse_stride,
                  int64_t p_batch_stride, int64_t p_head_stride, int64_t p_row_stride,
                  float softmax_scale, int is_causal, int dtype, bp_stream_t stream);

/*
 * bp_attn_probs_dropout -- bp_attn_probs that also reports the dropout mask of bp_flash_fwd_dropout called with
 * the same (p_dropout, rng_state): a DROPPED entry is stored with its sign bit set (-P, or -0.0), a kept one
 * as +P.  Same encoding idea as the reference's S_dmask (fmha_api.cpp:279; tests/test_flash_attn.py:181-236
 * decode it with `S >= 0`); here P is already normalised, so decode with the sign BIT, not with `< 0`.
 */
int bp_attn_probs_dropout(const void *q, const void *k, const float *softmax_lse, void *probs,
                          int batch, int nheads, int head_dim, int seqlen_q, int seqlen_k,
                          int64_t q_batch_stride, int64_t q_row_stride, int64_t q_head_stride,
                          int64_t k_batch_stride, int64_t k_row_stride, int64_t k_head_stride,
                          int64_t lse_stride,
                          int64_t p_batch_stride, int64_t p_head_stride, int64_t p_row_stride,
                          float softmax_scale, int is_causal, int dtype,
                          float p_dropout, const uint64_t *rng_state, bp_stream_t stream);

/*
 * bp_sense_lse -- log-sum-exp of every (sense, query) row of the causal sense attention:
 *   lse[b,l,t] = log sum_{s<=t} exp(scale * q_l[t].k_l[s])
 * First pass of bp_sense_alpha / bp_sense_mix, exported so callers can share one LSE between them
 * (and time the passes separately).  The reference computes this inside torch.softmax
 * (training/src/models/backpack.py:122).
 *   lse (batch, nsenses, roundup(seqlen,16)) fp32, natural log.
 * Sense widths (this entry point, bp_sense_alpha, bp_sense_mix, bp_sense_mix_weighted): 1 <= d_k <= 640.  Up to 128 the
 * attention-class kernels run (LDS-DMA path for 16-byte friendly layouts); 129 ... 640 -- the reference's few-sense
 * ablations, training/configs/experiment/owt/backpack-mini-flash-vecs-4.yaml (d_k = 160) and ...-vecs-1.yaml (640) --
 * the wide kernels of csrc/sense_wide.hip (any alignment; d_k % 8 == 0 with 16-byte aligned rows takes 16-byte loads).
 * At d_k = 160 / 640 exactly, on 16-byte friendly operands with seqlen % 32 == 0, the LDS-DMA ring kernels of
 * csrc/sense_wide_dma.hip run instead (bp_sense_lse, bp_sense_mix; bp_sense_mix_gather takes only these two widths beyond 128).
 * The backward entry points stay at d_k <= 128.
 */
int bp_sense_lse(const void *qk, float *lse, int batch, int seqlen, int nsenses, int d_k,
                 int64_t qk_batch_stride, int64_t qk_row_stride, int64_t qk_two_stride,
                 int64_t qk_sense_stride, float softmax_scale, int dtype, bp_stream_t stream);

/*
 * bp_sense_alpha -- Backpack contextualisation weights
 *   alpha[b,l,t,s] = softmax_s( q_l[t].k_l[s] * scale ) over s <= t, 0 for s > t.
 * Replaces the eager body of ContextSelfAttn.forward after its Wqkv projection
 * (training/src/models/backpack.py:112-122).
 *   qk     (batch, seqlen, 2, nsenses, d_k) 16-bit: the Wqkv output viewed as in backpack.py:111;
 *          element strides qk_batch/qk_row/qk_two/qk_sense, last stride 1
 *   alpha  (batch, nsenses, seqlen, seqlen) 16-bit contiguous, caller-allocated
 *   lse_ws fp32 scratch, batch * nsenses * roundup(seqlen,16) elements
 *   lse_ready  0: compute the LSE into lse_ws first;  1: lse_ws already holds bp_sense_lse's result
 */
int bp_sense_alpha(const void *qk, void *alpha, float *lse_ws, int lse_ready,
                   int batch, int seqlen, int nsenses, int d_k,
                   int64_t qk_batch_stride, int64_t qk_row_stride, int64_t qk_two_stride,
                   int64_t qk_sense_stride, float softmax_scale, int dtype, bp_stream_t stream);

/*
 * bp_sense_mix -- fused sense-weighted combination, alpha never materialised:
 *   out[b,t,:] = sum_l sum_{s<=t} alpha[b,l,t,s] * content[b,s,l,:]
 * Replaces `torch.sum(contextualization @ content, dim=1)` together with the softmax that
 * produced `contextualization` (training/src/models/backpack.py:305,313).
 *   qk       as in bp_sense_alpha
 *   content  (batch, seqlen, nsenses, d_out) 16-bit -- the (B,S,k*d) output of the content
 *            model's final MLP before its reshape/transpose (backpack.py:274-276); element strides
 *            c_batch/c_row/c_sense, last stride 1.  d_out is free (vocab-sized content works).
 *   out      (batch, seqlen, d_out) 16-bit, strides o_batch/o_row
 *   lse_ws   fp32 scratch, batch * nsenses * roundup(seqlen,16) elements
 *   lse_ready  as in bp_sense_alpha
 *   queue_ws   BP_QUEUE_WS_BYTES of device memory for the persistent launch, or NULL (see the top of this file)
 */
int bp_sense_mix(const void *qk, const void *content, void *out, float *lse_ws, int lse_ready,
                 int batch, int seqlen, int nsenses, int d_k, int d_out,
                 int64_t qk_batch_stride, int64_t qk_row_stride, int64_t qk_two_stride,
                 int64_t qk_sense_stride,
                 int64_t c_batch_stride, int64_t c_row_stride, int64_t c_sense_stride,
                 int64_t o_batch_stride, int64_t o_row_stride,
                 float softmax_scale, int dtype, void *queue_ws, bp_stream_t stream);

/*
 * bp_sense_mix_weighted -- bp_sense_mix with the intervention hook fused in:
 *   out[b,t,:] = sum_l sum_{s<=t} alpha[b,l,t,s] * key_weight[b,l,s] * content[b,s,l,:]
 * i.e. column s of sense l's contextualisation (equivalently: row s of that sense's content) is
 * scaled before the contraction.  Covers, without materialising alpha or a re-weighted copy of the
 * content, the reference's control experiments:
 *   - `contextualization[0, vector_index, :, index] *= percent` then `sum(contextualization @ content)`
 *     (training/src/test_genderbias.py:71-78),
 *   - `content * content_weights.transpose(1,2).unsqueeze(3)` then the same contraction
 *     (training/src/models/intervened_models.py:97-101).
 *   key_weight  (batch, nsenses, seqlen) fp32, unit stride along seqlen, element strides
 *               kw_batch/kw_sense; NULL = no weighting (then identical to bp_sense_mix)
 * All other arguments as bp_sense_mix.
 */
int bp_sense_mix_weighted(const void *qk, const void *content, const float *key_weight, void *out,
                          float *lse_ws, int lse_ready,
                          int batch, int seqlen, int nsenses, int d_k, int d_out,
                          int64_t qk_batch_stride, int64_t qk_row_stride, int64_t qk_two_stride,
                          int64_t qk_sense_stride,
                          int64_t c_batch_stride, int64_t c_row_stride, int64_t c_sense_stride,
                          int64_t kw_batch_stride, int64_t kw_sense_stride,
                          int64_t o_batch_stride, int64_t o_row_stride,
                          float softmax_scale, int dtype, void *queue_ws, bp_stream_t stream);

/*
 * bp_sense_mix_gather -- bp_sense_mix with the content rows taken from a TABLE through a row index:
 *   out[b,t,:] = sum_l sum_{s<=t} alpha[b,l,t,s] * table[row_index[b,s], l, :]
 * The sense vectors C_l(x_s) of the reference depend on the token x_s alone (training/src/models/backpack.py:251-276: word
 * embedding without positions, an Identity mixer, per-token MLPs), so inference can run the content network once per
 * DISTINCT token of a batch (table = its output for the sorted distinct ids, row_index = torch.unique's inverse) and never
 * materialise the (batch, seqlen, nsenses * d_out) content tensor the reference builds (:276, :313).
 *   table      (table_rows, nsenses, d_out), element strides t_row_stride / t_sense_stride, last dim contiguous
 *   row_index  (batch, seqlen) int32, unit stride along seqlen, element stride idx_batch_stride; 0 <= value < table_rows.
 *              NOT validated: the kernel clamps every index as an unsigned value to table_rows - 1, so a negative or too
 *              large index silently reads the table's LAST row (never memory outside the table); callers that need an
 *              error for bad ids check them before the call (the reference's nn.Embedding asserts on the device)
 * Restrictions (BP_ERR_SHAPE otherwise; callers gather the rows themselves and call bp_sense_mix): the 16-byte vector
 * path (d_k % 8 == 0, d_out % 8 == 0, aligned bases, strides multiples of 8), seqlen <= 4096 and table_rows <= 65536 (a
 * job's row indices are kept in 8 KB of LDS as u16; ABI 6 took any row count at seqlen <= 4096 / 2048), and
 * table_rows * t_row_stride * 2 bytes < 4 GiB (row offsets are 32-bit in the DMA instruction).  Senses wider than 128:
 * d_k = 160 or 640 with seqlen % 32 == 0 only (ABI 9; row indices are kept as u32, any table_rows); BP_ERR_HEAD_DIM otherwise.
 * All other arguments as bp_sense_mix.
 */
int bp_sense_mix_gather(const void *qk, const void *table, const int32_t *row_index, void *out,
                        float *lse_ws, int lse_ready,
                        int batch, int seqlen, int nsenses, int d_k, int d_out, int64_t table_rows,
                        int64_t qk_batch_stride, int64_t qk_row_stride, int64_t qk_two_stride,
                        int64_t qk_sense_stride,
                        int64_t t_row_stride, int64_t t_sense_stride, int64_t idx_batch_stride,
                        int64_t o_batch_stride, int64_t o_row_stride,
                        float softmax_scale, int dtype, void *queue_ws, bp_stream_t stream);

/*
 * bp_sense_mix_dc -- backward of bp_sense_mix with respect to the content:
 *   dcontent[b,s,l,:] = sum_{t>=s} alpha[b,l,t,s] * dout[b,t,:]
 * alpha is recomputed from qk and the saved log-sum-exp; no (batch, nsenses, seqlen, seqlen) tensor exists.  The
 * reference leaves this product to autograd through `torch.sum(contextualization @ content, dim=1)`
 * (training/src/models/backpack.py:313).
 *   qk        as in bp_sense_alpha, d_k % 8 == 0 (pad the projection, see ContextSelfAttn.project)
 *   dout      (batch, seqlen, d_out) 16-bit, strides do_batch/do_row, last stride 1, d_out % 8 == 0
 *   lse       (batch, nsenses, roundup(seqlen,16)) fp32: bp_sense_lse's result for this qk
 *   dcontent  (batch, seqlen, nsenses, d_out) 16-bit, strides c_batch/c_row/c_sense (the content's own layout)
 */
int bp_sense_mix_dc(const void *qk, const void *dout, const float *lse, void *dcontent,
                    int batch, int seqlen, int nsenses, int d_k, int d_out,
                    int64_t qk_batch_stride, int64_t qk_row_stride, int64_t qk_two_stride, int64_t qk_sense_stride,
                    int64_t do_batch_stride, int64_t do_row_stride,
                    int64_t c_batch_stride, int64_t c_row_stride, int64_t c_sense_stride,
                    float softmax_scale, int dtype, void *queue_ws, bp_stream_t stream);

/*
 * bp_sense_dq_dk -- backward of the sense weights with respect to qk, for ONE slab of 128 queries [t0, t0 + 128):
 *   D_l[t]  = sum_s alpha_l[t,s] dP_l[t,s]        dS_l[t,s] = alpha_l[t,s] (dP_l[t,s] - D_l[t])
 *   dq_l[t] = scale sum_s dS_l[t,s] k_l[s]   -> written to dqk[b, t, 0, l, :]   (rows of the slab)
 *   dk_l[s] += scale sum_{t in slab} dS_l[t,s] q_l[t]   -> ADDED to dk_acc[b, s, l, :] (fp32; zero it before the
 *              first slab, run the slabs one after the other on one stream: the sum is then deterministic)
 * where dP_l[t,s] = dout[b,t,:] . content[b,s,l,:] comes precomputed, TRANSPOSED, from the caller:
 *   dpt  (batch, N, 128) 16-bit, row index s * nsenses + l, column = query t0 + j, N = min(seqlen, t0 + 128) *
 *        nsenses: the plain GEMM  content.view(batch, seqlen*nsenses, d)[:, :N] @ dout[:, t0:t0+128].T  (columns of
 *        queries past the sequence: anything finite).  A (B, S*k, 128) buffer, 1/(S/128) of the alpha-sized ones.
 *   dsum_ws  (batch, nsenses, roundup(seqlen,16)) fp32 scratch (D of the slab's rows is put there)
 *   dqk      16-bit, laid out like qk (strides dqk_batch/dqk_row/dqk_sense, the q half at offset 0)
 * Replaces what autograd does for ContextSelfAttn.forward (training/src/models/backpack.py:116-122).
 */
int bp_sense_dq_dk(const void *qk, const void *dpt, const float *lse, float *dsum_ws, void *dqk, float *dk_acc,
                   int batch, int seqlen, int nsenses, int d_k, int t0,
                   int64_t qk_batch_stride, int64_t qk_row_stride, int64_t qk_two_stride, int64_t qk_sense_stride,
                   int64_t dpt_batch_stride,
                   int64_t dqk_batch_stride, int64_t dqk_row_stride, int64_t dqk_sense_stride,
                   int64_t dka_batch_stride, int64_t dka_row_stride, int64_t dka_sense_stride,
                   float softmax_scale, int dtype, bp_stream_t stream);

/*
 * bp_flash_bwd -- attention backward: dq, dk, dv from q, k, v, dout and the forward's out and softmax_lse.
 * Replaces flash_attn_cuda.bwd / mha_bwd (csrc/flash_attn/fmha_api.cpp:337-504) called by
 * _flash_attn_backward (flash_attn/flash_attn_interface.py:31-47), no-dropout path.  P is recomputed
 * from the LSE as upstream; the result is deterministic (no atomics).
 *   q, dout, out, dq  (total_q, nheads, head_dim); k, v, dk, dv (total_k, nheads, head_dim): 16-bit, last
 *                stride 1, 16-byte aligned rows, head_dim % 8 == 0 and <= 128
 *   softmax_lse  (batch, nheads, lse_stride) fp32
 *   dsum_ws      (batch, nheads, 2, lse_stride) fp32 workspace, contents undefined on entry: the kernels put the row
 *                statistics there, -D[b,h,i] = -sum_d dout_i[d] * out_i[d] (upstream's dsoftmax_sum,
 *                fmha_api.cpp:421) and -softmax_lse[b,h,i] / softmax_scale
 *   dsum_ws_floats  number of floats `dsum_ws` holds; must be >= bp_flash_bwd_ws_floats(batch, nheads, lse_stride)
 *                (BP_ERR_WORKSPACE otherwise: the workspace doubled between ABI 2 and 3, so its size is now an
 *                argument instead of a sentence in this comment)
 *   cu_seqlens_*       as in bp_flash_fwd (NULL = fixed length)
 */
int64_t bp_flash_bwd_ws_floats(int batch, int nheads, int64_t lse_stride);
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
                 bp_stream_t stream);

/*
 * bp_flash_bwd_dropout -- bp_flash_bwd for a forward that ran bp_flash_fwd_dropout: the p_dropout > 0 path of
 * flash_attn_cuda.bwd (fmha_api.cpp:337-504), which upstream feeds with the generator state saved by the
 * forward (flash_attn_interface.py:53-68: `rng_state`).  `out` must be the dropped-out forward output and
 * (p_dropout, rng_state) the forward's; the kernels regenerate the mask.
 */
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
                         float p_dropout, const uint64_t *rng_state, bp_stream_t stream);

/*
 * bp_add_layer_norm -- fused residual add + LayerNorm forward (eval path):
 *   x = x0 + x1 ;  z = (x - mean) * rsqrt(var + eps) * gamma + beta     (fp32 math)
 * Replaces dropout_layer_norm.dropout_add_ln_fwd with dropout_p = 0 and no rowscale / colscale /
 * subset (reference csrc/layer_norm/ln_api.cpp:83-254, called from
 * flash_attn/ops/layer_norm.py:9-25 by Block.forward, flash_attn/modules/block.py:83-104, and by
 * GPTModel.forward, flash_attn/models/gpt.py:236-240).
 *   x0       (rows, cols) contiguous, dtype `dtype` (fp16 / bf16)
 *   x1       residual in, (rows, cols) contiguous, fp32 if x1_is_f32 else `dtype`; may be NULL
 *   gamma, beta  (cols), fp32 if w_is_f32 else `dtype`
 *   z        (rows, cols) in `dtype` (the reference's otype = itype, ln_api.cpp:104)
 *   x_out    residual out = x0 + x1 rounded to its dtype (fp32 if xout_is_f32 else `dtype`); may be
 *            NULL.  z is computed from the unrounded fp32 sum (ln_fwd_kernels.cuh:131-133).
 * cols must be a multiple of 4 and <= 8192; all pointers 16-byte aligned.
 */
int bp_add_layer_norm(const void *x0, const void *x1, const void *gamma, const void *beta, void *z,
                      void *x_out, int64_t rows, int cols, float epsilon, int dtype, int x1_is_f32,
                      int xout_is_f32, int w_is_f32, bp_stream_t stream);

/*
 * bp_dropout_add_layer_norm -- the full forward of the reference's dropout_add_ln_fwd minus rowscale / colscale /
 * subset (csrc/layer_norm/ln_api.cpp:83-254; kernel ln_fwd_kernels.cuh:76,112-134):
 *   x = dropout(x0) / (1 - p) + x1 ;  z = LayerNorm(x)
 * Arguments as bp_add_layer_norm, plus
 *   x0_is_f32  x0 (and therefore z: otype = itype, ln_api.cpp:104) is fp32 instead of `dtype` -- the AMP case,
 *              where the fp32 embedding output enters the first LayerNorm; requires an fp32 residual stream
 *   dmask      optional (rows, cols) uint8 keep mask out (1 = kept; 4-byte aligned), written only when p_dropout > 0
 *              -- what `return_dropout_mask=True` hands back (flash_attn/ops/layer_norm.py:207-217)
 *   p_dropout, rng_state   as in bp_flash_fwd_dropout; here ONE stream per call with counter (row, column / 4)
 * rows must be < 2^32.
 */
int bp_dropout_add_layer_norm(const void *x0, const void *x1, const void *gamma, const void *beta, void *z,
                              void *x_out, uint8_t *dmask, int64_t rows, int cols, float epsilon, int dtype,
                              int x0_is_f32, int x1_is_f32, int xout_is_f32, int w_is_f32,
                              float p_dropout, const uint64_t *rng_state, bp_stream_t stream);

/*
 * bp_softmax_bwd_causal -- backward of the causal softmax behind the sense weights (training path of
 * ContextSelfAttn.forward, training/src/models/backpack.py:112-122, which the reference leaves to autograd):
 *   dscores[n,t,s] = scale * alpha[n,t,s] * (dalpha[n,t,s] - sum_{s'<=t} alpha[n,t,s'] dalpha[n,t,s'])  for s <= t,
 *   0 above the diagonal.  In place: `dalpha_inout` holds dalpha on entry and dscores on return.
 *   alpha, dalpha_inout  (n_matrices, seqlen, seqlen) 16-bit contiguous, 16-byte aligned;
 *   seqlen % 8 == 0 and <= 4096 (BP_ERR_SHAPE otherwise)
 */
int bp_softmax_bwd_causal(const void *alpha, void *dalpha_inout, int64_t n_matrices, int seqlen,
                          float softmax_scale, int dtype, bp_stream_t stream);

/*
 * bp_add_layer_norm_bwd -- backward of bp_add_layer_norm (eval-path subset of the reference's
 * dropout_add_ln_bwd, csrc/layer_norm/ln_api.cpp:256-408; Python side flash_attn/ops/layer_norm.py:131-152):
 *   dx = rs (dz*gamma - mean(dz*gamma) - xhat mean(dz*gamma*xhat)) + dx_in ;  dgamma = sum dz*xhat ; dbeta = sum dz
 * with mean / rstd recomputed from `x`, the summed stream the forward normalised (its x_out; x0 itself when
 * the forward had no residual).  dx0 and dx1 receive the same values in their own dtypes.
 *   dz, dx0      (rows, cols) 16-bit           dx_in, x, dx1  (rows, cols) residual dtype (dx_in / dx1 may be NULL)
 *   gamma, dgamma, dbeta  (cols) fp32 or 16-bit (w_is_f32)
 *   ws           fp32 workspace, 2 * BP_LN_BWD_WS_ROWS * cols elements
 * cols % 4 == 0 and <= 2048 (BP_ERR_SHAPE otherwise: callers differentiate the eager expression instead).
 */
#define BP_LN_BWD_WS_ROWS 1024
int bp_add_layer_norm_bwd(const void *dz, const void *dx_in, const void *x, const void *gamma,
                          void *dx0, void *dx1, void *dgamma, void *dbeta, float *ws,
                          int64_t rows, int cols, float epsilon, int dtype, int res_is_f32, int w_is_f32,
                          bp_stream_t stream);

/*
 * bp_dropout_add_layer_norm_bwd -- backward of bp_dropout_add_layer_norm (dropout_add_ln_bwd,
 * ln_api.cpp:256-408): as bp_add_layer_norm_bwd, with dx0 = dropout-masked dx / (1 - p) (dx1 stays dx).  The
 * mask is regenerated from the forward's (p_dropout, rng_state); the reference reads its saved dmask instead.
 *   x0_is_f32  dz and dx0 are fp32 (the forward's x0 / z dtype)
 */
int bp_dropout_add_layer_norm_bwd(const void *dz, const void *dx_in, const void *x, const void *gamma,
                                  void *dx0, void *dx1, void *dgamma, void *dbeta, float *ws,
                                  int64_t rows, int cols, float epsilon, int dtype, int x0_is_f32, int res_is_f32,
                                  int w_is_f32, float p_dropout, const uint64_t *rng_state, bp_stream_t stream);

/*
 * bp_dropout_add_layer_norm_scaled / _scaled_bwd -- bp_dropout_add_layer_norm{,_bwd} with the two scale vectors of the
 * reference's dropout_add_ln_fwd / _bwd (csrc/layer_norm/ln_api.cpp:83-254,256-408; kernels ln_fwd_kernels.cuh:99,123-125,
 * ln_bwd_kernels.cuh:183-195; Python side flash_attn/ops/layer_norm.py:207-217 `rowscale`, `layerscale`):
 *   x = dropout(x0 * rowscale[row]) / (1 - p) * colscale[col] + x1 ;  z = LayerNorm(x)
 *   dx0 = dx * rowscale[row] * mask / (1 - p) * colscale[col] ;  dcolscale[col] = sum_rows dx * rowscale[row] * mask / (1 - p) * x0
 *   rowscale  (rows) in x0's dtype (fp32 when x0_is_f32), or NULL      -- DropPath: Bernoulli(survival) / survival per row
 *   colscale  (cols) in gamma's dtype, or NULL                        -- LayerScale
 *   x0        backward only, the forward's x0 (needed for dcolscale; may be NULL without a colscale)
 *   dcolscale (cols) in gamma's dtype; required with a colscale
 *   ws        fp32 workspace of `ws_floats` elements >= bp_ln_bwd_ws_floats(cols, colscale != NULL) (BP_ERR_WORKSPACE otherwise)
 * With both vectors NULL these are the unscaled entry points (which forward to them).  The `subset` arguments of the
 * reference (x0_subset / out_subset / rowscale_const, a ViT token-dropping feature) are not part of this ABI.
 */
int64_t bp_ln_bwd_ws_floats(int cols, int has_colscale);
int bp_dropout_add_layer_norm_scaled(const void *x0, const void *x1, const void *gamma, const void *beta,
                                     const void *rowscale, const void *colscale, void *z, void *x_out, uint8_t *dmask,
                                     int64_t rows, int cols, float epsilon, int dtype, int x0_is_f32, int x1_is_f32,
                                     int xout_is_f32, int w_is_f32, float p_dropout, const uint64_t *rng_state,
                                     bp_stream_t stream);
int bp_dropout_add_layer_norm_scaled_bwd(const void *dz, const void *dx_in, const void *x, const void *x0,
                                         const void *gamma, const void *rowscale, const void *colscale,
                                         void *dx0, void *dx1, void *dgamma, void *dbeta, void *dcolscale,
                                         float *ws, int64_t ws_floats, int64_t rows, int cols, float epsilon, int dtype,
                                         int x0_is_f32, int res_is_f32, int w_is_f32, float p_dropout,
                                         const uint64_t *rng_state, bp_stream_t stream);

/*
 * bp_xentropy_fwd / bp_xentropy_bwd -- fused softmax cross-entropy over vocabulary-sized rows.
 * Replace xentropy_cuda_lib.forward(logits, labels, smoothing[, total_classes]) -> (losses, lse) and
 * xentropy_cuda_lib.backward(grad_loss, logits, lse, labels, smoothing, inplace, total_classes)
 * (flash_attn/losses/cross_entropy.py:37,54,103-105):
 *   lse_i  = log sum_j exp(x_ij)
 *   loss_i = (1 - s)(lse_i - x_i[y_i]) + s (lse_i - sum_j x_ij / total_classes)
 *   dx_ij  = g_i (exp(x_ij - lse_i) - (1 - s)[j == y_i] - s / total_classes)
 * A label outside [0, cols) has no x_i[y_i] term (shifted labels of the vocabulary-parallel caller,
 * cross_entropy.py:41-63); rows with the ignore index are zeroed by the caller (:39,:101).
 *   logits       (rows, cols) fp16 / bf16 / fp32 (dtype 0 / 1 / 2), row stride in elements, last stride 1
 *   labels       (rows) int64
 *   losses, lse  (rows) fp32
 *   grad_logits  (rows, cols) in the logits' dtype; MAY BE the logits buffer itself (inplace_backward)
 *   total_classes  <= 0: cols
 */
int bp_xentropy_fwd(const void *logits, const int64_t *labels, float *losses, float *lse,
                    int64_t rows, int cols, int64_t row_stride, float smoothing, int total_classes,
                    int dtype, bp_stream_t stream);
int bp_xentropy_bwd(const float *grad_losses, const void *logits, const float *lse, const int64_t *labels,
                    void *grad_logits, int64_t rows, int cols, int64_t row_stride, int64_t grad_row_stride,
                    float smoothing, int total_classes, int dtype, bp_stream_t stream);

/*
 * bp_bias_gelu_fwd / bp_bias_gelu_bwd / bp_column_sum -- the elementwise halves of the reference's fused dense
 * layers (flash_attn/ops/fused_dense.py:175-330 `FusedDenseGeluDenseFunc`, :27-108 `FusedDenseFunc`), i.e. what the
 * cuBLASLt epilogues of csrc/fused_dense_lib do around the GEMMs (fused_dense.cpp:195-197: `linear_gelu_forward`,
 * `bias_gelu_linear_dgrad_bgrad`, `linear_bias_wgrad`).  The GEMMs stay on the BLAS library.
 *   forward   y = gelu_tanh(x + bias);  pre_out (optional) = x + bias rounded to 16 bit -- y is then the GELU of that
 *             rounded value, so bp_bias_gelu_bwd differentiates exactly what the forward evaluated
 *   backward  dpre = grad * gelu_tanh'(pre);   dbias[c] = sum_r dpre[r,c]   (one pass; deterministic two-stage sum)
 *   column sum  dbias[c] = sum_r grad[r,c]     (bias gradient of a dense layer without activation)
 * gelu_tanh(x) = 0.5 x (1 + tanh(sqrt(2/pi) (x + 0.044715 x^3)))  (GPT-2 `gelu_new`, F.gelu(approximate='tanh')).
 *   x, grad, pre, pre_out, y, dpre   (rows, cols) 16-bit contiguous, cols % 8 == 0, 16-byte aligned;
 *                                    y may alias x, dpre may alias grad
 *   bias    (cols) 16-bit or NULL (pre_out then must be NULL too: the pre-activation is x itself)
 *   dbias   (cols) fp32 (dbias_is_f32) or 16-bit; NULL in bp_bias_gelu_bwd = no bias gradient wanted
 *   ws      fp32 workspace of bp_bias_grad_ws_floats(rows, cols) elements (contents undefined on entry)
 */
int64_t bp_bias_grad_ws_floats(int64_t rows, int cols);
int bp_bias_gelu_fwd(const void *x, const void *bias, void *pre_out, void *y, int64_t rows, int cols, int dtype,
                     bp_stream_t stream);
int bp_bias_gelu_bwd(const void *grad, const void *pre, void *dpre, void *dbias, float *ws, int64_t rows, int cols,
                     int dtype, int dbias_is_f32, bp_stream_t stream);
int bp_column_sum(const void *grad, void *dbias, float *ws, int64_t rows, int cols, int dtype, int dbias_is_f32,
                  bp_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* BP_HIP_H */

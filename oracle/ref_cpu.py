"""CPU oracle for the Backpack forward hot path.  TEST INFRASTRUCTURE ONLY.

This file restates, in plain PyTorch eager ops, what the reference
(john-hewitt/backpacks-flash-attn) computes on its non-fused path.  It is the
checker for the HIP kernels; it is NOT part of the product.  Only `tests/`,
`__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` may import it.

Parity status: PINNED.  `tests/golden/make_golden.py` imports the real reference
from /root/reference (in the build container), checks every function below
against it, and writes `tests/golden/*.npz`; `tests/test_oracle_golden.py`
re-checks this file against those vectors on every run.

Reference anchors (paths relative to the reference repo):
  * eager self attention          flash_attn/modules/mha.py:195-224
  * fp32 test oracle (-inf mask)  tests/test_flash_attn.py:129-178
  * logsumexp definition          csrc/flash_attn/src/fmha_fprop_kernel_1xN.h:592-596
  * sense weights alpha           training/src/models/backpack.py:107-122
  * sense combination             training/src/models/backpack.py:313
  * content module                training/src/models/backpack.py:251-276
  * GPT-2 trunk                   flash_attn/models/gpt.py:224-246, modules/block.py:70-106
  * per-layer softmax scale       flash_attn/models/gpt.py:47-50
  * LM head                       training/src/models/backpack.py:342-351
  * fused add + LayerNorm         flash_attn/ops/layer_norm.py:207-217, csrc/layer_norm/ln_fwd_kernels.cuh:116-175
"""
import math

import torch
import torch.nn.functional as F

MASK_VALUE = -10000.0  # additive mask constant of the eager reference (mha.py:212,219)


# --------------------------------------------------------------------------------------
# attention
# --------------------------------------------------------------------------------------
def _additive_causal(sq, sk, dtype, device):
    full = torch.full((sq, sk), MASK_VALUE, device=device)
    return torch.triu(full, 1).to(dtype)


def self_attention_eager(qkv, causal=False, softmax_scale=None, key_padding_mask=None):
    """Eager twin of the flash kernel: mha.py:195-224.

    qkv (B,S,3,H,D) -> (B,S,H,D).  K is scaled *before* the matmul in the storage
    dtype, masks are additive -10000, softmax runs in v.dtype.
    """
    b, s = qkv.shape[0], qkv.shape[1]
    q, k, v = qkv.unbind(dim=2)
    scale = softmax_scale or 1.0 / math.sqrt(q.shape[-1])
    scores = torch.einsum('bthd,bshd->bhts', q, k * scale)
    if key_padding_mask is not None:
        pad = torch.full((b, s), MASK_VALUE, dtype=scores.dtype, device=scores.device)
        pad.masked_fill_(key_padding_mask, 0.0)
        scores = scores + pad[:, None, None, :]
    if causal:
        scores = scores + _additive_causal(s, s, scores.dtype, scores.device)
    attn = torch.softmax(scores, dim=-1, dtype=v.dtype)
    return torch.einsum('bhts,bshd->bthd', attn, v)


def attention_fp32(q, k, v=None, causal=False, softmax_scale=None,
                   query_padding_mask=None, key_padding_mask=None,
                   upcast=True, reorder_ops=False, dropout_p=0.0, dropout_mask=None):
    """The reference's own test oracle: tests/test_flash_attn.py:129-178, extended with an
    explicit softmax_scale (the reference hard-codes 1/sqrt(d)) and with the row
    log-sum-exp the kernel returns (fmha_fprop_kernel_1xN.h:592-596).

    q (B,Sq,H,D), k/v (B,Sk,H,D).  Masks are -inf; causal is top-left aligned
    (col <= row, csrc/flash_attn/src/fmha/mask.h:57-70).
    dropout_mask (B,H,Sq,Sk) bool, True = keep, with dropout_p: the reference oracle's dropout arguments
    (tests/test_flash_attn.py:131,166-171): out = (attn masked by dropout_mask) @ (v / (1 - dropout_p)); the
    returned `attn` stays the UNdropped softmax, as upstream (:172-178).
    Returns (out (B,Sq,H,D) in q's dtype or None, attn (B,H,Sq,Sk), lse (B,H,Sq) fp32).
    """
    dtype_og = q.dtype
    if upcast:
        q, k = q.float(), k.float()
        v = v.float() if v is not None else None
    sq, sk = q.shape[1], k.shape[1]
    scale = softmax_scale if softmax_scale is not None else 1.0 / math.sqrt(q.shape[-1])
    if not reorder_ops:
        scores = torch.einsum('bthd,bshd->bhts', q * scale, k)
    else:
        scores = torch.einsum('bthd,bshd->bhts', q, k * scale)
    if key_padding_mask is not None:
        scores = scores.masked_fill(~key_padding_mask[:, None, None, :], float('-inf'))
    if causal:
        upper = torch.triu(torch.ones(sq, sk, dtype=torch.bool, device=q.device), 1)
        scores = scores.masked_fill(upper, float('-inf'))
    lse = torch.logsumexp(scores.float(), dim=-1)
    attn = torch.softmax(scores, dim=-1)
    # rows with no valid key: softmax gives NaN, the kernel gives zeros / lse=-inf
    attn = torch.nan_to_num(attn, nan=0.0)
    out = None
    if dropout_mask is not None:
        attn_used = attn.masked_fill(~dropout_mask, 0.0)
    else:
        attn_used = attn
    dropout_scaling = 1.0 / (1.0 - dropout_p)
    if v is not None:
        out = torch.einsum('bhts,bshd->bthd', attn_used, v * dropout_scaling)
        if query_padding_mask is not None:
            out = out.masked_fill(~query_padding_mask[:, :, None, None], 0.0)
        out = out.to(dtype_og)
    if query_padding_mask is not None:
        attn = attn.masked_fill(~query_padding_mask[:, None, :, None], 0.0)
    return out, attn.to(dtype_og), lse


def varlen_attention_fp32(q, k, v, cu_q, cu_k, causal=False, softmax_scale=None):
    """Unpadded (total, H, D) form of `attention_fp32`, one sequence at a time, as the
    kernel indexes it (csrc/flash_attn/src/fmha_kernel.h:44-75: rows cu[b]..cu[b+1]).
    Returns (out (total_q,H,D) in q's dtype, lse list of (H, sq_b) fp32)."""
    outs, lses = [], []
    nb = len(cu_q) - 1
    for b in range(nb):
        q0, q1 = int(cu_q[b]), int(cu_q[b + 1])
        k0, k1 = int(cu_k[b]), int(cu_k[b + 1])
        if q1 == q0:
            lses.append(torch.zeros(q.shape[1], 0))
            continue
        if k1 == k0:
            outs.append(torch.zeros_like(q[q0:q1]))
            lses.append(torch.full((q.shape[1], q1 - q0), float('-inf')))
            continue
        o, _, lse = attention_fp32(q[None, q0:q1], k[None, k0:k1], v[None, k0:k1], causal=causal,
                                   softmax_scale=softmax_scale)
        outs.append(o[0])
        lses.append(lse[0])
    return torch.cat(outs, dim=0), lses


# --------------------------------------------------------------------------------------
# Backpack sense weights and combination
# --------------------------------------------------------------------------------------
def sense_alpha_from_qk(qk, softmax_scale=None):
    """alpha from projected queries/keys: backpack.py:112-122.

    qk (B,S,2,k,d_k) -> alpha (B,k,S,S) in qk's dtype.  Additive -10000 causal mask,
    K scaled before the matmul, softmax in q's dtype -- exactly the eager op order.
    """
    s = qk.shape[1]
    q, k = qk.unbind(dim=2)
    scale = softmax_scale or 1.0 / math.sqrt(q.shape[-1])
    scores = torch.einsum('bthd,bshd->bhts', q, k * scale)
    scores = scores + _additive_causal(s, s, scores.dtype, scores.device)
    return torch.softmax(scores, dim=-1, dtype=q.dtype)


def context_self_attn(encoded, w, bias, num_content_vectors, softmax_scale=None):
    """ContextSelfAttn.forward: backpack.py:107-122.  encoded (B,S,d); w (2d,d); bias (2d)."""
    b, s, d = encoded.shape
    qk = F.linear(encoded, w, bias).reshape(b, s, 2, num_content_vectors, d // num_content_vectors)
    return sense_alpha_from_qk(qk, softmax_scale)


def sense_mix(alpha, content):
    """backpack.py:313: sum over senses of alpha_l @ C_l.
    alpha (B,k,S,S), content (B,k,S,dout) -> (B,S,dout)."""
    return torch.sum(alpha @ content, dim=1)


def sense_mix_from_qk_fp32(qk, content, softmax_scale=None):
    """fp32 composition of the two functions above (what the fused kernel computes):
    qk (B,S,2,k,d_k), content (B,k,S,dout) or its (B,S,k,dout) storage -> (B,S,dout) fp32."""
    alpha = sense_alpha_from_qk(qk.float(), softmax_scale)
    return sense_mix(alpha, content.float())


def add_layer_norm_fp32(x0, x1, gamma, beta, eps, residual_dtype=None):
    """Eval-mode dropout_add_layer_norm (flash_attn/ops/layer_norm.py:207-217; kernel
    csrc/layer_norm/ln_fwd_kernels.cuh:116-175): x = x0 + x1 in fp32, z = LN(x) in fp32 from the
    UNROUNDED sum, z stored in x0's dtype, x in the residual dtype (x1's, else `residual_dtype`).
    Statistics as the kernel: mean, then centred sum of squares / n, rsqrt(var + eps)."""
    x = x0.float() + (x1.float() if x1 is not None else 0.0)
    mu = x.mean(-1, keepdim=True)
    var = ((x - mu) ** 2).mean(-1, keepdim=True)
    z = (x - mu) * torch.rsqrt(var + eps) * gamma.float() + beta.float()
    rdt = x1.dtype if x1 is not None else (residual_dtype or x0.dtype)
    return z.to(x0.dtype), x.to(rdt)


# --------------------------------------------------------------------------------------
# whole-model eager forward, functional over a reference-named state dict
# --------------------------------------------------------------------------------------
def _ln(x, sd, prefix, eps):
    return F.layer_norm(x, (x.shape[-1],), sd[prefix + '.weight'], sd[prefix + '.bias'], eps)


def _mlp(x, sd, prefix):
    # flash_attn/modules/mlp.py:26-30 with activation = tanh-approximate GELU
    # ('gelu_new' -> approximate='tanh', gpt.py:87-89 / backpack.py:70-72)
    h = F.linear(x, sd[prefix + '.fc1.weight'], sd[prefix + '.fc1.bias'])
    h = F.gelu(h, approximate='tanh')
    return F.linear(h, sd[prefix + '.fc2.weight'], sd[prefix + '.fc2.bias'])


def _prenorm_block(hidden, residual, sd, prefix, eps, mixer):
    # flash_attn/modules/block.py:70-106 (prenorm, no fused LN, dropout off)
    mixed = mixer(hidden)
    residual = mixed + residual
    hidden = _ln(residual.to(sd[prefix + '.norm1.weight'].dtype), sd, prefix + '.norm1', eps)
    mlp_out = _mlp(hidden, sd, prefix + '.mlp')
    residual = mlp_out + residual
    hidden = _ln(residual.to(sd[prefix + '.norm2.weight'].dtype), sd, prefix + '.norm2', eps)
    return hidden, residual


def gpt2_trunk(sd, cfg, input_ids, prefix='transformer.gpt2_model', return_attn_inputs=False):
    """GPTModel.forward: gpt.py:224-246 with eager attention (use_flash_attn False)."""
    eps = cfg['layer_norm_epsilon']
    n_head = cfg['n_head']
    d = cfg['n_embd']
    dh = d // n_head
    b, s = input_ids.shape
    emb = F.embedding(input_ids, sd[prefix + '.embeddings.word_embeddings.weight'])
    pos = torch.arange(s, dtype=torch.long, device=input_ids.device)
    emb = emb + F.embedding(pos, sd[prefix + '.embeddings.position_embeddings.weight'])
    residual = emb.float()
    hidden = _ln(residual.to(sd[prefix + '.ln_0.weight'].dtype), sd, prefix + '.ln_0', eps)
    captured = []
    for i in range(cfg['n_layer']):
        lp = f'{prefix}.layers.{i}'
        scale = dh ** -0.5
        if cfg.get('scale_attn_by_inverse_layer_idx', False):
            scale /= float(i + 1)

        def mixer(x, lp=lp, scale=scale):
            qkv = F.linear(x, sd[lp + '.mixer.Wqkv.weight'], sd[lp + '.mixer.Wqkv.bias'])
            qkv = qkv.reshape(b, s, 3, n_head, dh)
            if return_attn_inputs:
                captured.append((qkv, scale))
            ctx = self_attention_eager(qkv, causal=True, softmax_scale=scale)
            return F.linear(ctx.reshape(b, s, d), sd[lp + '.mixer.out_proj.weight'],
                            sd[lp + '.mixer.out_proj.bias'])

        hidden, residual = _prenorm_block(hidden, residual, sd, lp, eps, mixer)
    if return_attn_inputs:
        return hidden, captured
    return hidden


def content_model(sd, cfg, input_ids, prefix='transformer.content_model',
                  emb_key='transformer.gpt2_model.embeddings.word_embeddings.weight'):
    """BackpackContentModule.forward: backpack.py:251-276 -> C (B,k,S,d) (a view)."""
    eps = cfg['layer_norm_epsilon']
    k = cfg['num_content_vectors']
    d = cfg['n_embd']
    hidden = F.embedding(input_ids, sd[emb_key])  # word embeddings only, no positions (:258)
    residual = hidden.float()
    hidden = _ln(residual.to(sd[prefix + '.ln_0.weight'].dtype), sd, prefix + '.ln_0', eps)
    hidden, residual = _prenorm_block(hidden, residual, sd, prefix + '.layers.0', eps,
                                      mixer=lambda x: x)  # Identity mixer (:125-143)
    out = _mlp(hidden, sd, prefix + '.final_mlp')  # (B,S,k*d)
    b, s, _ = out.shape
    return out.reshape(b, s, k, d).transpose(1, 2)


def backpack_forward(sd, cfg, input_ids, return_stages=False):
    """BackpackLMHeadModel.forward: backpack.py:297-314,342-351.  `sd` uses the reference's
    state-dict key names; `cfg` is a plain dict (n_embd, n_head, n_layer,
    num_content_vectors, layer_norm_epsilon, scale_attn_by_inverse_layer_idx)."""
    h = gpt2_trunk(sd, cfg, input_ids)
    alpha = context_self_attn(h, sd['transformer.contextualization_attn.Wqkv.weight'],
                              sd['transformer.contextualization_attn.Wqkv.bias'],
                              cfg['num_content_vectors'])
    content = content_model(sd, cfg, input_ids)
    hidden = sense_mix(alpha, content)
    logits = F.linear(hidden, sd['lm_head.weight'])
    if return_stages:
        return dict(trunk=h, alpha=alpha, content=content, hidden=hidden, logits=logits)
    return logits


# --------------------------------------------------------------------------------------
# reference-style initialisation for synthetic benchmarks (backpack.py:186-204,246,333)
# --------------------------------------------------------------------------------------
def init_state_dict(cfg, seed=0, dtype=torch.float32):
    """Random weights with the reference's init scheme: N(0, 0.02) for Linear/Embedding
    weights, zero biases, LN (1, 0); out_proj / fc2 weights N(0, 0.02/sqrt(2*n_layer))
    everywhere, because BackpackLMHeadModel re-applies _init_weights with the trunk's
    n_layer over every sub-module last (backpack.py:332-334)."""
    g = torch.Generator().manual_seed(seed)
    d, L, k = cfg['n_embd'], cfg['n_layer'], cfg['num_content_vectors']
    V, P = cfg['vocab_size'], cfg['n_positions']
    inner = cfg.get('n_inner') or 4 * d
    final_inner = d if cfg.get('shrink_final_inner') else inner
    sd = {}

    def normal(shape, std=0.02):
        return (torch.randn(shape, generator=g) * std).to(dtype)

    def ln(prefix):
        sd[prefix + '.weight'] = torch.ones(d, dtype=dtype)
        sd[prefix + '.bias'] = torch.zeros(d, dtype=dtype)

    def mlp(prefix, hidden, out, resid_std):
        sd[prefix + '.fc1.weight'] = normal((hidden, d))
        sd[prefix + '.fc1.bias'] = torch.zeros(hidden, dtype=dtype)
        sd[prefix + '.fc2.weight'] = normal((out, hidden), resid_std)
        sd[prefix + '.fc2.bias'] = torch.zeros(out, dtype=dtype)

    t = 'transformer.gpt2_model'
    sd[t + '.embeddings.word_embeddings.weight'] = normal((V, d))
    sd[t + '.embeddings.position_embeddings.weight'] = normal((P, d))
    ln(t + '.ln_0')
    std_r = 0.02 / math.sqrt(2 * L)
    for i in range(L):
        lp = f'{t}.layers.{i}'
        sd[lp + '.mixer.Wqkv.weight'] = normal((3 * d, d))
        sd[lp + '.mixer.Wqkv.bias'] = torch.zeros(3 * d, dtype=dtype)
        sd[lp + '.mixer.out_proj.weight'] = normal((d, d), std_r)
        sd[lp + '.mixer.out_proj.bias'] = torch.zeros(d, dtype=dtype)
        ln(lp + '.norm1')
        mlp(lp + '.mlp', inner, d, std_r)
        ln(lp + '.norm2')
    c = 'transformer.content_model'
    ln(c + '.ln_0')
    ln(c + '.layers.0.norm1')
    mlp(c + '.layers.0.mlp', final_inner, d, std_r)
    ln(c + '.layers.0.norm2')
    mlp(c + '.final_mlp', final_inner, k * d, std_r)
    sd['transformer.contextualization_attn.Wqkv.weight'] = normal((2 * d, d))
    sd['transformer.contextualization_attn.Wqkv.bias'] = torch.zeros(2 * d, dtype=dtype)
    sd['lm_head.weight'] = sd[t + '.embeddings.word_embeddings.weight']
    return sd


CONFIGS = {
    # training/configs/model/gpt2model/gpt2-{micro,mini,small}.yaml + backpack.yaml
    'micro': dict(n_embd=384, n_head=6, n_layer=6, num_content_vectors=16),
    'mini': dict(n_embd=640, n_head=8, n_layer=8, num_content_vectors=16),
    'mini-k64': dict(n_embd=640, n_head=8, n_layer=8, num_content_vectors=64,
                     shrink_final_inner=True),
    # the few-sense ends of the same ablation (training/configs/experiment/owt/backpack-mini-flash-vecs-{4,1}.yaml):
    # sense widths d_k = 160 and 640
    'mini-k4': dict(n_embd=640, n_head=8, n_layer=8, num_content_vectors=4, shrink_final_inner=True),
    'mini-k1': dict(n_embd=640, n_head=8, n_layer=8, num_content_vectors=1, shrink_final_inner=True),
    'small': dict(n_embd=768, n_head=12, n_layer=12, num_content_vectors=16),
}


def make_config(name, n_positions=1024, vocab_size=50264):
    cfg = dict(CONFIGS[name])
    cfg.update(n_positions=n_positions, vocab_size=vocab_size, layer_norm_epsilon=1e-5,
               scale_attn_by_inverse_layer_idx=True)
    return cfg


# ---------------------------------------------------------------------------------------------
# Control experiments built on the sense contraction:
# training/src/models/intervened_models.py.  Restated with the reference's own tensor algebra
# (full-vocabulary gather included) so the product's cheaper formulations are checked against it.
# ---------------------------------------------------------------------------------------------
def content_soft_mask(content_weights, input_ids, scores):
    """intervened_models.py:9-20.  content_weights (V, k), input_ids (B, S), scores (B, S, k)
    -> per-token, per-sense weights (B, S, k): w * score + (1 - score)."""
    picked = content_weights[input_ids]                                  # gather over the vocabulary axis
    return picked * scores + torch.ones_like(picked) * (1 - scores)


def mask_annealing_scores(lm_head_weight, input_ids, content, annealing_scale=0.1, upweight_nearby=True):
    """intervened_models.py:29-53.  content (B, k, S, d) -> scores (B, k, S):
    sigmoid(-scale * sum_j relu(content[b,l,i] . E[ids[b,j]]) + 6), optionally x (1 + i/100)."""
    b, s = input_ids.shape
    vocab_logits = torch.relu(content @ lm_head_weight.t())              # (B, k, S, V)
    index = input_ids.reshape(b, 1, 1, s).expand(-1, content.shape[1], s, -1)
    sims = torch.gather(vocab_logits, dim=3, index=index)               # (B, k, S, S)
    sims = torch.where(sims > 0.0, sims, torch.zeros_like(sims)).sum(dim=3)
    scores = torch.sigmoid(-annealing_scale * sims + 6)
    if upweight_nearby:
        scores = scores * (1 + torch.arange(s) / 100).reshape(1, 1, s)
    return scores


def _intervention_stages(sd, cfg, input_ids):
    st = backpack_forward(sd, cfg, input_ids, return_stages=True)
    return st['alpha'], st['content'], sd['lm_head.weight']


def weighted_backpack_logits(sd, cfg, input_ids, content_weights, annealing_scale=0.1, anneal=True,
                             upweight_nearby=True):
    """WeightedBackpackLMHeadModel.forward, intervened_models.py:70-105."""
    alpha, content, w_lm = _intervention_stages(sd, cfg, input_ids)
    if anneal:
        scores = mask_annealing_scores(w_lm, input_ids, content, annealing_scale, upweight_nearby).transpose(1, 2)
    else:
        scores = torch.ones(content.shape[0], content.shape[2], content.shape[1])
    weights = content_soft_mask(content_weights, input_ids, scores)       # (B, S, k)
    content = content * weights.transpose(1, 2).unsqueeze(3)
    hidden = torch.sum(alpha @ content, dim=1)
    return hidden @ w_lm.t()


def negative_weighted_backpack_logits(sd, cfg, input_ids, content_weights, annealing_scale=0.1, anneal=True,
                                      upweight_nearby=True):
    """NegativeWeightedBackpackLMHeadModel.forward, intervened_models.py:120-165: per (sense, position) the
    2 % most negative re-weighted vocabulary logits replace the plain ones; the contraction runs on
    vocabulary-sized content."""
    alpha, content, w_lm = _intervention_stages(sd, cfg, input_ids)
    if anneal:
        scores = mask_annealing_scores(w_lm, input_ids, content, annealing_scale, upweight_nearby).transpose(1, 2)
    else:
        scores = torch.ones(content.shape[0], content.shape[2], content.shape[1])
    weights = content_soft_mask(content_weights, input_ids, scores)
    weighted = content * weights.transpose(1, 2).unsqueeze(3)
    logits_c = content @ w_lm.t()
    logits_w = weighted @ w_lm.t()
    q = torch.quantile(logits_w.float(), q=0.02, keepdim=True, dim=-1)
    logits_c = torch.where(logits_w < q, logits_w, logits_c)
    return torch.sum(alpha @ logits_c, dim=1)


def replaced_word_logits(sd, cfg, input_ids, sense_dict):
    """ReplacedWordLMHeadModel.forward, intervened_models.py:175-199: every position whose token is a
    key of `sense_dict` gets that entry's (k, d) sense vectors instead of its own."""
    alpha, content, w_lm = _intervention_stages(sd, cfg, input_ids)
    content = content.clone()
    for bi in range(content.shape[0]):
        for si in range(content.shape[2]):
            word = int(input_ids[bi, si])
            if word in sense_dict:
                content[bi, :, si, :] = sense_dict[word]
    return torch.sum(alpha @ content, dim=1) @ w_lm.t()


# ---------------------------------------------------------------------------------------------
# Fused softmax cross-entropy.  The native kernel of the reference
# (xentropy_cuda_lib, csrc/xentropy) is not buildable here; its contract is spelled out at its call site,
# flash_attn/losses/cross_entropy.py:57-63: with smoothing s the per-row loss is
#   (1 - s) * (lse - x[y]) + s * (lse - sum_j x_j / total_classes),
# rows whose label is the ignore index give 0 (:39).  The reference's own test pins it to
# torch.nn.CrossEntropyLoss(label_smoothing=s) on fp32 logits (tests/losses/test_cross_entropy.py:31-41);
# tests/test_oracle_golden.py checks this restatement against that same oracle.
# ---------------------------------------------------------------------------------------------
def softmax_cross_entropy(logits, labels, smoothing=0.0, ignored_index=-100, total_classes=None):
    x = logits.float()
    total = total_classes or x.shape[1]
    lse = torch.logsumexp(x, dim=1)
    inside = (labels >= 0) & (labels < x.shape[1])
    picked = x.gather(1, labels.clamp(0, x.shape[1] - 1).unsqueeze(1)).squeeze(1)
    losses = smoothing * (lse - x.sum(dim=1) / total)
    losses = losses + torch.where(inside, (1 - smoothing) * (lse - picked), torch.zeros_like(lse))
    return losses.masked_fill(labels == ignored_index, 0), lse


def softmax_cross_entropy_grad(grad_losses, logits, labels, smoothing=0.0, ignored_index=-100,
                               total_classes=None):
    """d loss_i / d x_ij = g_i (softmax_ij - (1 - s)[j == y_i] - s / total_classes)  (cross_entropy.py:99-106)."""
    x = logits.float()
    total = total_classes or x.shape[1]
    g = grad_losses.float().masked_fill(labels == ignored_index, 0)
    d = torch.softmax(x, dim=1) - smoothing / total
    inside = (labels >= 0) & (labels < x.shape[1])
    rows = torch.nonzero(inside).squeeze(1)
    d[rows, labels[rows]] -= (1 - smoothing)
    return d * g.unsqueeze(1)

"""Python API of the fused attention -- mirror of the reference's
flash_attn/flash_attn_interface.py (public functions :242-380), with `flash_attn_cuda.fwd`
(:23-26) replaced by bp_hip.flash_fwd (C ABI bp_flash_fwd, include/bp_hip.h) and
`flash_attn_cuda.bwd` (:38-43) by bp_hip.flash_bwd (bp_flash_bwd).

The HIP backward covers the same head dims as the forward fast path (% 8 == 0, <= 128); for anything
else the autograd Functions RAISE in backward unless the caller opted in to differentiating an eager recomputation
(`with bp_hip.allow_eager_fallback():`, dropout_p = 0 only).
Attention dropout runs inside the kernels (bp_flash_fwd_dropout / bp_flash_bwd_dropout): the forward draws a
two-word generator state on the device from torch's CUDA generator and saves it for backward, where the
reference saves and restores the whole CUDA RNG state around its kernel (:56-57,:74-81).
"""
import torch

import bp_hip


def _get_block_size(device, head_dim, is_dropout):
    # kept for API compatibility (flash_attn_interface.py:8-10); the HIP kernel tiles 64 keys
    assert head_dim % 8 == 0 and head_dim <= 128
    return 64


def _flash_attn_forward(q, k, v, out, cu_seqlens_q, cu_seqlens_k, max_seqlen_q, max_seqlen_k,
                        dropout_p, softmax_scale, causal, return_softmax, num_splits=0,
                        generator=None, rng_state=None):
    """Same contract as the reference's helper (:13-28): writes `out` in place and returns
    (out, softmax_lse, S_dmask).  `S_dmask` here is the NORMALISED probability tensor
    (b, h, max_seqlen_q, max_seqlen_k), only for fixed-length batches (testing aid, as upstream); with
    dropout, dropped entries carry a set sign bit (the reference encodes its mask in the sign of S_dmask too,
    tests/test_flash_attn.py:181-236).  `rng_state`: bp_hip.new_rng_state(); required when dropout_p > 0 and
    the mask must be reproducible (backward).  `generator` (the reference's argument, :16,26): the state words are
    drawn from it instead of torch's default CUDA generator."""
    if dropout_p > 0.0 and rng_state is None:
        rng_state = bp_hip.new_rng_state(q.device, generator)
    softmax_lse = bp_hip.flash_fwd(q, k, v, out, cu_seqlens_q, cu_seqlens_k, max_seqlen_q,
                                   max_seqlen_k, softmax_scale, causal, dropout_p, rng_state)
    S_dmask = None
    if return_softmax:
        batch = cu_seqlens_q.numel() - 1 if cu_seqlens_q is not None else q.shape[0] // max_seqlen_q
        if q.shape[0] == batch * max_seqlen_q and k.shape[0] == batch * max_seqlen_k:
            qb = q.unflatten(0, (batch, max_seqlen_q))
            kb = k.unflatten(0, (batch, max_seqlen_k))
            S_dmask = bp_hip.attn_probs(qb, kb, softmax_lse, softmax_scale, causal, dropout_p, rng_state)
    return out, softmax_lse, S_dmask


def _flash_attn_backward(dout, q, k, v, out, softmax_lse, dq, dk, dv, cu_seqlens_q, cu_seqlens_k,
                         max_seqlen_q, max_seqlen_k, dropout_p, softmax_scale, causal, num_splits=0,
                         generator=None, rng_state=None):
    """Same contract as the reference's helper (:31-47): fills dq, dk, dv in place.  With dropout, `rng_state`
    is the forward's (the reference restores the saved CUDA RNG state instead, :74-81); a `generator` is accepted
    for signature compatibility only -- the mask is a function of `rng_state`, nothing is drawn here."""
    if dropout_p > 0.0 and rng_state is None:
        raise RuntimeError('flash_attn (gfx950 build): backward with dropout needs the rng_state the forward used')
    bp_hip.flash_bwd(dout, q, k, v, out, softmax_lse, dq, dk, dv, cu_seqlens_q, cu_seqlens_k,
                     max_seqlen_q, max_seqlen_k, softmax_scale, causal, dropout_p, rng_state)
    return dq, dk, dv


def _eager_varlen(q, k, v, cu_q, cu_k, softmax_scale, causal, max_q=None, max_k=None):
    """Differentiable recomputation, used by backward() for head dims the HIP backward lacks."""
    outs = []
    if cu_q is None:   # fixed-length batch
        cu_q = torch.arange(0, q.shape[0] + 1, max_q)
        cu_k = torch.arange(0, k.shape[0] + 1, max_k)
    cu_q, cu_k = cu_q.tolist(), cu_k.tolist()
    for b in range(len(cu_q) - 1):
        qb, kb, vb = q[cu_q[b]:cu_q[b + 1]], k[cu_k[b]:cu_k[b + 1]], v[cu_k[b]:cu_k[b + 1]]
        s = torch.einsum('thd,shd->hts', qb.float(), kb.float()) * softmax_scale
        if causal:
            mask = torch.ones(s.shape[-2:], dtype=torch.bool, device=s.device).triu(1)
            s = s.masked_fill(mask, float('-inf'))
        p = torch.softmax(s, dim=-1) if kb.shape[0] > 0 else s
        outs.append(torch.einsum('hts,shd->thd', p, vb.float()).to(q.dtype))
    return torch.cat(outs, dim=0)


class _FlashAttnFuncBase(torch.autograd.Function):
    """forward = HIP kernel; backward = HIP kernels (head dim % 8 == 0) or autograd through
    `_eager_varlen` (see module docstring)."""

    @staticmethod
    def _fwd(ctx, q, k, v, cu_q, cu_k, max_q, max_k, dropout_p, softmax_scale, causal, return_softmax):
        if softmax_scale is None:
            softmax_scale = q.shape[-1] ** (-0.5)
        rng_state = bp_hip.new_rng_state(q.device) if dropout_p > 0.0 else None
        out, lse, S = _flash_attn_forward(q, k, v, torch.empty_like(q), cu_q, cu_k, max_q, max_k,
                                          dropout_p, softmax_scale, causal, return_softmax, rng_state=rng_state)
        ctx.dropout_p, ctx.rng_state = dropout_p, rng_state
        ctx.softmax_scale, ctx.causal = softmax_scale, causal
        ctx.max_q, ctx.max_k = max_q, max_k
        ctx.fallback_recorded = bp_hip.eager_fallback_allowed()
        return out, lse, S

    @staticmethod
    def _bwd(ctx, dout, q, k, v, cu_q, cu_k, out=None, lse=None, grads=None):
        """grads: preallocated (dq, dk, dv) views (the packed Functions hand in slices of one buffer,
        as the reference does at :77-84)."""
        if out is not None and bp_hip.flash_bwd_supported(q):
            dq, dk, dv = grads if grads is not None else (torch.empty_like(q), torch.empty_like(k),
                                                          torch.empty_like(v))
            return _flash_attn_backward(dout, q, k, v, out, lse, dq, dk, dv, cu_q, cu_k, ctx.max_q,
                                        ctx.max_k, ctx.dropout_p, ctx.softmax_scale, ctx.causal,
                                        rng_state=ctx.rng_state)
        if ctx.dropout_p > 0.0:
            raise RuntimeError('flash_attn (gfx950 build): attention dropout needs head_dim % 8 == 0 and <= 128')
        if not bp_hip.eager_fallback_allowed(ctx):
            raise RuntimeError(
                'flash_attn (gfx950 build): the HIP backward takes head_dim %% 8 == 0 and <= 128 (got %d); differentiating '
                'an eager recomputation instead is opt-in: `with bp_hip.allow_eager_fallback():` around the forward or backward()'
                % q.shape[-1])
        with torch.enable_grad():
            q_, k_, v_ = (t.detach().requires_grad_() for t in (q, k, v))
            out = _eager_varlen(q_, k_, v_, cu_q, cu_k, ctx.softmax_scale, ctx.causal, ctx.max_q, ctx.max_k)
            return torch.autograd.grad(out, (q_, k_, v_), dout)


class FlashAttnQKVPackedFunc(_FlashAttnFuncBase):

    @staticmethod
    def forward(ctx, qkv, cu_seqlens, max_seqlen, dropout_p, softmax_scale, causal, return_softmax):
        out, lse, S = _FlashAttnFuncBase._fwd(ctx, qkv[:, 0], qkv[:, 1], qkv[:, 2], cu_seqlens,
                                              cu_seqlens, max_seqlen, max_seqlen, dropout_p,
                                              softmax_scale, causal, return_softmax)
        ctx.save_for_backward(qkv, cu_seqlens, out, lse)
        return out if not return_softmax else (out, lse, S)

    @staticmethod
    def backward(ctx, dout, *args):
        qkv, cu, out, lse = ctx.saved_tensors
        dqkv = torch.empty_like(qkv)
        dq, dk, dv = _FlashAttnFuncBase._bwd(ctx, dout, qkv[:, 0], qkv[:, 1], qkv[:, 2], cu, cu, out, lse,
                                             (dqkv[:, 0], dqkv[:, 1], dqkv[:, 2]))
        if dq.data_ptr() != dqkv.data_ptr():          # eager fallback returned fresh tensors
            dqkv = torch.stack([dq, dk, dv], dim=1)
        return dqkv, None, None, None, None, None, None


class FlashAttnKVPackedFunc(_FlashAttnFuncBase):

    @staticmethod
    def forward(ctx, q, kv, cu_seqlens_q, cu_seqlens_k, max_seqlen_q, max_seqlen_k, dropout_p,
                softmax_scale, causal, return_softmax):
        out, lse, S = _FlashAttnFuncBase._fwd(ctx, q, kv[:, 0], kv[:, 1], cu_seqlens_q, cu_seqlens_k,
                                              max_seqlen_q, max_seqlen_k, dropout_p, softmax_scale,
                                              causal, return_softmax)
        ctx.save_for_backward(q, kv, cu_seqlens_q, cu_seqlens_k, out, lse)
        return out if not return_softmax else (out, lse, S)

    @staticmethod
    def backward(ctx, dout, *args):
        q, kv, cu_q, cu_k, out, lse = ctx.saved_tensors
        dkv = torch.empty_like(kv)
        dq, dk, dv = _FlashAttnFuncBase._bwd(ctx, dout, q, kv[:, 0], kv[:, 1], cu_q, cu_k, out, lse,
                                             (torch.empty_like(q), dkv[:, 0], dkv[:, 1]))
        if dk.data_ptr() != dkv.data_ptr():
            dkv = torch.stack([dk, dv], dim=1)
        return dq, dkv, None, None, None, None, None, None, None, None


class FlashAttnFunc(_FlashAttnFuncBase):

    @staticmethod
    def forward(ctx, q, k, v, cu_seqlens_q, cu_seqlens_k, max_seqlen_q, max_seqlen_k, dropout_p,
                softmax_scale, causal, return_softmax):
        out, lse, S = _FlashAttnFuncBase._fwd(ctx, q, k, v, cu_seqlens_q, cu_seqlens_k, max_seqlen_q,
                                              max_seqlen_k, dropout_p, softmax_scale, causal,
                                              return_softmax)
        ctx.save_for_backward(q, k, v, cu_seqlens_q, cu_seqlens_k, out, lse)
        return out if not return_softmax else (out, lse, S)

    @staticmethod
    def backward(ctx, dout, *args):
        q, k, v, cu_q, cu_k, out, lse = ctx.saved_tensors
        dq, dk, dv = _FlashAttnFuncBase._bwd(ctx, dout, q, k, v, cu_q, cu_k, out, lse)
        return dq, dk, dv, None, None, None, None, None, None, None, None


def flash_attn_unpadded_qkvpacked_func(qkv, cu_seqlens, max_seqlen, dropout_p, softmax_scale=None,
                                       causal=False, return_attn_probs=False):
    """qkv (total, 3, nheads, headdim); cu_seqlens int32 (batch+1); returns out (total, nheads,
    headdim) [, softmax_lse (batch, nheads, seqlen), probs].  Reference: :242-267.
    Extension: cu_seqlens=None means a fixed-length batch of total // max_seqlen sequences (no index tensor)."""
    return FlashAttnQKVPackedFunc.apply(qkv, cu_seqlens, max_seqlen, dropout_p, softmax_scale,
                                        causal, return_attn_probs)


def flash_attn_unpadded_kvpacked_func(q, kv, cu_seqlens_q, cu_seqlens_k, max_seqlen_q, max_seqlen_k,
                                      dropout_p, softmax_scale=None, causal=False,
                                      return_attn_probs=False):
    """q (total_q, nheads, headdim), kv (total_k, 2, nheads, headdim).  Reference: :270-300."""
    return FlashAttnKVPackedFunc.apply(q, kv, cu_seqlens_q, cu_seqlens_k, max_seqlen_q, max_seqlen_k,
                                       dropout_p, softmax_scale, causal, return_attn_probs)


def flash_attn_unpadded_func(q, k, v, cu_seqlens_q, cu_seqlens_k, max_seqlen_q, max_seqlen_k,
                             dropout_p, softmax_scale=None, causal=False, return_attn_probs=False):
    """q (total_q, nheads, headdim), k, v (total_k, nheads, headdim).  Reference: :303-334."""
    return FlashAttnFunc.apply(q, k, v, cu_seqlens_q, cu_seqlens_k, max_seqlen_q, max_seqlen_k,
                               dropout_p, softmax_scale, causal, return_attn_probs)


def flash_attn_unpadded_qkvpacked_split_func(qkv, cu_seqlens, max_seqlen0, max_seqlen1, batch_size0,
                                             dropout_p, softmax_scale=None, causal=False,
                                             return_attn_probs=False):
    """Reference :337-371 splits the batch into two launches on two streams so short sequences
    do not pay for long ones.  The HIP kernel exits per (sequence, tile) on its own, so one launch
    covers both parts; the signature is kept."""
    return flash_attn_unpadded_qkvpacked_func(qkv, cu_seqlens, max(max_seqlen0, max_seqlen1),
                                              dropout_p, softmax_scale, causal, return_attn_probs)


def flash_attn_func(qkv, cu_seqlens, dropout_p, max_s, softmax_scale=None, causal=False,
                    return_attn_probs=False):
    """Backward-compatibility alias (reference :374-380)."""
    return flash_attn_unpadded_qkvpacked_func(qkv, cu_seqlens, max_s, dropout_p, softmax_scale,
                                              causal, return_attn_probs)

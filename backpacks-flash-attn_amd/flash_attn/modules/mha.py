"""Attention modules -- API mirror of the reference's flash_attn/modules/mha.py for the serial
(non tensor-parallel) path: FlashSelfAttention / FlashCrossAttention (HIP kernel), their eager
twins SelfAttention / CrossAttention (the reference's own non-fused path, any device), and MHA.
ParallelMHA, rotary embeddings, dwconv and the Triton variants are out of scope."""
import math
from functools import partial

import torch
import torch.nn as nn
import torch.nn.functional as F

from flash_attn.flash_attn_interface import (flash_attn_unpadded_kvpacked_func,
                                             flash_attn_unpadded_qkvpacked_func)
from flash_attn.ops.fused_dense import FusedDense

_NEG = -10000.0  # additive mask value of the eager path (reference mha.py:212,219)


class FlashSelfAttention(nn.Module):
    """Fused softmax attention on packed qkv.  Reference: mha.py:34-100."""

    def __init__(self, causal=False, softmax_scale=None, attention_dropout=0.0, triton=False):
        super().__init__()
        assert not triton, 'the gfx950 build has no Triton path'
        self.causal = causal
        self.softmax_scale = softmax_scale
        self.dropout_p = attention_dropout
        self.triton = False

    def forward(self, qkv, causal=None, cu_seqlens=None, max_seqlen=None):
        """qkv (B,S,3,H,D) -> (B,S,H,D); or (total,3,H,D) + cu_seqlens int32 + max_seqlen
        -> (total,H,D)."""
        assert qkv.dtype in (torch.float16, torch.bfloat16)
        assert qkv.is_cuda
        causal = self.causal if causal is None else causal
        p_drop = self.dropout_p if self.training else 0.0
        if cu_seqlens is not None:
            assert cu_seqlens.dtype == torch.int32
            assert max_seqlen is not None and isinstance(max_seqlen, int)
            return flash_attn_unpadded_qkvpacked_func(qkv, cu_seqlens, max_seqlen, p_drop,
                                                      softmax_scale=self.softmax_scale, causal=causal)
        batch, seqlen = qkv.shape[0], qkv.shape[1]
        # fixed-length batch: cu_seqlens=None tells the C ABI that sequence b occupies rows [b*S, (b+1)*S)
        # (bp_flash_fwd / bp_flash_bwd, include/bp_hip.h) -- the reference builds an arange here on every
        # call (mha.py:91-94); no tensor means nothing a captured HIP graph could be left pointing at
        out = flash_attn_unpadded_qkvpacked_func(
            qkv.flatten(0, 1), None, seqlen, p_drop, softmax_scale=self.softmax_scale, causal=causal)
        return out.unflatten(0, (batch, seqlen))


class FlashCrossAttention(nn.Module):
    """Fused attention with separate q and packed kv.  Reference: mha.py:103-176."""

    def __init__(self, causal=False, softmax_scale=None, attention_dropout=0.0, triton=False):
        super().__init__()
        assert not triton, 'the gfx950 build has no Triton path'
        self.causal = causal
        self.softmax_scale = softmax_scale
        self.dropout_p = attention_dropout
        self.triton = False

    def forward(self, q, kv, causal=None, cu_seqlens=None, max_seqlen=None, cu_seqlens_k=None,
                max_seqlen_k=None):
        """q (B,Sq,H,D), kv (B,Sk,2,H,D); or the unpadded forms with both cu_seqlens."""
        assert q.dtype in (torch.float16, torch.bfloat16)
        assert q.is_cuda and kv.is_cuda
        causal = self.causal if causal is None else causal
        p_drop = self.dropout_p if self.training else 0.0
        if cu_seqlens is not None:
            assert cu_seqlens.dtype == torch.int32 and isinstance(max_seqlen, int)
            assert cu_seqlens_k is not None and cu_seqlens_k.dtype == torch.int32
            assert max_seqlen_k is not None
            return flash_attn_unpadded_kvpacked_func(q, kv, cu_seqlens, cu_seqlens_k, max_seqlen,
                                                     max_seqlen_k, p_drop,
                                                     softmax_scale=self.softmax_scale, causal=causal)
        batch, sq, sk = q.shape[0], q.shape[1], kv.shape[1]
        assert kv.shape[0] == batch and kv.shape[3] == q.shape[2] and kv.shape[4] == q.shape[3]
        out = flash_attn_unpadded_kvpacked_func(
            q.flatten(0, 1), kv.flatten(0, 1), None, None, sq, sk, p_drop,
            softmax_scale=self.softmax_scale, causal=causal)
        return out.unflatten(0, (batch, sq))


def _eager_attention(q, k, v, softmax_scale, causal, key_padding_mask, p_drop):
    """The reference's non-fused arithmetic, in its op order: scale K, additive -10000 masks,
    softmax in v.dtype (mha.py:206-224)."""
    scale = softmax_scale or 1.0 / math.sqrt(q.shape[-1])
    scores = torch.einsum('bthd,bshd->bhts', q, k * scale)
    if key_padding_mask is not None:
        pad = torch.full(key_padding_mask.shape, _NEG, dtype=scores.dtype, device=scores.device)
        pad.masked_fill_(key_padding_mask, 0.0)
        scores = scores + pad[:, None, None, :]
    if causal:
        sq, sk = scores.shape[-2], scores.shape[-1]
        mask = torch.triu(torch.full((sq, sk), _NEG, device=scores.device), 1)
        scores = scores + mask.to(dtype=scores.dtype)
    attn = torch.softmax(scores, dim=-1, dtype=v.dtype)
    attn = F.dropout(attn, p_drop)
    return torch.einsum('bhts,bshd->bthd', attn, v)


class SelfAttention(nn.Module):
    """Eager twin of FlashSelfAttention (any dtype, any device).  Reference: mha.py:179-224."""

    def __init__(self, causal=False, softmax_scale=None, attention_dropout=0.0):
        super().__init__()
        self.causal = causal
        self.softmax_scale = softmax_scale
        self.dropout_p = attention_dropout

    def forward(self, qkv, causal=None, key_padding_mask=None):
        causal = self.causal if causal is None else causal
        q, k, v = qkv.unbind(dim=2)
        return _eager_attention(q, k, v, self.softmax_scale, causal, key_padding_mask,
                                self.dropout_p if self.training else 0.0)


class CrossAttention(nn.Module):
    """Eager twin of FlashCrossAttention.  Reference: mha.py:227-276."""

    def __init__(self, causal=False, softmax_scale=None, attention_dropout=0.0):
        super().__init__()
        self.causal = causal
        self.softmax_scale = softmax_scale
        self.dropout_p = attention_dropout

    def forward(self, q, kv, causal=None, key_padding_mask=None):
        causal = self.causal if causal is None else causal
        assert kv.shape[0] == q.shape[0] and kv.shape[3] == q.shape[2] and kv.shape[4] == q.shape[3]
        k, v = kv.unbind(dim=2)
        return _eager_attention(q, k, v, self.softmax_scale, causal, key_padding_mask,
                                self.dropout_p if self.training else 0.0)


class LinearResidual(nn.Linear):
    """nn.Linear that also hands back its input (reference mha.py:279-284)."""

    def forward(self, input):
        return super().forward(input), input


class MHA(nn.Module):
    """Multi-head self attention: Wqkv -> inner attention -> out_proj.  Reference: mha.py:287-467.
    State-dict keys (`Wqkv.*`, `out_proj.*`) match the reference so its checkpoints load.
    Self-attention only; `use_flash_attn` picks the HIP kernel, otherwise the eager twin runs."""

    def __init__(self, embed_dim, num_heads, cross_attn=False, bias=True, dropout=0.0,
                 softmax_scale=None, causal=False, layer_idx=None, dwconv=False, rotary_emb_dim=0,
                 rotary_emb_scale_base=0, fused_bias_fc=False, use_flash_attn=False,
                 return_residual=False, checkpointing=False, device=None, dtype=None):
        factory_kwargs = {'device': device, 'dtype': dtype}
        super().__init__()
        if cross_attn or dwconv or rotary_emb_dim > 0:
            raise NotImplementedError('gfx950 build: MHA covers the Backpack/GPT-2 self-attention '
                                      'path only (no cross_attn / dwconv / rotary)')
        self.embed_dim = embed_dim
        self.cross_attn = False
        self.causal = causal
        self.layer_idx = layer_idx
        self.dwconv = False
        self.rotary_emb_dim = 0
        self.use_flash_attn = use_flash_attn
        self.return_residual = return_residual
        self.checkpointing = checkpointing
        self.num_heads = num_heads
        assert embed_dim % num_heads == 0, 'embed_dim must be divisible by num_heads'
        self.head_dim = embed_dim // num_heads
        # fused_bias_fc selects FusedDense, as upstream (mha.py:330-337): same parameters, bias gradient by bp_column_sum
        linear_cls = FusedDense if fused_bias_fc else nn.Linear
        if return_residual:
            qkv_cls = partial(FusedDense, return_residual=True) if fused_bias_fc else LinearResidual
        else:
            qkv_cls = linear_cls
        self.Wqkv = qkv_cls(embed_dim, 3 * embed_dim, bias=bias, **factory_kwargs)
        attn_cls = FlashSelfAttention if use_flash_attn else SelfAttention
        cross_cls = FlashCrossAttention if use_flash_attn else CrossAttention
        self.inner_attn = attn_cls(causal=causal, softmax_scale=softmax_scale, attention_dropout=dropout)
        self.inner_cross_attn = cross_cls(causal=causal, softmax_scale=softmax_scale,
                                          attention_dropout=dropout)
        self.out_proj = linear_cls(embed_dim, embed_dim, **factory_kwargs)

    def forward(self, x, x_kv=None, key_padding_mask=None, cu_seqlens=None, max_seqlen=None,
                inference_params=None, **kwargs):
        """x (batch, seqlen, hidden) or, with cu_seqlens/max_seqlen, (total, hidden)."""
        if inference_params is not None:
            raise NotImplementedError('gfx950 build: KV-cache decoding is out of scope; the '
                                      "reference's own generation re-runs the full forward")
        if cu_seqlens is not None:
            assert max_seqlen is not None and key_padding_mask is None and self.use_flash_attn
        if key_padding_mask is not None:
            assert cu_seqlens is None and max_seqlen is None and not self.use_flash_attn
        if self.use_flash_attn:
            kwargs = {'cu_seqlens': cu_seqlens, 'max_seqlen': max_seqlen, **kwargs}
        else:
            kwargs = {'key_padding_mask': key_padding_mask, **kwargs}
        if self.return_residual:
            qkv, x = self.Wqkv(x)
        else:
            qkv = self.Wqkv(x)
        qkv = qkv.unflatten(-1, (3, self.num_heads, self.head_dim))
        if self.checkpointing:
            context = torch.utils.checkpoint.checkpoint(self.inner_attn, qkv, **kwargs)
        else:
            context = self.inner_attn(qkv, **kwargs)
        out = self.out_proj(context.flatten(-2))
        return out if not self.return_residual else (out, x)

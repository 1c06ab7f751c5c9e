"""`FlashAttention` / `FlashMHA` modules -- mirror of the reference's flash_attn/flash_attention.py."""
import torch
import torch.nn as nn

from flash_attn.bert_padding import pad_input, unpad_input
from flash_attn.flash_attn_interface import flash_attn_unpadded_qkvpacked_func


class FlashAttention(nn.Module):
    """softmax(scale * Q K^T) V on packed qkv.  Reference: flash_attention.py:11-71."""

    def __init__(self, softmax_scale=None, attention_dropout=0.0, device=None, dtype=None):
        super().__init__()
        self.softmax_scale = softmax_scale
        self.dropout_p = attention_dropout

    def forward(self, qkv, key_padding_mask=None, causal=False, cu_seqlens=None, max_s=None,
                need_weights=False):
        """qkv (B,S,3,H,D), or (nnz,3,H,D) together with cu_seqlens/max_s.
        key_padding_mask (B,S) bool, True = keep.  Returns (output, None)."""
        assert not need_weights
        assert qkv.dtype in (torch.float16, torch.bfloat16)
        assert qkv.is_cuda
        p_drop = self.dropout_p if self.training else 0.0
        if cu_seqlens is not None:
            assert max_s is not None
            return flash_attn_unpadded_qkvpacked_func(qkv, cu_seqlens, max_s, p_drop,
                                                      softmax_scale=self.softmax_scale,
                                                      causal=causal), None
        batch, seqlen = qkv.shape[0], qkv.shape[1]
        if key_padding_mask is None:
            # cu_seqlens=None = fixed-length batch (the reference builds an arange, flash_attention.py:47-49)
            out = flash_attn_unpadded_qkvpacked_func(qkv.flatten(0, 1), None, seqlen, p_drop,
                                                     softmax_scale=self.softmax_scale, causal=causal)
            return out.unflatten(0, (batch, seqlen)), None
        nheads, hd = qkv.shape[-2], qkv.shape[-1]
        rows, indices, cu, max_len = unpad_input(qkv.flatten(2), key_padding_mask)
        out = flash_attn_unpadded_qkvpacked_func(rows.unflatten(-1, (3, nheads, hd)), cu, max_len,
                                                 p_drop, softmax_scale=self.softmax_scale,
                                                 causal=causal)
        out = pad_input(out.flatten(1), indices, batch, seqlen)
        return out.unflatten(-1, (nheads, hd)), None


class FlashMHA(nn.Module):
    """Wqkv -> FlashAttention -> out_proj.  Reference: flash_attention.py:74-101."""

    def __init__(self, embed_dim, num_heads, bias=True, batch_first=True, attention_dropout=0.0,
                 causal=False, device=None, dtype=None, **kwargs):
        assert batch_first
        factory_kwargs = {'device': device, 'dtype': dtype}
        super().__init__()
        self.embed_dim, self.causal, self.num_heads = embed_dim, causal, num_heads
        assert embed_dim % num_heads == 0
        self.head_dim = embed_dim // num_heads
        assert self.head_dim % 8 == 0 and self.head_dim <= 128
        self.Wqkv = nn.Linear(embed_dim, 3 * embed_dim, bias=bias, **factory_kwargs)
        self.inner_attn = FlashAttention(attention_dropout=attention_dropout, **factory_kwargs)
        self.out_proj = nn.Linear(embed_dim, embed_dim, bias=bias, **factory_kwargs)

    def forward(self, x, key_padding_mask=None, need_weights=False):
        qkv = self.Wqkv(x).unflatten(-1, (3, self.num_heads, self.head_dim))
        ctx, weights = self.inner_attn(qkv, key_padding_mask=key_padding_mask,
                                       need_weights=need_weights, causal=self.causal)
        return self.out_proj(ctx.flatten(-2)), weights

"""Pad / unpad helpers for variable-length batches -- same results as the reference's
flash_attn/bert_padding.py:97-132 (plain torch indexing; autograd comes from torch)."""
import torch
import torch.nn.functional as F


def index_first_axis(x, indices):
    return x[indices]


def index_put_first_axis(values, indices, first_axis_dim):
    out = torch.zeros((first_axis_dim,) + tuple(values.shape[1:]), device=values.device,
                      dtype=values.dtype)
    return out.index_put((indices,), values)


def unpad_input(hidden_states, attention_mask):
    """hidden_states (batch, seqlen, ...), attention_mask (batch, seqlen) with 1 = valid.
    Returns (rows (total_nnz, ...), indices (total_nnz,), cu_seqlens int32 (batch+1,), max_len)."""
    lengths = attention_mask.sum(dim=-1, dtype=torch.int32)
    indices = torch.nonzero(attention_mask.flatten(), as_tuple=False).flatten()
    cu_seqlens = F.pad(torch.cumsum(lengths, dim=0, dtype=torch.int32), (1, 0))
    flat = hidden_states.reshape((-1,) + tuple(hidden_states.shape[2:]))
    return index_first_axis(flat, indices), indices, cu_seqlens, int(lengths.max().item())


def pad_input(hidden_states, indices, batch, seqlen):
    """Inverse of unpad_input: (total_nnz, ...) -> (batch, seqlen, ...) with zeros in the holes."""
    out = index_put_first_axis(hidden_states, indices, batch * seqlen)
    return out.reshape((batch, seqlen) + tuple(hidden_states.shape[1:]))

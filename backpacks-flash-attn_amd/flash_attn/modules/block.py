"""Pre-norm residual block -- mirror of the reference's flash_attn/modules/block.py:22-106 for the
configuration Backpack / GPT-2 use (prenorm=True, drop_path=0).  Sub-module names (mixer, norm1,
mlp, norm2, dropout1/2) match the reference so checkpoints load.

Order of operations (reference comment block.py:70-76): the block receives (hidden, residual),
runs  mixer -> dropout -> add -> LN  and  mlp -> dropout -> add -> LN,  and returns both the LN
output and the fp32 residual stream."""
from functools import partial
from typing import Optional

import torch
import torch.nn as nn
from torch import Tensor

from flash_attn.modules.mha import MHA
from flash_attn.modules.mlp import Mlp
from flash_attn.ops.layer_norm import dropout_add_layer_norm


class Block(nn.Module):

    def __init__(self, dim, mixer_cls=None, mlp_cls=None, norm_cls=nn.LayerNorm,
                 dropout_cls=nn.Dropout, prenorm=True, resid_dropout=0., drop_path=0.,
                 fused_dropout_add_ln=False, return_residual=False, sequence_parallel=False):
        super().__init__()
        if not prenorm or drop_path != 0. or return_residual or sequence_parallel:
            raise NotImplementedError('gfx950 build: Block covers prenorm=True, drop_path=0 only')
        self.prenorm = True
        # fused_dropout_add_ln: add + LayerNorm in ONE HIP launch (bp_add_layer_norm) instead of the
        # three torch kernels of the unfused sequence -- the reference's own switch (block.py:24,82-90)
        self.fused_dropout_add_ln = fused_dropout_add_ln
        self.return_residual = False
        if mixer_cls is None:
            mixer_cls = partial(MHA, num_heads=dim // 64)
        if mlp_cls is None:
            mlp_cls = partial(Mlp, hidden_features=4 * dim)
        self.mixer = mixer_cls(dim)
        self.dropout1 = dropout_cls(resid_dropout)
        self.norm1 = norm_cls(dim)
        self.mlp = mlp_cls(dim)
        if not isinstance(self.mlp, nn.Identity):
            self.dropout2 = dropout_cls(resid_dropout)
            self.norm2 = norm_cls(dim)

    def forward(self, hidden_states: Tensor, residual: Optional[Tensor] = None, mixer_kwargs=None):
        assert residual is not None, 'prenorm block needs the running residual'
        mixed = self.mixer(hidden_states, **(mixer_kwargs or {}))
        if self.fused_dropout_add_ln:
            hidden_states, residual = dropout_add_layer_norm(
                mixed, residual, self.norm1.weight, self.norm1.bias,
                self.dropout1.p if self.training else 0.0, self.norm1.eps, prenorm=True)
        else:
            residual = self.dropout1(mixed) + residual
            hidden_states = self.norm1(residual.to(dtype=self.norm1.weight.dtype))
        if not isinstance(self.mlp, nn.Identity):
            mlp_out = self.mlp(hidden_states)
            if self.fused_dropout_add_ln:
                hidden_states, residual = dropout_add_layer_norm(
                    mlp_out, residual, self.norm2.weight, self.norm2.bias,
                    self.dropout2.p if self.training else 0.0, self.norm2.eps, prenorm=True)
            else:
                residual = self.dropout2(mlp_out) + residual
                hidden_states = self.norm2(residual.to(dtype=self.norm2.weight.dtype))
        return hidden_states, residual

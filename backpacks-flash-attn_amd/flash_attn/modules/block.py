"""Pre-norm residual block -- mirror of the reference's flash_attn/modules/block.py:22-106 for the
configuration Backpack / GPT-2 use (prenorm=True).  Sub-module names (mixer, norm1, mlp, norm2,
dropout1/2, drop_path1/2) match the reference so checkpoints load.

Order of operations (reference comment block.py:70-76): the block receives (hidden, residual),
runs  mixer -> dropout -> drop-path -> add -> LN  and  mlp -> dropout -> drop-path -> add -> LN,  and
returns both the LN output and the fp32 residual stream.  With `fused_dropout_add_ln` the drop-path
factor of a sample travels into the fused kernel as its `rowscale` (reference :82-90, :96-105;
bp_dropout_add_layer_norm_scaled), so stochastic depth costs no extra pass over the activations."""
from functools import partial
from typing import Optional

import torch
import torch.nn as nn
from torch import Tensor

from flash_attn.modules.mha import MHA
from flash_attn.modules.mlp import Mlp
from flash_attn.ops.layer_norm import dropout_add_layer_norm


class StochasticDepth(nn.Module):
    """torchvision.ops.StochasticDepth (the reference imports it, block.py:11): in training a whole row -- a sample,
    mode 'row' -- or the whole batch ('batch') is dropped with probability p and the survivors are scaled by 1 / (1 - p).
    Restated here because torchvision is not part of the image; no parameters, so state dicts are unaffected."""

    def __init__(self, p: float, mode: str = 'row'):
        super().__init__()
        if not 0.0 <= p <= 1.0:
            raise ValueError(f'drop probability has to be between 0 and 1, but got {p}')
        if mode not in ('batch', 'row'):
            raise ValueError(f"mode has to be either 'batch' or 'row', but got {mode}")
        self.p, self.mode = p, mode

    def forward(self, x: Tensor) -> Tensor:
        if not self.training or self.p == 0.0:
            return x
        survival = 1.0 - self.p
        size = [x.shape[0]] + [1] * (x.dim() - 1) if self.mode == 'row' else [1] * x.dim()
        noise = torch.empty(size, dtype=x.dtype, device=x.device).bernoulli_(survival)
        if survival > 0.0:
            noise.div_(survival)
        return x * noise

    def extra_repr(self):
        return f'p={self.p}, mode={self.mode}'


class Block(nn.Module):

    def __init__(self, dim, mixer_cls=None, mlp_cls=None, norm_cls=nn.LayerNorm,
                 dropout_cls=nn.Dropout, prenorm=True, resid_dropout=0., drop_path=0.,
                 fused_dropout_add_ln=False, return_residual=False, sequence_parallel=False):
        super().__init__()
        if not prenorm or return_residual or sequence_parallel:
            raise NotImplementedError('gfx950 build: Block covers prenorm=True without return_residual / sequence_parallel')
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
        self.drop_path1 = StochasticDepth(drop_path, mode='row')
        self.norm1 = norm_cls(dim)
        self.mlp = mlp_cls(dim)
        if not isinstance(self.mlp, nn.Identity):
            self.dropout2 = dropout_cls(resid_dropout)
            self.drop_path2 = StochasticDepth(drop_path, mode='row')
            self.norm2 = norm_cls(dim)

    def _rowscale(self, drop_path, branch):
        """The (B, S) factor the fused add + LayerNorm multiplies the branch's rows by (reference :82-90): None unless
        stochastic depth is active."""
        if drop_path.p == 0 or not self.training:
            return None
        return drop_path(torch.ones(branch.shape[:-1], device=branch.device, dtype=branch.dtype))

    def forward(self, hidden_states: Tensor, residual: Optional[Tensor] = None, mixer_kwargs=None):
        assert residual is not None, 'prenorm block needs the running residual'
        mixed = self.mixer(hidden_states, **(mixer_kwargs or {}))
        if self.fused_dropout_add_ln:
            hidden_states, residual = dropout_add_layer_norm(
                mixed, residual, self.norm1.weight, self.norm1.bias,
                self.dropout1.p if self.training else 0.0, self.norm1.eps,
                rowscale=self._rowscale(self.drop_path1, mixed), prenorm=True)
        else:
            residual = self.drop_path1(self.dropout1(mixed)) + residual
            hidden_states = self.norm1(residual.to(dtype=self.norm1.weight.dtype))
        if not isinstance(self.mlp, nn.Identity):
            mlp_out = self.mlp(hidden_states)
            if self.fused_dropout_add_ln:
                hidden_states, residual = dropout_add_layer_norm(
                    mlp_out, residual, self.norm2.weight, self.norm2.bias,
                    self.dropout2.p if self.training else 0.0, self.norm2.eps,
                    rowscale=self._rowscale(self.drop_path2, mlp_out), prenorm=True)
            else:
                residual = self.drop_path2(self.dropout2(mlp_out)) + residual
                hidden_states = self.norm2(residual.to(dtype=self.norm2.weight.dtype))
        return hidden_states, residual

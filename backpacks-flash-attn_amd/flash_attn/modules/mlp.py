"""Two-layer MLP -- mirror of the reference's flash_attn/modules/mlp.py:13-30: same constructor arguments and
the parameter names fc1 / fc2 that its checkpoints use."""
from functools import partial

import torch.nn as nn
import torch.nn.functional as F


class Mlp(nn.Module):
    """x -> fc2(activation(fc1(x))); with return_residual the input is handed back alongside (prenorm blocks)."""

    def __init__(self, in_features, hidden_features=None, out_features=None, activation=F.gelu,
                 return_residual=False, device=None, dtype=None):
        super().__init__()
        width = hidden_features if hidden_features else in_features
        linear = partial(nn.Linear, device=device, dtype=dtype)
        self.fc1 = linear(in_features, width)
        self.fc2 = linear(width, out_features if out_features else in_features)
        self.activation, self.return_residual = activation, return_residual

    def forward(self, x):
        out = self.fc2(self.activation(self.fc1(x)))
        if self.return_residual:
            return out, x
        return out


from flash_attn.ops.fused_dense import FusedDenseGeluDense  # noqa: E402,F401  (re-exported here, as upstream does)

ParallelFusedDenseGeluDense = None   # tensor parallelism is out of scope

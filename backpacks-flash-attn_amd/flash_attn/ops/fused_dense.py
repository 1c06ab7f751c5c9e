"""Dense layers with fused epilogues -- mirror of the parts of the reference's
flash_attn/ops/fused_dense.py that the Backpack / GPT-2 path instantiates: `FusedDense` (:110-129,
GEMM + bias) and `FusedDenseGeluDense` (:356-404, GEMM + bias + tanh-GELU epilogue, then GEMM + bias).
The reference drives cuBLASLt epilogues from its own extension (csrc/fused_dense_lib); on ROCm the same
epilogues are reached through torch (hipBLASLt): `torch._addmm_activation(bias, x, W^T, use_gelu=True)`
computes gelu_tanh(x @ W^T + bias) in one GEMM launch.  Dense layers are out of scope for hand-written
kernels -- this file only keeps the call sites and state-dict keys."""
import torch
import torch.nn as nn
import torch.nn.functional as F


class FusedDense(nn.Linear):
    """nn.Linear with the reference's constructor / optional residual return."""

    def __init__(self, in_features, out_features, bias=True, return_residual=False, device=None,
                 dtype=None):
        super().__init__(in_features, out_features, bias=bias, device=device, dtype=dtype)
        self.return_residual = return_residual

    def forward(self, x, process_group=None):
        assert process_group is None
        out = F.linear(x, self.weight, self.bias)
        return out if not self.return_residual else (out, x)


def fused_dense_gelu_dense_func(x, weight1, weight2, bias1=None, bias2=None, save_pre_act=True,
                                return_residual=False, checkpoint_lvl=0, heuristic=0,
                                process_group=None):
    assert process_group is None
    lead = x.shape[:-1]
    x2 = x.reshape(-1, x.shape[-1])
    if (x2.is_cuda and bias1 is not None and x2.dtype in (torch.float16, torch.bfloat16)
            and weight1.dtype == x2.dtype and bias1.dtype == x2.dtype       # not under AMP (fp32 parameters)
            and not torch.is_grad_enabled()):
        hidden = torch._addmm_activation(bias1, x2, weight1.t(), use_gelu=True)   # tanh GELU epilogue
    else:
        hidden = F.gelu(F.linear(x2, weight1, bias1), approximate='tanh')
    out = F.linear(hidden, weight2, bias2).reshape(*lead, weight2.shape[0])
    return out if not return_residual else (out, x)


class FusedDenseGeluDense(nn.Module):
    """fc1 -> tanh-GELU -> fc2 with the GELU in fc1's GEMM epilogue (keys fc1.*, fc2.*)."""

    def __init__(self, in_features, hidden_features, out_features=None, bias1=True, bias2=True,
                 return_residual=False, checkpoint_lvl=0, heuristic=0, device=None, dtype=None):
        assert checkpoint_lvl in (0, 1, 2)
        super().__init__()
        out_features = out_features or in_features
        self.return_residual = return_residual
        self.checkpoint_lvl = checkpoint_lvl
        self.heuristic = heuristic
        self.fc1 = nn.Linear(in_features, hidden_features, bias=bias1, device=device, dtype=dtype)
        self.fc2 = nn.Linear(hidden_features, out_features, bias=bias2, device=device, dtype=dtype)

    def forward(self, x, process_group=None):
        return fused_dense_gelu_dense_func(x, self.fc1.weight, self.fc2.weight, self.fc1.bias,
                                           self.fc2.bias, save_pre_act=self.training,
                                           return_residual=self.return_residual)

"""Dense layers with fused epilogues -- mirror of the parts of the reference's flash_attn/ops/fused_dense.py that
the Backpack / GPT-2 path instantiates: `FusedDense` / `fused_dense_func` / `FusedDenseFunc` (:20-129) and
`FusedDenseGeluDense` / `fused_dense_gelu_dense_func` / `FusedDenseGeluDenseFunc` (:175-404).

The reference drives cuBLASLt epilogues from its own extension (csrc/fused_dense_lib): GEMM + bias + GELU with the
pre-activation as a second output, dGELU + bias gradient in the epilogue of the backward GEMM, and the bias gradient
inside the weight-gradient GEMM.  Here the GEMMs stay on the BLAS library through torch (SURVEY.md section 2 row 8:
dense layers are not hand-written) and everything AROUND them is one pass of a HIP kernel (csrc/bias_gelu.hip):

    inference (no grad, 16-bit)  gelu(x W1^T + b1) in ONE library launch: `torch._addmm_activation`
    training forward             pre = x W1^T + b1 (bias in the GEMM epilogue) -> bp_bias_gelu_fwd -> hidden
    training backward            g W2 -> bp_bias_gelu_bwd: dGELU and the b1 gradient in the same pass;
                                 every other bias gradient -> bp_column_sum (deterministic two-stage sums)

Tensor parallelism (`process_group`) is out of scope and raises.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

import bp_hip


def _autocast_dtype(x):
    """The dtype the reference's @custom_fwd bodies cast to (fused_dense.py:33-36,196-200), or None."""
    if x.is_cuda and torch.is_autocast_enabled('cuda'):
        return torch.get_autocast_dtype('cuda')
    return None


def _eligible(x, *params):
    """The reference's gate for its fused path (:97-104,:339-347) without the 64k-row limit of its extension: 16-bit
    CUDA tensors, or fp32 ones under autocast."""
    if not x.is_cuda or any(p is not None and not p.is_cuda for p in params):
        return False
    return x.dtype in (torch.float16, torch.bfloat16) or (x.dtype == torch.float32 and torch.is_autocast_enabled('cuda'))


def _bias_grad(grad2d, bias_dtype):
    """Column sums of a (rows, cols) gradient; `bias_dtype`: dtype of the (possibly fp32 master) bias parameter."""
    if bp_hip.bias_gelu_supported(grad2d):
        return bp_hip.column_sum(grad2d, torch.float32 if bias_dtype == torch.float32 else grad2d.dtype)
    return grad2d.sum(0)


class FusedDenseFunc(torch.autograd.Function):
    """x W^T + b; backward: dx = g W, dW = g^T x (BLAS), db = bp_column_sum(g)."""

    @staticmethod
    def forward(ctx, x, weight, bias, return_residual=False, process_group=None):
        if process_group is not None:
            raise NotImplementedError('tensor parallelism is out of scope for the gfx950 build')
        cast = _autocast_dtype(x)
        ctx.bias_dtype = bias.dtype if bias is not None else None
        if cast is not None:
            x, weight = x.to(cast), weight.to(cast)
            bias = bias.to(cast) if bias is not None else None
        x = x.contiguous()
        ctx.return_residual = return_residual
        ctx.save_for_backward(x if ctx.needs_input_grad[1] else None, weight)
        out = F.linear(x, weight, bias)
        return out if not return_residual else (out, x)

    @staticmethod
    def backward(ctx, grad_output, *extra):
        x, weight = ctx.saved_tensors
        g = grad_output.contiguous().reshape(-1, grad_output.shape[-1])
        grad_input = grad_weight = grad_bias = None
        if ctx.needs_input_grad[0]:
            if ctx.return_residual:   # the gradient of the returned residual folds into the GEMM (:76-78)
                grad_input = torch.addmm(extra[0].reshape(g.shape[0], -1), g, weight)
            else:
                grad_input = g @ weight
            grad_input = grad_input.reshape(*grad_output.shape[:-1], weight.shape[1])
        if ctx.needs_input_grad[1]:
            grad_weight = g.t() @ x.reshape(g.shape[0], -1)
        if ctx.needs_input_grad[2]:
            grad_bias = _bias_grad(g, ctx.bias_dtype)
        return grad_input, grad_weight, grad_bias, None, None


def fused_dense_func(x, weight, bias=None, return_residual=False, process_group=None):
    if process_group is not None:
        raise NotImplementedError('tensor parallelism is out of scope for the gfx950 build')
    if _eligible(x, weight, bias) and torch.is_grad_enabled() and (
            x.requires_grad or weight.requires_grad or (bias is not None and bias.requires_grad)):
        return FusedDenseFunc.apply(x, weight, bias, return_residual, None)
    out = F.linear(x, weight, bias)
    return out if not return_residual else (out, x)


class FusedDense(nn.Linear):
    """nn.Linear with the reference's constructor / optional residual return (:110-129)."""

    def __init__(self, in_features, out_features, bias=True, return_residual=False, device=None,
                 dtype=None):
        super().__init__(in_features, out_features, bias=bias, device=device, dtype=dtype)
        self.return_residual = return_residual

    def forward(self, x, process_group=None):
        return fused_dense_func(x, self.weight, self.bias, return_residual=self.return_residual,
                                process_group=process_group)


def _gelu(pre):
    """tanh-GELU of a 16-bit (rows, cols) tensor: the HIP kernel where it applies, torch otherwise."""
    if bp_hip.bias_gelu_supported(pre):
        return bp_hip.bias_gelu_fwd(pre)[0]
    return F.gelu(pre, approximate='tanh')


class FusedDenseGeluDenseFunc(torch.autograd.Function):
    """fc2(gelu_tanh(fc1(x))) with the reference's three checkpoint levels (:186-192): 0 keeps the pre-activation and
    the hidden state, 1 recomputes the hidden state, 2 recomputes both in backward."""

    @staticmethod
    def forward(ctx, x, weight1, bias1, weight2, bias2, save_pre_act=True, return_residual=False,
                checkpoint_lvl=0, heuristic=0, process_group=None):
        if process_group is not None:
            raise NotImplementedError('tensor parallelism is out of scope for the gfx950 build')
        ctx.bias_dtypes = tuple(b.dtype if b is not None else None for b in (bias1, bias2))
        cast = _autocast_dtype(x)
        if cast is not None:
            x, weight1, weight2 = x.to(cast), weight1.to(cast), weight2.to(cast)
            bias1 = bias1.to(cast) if bias1 is not None else None
            bias2 = bias2.to(cast) if bias2 is not None else None
        if not save_pre_act:
            checkpoint_lvl = 2
        x = x.contiguous()
        x2 = x.reshape(-1, x.shape[-1])
        pre = F.linear(x2, weight1, bias1)          # bias in the GEMM's epilogue
        hidden = _gelu(pre)
        out = F.linear(hidden, weight2, bias2)
        ctx.checkpoint_lvl, ctx.return_residual = checkpoint_lvl, return_residual
        if checkpoint_lvl == 0:
            ctx.save_for_backward(x, weight1, weight2, pre, hidden)
        elif checkpoint_lvl == 1:
            ctx.save_for_backward(x, weight1, weight2, pre)
        else:
            ctx.save_for_backward(x, weight1, weight2, bias1)
        out = out.reshape(*x.shape[:-1], out.shape[-1])
        return out if not return_residual else (out, x)

    @staticmethod
    def backward(ctx, grad_output, *extra):
        x, weight1, weight2, *rest = ctx.saved_tensors
        x2 = x.reshape(-1, x.shape[-1])
        if ctx.checkpoint_lvl == 0:
            pre, hidden = rest
        elif ctx.checkpoint_lvl == 1:
            pre, = rest
            hidden = _gelu(pre)
        else:
            pre = F.linear(x2, weight1, rest[0])
            hidden = _gelu(pre)
        bias1_dtype, bias2_dtype = ctx.bias_dtypes
        g = grad_output.contiguous().reshape(-1, grad_output.shape[-1])
        grad_weight2 = g.t() @ hidden if ctx.needs_input_grad[3] else None
        grad_bias2 = _bias_grad(g, bias2_dtype) if ctx.needs_input_grad[4] else None
        grad_hidden = g @ weight2
        want_b1 = ctx.needs_input_grad[2]
        if bp_hip.bias_gelu_supported(grad_hidden):
            b1_dtype = None
            if want_b1:
                b1_dtype = torch.float32 if bias1_dtype == torch.float32 else grad_hidden.dtype
            grad_pre, grad_bias1 = bp_hip.bias_gelu_bwd(grad_hidden, pre, b1_dtype, inplace=True)
        else:
            with torch.enable_grad():
                p_ = pre.detach().requires_grad_()
                grad_pre, = torch.autograd.grad(F.gelu(p_, approximate='tanh'), p_, grad_hidden)
            grad_bias1 = grad_pre.sum(0) if want_b1 else None
        grad_input = None
        if ctx.needs_input_grad[0]:
            if ctx.return_residual:
                grad_input = torch.addmm(extra[0].reshape(x2.shape), grad_pre, weight1)
            else:
                grad_input = grad_pre @ weight1
            grad_input = grad_input.reshape(x.shape)
        grad_weight1 = grad_pre.t() @ x2 if ctx.needs_input_grad[1] else None
        return (grad_input, grad_weight1, grad_bias1, grad_weight2, grad_bias2, None, None, None, None, None)


def fused_dense_gelu_dense_func(x, weight1, weight2, bias1=None, bias2=None, save_pre_act=True,
                                return_residual=False, checkpoint_lvl=0, heuristic=0,
                                process_group=None):
    if process_group is not None:
        raise NotImplementedError('tensor parallelism is out of scope for the gfx950 build')
    params = (weight1, weight2, bias1, bias2)
    if _eligible(x, *params):
        if torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in (x,) + params):
            return FusedDenseGeluDenseFunc.apply(x, weight1, bias1, weight2, bias2, save_pre_act, return_residual,
                                                 checkpoint_lvl, heuristic, None)
        cast = _autocast_dtype(x)
        x2 = x.reshape(-1, x.shape[-1])
        if cast is not None:
            x2, weight1, weight2 = x2.to(cast), weight1.to(cast), weight2.to(cast)
            bias1 = bias1.to(cast) if bias1 is not None else None
            bias2 = bias2.to(cast) if bias2 is not None else None
        if bias1 is not None:
            hidden = torch._addmm_activation(bias1, x2, weight1.t(), use_gelu=True)   # tanh GELU in the epilogue
        else:
            hidden = _gelu(x2 @ weight1.t())
        out = F.linear(hidden, weight2, bias2).reshape(*x.shape[:-1], weight2.shape[0])
        return out if not return_residual else (out, x)
    out = F.linear(F.gelu(F.linear(x, weight1, bias1), approximate='tanh'), weight2, bias2)
    return out if not return_residual else (out, x)


class FusedDenseGeluDense(nn.Module):
    """fc1 -> tanh-GELU -> fc2 (state-dict keys fc1.*, fc2.*), constructor as the reference's (:356-386)."""

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
                                           return_residual=self.return_residual,
                                           checkpoint_lvl=self.checkpoint_lvl, heuristic=self.heuristic,
                                           process_group=process_group)

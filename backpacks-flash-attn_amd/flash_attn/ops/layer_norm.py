"""Fused (dropout +) residual add + LayerNorm -- mirror of the reference's
flash_attn/ops/layer_norm.py (`dropout_add_layer_norm` :207-217, `layer_norm` :203-204,
`DropoutAddLayerNorm` :232-252) on the HIP kernel bp_add_layer_norm.

Dropout runs inside the kernel (bp_dropout_add_layer_norm: counter-based bits from a two-word generator state
drawn on the device, regenerated in backward instead of reading a saved mask back); x0 may be 16-bit or fp32 (the
AMP case: fp32 embedding output into the first LayerNorm).  rowscale / layerscale (DropPath, LayerScale; reference
:102-150,207-217) are inside the same kernels (bp_dropout_add_layer_norm_scaled{,_bwd}, round 4) although no Backpack /
GPT-2 config uses them; the `subset` variant (:153-200, ViT token dropping) is not provided.  Backward is the HIP kernel
for rows up to 2048 columns
(statistics recomputed from the saved summed stream); wider rows raise unless the caller opted in to differentiating
the eager expression (`bp_hip.allow_eager_fallback()`)."""
import torch
import torch.nn.functional as F
from torch.nn import init

import bp_hip


class DropoutAddLayerNormFn(torch.autograd.Function):

    @staticmethod
    def forward(ctx, x0, x1, gamma, beta, rowscale, colscale, dropout_p, epsilon, residual_in_fp32, prenorm,
                return_dmask=False):
        residual_dtype = x1.dtype if x1 is not None else (torch.float32 if residual_in_fp32 else x0.dtype)
        rng_state = bp_hip.new_rng_state(x0.device) if dropout_p > 0.0 else None
        # the summed stream is needed by backward (statistics are recomputed from it) whenever it is not x0 itself
        # (reference ln_fwd_kernels.cuh:50 `save_x`)
        need_x = (prenorm or x1 is not None or residual_dtype != x0.dtype or dropout_p > 0.0
                  or rowscale is not None or colscale is not None)
        outs = bp_hip.add_layer_norm(x0, x1, gamma, beta, epsilon, residual_dtype=residual_dtype,
                                     return_residual=need_x, dropout_p=dropout_p, rng_state=rng_state,
                                     return_dropout_mask=return_dmask, rowscale=rowscale, colscale=colscale)
        outs = outs if isinstance(outs, tuple) else (outs,)
        z = outs[0]
        x = outs[1] if need_x else None
        dmask = outs[-1] if return_dmask else None
        # (x0 itself only when the gradient of colscale needs it, as upstream :119-120)
        ctx.save_for_backward(x if x is not None else x0, gamma, beta, rowscale, colscale,
                              x0 if colscale is not None else None)
        ctx.eps, ctx.prenorm, ctx.residual_dtype = epsilon, prenorm, residual_dtype
        ctx.x0_dtype, ctx.has_x1 = x0.dtype, x1 is not None
        ctx.dropout_p, ctx.rng_state = dropout_p, rng_state
        ctx.fallback_recorded = bp_hip.eager_fallback_allowed()
        result = (z, x) if prenorm else (z,)
        if return_dmask:
            ctx.mark_non_differentiable(dmask)
            result = result + (dmask,)
        return result if len(result) > 1 else result[0]

    @staticmethod
    def backward(ctx, dz, *args):
        xsum, gamma, beta, rowscale, colscale, x0 = ctx.saved_tensors
        dx = args[0] if (ctx.prenorm and args) else None
        if bp_hip.add_layer_norm_bwd_supported(ctx.x0_dtype, xsum.shape[-1]):
            if dx is not None and dx.dtype != xsum.dtype:
                dx = dx.to(xsum.dtype)
            dx0, dx1, dg, dbt, *rest = bp_hip.add_layer_norm_bwd(dz, dx, xsum, gamma, ctx.eps, want_dx1=ctx.has_x1,
                                                                 dropout_p=ctx.dropout_p, rng_state=ctx.rng_state,
                                                                 rowscale=rowscale, colscale=colscale, x0=x0)
            return (dx0.view_as(dz), (dx1.view_as(xsum) if ctx.has_x1 else None), dg, dbt, None,
                    (rest[0] if colscale is not None else None), None, None, None, None, None)
        if ctx.dropout_p > 0.0 or rowscale is not None or colscale is not None:
            raise RuntimeError('dropout_add_layer_norm (gfx950 build): fused dropout / rowscale / layerscale need <= 2048 columns')
        if not bp_hip.eager_fallback_allowed(ctx):
            raise RuntimeError('dropout_add_layer_norm (gfx950 build): the HIP backward takes rows of up to 2048 columns (got '
                               '%d); differentiating the eager expression instead is opt-in: '
                               '`with bp_hip.allow_eager_fallback():` around the forward or backward()' % xsum.shape[-1])
        with torch.enable_grad():   # wide rows: differentiate the eager expression
            a = xsum.detach().float().requires_grad_()
            g, bt = gamma.detach().requires_grad_(), beta.detach().requires_grad_()
            z = F.layer_norm(a, (a.shape[-1],), g.float(), bt.float(), ctx.eps).to(ctx.x0_dtype)
            outs, grads = [z], [dz]
            if dx is not None:
                outs.append(a.to(ctx.residual_dtype))
                grads.append(dx)
            da, dg, dbt = torch.autograd.grad(outs, [a, g, bt], grads)
        return (da.to(ctx.x0_dtype), (da.to(ctx.residual_dtype) if ctx.has_x1 else None), dg, dbt,
                None, None, None, None, None, None, None)


def dropout_add_layer_norm(x0, x1, weight, bias, dropout_p, epsilon, rowscale=None, layerscale=None,
                           prenorm=False, residual_in_fp32=False, return_dropout_mask=False):
    """z = LayerNorm(dropout(x0) + x1); with prenorm=True returns (z, dropout(x0) + x1); with
    return_dropout_mask=True the uint8 keep mask is appended (reference :207-217).
    residual_in_fp32 only has an effect if x1 is None, otherwise the residual dtype is x1.dtype."""
    assert x0.is_cuda and x0.dtype in (torch.float16, torch.bfloat16, torch.float32)
    return DropoutAddLayerNormFn.apply(x0, x1, weight, bias, rowscale, layerscale, float(dropout_p), epsilon,
                                       residual_in_fp32, prenorm, return_dropout_mask)


def layer_norm(x, weight, bias, epsilon):
    return DropoutAddLayerNormFn.apply(x, None, weight, bias, None, None, 0.0, epsilon, False, False)


class DropoutAddLayerNorm(torch.nn.Module):

    def __init__(self, hidden_size, prenorm=False, p=0.0, eps=1e-5, residual_in_fp32=False,
                 device=None, dtype=None):
        super().__init__()
        self.prenorm, self.p, self.epsilon, self.residual_in_fp32 = prenorm, p, eps, residual_in_fp32
        self.weight = torch.nn.Parameter(torch.empty(hidden_size, device=device, dtype=dtype))
        self.bias = torch.nn.Parameter(torch.empty(hidden_size, device=device, dtype=dtype))
        self.reset_parameters()

    def reset_parameters(self):
        init.ones_(self.weight)
        init.zeros_(self.bias)

    def forward(self, x0, x1=None):
        return dropout_add_layer_norm(x0, x1, self.weight, self.bias, self.p if self.training else 0.0,
                                      self.epsilon, prenorm=self.prenorm,
                                      residual_in_fp32=self.residual_in_fp32)

"""Fused softmax cross-entropy -- mirror of the reference's flash_attn/losses/cross_entropy.py (same class
names and arguments), with `xentropy_cuda_lib.forward/backward` (:37,:54,:103) replaced by the HIP kernels
behind bp_xentropy_fwd / bp_xentropy_bwd (include/bp_hip.h): one streaming pass over the logits each way,
losses and the row log-sum-exp in fp32, gradient optionally written over the logits (`inplace_backward`).

The vocabulary-parallel branch (process_group, :41-91) belongs to tensor parallelism, which the Backpack path
never uses (`process_group=None` throughout): passing a process group raises.
"""
import torch
import torch.nn as nn

import bp_hip


class SoftmaxCrossEntropyLossFn(torch.autograd.Function):

    @staticmethod
    def forward(ctx, logits, labels, smoothing=0.0, ignored_index=-100, inplace_backward=False,
                process_group=None):
        """logits (batch, vocab_size) on the GPU, labels (batch,)."""
        if process_group is not None:
            raise NotImplementedError('gfx950 build: vocabulary-parallel cross entropy (tensor parallelism) is '
                                      'out of scope; the Backpack path runs with process_group=None')
        rows, classes = logits.shape
        assert labels.shape == (rows,)
        skip = labels == ignored_index
        losses, lse = bp_hip.xentropy_fwd(logits, labels, smoothing, classes)
        losses.masked_fill_(skip, 0)
        ctx.save_for_backward(logits, lse, labels)
        ctx.smoothing, ctx.ignored_index, ctx.inplace_backward = smoothing, ignored_index, inplace_backward
        ctx.total_classes = classes
        return losses

    @staticmethod
    def backward(ctx, grad_loss):
        logits, lse, labels = ctx.saved_tensors
        grad_loss = grad_loss.contiguous().masked_fill(labels == ctx.ignored_index, 0)
        grad_logits = bp_hip.xentropy_bwd(grad_loss, logits, lse, labels, ctx.smoothing, ctx.inplace_backward,
                                          ctx.total_classes)
        return grad_logits, None, None, None, None, None, None


class CrossEntropyLoss(nn.Module):

    def __init__(self, ignore_index=-100, reduction='mean', label_smoothing=0.0, inplace_backward=False,
                 process_group=None):
        super().__init__()
        if reduction not in ['mean', 'none']:
            raise NotImplementedError("Only support reduction = 'mean' or 'none'")
        self.ignore_index = ignore_index
        self.reduction = reduction
        self.label_smoothing = label_smoothing
        self.inplace_backward = inplace_backward
        self.process_group = process_group

    def forward(self, input, target):
        assert input.is_cuda and target.is_cuda
        loss = SoftmaxCrossEntropyLossFn.apply(input, target, self.label_smoothing, self.ignore_index,
                                               self.inplace_backward, self.process_group)
        if self.reduction == 'mean':
            return loss.sum() / (target != self.ignore_index).sum()
        return loss


CrossEntropyLossApex = CrossEntropyLoss   # name used by the reference's tests (tests/losses/test_cross_entropy.py:9)

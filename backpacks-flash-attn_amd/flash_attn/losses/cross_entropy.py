"""Fused softmax cross-entropy -- mirror of the reference's flash_attn/losses/cross_entropy.py (same class
names and arguments), with `xentropy_cuda_lib.forward/backward` (:37,:54,:103) replaced by the HIP kernels
behind bp_xentropy_fwd / bp_xentropy_bwd (include/bp_hip.h): one streaming pass over the logits each way,
losses and the row log-sum-exp in fp32, gradient optionally written over the logits (`inplace_backward`).

The vocabulary-parallel branch (process_group, :41-91) keeps the reference's algebra -- each rank computes its
local loss and LSE, one all-gather of the LSEs and one all-reduce of the losses give the global values -- on
torch.distributed (RCCL on ROCm).
"""
import torch
import torch.nn as nn

import bp_hip


def _merge_vocab_shards(losses, lse_shard, labels, shard_size, smoothing, group):
    """Global losses / LSE from per-shard ones (reference algebra, :65-91): with L = logsumexp over shards of
    the shard LSEs and o the shard owning the label,
        loss = loss_partial_sum + (1 - s)(L - lse_o) + s (L - sum_shards lse)."""
    world = torch.distributed.get_world_size(group)
    rows = labels.shape[0]
    gathered = torch.empty(world, rows, dtype=lse_shard.dtype, device=lse_shard.device)
    torch.distributed.all_gather_into_tensor(gathered, lse_shard.contiguous(), group=group)
    reduce_losses = torch.distributed.all_reduce(losses, op=torch.distributed.ReduceOp.SUM, group=group,
                                                 async_op=True)
    lse = torch.logsumexp(gathered, dim=0)
    owner = torch.div(labels, shard_size, rounding_mode='floor').clamp_(0, world - 1)
    lse_owner = gathered.gather(0, owner.unsqueeze(0)).squeeze(0)
    correction = (1 - smoothing) * (lse - lse_owner)
    if smoothing != 0.0:
        correction = correction + smoothing * (lse - gathered.sum(dim=0))
    reduce_losses.wait()
    losses += correction
    return losses, lse


class SoftmaxCrossEntropyLossFn(torch.autograd.Function):

    @staticmethod
    def forward(ctx, logits, labels, smoothing=0.0, ignored_index=-100, inplace_backward=False,
                process_group=None):
        """logits (batch, vocab_size) on the GPU, labels (batch,).  With a process group every rank holds
        one contiguous slice of the vocabulary."""
        rows, shard_size = logits.shape
        assert labels.shape == (rows,)
        shards = 1 if process_group is None else torch.distributed.get_world_size(process_group)
        skip = labels == ignored_index
        kernel_labels = labels
        if shards > 1:
            # labels owned by another shard fall outside [0, shard_size): the kernel then leaves out the
            # target-logit term -- what the reference's shifted labels achieve (:41-63)
            offset = torch.distributed.get_rank(process_group) * shard_size
            kernel_labels = torch.where(skip, labels, labels - offset)
        losses, lse = bp_hip.xentropy_fwd(logits, kernel_labels, smoothing, shards * shard_size)
        losses.masked_fill_(skip, 0)
        if shards > 1:
            losses, lse = _merge_vocab_shards(losses, lse, labels, shard_size, smoothing, process_group)
            losses.masked_fill_(skip, 0)
        ctx.save_for_backward(logits, lse, kernel_labels)
        ctx.smoothing, ctx.ignored_index, ctx.inplace_backward = smoothing, ignored_index, inplace_backward
        ctx.total_classes = shards * shard_size
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

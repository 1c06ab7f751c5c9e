"""Evaluation loss / perplexity of a (Backpack) LM head model without the full logits tensor.

The reference evaluates with `logits = model(ids).logits` followed by a cross-entropy over (B*S, vocab)
(training/src/tasks/seq.py; metric training/src/metrics/perplexity.py): at Backpack-Small that is 103 MB of
bf16 logits per 1024-token sample, which -- not the model -- caps the evaluation batch.  Here the LM head and
the loss run over row chunks: (chunk, d) x (d, vocab) -> fused cross-entropy (bp_xentropy_fwd) -> two scalars
kept, the chunk's logits dropped.  Peak memory is one chunk of logits whatever the batch size.
"""
import torch
import torch.nn.functional as F

import bp_hip


@torch.no_grad()
def lm_loss_chunked(model, input_ids, labels=None, chunk_tokens=16384, ignore_index=-100):
    """Mean next-token cross-entropy of `model` (BackpackLMHeadModel / GPTLMHeadModel mirror) on `input_ids`.

    labels (B,S) default to the inputs shifted left (last position ignored).  Returns (loss, n_tokens): a 0-d
    fp32 tensor and the number of scored positions; perplexity = exp(loss).
    """
    if labels is None:
        labels = torch.full_like(input_ids, ignore_index)
        labels[:, :-1] = input_ids[:, 1:]
    hidden = model.transformer(input_ids)                       # (B,S,d)
    rows = hidden.reshape(-1, hidden.shape[-1])
    flat_labels = labels.reshape(-1)
    total = torch.zeros((), dtype=torch.float32, device=rows.device)
    for lo in range(0, rows.shape[0], chunk_tokens):
        logits = model.lm_head(rows[lo:lo + chunk_tokens])
        y = flat_labels[lo:lo + chunk_tokens]
        if logits.is_cuda:
            losses, _ = bp_hip.xentropy_fwd(logits, y)
            total += losses.masked_fill_(y == ignore_index, 0).sum()
        else:   # the reference's non-fused mode
            total += F.cross_entropy(logits.float(), y, ignore_index=ignore_index, reduction='sum')
    count = (flat_labels != ignore_index).sum()
    return total / count.clamp(min=1), int(count)

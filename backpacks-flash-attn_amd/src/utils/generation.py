"""Greedy decoding / sampling on top of the full forward -- mirror of the reference's
training/src/utils/generation.py:23-92.  As upstream there is no KV cache: every step re-runs the
whole forward on the grown prefix (so every step exercises the HIP attention and sense-mix kernels).
Differences kept deliberately small: the result is a plain dataclass instead of the
transformers `*DecoderOnlyOutput` classes (removed in transformers 5), and the appended token is
`unsqueeze(1)` so batch sizes > 1 work (the reference's `unsqueeze(0)` in greedy_decode, :68, only
concatenates for batch 1; identical result there).

Index contract kept bit for bit: the reference never appends the LAST token it picks (:64-72 -- `seqlen`
starts at prompt + 1 and the loop appends only while `seqlen < max_length`), so `sequences` has
max(prompt_len, max_length - 1) columns, not max_length as its docstring says; `scores` holds the first
step's logits only (:59).  Same here."""
from dataclasses import dataclass, field
from typing import Optional, Tuple

import torch


@dataclass
class InferenceParams:
    """Kept for signature compatibility (reference :11-20); unused because nothing is cached."""
    max_sequence_len: int
    max_batch_size: int
    sequence_len_offset: int = 0
    batch_size_offset: int = 0
    key_value_memory_dict: dict = field(default_factory=dict)


@dataclass
class DecoderOnlyOutput:
    sequences: torch.Tensor
    scores: Optional[Tuple[torch.Tensor, ...]] = None


def _decode(input_ids, model, max_length, pick):
    """The reference's loop, statement for statement (:56-72 / :31-44)."""
    seqlen_og = input_ids.shape[1]
    with torch.inference_mode():
        logits = model(input_ids).logits[:, -1]
        scores = [logits]                       # upstream records the first step's scores only (:59,:32)
        next_token = pick(logits)
        seqlen = seqlen_og + 1
        while seqlen < max_length:
            input_ids = torch.cat((input_ids, next_token.unsqueeze(1)), dim=1)
            logits = model(input_ids).logits[:, -1]
            next_token = pick(logits)           # the pick of the final iteration is dropped, as upstream
            seqlen += 1
    return DecoderOnlyOutput(sequences=input_ids, scores=tuple(scores))


def _decode_graphed(input_ids, model, max_length, pick):
    """The same loop on ONE captured forward (HIP graph): the prefix lives in a fixed (batch, width) buffer padded with
    token 0, width = the final sequence length.  Every layer of the model is causal or per-token, so the logits of
    position t do not depend on what is stored behind it: replaying the full-width forward and reading row t gives what
    the reference computes on the grown prefix, for ~150 kernel launches less host work per token (the decode loop of a
    small model is launch-bound: there is no KV cache upstream either).  Greedy tokens can differ from the eager loop
    only where two logits tie to within the rounding of a differently tiled GEMM."""
    batch, seqlen_og = input_ids.shape
    width = max(seqlen_og, max_length - 1)
    buf = torch.zeros((batch, width), dtype=input_ids.dtype, device=input_ids.device)
    buf[:, :seqlen_og] = input_ids
    with torch.inference_mode():
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(2):                      # warm-up outside the capture (workspaces, library handles)
                model(buf)
        torch.cuda.current_stream().wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            static_logits = model(buf).logits
        graph.replay()
        logits = static_logits[:, seqlen_og - 1].clone()
        scores = [logits]
        next_token = pick(logits)
        seqlen = seqlen_og + 1
        while seqlen < max_length:
            buf[:, seqlen - 1] = next_token
            graph.replay()
            next_token = pick(static_logits[:, seqlen - 1])
            seqlen += 1
    return DecoderOnlyOutput(sequences=buf[:, :max(seqlen_og, seqlen - 1)].clone(), scores=tuple(scores))


def greedy_decode(input_ids, model, max_length, cg=False):
    """input_ids (batch, seq_len) -> sequences (batch, max_length - 1): argmax continuation.
    cg=True: one captured full-width forward replayed per token (CUDA tensors only), see _decode_graphed."""
    pick = lambda logits: torch.argmax(logits, dim=-1)   # noqa: E731
    if cg and input_ids.is_cuda:
        return _decode_graphed(input_ids, model, max_length, pick)
    return _decode(input_ids, model, max_length, pick)


def sample(input_ids, model, max_length, cg=False):
    """Ancestral sampling from softmax(logits) (reference :23-48)."""
    def pick(logits):
        return torch.distributions.Categorical(logits=torch.log_softmax(logits.float(), dim=-1)).sample()
    if cg and input_ids.is_cuda:
        return _decode_graphed(input_ids, model, max_length, pick)
    return _decode(input_ids, model, max_length, pick)


class GenerationMixin:

    def generate(self, input_ids, max_length, return_dict_in_generate=False, output_scores=False, cg=False):
        output = greedy_decode(input_ids, self, max_length, cg=cg)
        if not output_scores:
            output.scores = None
        return output if return_dict_in_generate else output.sequences

    def sample(self, input_ids, max_length, return_dict_in_generate=False, output_scores=False, cg=False):
        output = sample(input_ids, self, max_length, cg=cg)
        if not output_scores:
            output.scores = None
        return output if return_dict_in_generate else output.sequences

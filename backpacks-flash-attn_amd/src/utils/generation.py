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


def greedy_decode(input_ids, model, max_length):
    """input_ids (batch, seq_len) -> sequences (batch, max_length - 1): argmax continuation."""
    return _decode(input_ids, model, max_length, lambda logits: torch.argmax(logits, dim=-1))


def sample(input_ids, model, max_length):
    """Ancestral sampling from softmax(logits) (reference :23-48)."""
    def pick(logits):
        return torch.distributions.Categorical(logits=torch.log_softmax(logits.float(), dim=-1)).sample()
    return _decode(input_ids, model, max_length, pick)


class GenerationMixin:

    def generate(self, input_ids, max_length, return_dict_in_generate=False, output_scores=False):
        output = greedy_decode(input_ids, self, max_length)
        if not output_scores:
            output.scores = None
        return output if return_dict_in_generate else output.sequences

    def sample(self, input_ids, max_length, return_dict_in_generate=False, output_scores=False):
        output = sample(input_ids, self, max_length)
        if not output_scores:
            output.scores = None
        return output if return_dict_in_generate else output.sequences

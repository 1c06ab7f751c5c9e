"""Control experiments on a trained Backpack -- MI355X-native mirror of the reference's
training/src/models/intervened_models.py (same public names and constructor arguments:
create_content_soft_mask :9-20, get_sense_vector_of_word :23-26, mask_annealing :29-53,
WeightedBackpackLMHeadModel :58-105, NegativeWeightedBackpackLMHeadModel :108-165,
ReplacedWordLMHeadModel :168-199).

All three models change what enters the sense contraction `sum_l alpha_l @ C_l`.  The reference redoes
that contraction in eager ops on a materialised alpha (B,k,S,S) and an edited copy of the content; here,
when the wrapped network runs its HIP path (`config.use_flash_attn`), it stays ONE fused launch:
  * per-(token, sense) weights go in as `key_weight` (C ABI bp_sense_mix_weighted) -- no weighted copy of
    the 25 MB/sample content tensor, no alpha;
  * vocabulary-sized "content logits" use the same kernel with d_out = vocab;
  * the similarity term of the annealing only needs the logits of the tokens that occur in the sequence,
    so it is a (S x S) product per sense against the gathered embeddings instead of a gather out of a
    (B,k,S,vocab) tensor (identical numbers, 50264/S times less work and memory).
With `use_flash_attn=False` the reference's eager op sequence runs (any device / dtype).
"""
from collections import namedtuple

import torch
from torch import nn

import bp_hip
from src.utils.generation import GenerationMixin

CausalLMOutput = namedtuple('CausalLMOutput', ['logits'])


def create_content_soft_mask(content_weights, input_ids, scores):
    """content_weights (vocab, k), input_ids (B,S), scores (B,S,k) -> weights (B,S,k) =
    w[token] * score + (1 - score)   (reference :9-20)."""
    picked = content_weights.to(scores.device)[input_ids]
    return picked * scores + (1 - scores)


def get_sense_vector_of_word(word_id, model, sense_index):
    """Sense vector `sense_index` of token `word_id` (reference :23-26; a sense vector does not depend on
    the context, so one position is enough -- the reference fills a whole n_positions row)."""
    ids = torch.as_tensor(word_id, device=model.lm_head.weight.device).reshape(1, 1).long()
    senses = model.transformer.content_model(ids)          # (1, k, 1, d)
    return senses[0, sense_index, 0, :]


def mask_annealing(model, input_ids, target_vector, content, annealing_scale=0.1, upweight_nearby=True):
    """scores (B,k,S) = sigmoid(-scale * sum_j relu(content[b,l,i] . E[ids[b,j]]) + 6) [* (1 + i/100)]
    (reference :29-53; `target_vector` is unused there as well)."""
    seqlen = input_ids.shape[1]
    emb = model.lm_head.weight[input_ids]                                      # (B, S, d)
    sims = torch.relu(content @ emb.transpose(1, 2).unsqueeze(1)).sum(dim=3)   # (B,k,S,d)@(B,1,d,S) -> sum_j
    scores = torch.sigmoid(-annealing_scale * sims + 6)
    if upweight_nearby:
        scores = scores * (1 + torch.arange(seqlen, device=scores.device) / 100).reshape(1, 1, seqlen)
    return scores


class _Intervened(nn.Module, GenerationMixin):
    """Shared plumbing: the three stages of the wrapped network, and the contraction."""

    def _stages(self, input_ids, position_ids, inference_params):
        t = self.backpack_network.transformer
        hidden = t.gpt2_model(input_ids, position_ids=position_ids, inference_params=inference_params)
        content = t.content_model(input_ids, position_ids, inference_params)      # (B,k,S,d) view
        return t, hidden, content

    @staticmethod
    def _mix(t, hidden, content, key_weight=None):
        """sum_l (alpha_l * key_weight_l) @ content_l; content (B,k,S,d_out), key_weight (B,k,S) or None."""
        attn = t.contextualization_attn
        if t.fused_senses:
            return bp_hip.sense_mix(attn.project(hidden), content.transpose(1, 2), attn.scale(),
                                    key_weight=key_weight)
        alpha = attn(hidden)                                                      # (B,k,S,S)
        if key_weight is not None:
            content = content * key_weight.unsqueeze(3).to(content.dtype)
        return torch.sum(alpha @ content, dim=1)

    def _weights(self, input_ids, content):
        """(B,k,S) per-token, per-sense weights of the soft mask (reference :83-99)."""
        if self.anneal:
            scores = mask_annealing(self.backpack_network, input_ids, self.target_weight, content,
                                    self.annealing_scale, self.upweight_nearby).transpose(1, 2)
        else:
            b, k, s, _ = content.shape
            scores = torch.ones(b, s, k, device=content.device)
        return create_content_soft_mask(self.content_weights, input_ids, scores.float()).transpose(1, 2)


class WeightedBackpackLMHeadModel(_Intervened):
    """Sense vectors re-weighted per (token, sense) before the contraction (reference :58-105)."""

    def __init__(self, backpack_network, content_weights, target_weight, annealing_scale, anneal=True,
                 upweight_nearby=True):
        super().__init__()
        self.backpack_network = backpack_network
        self.content_weights = content_weights          # (vocab, k)
        self.target_weight = target_weight
        self.annealing_scale = annealing_scale
        self.anneal = anneal
        self.upweight_nearby = upweight_nearby

    def forward(self, input_ids, position_ids=None, inference_params=None):
        t, hidden, content = self._stages(input_ids, position_ids, inference_params)
        mixed = self._mix(t, hidden, content, self._weights(input_ids, content))
        return CausalLMOutput(logits=self.backpack_network.lm_head(mixed))


class NegativeWeightedBackpackLMHeadModel(WeightedBackpackLMHeadModel):
    """Per (sense, position) the 2 % most negative re-weighted vocabulary logits replace the plain ones,
    then the contraction runs on vocabulary-sized content (reference :108-165)."""

    def forward(self, input_ids, position_ids=None, inference_params=None):
        t, hidden, content = self._stages(input_ids, position_ids, inference_params)
        weights = self._weights(input_ids, content)                              # (B,k,S)
        # everything in the content's storage order (B,S,k,.) -- the order the kernel reads -- so the
        # vocabulary-sized tensors are produced once and never transposed in memory
        c_st = content.transpose(1, 2)                                           # (B,S,k,d), contiguous
        w_lm_t = self.backpack_network.lm_head.weight.t()
        logits_c = c_st @ w_lm_t                                                 # (B,S,k,V)
        logits_w = (c_st * weights.transpose(1, 2).unsqueeze(3).to(c_st.dtype)) @ w_lm_t
        cut = torch.quantile(logits_w.float(), q=0.02, keepdim=True, dim=-1)
        logits_c = torch.where(logits_w < cut, logits_w, logits_c).transpose(1, 2)   # (B,k,S,V) view
        return CausalLMOutput(logits=self._mix(t, hidden, logits_c))


class ReplacedWordLMHeadModel(_Intervened):
    """Tokens listed in `sense_dict` {token id: (k, d) tensor} contribute those sense vectors instead of
    their own (reference :168-199)."""

    def __init__(self, backpack_network, sense_dict):
        super().__init__()
        self.backpack_network = backpack_network
        self.sense_dict = sense_dict

    def replace_content(self, input_ids, content):
        content = content.clone()
        for word, senses in self.sense_dict.items():           # one masked assignment per listed word
            hit = (input_ids == word)                            # (B,S)
            if bool(hit.any()):
                b_idx, s_idx = hit.nonzero(as_tuple=True)
                content[b_idx, :, s_idx, :] = senses.to(content.device, content.dtype)
        return content

    def forward(self, input_ids, position_ids=None, inference_params=None):
        t, hidden, content = self._stages(input_ids, position_ids, inference_params)
        content = self.replace_content(input_ids, content)
        if t.fused_senses and content.transpose(1, 2).stride(-1) != 1:
            content = content.contiguous()
        mixed = self._mix(t, hidden, content)
        return CausalLMOutput(logits=self.backpack_network.lm_head(mixed))

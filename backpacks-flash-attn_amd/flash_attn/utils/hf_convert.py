"""Hugging Face GPT-2 <-> flash-layout weight conversion -- mirror of the reference's
`remap_state_dict_gpt2` (training/src/models/backpack.py:354-409, same function in
training/demo_convert.py:87-143), `remap_state_dict_flash` (training/demo_convert.py:22-85) and
`load_non_optimized_model` (:7-20).

Both directions are one rename table plus three value rules:
  * HF `Conv1D` weights are (in, out): the four projection matrices are transposed (views, as upstream);
  * the word embedding is zero-padded to the (padded) `config.vocab_size` going in, left padded going out;
  * the LayerNorms shift by half a block: HF applies ln_1 / ln_2 BEFORE attention / MLP and ln_f at the end,
    the pre-norm Block (flash_attn/modules/block.py) applies norm1 / norm2 AFTER them and the model owns ln_0:
        ln_0 = h.0.ln_1      layers.d.norm1 = h.d.ln_2      layers.d.norm2 = h.(d+1).ln_1      last norm2 = ln_f
The causal-mask buffers `h.d.attn.bias` (and `masked_bias`) carry no weights and are dropped.
Pinned by tests/golden/g7_hf_remap.npz (outputs of the reference's own functions).
"""
import re
from collections import OrderedDict

import torch.nn.functional as F

_PER_LAYER = (   # (HF name under h.<d>., flash name under transformer.layers.<d>., transpose)
    ('attn.c_attn.weight', 'mixer.Wqkv.weight', True),
    ('attn.c_attn.bias', 'mixer.Wqkv.bias', False),
    ('attn.c_proj.weight', 'mixer.out_proj.weight', True),
    ('attn.c_proj.bias', 'mixer.out_proj.bias', False),
    ('mlp.c_fc.weight', 'mlp.fc1.weight', True),
    ('mlp.c_fc.bias', 'mlp.fc1.bias', False),
    ('mlp.c_proj.weight', 'mlp.fc2.weight', True),
    ('mlp.c_proj.bias', 'mlp.fc2.bias', False),
)


def _norm_pairs(n_layer):
    """[(HF LayerNorm prefix, flash LayerNorm prefix)] for the half-block shift described above."""
    pairs = [('h.0.ln_1', 'transformer.ln_0'), ('ln_f', 'transformer.layers.%d.norm2' % (n_layer - 1))]
    for d in range(n_layer):
        pairs.append(('h.%d.ln_2' % d, 'transformer.layers.%d.norm1' % d))
        if d > 0:
            pairs.append(('h.%d.ln_1' % d, 'transformer.layers.%d.norm2' % (d - 1)))
    return pairs


def remap_state_dict_gpt2(state_dict, config):
    """HF `GPT2Model.state_dict()` (keys `wte.weight`, `wpe.weight`, `h.<d>...`, `ln_f...`) -> the state dict
    of a flash `GPTLMHeadModel` (`transformer.*` + tied `lm_head.weight`).  Keys the table does not know are
    passed through unchanged, as upstream's regex renames do."""
    src = dict(state_dict)
    out = OrderedDict()
    n_layer = config.num_hidden_layers
    wte = src.pop('wte.weight')
    wte = F.pad(wte, (0, 0, 0, config.vocab_size - wte.shape[0]))   # vocab padded to a multiple of 8, say
    out['transformer.embeddings.word_embeddings.weight'] = wte
    out['lm_head.weight'] = wte
    for key in [k for k in src if k.startswith('wpe.')]:
        out['transformer.embeddings.position_embeddings.' + key[len('wpe.'):]] = src.pop(key)
    for hf, flash in _norm_pairs(n_layer):
        for leaf in ('weight', 'bias'):
            out['%s.%s' % (flash, leaf)] = src.pop('%s.%s' % (hf, leaf))
    for d in range(n_layer):
        # causal-mask buffers, not weights (upstream :396 pops `attn.bias` unconditionally; recent `transformers`
        # register both as non-persistent, so their state dicts no longer carry them)
        src.pop('h.%d.attn.bias' % d, None)
        src.pop('h.%d.attn.masked_bias' % d, None)
        for hf, flash, transpose in _PER_LAYER:
            value = src.pop('h.%d.%s' % (d, hf))
            out['transformer.layers.%d.%s' % (d, flash)] = value.t() if transpose else value
    out.update(src)
    return out


def remap_state_dict_flash(state_dict, config):
    """Flash `GPTLMHeadModel.state_dict()` -> HF `GPT2LMHeadModel` naming (`transformer.wte.weight`, ...,
    `lm_head.weight`); the embedding keeps its padded rows, as upstream (demo_convert.py:31-35).  The mask
    buffers HF expects (`.attn.bias`, `.attn.masked_bias`) are not produced -- upstream copies them from a
    freshly built HF model (:194-199)."""
    src = dict(state_dict)
    out = OrderedDict()
    n_layer = config.num_hidden_layers
    wte = src.pop('transformer.embeddings.word_embeddings.weight')
    src.pop('lm_head.weight')
    for key in [k for k in src if k.startswith('transformer.embeddings.position_embeddings.')]:
        out['transformer.wpe.' + key[len('transformer.embeddings.position_embeddings.'):]] = src.pop(key)
    out['transformer.wte.weight'] = wte
    for hf, flash in _norm_pairs(n_layer):
        for leaf in ('weight', 'bias'):
            out['transformer.%s.%s' % (hf, leaf)] = src.pop('%s.%s' % (flash, leaf))
    for d in range(n_layer):
        for hf, flash, transpose in _PER_LAYER:
            value = src.pop('transformer.layers.%d.%s' % (d, flash))
            out['transformer.h.%d.%s' % (d, hf)] = value.t() if transpose else value
    for key, value in src.items():               # anything else keeps its name under `transformer.`
        out['transformer.' + key] = value
    out['lm_head.weight'] = wte
    return out


def load_non_optimized_model(model, device=None):
    """Rebuild `model` with every `fused*` / `*flash*` config switch off and the same weights: the reference's
    way to run on a device without its native kernels (demo_convert.py:7-20; there it also moves the model to
    'cuda', here the device is kept unless one is given).  The result runs the eager op sequence on any
    device -- it is the model the parity tests compare the HIP path against."""
    config = model.config
    for k in list(vars(config)):
        if 'fused' in k or 'flash' in k:
            setattr(config, k, False)
    new = type(model)(config)
    new.load_state_dict(model.state_dict())
    if device is None:
        device = next(model.parameters()).device
    return new.to(device)


def gpt2_trunk_state_dict(hf_state_dict, config):
    """What `BackpackModel.from_pretrained` loads into its `gpt2_model` (a `GPTModel`): the remapped dict
    without the LM head and without the `transformer.` prefix.  Accepts both `GPT2Model` and
    `GPT2LMHeadModel` checkpoints (the latter prefix every key with `transformer.`)."""
    if any(k.startswith('transformer.') for k in hf_state_dict):
        hf_state_dict = {re.sub(r'^transformer\.', '', k): v for k, v in hf_state_dict.items()
                         if k != 'lm_head.weight'}
    remapped = remap_state_dict_gpt2(hf_state_dict, config)
    remapped.pop('lm_head.weight')
    return OrderedDict((re.sub(r'^transformer\.', '', k), v) for k, v in remapped.items())

"""GPT-2 trunk -- mirror of the reference's flash_attn/models/gpt.py for the serial path:
`create_mixer_cls` (:44-69, incl. the 1/(layer_idx+1) softmax scale), `create_mlp_cls` (:72-108),
`create_block` (:111-122), `GPTModel` (:175-246), `GPTLMHeadModel` (:249-282).
Linear layers / LayerNorm / embeddings are torch ops on ROCm (hipBLASLt); every layer's attention
is one launch of the HIP flash kernel when `config.use_flash_attn` is set."""
import math
from collections import namedtuple
from functools import partial

import torch
import torch.nn as nn
import torch.nn.functional as F
from transformers import GPT2Config

from flash_attn.modules.block import Block
from flash_attn.modules.embedding import GPT2Embeddings
from flash_attn.modules.mha import MHA
from flash_attn.modules.mlp import FusedDenseGeluDense, Mlp
from flash_attn.ops.layer_norm import dropout_add_layer_norm
from flash_attn.utils.pretrained import state_dict_from_pretrained
from flash_attn.utils.hf_convert import gpt2_trunk_state_dict, remap_state_dict_gpt2


def create_mixer_cls(config, layer_idx=None, process_group=None, device=None, dtype=None):
    assert process_group is None, 'tensor parallelism is out of scope for the Backpack path'
    head_dim = getattr(config, 'head_dim', config.hidden_size // config.num_attention_heads)
    softmax_scale = 1.0 if not config.scale_attn_weights else head_dim ** (-0.5)
    if config.scale_attn_by_inverse_layer_idx:
        assert layer_idx is not None
        softmax_scale /= float(layer_idx + 1)
    return partial(MHA, num_heads=config.num_attention_heads, dropout=config.attn_pdrop,
                   softmax_scale=softmax_scale, causal=True, layer_idx=layer_idx,
                   use_flash_attn=getattr(config, 'use_flash_attn', False),
                   fused_bias_fc=getattr(config, 'fused_bias_fc', False),
                   device=device, dtype=dtype)


def _activation(config):
    assert config.activation_function in ('gelu', 'gelu_new', 'gelu_fast')
    approximate = 'tanh' if config.activation_function in ('gelu_new', 'gelu_fast') else 'none'
    return partial(F.gelu, approximate=approximate)


def create_mlp_cls(config, layer_idx=None, process_group=None, device=None, dtype=None):
    assert process_group is None
    inner_dim = config.n_inner if config.n_inner is not None else 4 * config.hidden_size
    if getattr(config, 'fused_dense_gelu_dense', False):
        assert config.activation_function in ('gelu_new', 'gelu_fast'), \
            'fused_dense_gelu_dense only supports approximate gelu'          # reference gpt.py:75-78
        return partial(FusedDenseGeluDense, hidden_features=inner_dim, device=device, dtype=dtype)
    return partial(Mlp, hidden_features=inner_dim, activation=_activation(config), device=device,
                   dtype=dtype)


def create_block(config, layer_idx=None, process_group=None, device=None, dtype=None):
    mixer_cls = create_mixer_cls(config, layer_idx, process_group=process_group, device=device, dtype=dtype)
    mlp_cls = create_mlp_cls(config, layer_idx, process_group=process_group, device=device, dtype=dtype)
    norm_cls = partial(nn.LayerNorm, eps=config.layer_norm_epsilon, device=device, dtype=dtype)
    block = Block(config.hidden_size, mixer_cls, mlp_cls, norm_cls=norm_cls, prenorm=True,
                  resid_dropout=config.resid_pdrop,
                  fused_dropout_add_ln=getattr(config, 'fused_dropout_add_ln', False))
    block.layer_idx = layer_idx
    return block


def _init_weights(module, n_layer, initializer_range=0.02, rescale_prenorm_residual=True):
    """GPT-2 initialisation (reference gpt.py:153-172): N(0, std) for Linear / Embedding weights,
    zero biases, and N(0, std / sqrt(2 * n_layer)) for the projections that write into the
    residual stream (out_proj, fc2)."""
    if isinstance(module, nn.Linear):
        nn.init.normal_(module.weight, std=initializer_range)
        if module.bias is not None:
            nn.init.zeros_(module.bias)
    elif isinstance(module, nn.Embedding):
        nn.init.normal_(module.weight, std=initializer_range)
    if rescale_prenorm_residual:
        for name, p in module.named_parameters():
            if name in ('out_proj.weight', 'fc2.weight'):
                nn.init.normal_(p, mean=0.0, std=initializer_range / math.sqrt(2 * n_layer))


def _pad_vocab(config):
    multiple = getattr(config, 'pad_vocab_size_multiple', 1)
    if config.vocab_size % multiple != 0:
        config.vocab_size += multiple - config.vocab_size % multiple
    return multiple


class GPTPreTrainedModel(nn.Module):

    def __init__(self, config, *inputs, **kwargs):
        super().__init__()
        if not isinstance(config, GPT2Config):
            raise ValueError('config must be a transformers.GPT2Config, got %r' % type(config))
        self.config = config

    @classmethod
    def from_pretrained(cls, model_name, config, *inputs, state_dict=None, **kwargs):
        """Build the model and load Hugging Face GPT-2 weights into it (reference gpt.py:140-151).
        `state_dict`: an already loaded HF state dict (skips the lookup of `model_name`).  A `GPTLMHeadModel`
        takes the remapped dict as it is; a bare `GPTModel` takes it without the `transformer.` prefix and
        the head (upstream's strict load only works for the former)."""
        model = cls(config, *inputs, **kwargs)
        hf = state_dict if state_dict is not None else state_dict_from_pretrained(model_name)
        if hasattr(model, 'lm_head'):
            hf = {(k[len('transformer.'):] if k.startswith('transformer.') else k): v for k, v in hf.items()
                  if k != 'lm_head.weight'}
            model.load_state_dict(remap_state_dict_gpt2(hf, config))
            model.tie_weights()
        else:
            model.load_state_dict(gpt2_trunk_state_dict(hf, config))
        return model


class GPTModel(GPTPreTrainedModel):
    """embeddings -> fp32 residual stream -> ln_0 -> n_layer pre-norm blocks.  gpt.py:175-246."""

    def __init__(self, config: GPT2Config, process_group=None, device=None, dtype=None):
        super().__init__(config)
        assert process_group is None, 'tensor parallelism is out of scope for the Backpack path'
        factory_kwargs = {'device': device, 'dtype': dtype}
        self.process_group = None
        self.pad_vocab_size_multiple = _pad_vocab(config)
        self.embeddings = GPT2Embeddings(config.hidden_size, config.vocab_size,
                                         config.max_position_embeddings, **factory_kwargs)
        self.emb_drop = nn.Dropout(config.embd_pdrop)
        self.fused_dropout_add_ln = getattr(config, 'fused_dropout_add_ln', False)
        self.ln_0 = nn.LayerNorm(config.hidden_size, eps=config.layer_norm_epsilon, **factory_kwargs)
        self.layers = nn.ModuleList([create_block(config, layer_idx=i, **factory_kwargs)
                                     for i in range(config.num_hidden_layers)])
        self.apply(partial(_init_weights, n_layer=config.num_hidden_layers,
                           initializer_range=config.initializer_range))

    def forward(self, input_ids, position_ids=None, inference_params=None):
        assert inference_params is None, 'KV-cache decoding is out of scope'
        hidden = self.embeddings(input_ids, position_ids=position_ids)
        if self.fused_dropout_add_ln:
            # one HIP launch: residual (fp32) = hidden, hidden = LN(residual)   (reference gpt.py:236-240)
            hidden, residual = dropout_add_layer_norm(
                hidden, None, self.ln_0.weight, self.ln_0.bias,
                self.emb_drop.p if self.training else 0.0, self.ln_0.eps, prenorm=True,
                residual_in_fp32=True)
        else:
            residual = self.emb_drop(hidden).float()   # residual stream stays fp32 (gpt.py:231-234)
            hidden = self.ln_0(residual.to(dtype=self.ln_0.weight.dtype))
        for layer in self.layers:
            hidden, residual = layer(hidden, residual)
        return hidden


class GPTLMHeadModel(GPTPreTrainedModel):
    """GPTModel + tied LM head.  gpt.py:249-282."""

    def __init__(self, config: GPT2Config, process_group=None, device=None, dtype=None):
        super().__init__(config)
        self.process_group = None
        self.transformer = GPTModel(config, device=device, dtype=dtype)
        self.lm_head = nn.Linear(config.n_embd, config.vocab_size, bias=False, device=device, dtype=dtype)
        self.apply(partial(_init_weights, n_layer=config.num_hidden_layers,
                           initializer_range=config.initializer_range))
        self.tie_weights()

    def tie_weights(self):
        self.lm_head.weight = self.transformer.embeddings.word_embeddings.weight

    def forward(self, input_ids, position_ids=None, inference_params=None):
        hidden = self.transformer(input_ids, position_ids=position_ids)
        CausalLMOutput = namedtuple('CausalLMOutput', ['logits'])
        return CausalLMOutput(logits=self.lm_head(hidden))

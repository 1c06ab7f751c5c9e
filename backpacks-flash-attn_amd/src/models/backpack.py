"""Backpack language model (Hewitt et al., ACL 2023) -- MI355X-native mirror of the reference's
training/src/models/backpack.py: same classes, constructor arguments, attribute names and
state-dict keys (BackpackConfig :146-154, ContextSelfAttn :94-122, BackpackContentModule :207-276,
BackpackModel :278-314, BackpackLMHeadModel :318-351).

What differs is HOW the forward runs when `config.use_flash_attn` is set (the reference's own
switch for its native path, flash_attn/models/gpt.py:56):
  * every trunk layer's attention is one launch of the HIP flash kernel (bp_flash_fwd);
  * `BackpackModel.forward` never materialises the (B,k,S,S) sense weights: the causal softmax and
    `torch.sum(alpha @ content, dim=1)` (reference :305,:313) are one fused HIP contraction
    (bp_sense_mix);
  * `ContextSelfAttn.forward` still returns alpha (B,k,S,S) for the callers that edit it
    (training/src/models/intervened_models.py:78-101), produced by bp_sense_alpha.
With `use_flash_attn=False` the modules run the reference's eager op sequence (any device/dtype) --
its "non-optimized" mode (training/demo_convert.py:7-20).  The flag alone decides; nothing falls
back silently, and the HIP path raises if its inputs are not 16-bit CUDA tensors.
"""
import math
import warnings
from collections import namedtuple
from functools import partial

import torch
import torch.nn as nn
import torch.nn.functional as F
from transformers import GPT2Config

import bp_hip
from flash_attn.models.gpt import GPTModel, GPTPreTrainedModel, _activation, _init_weights, _pad_vocab
from flash_attn.modules.block import Block
from flash_attn.modules.mlp import FusedDenseGeluDense, Mlp
from flash_attn.ops.fused_dense import FusedDense, fused_dense_func
from flash_attn.ops.layer_norm import dropout_add_layer_norm
from flash_attn.utils.pretrained import state_dict_from_pretrained
from src.utils.generation import GenerationMixin
from src.utils.hf_convert import gpt2_trunk_state_dict, remap_state_dict_gpt2  # noqa: F401  (re-exported, reference :354)


class BackpackConfig(GPT2Config):

    def __init__(self, num_content_vectors=16, **kwargs):
        self.num_content_vectors = num_content_vectors
        super().__init__(**kwargs)


def create_content_mlp_cls(config, layer_idx=None, expand_out=False, process_group=None,
                           device=None, dtype=None):
    """MLP d -> inner -> (k*d if expand_out else d); inner = d when `shrink_final_inner`
    (reference :53-92)."""
    assert process_group is None
    inner_dim = config.n_inner if config.n_inner is not None else 4 * config.hidden_size
    if getattr(config, 'shrink_final_inner', None):
        inner_dim = config.hidden_size
    outer_dim = config.num_content_vectors * config.hidden_size if expand_out else config.hidden_size
    if getattr(config, 'fused_dense_gelu_dense', False):
        assert config.activation_function in ('gelu_new', 'gelu_fast')      # reference :60-62
        return partial(FusedDenseGeluDense, hidden_features=inner_dim, out_features=outer_dim,
                       device=device, dtype=dtype)
    return partial(Mlp, hidden_features=inner_dim, out_features=outer_dim,
                   activation=_activation(config), device=device, dtype=dtype)


class Identity(nn.Identity):
    """Mixer of the content model's block: passes its input through (reference :125-128)."""

    def forward(self, x, **kwargs):
        return x


def create_nomix_block(config, expand_out=False, layer_idx=None, process_group=None, device=None,
                       dtype=None):
    mlp_cls = create_content_mlp_cls(config, layer_idx, expand_out, device=device, dtype=dtype)
    norm_cls = partial(nn.LayerNorm, eps=config.layer_norm_epsilon, device=device, dtype=dtype)
    block = Block(config.hidden_size, Identity, mlp_cls, norm_cls=norm_cls, prenorm=True,
                  resid_dropout=config.resid_pdrop,
                  fused_dropout_add_ln=getattr(config, 'fused_dropout_add_ln', False))
    block.layer_idx = layer_idx
    return block


class ContextSelfAttn(nn.Module):
    """num_content_vectors causal attention maps per pair of positions (reference :94-122).

    forward(encoded (B,S,d)) -> alpha (B,k,S,S) in the activation dtype.
    `project(encoded)` -> qk (B,S,2,k,d/k) is the half that `BackpackModel` feeds to the fused mix.
    """

    def __init__(self, num_content_vectors, embed_dim, device=None, dtype=None, use_hip=False):
        super().__init__()
        # FusedDense as in the reference (:102): nn.Linear's parameters, bias gradient by bp_column_sum in training
        self.Wqkv = FusedDense(embed_dim, 2 * embed_dim, device=device, dtype=dtype)
        self.num_content_vectors = num_content_vectors
        self.softmax_scale = None
        self.use_hip = use_hip
        # Sense widths d_k = d / k: up to 128 (after widening to a multiple of 8) the LDS-DMA kernels and the fused backward;
        # 129 ... 640 -- the reference's few-sense ablations, backpack-mini-flash-vecs-4.yaml: d_k = 160, vecs-1: 640 -- the
        # wide kernels of csrc/sense_wide.hip (forward fused as well; backward through the alpha-rebuilding route, alpha
        # being small with few senses: k S^2 per sample).  Beyond 640 (no reference config) the reference's own eager op
        # sequence runs on the GPU, with one warning; the trunk keeps its kernels either way.
        self.fused = bool(use_hip) and -(-(embed_dim // num_content_vectors) // 8) * 8 <= bp_hip.SENSE_MAX_DK
        if use_hip and not self.fused:
            warnings.warn(f'Backpack: {num_content_vectors} sense(s) at width {embed_dim} give d_k = '
                          f'{embed_dim // num_content_vectors} > {bp_hip.SENSE_MAX_DK}: the sense weights and their combination '
                          'run as the eager op sequence on the GPU (the HIP sense kernels cover d_k <= '
                          f'{bp_hip.SENSE_MAX_DK}); the trunk keeps its HIP kernels')

    def project(self, encoded):
        """encoded (B,S,d) -> qk (B,S,2,k,d_k).  On the HIP path a d_k that is not a multiple of 8 (the Mini
        k=64 ablation: d_k = 10, training/configs/experiment/owt/backpack-mini-flash-vecs-64.yaml) is widened
        to the next multiple with ZERO columns: the LDS-DMA kernels move 16-byte chunks, zeros add nothing to
        q.k, and padding the projection's rows (1.3 M weights) instead of its output (2.6 KB per token, three
        consumers) leaves autograd and the state-dict keys untouched.  Pair it with `scale()`."""
        b, s, d = encoded.shape
        k = self.num_content_vectors
        dk = d // k
        pad = (-dk) % 8 if self.fused else 0
        if pad == 0:
            return self.Wqkv(encoded).reshape(b, s, 2, k, dk)
        w, bias = self._padded_projection(k, dk, pad, d)
        return fused_dense_func(encoded, w, bias).view(b, s, 2, k, dk + pad)

    def _padded_projection(self, k, dk, pad, d):
        """`Wqkv` with `pad` zero rows appended to every sense's q and k block.  Under autograd the pad is part of the
        graph (gradients flow back to the unpadded parameter); without it (inference) the padded copy is kept until the
        parameter changes -- `_version` moves on every in-place update, `data_ptr` on a reload / `.to()`."""
        weight, bias = self.Wqkv.weight, self.Wqkv.bias

        def pad_now():
            return (F.pad(weight.view(2, k, dk, d), (0, 0, 0, pad)).view(2 * k * (dk + pad), d),
                    F.pad(bias.view(2, k, dk), (0, pad)).view(-1))

        if torch.is_grad_enabled() and (weight.requires_grad or bias.requires_grad):
            return pad_now()
        # While a HIP graph is being captured the pad kernels must be PART of the graph: a cached copy made outside
        # (e.g. by GraphedForward's warm-up forwards) would be what every replay reads, also after the weights were
        # updated in place.  Inference tensors carry no version counter: nothing to key a cache on, pad per call.
        if (weight.is_cuda and torch.cuda.is_current_stream_capturing()) or weight.is_inference() \
                or bias.is_inference():
            with torch.no_grad():
                return pad_now()
        key = (weight.data_ptr(), weight._version, bias.data_ptr(), bias._version, weight.dtype, weight.device)
        cached = getattr(self, '_padded_cache', None)
        if cached is None or cached[0] != key:
            with torch.no_grad():
                cached = (key,) + pad_now()
            self._padded_cache = cached
        return cached[1], cached[2]

    def scale(self):
        """softmax scale of the TRUE sense width d/k (reference :117), whatever `project` padded to."""
        d_k = self.Wqkv.in_features // self.num_content_vectors
        return self.softmax_scale or 1.0 / math.sqrt(d_k)

    def forward(self, encoded):
        qk = self.project(encoded)
        if self.fused:
            return bp_hip.sense_alpha_autograd(qk, self.scale())
        seqlen = qk.shape[1]
        q, k = qk.unbind(dim=2)
        scale = self.softmax_scale or 1.0 / math.sqrt(q.shape[-1])
        scores = torch.einsum('bthd,bshd->bhts', q, k * scale)
        mask = torch.triu(torch.full((seqlen, seqlen), -10000.0, device=scores.device), 1)
        scores = scores + mask.to(dtype=scores.dtype)
        return torch.softmax(scores, dim=-1, dtype=q.dtype)


class BackpackPreTrainedModel(nn.Module):

    def __init__(self, config, *inputs, **kwargs):
        super().__init__()
        if not isinstance(config, BackpackConfig):
            raise ValueError('config must be a BackpackConfig, got %r' % type(config))
        self.config = config

    @classmethod
    def from_pretrained(cls, model_name, config, *inputs, state_dict=None, **kwargs):
        """Build the model and initialise its GPT-2 TRUNK from Hugging Face GPT-2 weights; the content
        model and the sense attention keep their fresh initialisation (reference :172-183).  Upstream hands
        the `transformer.`-prefixed remapped dict to `model.gpt2_model`, whose keys carry no such prefix (and
        which only `BackpackModel` has), so its strict load cannot succeed as written; here the prefix and
        the LM head are dropped first and `BackpackLMHeadModel` is handled too (trunk = transformer.gpt2_model,
        whose word embedding is the tied head)."""
        return _load_gpt2_trunk(cls(config, *inputs, **kwargs), model_name, config, state_dict)


def _load_gpt2_trunk(model, model_name, config, state_dict=None):
    hf = state_dict if state_dict is not None else state_dict_from_pretrained(model_name)
    trunk = model.gpt2_model if hasattr(model, 'gpt2_model') else model.transformer.gpt2_model
    trunk.load_state_dict(gpt2_trunk_state_dict(hf, config))
    if hasattr(model, 'tie_weights'):
        model.tie_weights()
    return model


class BackpackContentModule(nn.Module):
    """Sense vectors C(x): word embedding (no positions) -> LN -> one no-mix block -> final MLP
    d -> k*d, returned as the (B,k,S,d) view of the contiguous (B,S,k*d) buffer (reference :251-276).
    That buffer layout is the input contract of bp_sense_mix."""

    def __init__(self, config, num_content_vectors, embeddings, process_group=None, device=None,
                 dtype=None):
        super().__init__()
        assert process_group is None
        factory_kwargs = {'device': device, 'dtype': dtype}
        self.num_content_vectors = num_content_vectors
        self.embeddings = embeddings
        self.process_group = None
        self.n_embd = config.n_embd
        self.fused_dropout_add_ln = getattr(config, 'fused_dropout_add_ln', False)
        self.ln_0 = nn.LayerNorm(config.hidden_size, eps=config.layer_norm_epsilon, **factory_kwargs)
        n_layers = 1
        self.layers = nn.ModuleList([create_nomix_block(config, layer_idx=i, expand_out=False,
                                                        **factory_kwargs) for i in range(n_layers)])
        self.final_mlp = create_content_mlp_cls(config, layer_idx=n_layers + 1, expand_out=True,
                                                **factory_kwargs)(config.n_embd)
        self.emb_drop = nn.Dropout(config.embd_pdrop)
        self.apply(partial(_init_weights, n_layer=n_layers, initializer_range=config.initializer_range))

    def forward(self, input_ids, position_ids=None, inference_params=None):
        hidden = self.embeddings.word_embeddings(input_ids)          # no positions (reference :258)
        if self.fused_dropout_add_ln:
            hidden, residual = dropout_add_layer_norm(                # reference :263-268
                hidden, None, self.ln_0.weight, self.ln_0.bias,
                self.emb_drop.p if self.training else 0.0, self.ln_0.eps, prenorm=True,
                residual_in_fp32=True)
        else:
            residual = self.emb_drop(hidden).float()
            hidden = self.ln_0(residual.to(dtype=self.ln_0.weight.dtype))
        for layer in self.layers:
            hidden, residual = layer(hidden, residual)
        hidden = self.final_mlp(hidden)                               # (B, S, k*d)
        bs, s, _ = hidden.shape
        return hidden.reshape(bs, s, self.num_content_vectors, self.n_embd).transpose(1, 2)


class BackpackModel(GPTPreTrainedModel):

    def __init__(self, config: BackpackConfig, process_group=None, device=None, dtype=None):
        super().__init__(config)
        assert process_group is None, 'tensor parallelism is out of scope for the Backpack path'
        factory_kwargs = {'device': device, 'dtype': dtype}
        self.process_group = None
        assert config.activation_function in ('gelu', 'gelu_new', 'gelu_fast')
        self.pad_vocab_size_multiple = _pad_vocab(config)
        self.use_hip = bool(getattr(config, 'use_flash_attn', False))
        self.num_content_vectors = config.num_content_vectors
        self.dedup_content = bool(getattr(config, 'dedup_content', True))
        self.sense_table_mode = getattr(config, 'sense_table', 'cached')      # 'cached' | 'batch' | 'off', see below
        assert self.sense_table_mode in ('cached', 'batch', 'off')
        # 'batch' mode pays from about 0.65 x vocab positions up with uniformly random ids (Small, S = 1024: equal at 32 k
        # positions, -6 % at 49 k, -12 % at 100 k; profiles/r05_a_content_modes_small.jsonl); text repeats tokens and pays earlier
        self.dedup_min_positions = int(getattr(config, 'dedup_min_positions', config.vocab_size))
        self._sense_table = None
        self.gpt2_model = GPTModel(config, **factory_kwargs)
        self.content_model = BackpackContentModule(config, self.num_content_vectors,
                                                   self.gpt2_model.embeddings, **factory_kwargs)
        self.embeddings = self.gpt2_model.embeddings   # shared with the contextualisation model
        self.contextualization_attn = ContextSelfAttn(self.num_content_vectors, config.n_embd,
                                                      use_hip=self.use_hip, **factory_kwargs)
        # False beyond d_k = 640 and off the HIP path: eager sense weights + combination
        self.fused_senses = self.contextualization_attn.fused

    @classmethod
    def from_pretrained(cls, model_name, config, *inputs, state_dict=None, **kwargs):
        """HF GPT-2 weights into the trunk only (see BackpackPreTrainedModel.from_pretrained)."""
        return _load_gpt2_trunk(cls(config, *inputs, **kwargs), model_name, config, state_dict)

    # ---- inference: the content network once per TOKEN instead of once per position ---------------------------------
    # The sense vectors C_l(x_j) are a function of the token alone (no positions, reference :258; an Identity mixer,
    # :130-143; everything else per-token LayerNorm / MLP).  In inference (HIP path, eval(), no autograd graph) the
    # content network therefore needs one row per token id, and the mix kernel reads the rows through an index
    # (bp_sense_mix_gather).  Two tables, chosen by `config.sense_table` / `self.sense_table_mode`:
    #   'cached' (default)  the WHOLE vocabulary, built once per weight version (`sense_table()`); index = the ids.
    #                       Static shapes, no host sync: every batch size, graph capture, generate(cg=True).
    #   'batch'             the distinct ids of THIS batch (torch.unique, a host sync), rebuilt every forward; taken from
    #                       `dedup_min_positions` positions up (default: one per vocabulary entry; below, per position is as fast).
    #   'off'               every position through the content network (the reference's order; also `dedup_content=False`).
    # Mathematically identical to the per-position order; bits can differ where the BLAS GEMMs round a row differently
    # when the row count changes.  Training always runs per position (dropout inside the content network, autograd).

    def _token_table_allowed(self, input_ids):
        return (self.dedup_content and self.sense_table_mode != 'off' and self.fused_senses and not self.training
                and not torch.is_grad_enabled() and input_ids.is_cuda)

    def _dedup_applies(self, input_ids):
        """True when this forward builds the table of the batch's distinct tokens ('batch' mode, or 'cached' mode whose
        table cannot be kept -- inference-mode parameters).  Never while a stream is being captured: torch.unique has a
        data-dependent shape."""
        if not self._token_table_allowed(input_ids) or torch.cuda.is_current_stream_capturing():
            return False
        if self.sense_table_mode == 'cached' and self.sense_table() is not None:
            return False
        return input_ids.numel() >= self.dedup_min_positions

    def _sense_table_key(self, fingerprint=False):
        params = list(self.content_model.parameters())
        if any(p.is_inference() for p in params):
            return None                       # no version counter to key a cache on
        key = tuple((p.data_ptr(), p._version, p.dtype, p.device) for p in params)
        if fingerprint:
            # `p.data.copy_()` / `p.data.mul_()` (the reference's own EMA swap, training/src/utils/ema.py:121,165) write the
            # storage WITHOUT moving `p._version`: only the values themselves tell.  One fp32 sum per parameter, one
            # device-to-host copy (a synchronisation: callers decide when that is affordable, `_verify_applies`).
            with torch.no_grad():
                sums = torch.stack([torch.sum(p.detach(), dtype=torch.float32) for p in params])
            key = key + (tuple(sums.tolist()),)
        return key

    def _verify_applies(self, input_ids):
        """Whether this forward also checks the cached table against the parameter VALUES (`config.sense_table_verify`:
        True / False / 'auto').  'auto' (default): forwards of at least `sense_table_verify_min_positions` positions
        (16 384), where one ~40 us reduction and a host synchronisation are noise -- bulk evaluation, which is where
        weights get swapped through `.data` -- and never while a stream is capturing or for small latency-bound
        forwards (decoding), which rely on `invalidate_sense_table()` after such an update."""
        mode = getattr(self.config, 'sense_table_verify', 'auto')
        if mode in (False, None, 'off') or not input_ids.is_cuda or torch.cuda.is_current_stream_capturing():
            return False
        if mode is True or mode == 'on':
            return True
        return input_ids.numel() >= int(getattr(self.config, 'sense_table_verify_min_positions', 16384))

    def invalidate_sense_table(self):
        """Mark the cached whole-vocabulary sense table stale: the next eval forward (or `refresh_inference_caches()`)
        rebuilds it IN PLACE.  Needed after parameter updates that bypass autograd's version counter -- `p.data.copy_()`,
        `p.data.mul_()`, writes through `untyped_storage()` -- when the forward is too small for the automatic value check
        (see `_verify_applies`).  In-place updates through the parameter itself, `load_state_dict`, `.to()` are seen
        without it."""
        if self._sense_table is not None:
            self._sense_table = (None, self._sense_table[1], None)

    def pin_sense_table(self, pinned=True):
        """Keep the table's storage across `.train()` (a captured HIP graph holds its address: bp_hip.GraphedForward pins
        it); an unpinned table is dropped when training starts."""
        self._sense_table_pinned = bool(pinned)

    def sense_table(self, verify=False):
        """(vocab rows, k, d): the content network's output for EVERY row of the word embedding, kept until a parameter of
        the content model changes (`_version` moves on every in-place update, `data_ptr` on a reload / `.to()`; with
        `verify` also the parameter values, which catches `.data` updates), invalidated by `.train()`.  A refresh writes
        into the SAME storage, so a captured HIP graph that reads the table sees the new rows (bp_hip.GraphedForward
        pins the storage and calls `refresh_inference_caches()` in front of every replay; `generate(cg=True)` captures
        and replays inside one call, weights cannot change in between).  Returns None when nothing can be kept
        (inference-mode parameters), when the table would have to be built while a stream is capturing, or when the
        build runs out of memory (one warning; the forward then takes the per-position order).  1.2 GB at
        Backpack-Small, 4.1 GB at Mini k = 64; ~5 ms to build."""
        key = self._sense_table_key()
        if key is None:
            return None
        cached = self._sense_table
        if cached is not None and cached[0] == key:
            if not verify:
                return cached[1]
            full = self._sense_table_key(fingerprint=True)
            if cached[2] is not None and cached[2] == full:
                return cached[1]
            # (a table built without a fingerprint cannot vouch for the values: rebuild once, then it can)
        weight = self.embeddings.word_embeddings.weight
        if weight.is_cuda and torch.cuda.is_current_stream_capturing():
            return None
        full = self._sense_table_key(fingerprint=True) if verify else None
        with torch.inference_mode(False), torch.no_grad():
            was_training = self.content_model.training
            self.content_model.eval()                 # (also the embedding module it shares with the trunk)
            try:
                ids = torch.arange(weight.shape[0], device=weight.device).unsqueeze(0)
                rows = self.content_model(ids)[0].transpose(0, 1)      # (V,k,d): the (1,V,k*d) block as it lies
                if cached is not None and cached[1].shape == rows.shape and cached[1].dtype == rows.dtype \
                        and cached[1].device == rows.device:
                    cached[1].copy_(rows)
                    rows = cached[1]
            except torch.OutOfMemoryError:
                if not getattr(self, '_sense_table_oom_warned', False):
                    warnings.warn('Backpack: not enough memory for the whole-vocabulary sense table; the forward runs '
                                  'the content network per position')
                    self._sense_table_oom_warned = True
                return None
            finally:
                self.content_model.train(was_training)
        self._sense_table = (key, rows, full)
        return rows

    def refresh_inference_caches(self):
        """Bring the cached sense table up to date with the weights (a no-op when it is); call in front of replaying a
        captured graph of this model after an in-place weight update (after a `.data` update: `invalidate_sense_table()`
        first)."""
        if self.sense_table_mode == 'cached' and self.dedup_content and self.fused_senses and not self.training:
            self.sense_table()

    def train(self, mode=True):
        if mode and self._sense_table is not None:
            # training runs per position; do not hold 1-4 GB of stale rows -- unless a captured graph reads this storage
            self._sense_table = (None, self._sense_table[1], None) if getattr(self, '_sense_table_pinned', False) else None
        return super().train(mode)

    def _table_of_unique_tokens(self, input_ids):
        """(rows (U, k, d) = the content network's output for the sorted distinct ids, index (B, S) of every position's row)"""
        uniq, inverse = torch.unique(input_ids, return_inverse=True)
        table = self.content_model(uniq.unsqueeze(0))                  # (1,k,U,d) view of one (1,U,k*d) block
        return table[0].transpose(0, 1), inverse                        # (U,k,d): the block as it lies, no copy

    def _content_of_unique_tokens(self, input_ids):
        rows, inverse = self._table_of_unique_tokens(input_ids)
        u, k, d = rows.shape
        content = torch.nn.functional.embedding(inverse, rows.reshape(u, k * d))   # (B,S,k*d): every position's row
        return content.view(*input_ids.shape, k, d).transpose(1, 2)    # (B,k,S,d) view, as content_model returns it

    def _mix_from_table(self, hidden, rows, index):
        qk = self.contextualization_attn.project(hidden)
        if bp_hip.sense_mix_gather_supported(qk, rows, index.shape[1]):
            # the mix kernel reads the table rows itself: no (B,S,k,d) content tensor at all
            return bp_hip.sense_mix_gather(qk, rows, index.to(torch.int32), self.contextualization_attn.scale())
        # shapes the gathering kernel does not take: torch gathers the rows into the (B,S,k,d) tensor the dense kernel
        # reads -- said once per model, with the limit that was hit (bp_hip.sense_mix_gather_limits)
        if not getattr(self, '_gather_fallback_said', False):
            self._gather_fallback_said = True
            warnings.warn('Backpack: the sense table is gathered by torch into a (B, S, k*d) tensor instead of inside the '
                          'mix kernel: ' + bp_hip.sense_mix_gather_limits(qk, rows, index.shape[1]))
        content = torch.nn.functional.embedding(index, rows.reshape(rows.shape[0], -1))
        return bp_hip.sense_mix(qk, content.view(*index.shape, *rows.shape[1:]), self.contextualization_attn.scale())

    # ---- the reference's order of operations (every position through the content network) at HBM-filling batches -----
    # The contraction is per sample (reference :297-314 has no cross-sample term), so without an autograd graph the
    # content network and the mix run over CHUNKS of samples: the (B,S,k*d) content tensor (25 MB per sample at Small,
    # 52 GB at B = 2048) exists for one chunk at a time.  Same kernels, same per-sample arithmetic; what can differ from
    # the whole-batch call is how the BLAS GEMMs round a row when their row count changes (as with the token tables).

    def _chunked_content_applies(self, input_ids):
        if not (self.fused_senses and input_ids.is_cuda and not torch.is_grad_enabled()) or input_ids.dim() != 2:
            return False
        return input_ids.shape[0] > self._content_chunk_samples(input_ids.shape[1])

    def _content_chunk_samples(self, seqlen):
        # default: 262144 positions per chunk (the content GEMMs are at their rate from ~128 k rows up; 6.4 GB of content
        # at Small); `config.content_chunk_positions` overrides, 0 / None = never chunk
        positions = getattr(self.config, 'content_chunk_positions', 262144)
        if not positions:
            return 1 << 62
        return max(1, int(positions) // max(int(seqlen), 1))

    def _mix_per_position_chunked(self, hidden, input_ids, position_ids, inference_params):
        qk = self.contextualization_attn.project(hidden)               # (B,S,2,k,d_k): 2 d per position, whole batch
        scale = self.contextualization_attn.scale()
        out = torch.empty(hidden.shape, dtype=hidden.dtype, device=hidden.device)
        step = self._content_chunk_samples(input_ids.shape[1])
        for b0 in range(0, input_ids.shape[0], step):
            sl = slice(b0, min(b0 + step, input_ids.shape[0]))
            pos = position_ids[sl] if position_ids is not None and position_ids.shape[0] == input_ids.shape[0] \
                else position_ids
            content = self.content_model(input_ids[sl], pos, inference_params)     # (c,k,S,d) view of (c,S,k*d)
            bp_hip.sense_mix(qk[sl], content.transpose(1, 2), scale, out=out[sl])
            del content
        return out

    def forward(self, input_ids, position_ids=None, inference_params=None):
        contextl_hidden_states = self.gpt2_model(input_ids, position_ids=position_ids,
                                                 inference_params=inference_params)
        if self._token_table_allowed(input_ids):
            if self.sense_table_mode == 'cached':
                rows = self.sense_table(verify=self._verify_applies(input_ids))
                if rows is not None:
                    return self._mix_from_table(contextl_hidden_states, rows, input_ids)
            if self._dedup_applies(input_ids):
                rows, inverse = self._table_of_unique_tokens(input_ids)
                return self._mix_from_table(contextl_hidden_states, rows, inverse)
        if self._chunked_content_applies(input_ids):
            return self._mix_per_position_chunked(contextl_hidden_states, input_ids, position_ids, inference_params)
        content = self.content_model(input_ids, position_ids, inference_params)   # (B,k,S,d) view
        if self.fused_senses:
            # fused: softmax_causal(q_l k_l^T) @ C_l summed over senses, alpha never stored
            qk = self.contextualization_attn.project(contextl_hidden_states)
            return bp_hip.sense_mix_autograd(qk, content.transpose(1, 2),
                                             self.contextualization_attn.scale())
        contextualization = self.contextualization_attn(contextl_hidden_states)   # (B,k,S,S)
        return _combine_senses(contextualization, content)                         # (B,S,d)


def _combine_senses(contextualization, content):
    """The reference's eager sense combination, torch.sum(contextualization @ content, dim=1)
    (training/src/models/backpack.py:313).  One exception: 16-bit tensors on the GPU with gradients enabled take the
    einsum restatement (identical to 1.2e-7 in fp32; here the sum over the senses is
    accumulated in fp32 inside one GEMM instead of being a 16-bit sum of 16-bit products).  On ROCm 7.2 the BLAS
    backward of the batched `contextualization @ content` in bf16 takes a memory fault at Backpack-Small dimensions
    (k 16, S 1024, d 768 -- with or without a contiguous content, with or without a materialised gradient of the
    sum; fp32 and the einsum form do not: scripts/debug/r03_blas_fault.py, profiles/r03_h_blas_fault.txt).  The HIP
    path (use_flash_attn) never touches any of this."""
    if contextualization.is_cuda and contextualization.dtype != torch.float32 and torch.is_grad_enabled() \
            and (contextualization.requires_grad or content.requires_grad):
        return torch.einsum('blts,blsd->btd', contextualization, content)
    return torch.sum(contextualization @ content, dim=1)


class BackpackLMHeadModel(BackpackPreTrainedModel, GenerationMixin):

    def __init__(self, config: BackpackConfig, process_group=None, device=None, dtype=None):
        super().__init__(config)
        assert process_group is None
        self.process_group = None
        self.transformer = BackpackModel(config, device=device, dtype=dtype)
        self.lm_head = nn.Linear(config.n_embd, config.vocab_size, bias=False, device=device, dtype=dtype)
        self.apply(partial(_init_weights, n_layer=config.num_hidden_layers,
                           initializer_range=config.initializer_range))
        self.tie_weights()

    def tie_weights(self):
        # tied with the word embeddings of both the trunk and the content model (reference :339-340)
        self.lm_head.weight = self.transformer.embeddings.word_embeddings.weight

    def forward(self, input_ids, position_ids=None, inference_params=None, logits_out=None):
        """`logits_out` (not in the reference's signature, :342-351): a caller-owned (B, S, vocab) buffer the LM head
        writes into -- inference at HBM-filling batches, where the logits are most of the memory (103 MB per sample at
        Small / S = 1024) and a fresh allocation per step is what the caching allocator cannot re-place."""
        hidden_states = self.transformer(input_ids, position_ids=position_ids,
                                         inference_params=inference_params)
        CausalLMOutput = namedtuple('CausalLMOutput', ['logits'])
        if logits_out is None:
            return CausalLMOutput(logits=self.lm_head(hidden_states))
        if torch.is_grad_enabled() and (hidden_states.requires_grad or self.lm_head.weight.requires_grad):
            raise RuntimeError('logits_out is an inference-only argument (run under torch.no_grad())')
        want = hidden_states.shape[:-1] + (self.lm_head.weight.shape[0],)
        if tuple(logits_out.shape) != tuple(want) or logits_out.dtype != hidden_states.dtype \
                or not logits_out.is_contiguous():
            raise RuntimeError(f'logits_out must be a contiguous {tuple(want)} tensor of {hidden_states.dtype}')
        # `out=` bypasses autocast and module hooks: the operands must already agree (fp32 weights under bf16 autocast
        # would need a cast of the 39 M-entry matrix per call -- convert the model instead), and there is no bias to add
        if self.lm_head.bias is not None or self.lm_head.weight.dtype != hidden_states.dtype:
            raise RuntimeError(f'logits_out needs a bias-free lm_head whose weight has the activation dtype '
                               f'({self.lm_head.weight.dtype} vs {hidden_states.dtype}); call model.to(dtype) first')
        x2, o2 = hidden_states.reshape(-1, hidden_states.shape[-1]), logits_out.view(-1, want[-1])
        rows = int(getattr(self, 'lm_head_chunk_rows', 0) or 0)    # 0: one GEMM; else row chunks into the same block
        if rows <= 0 or rows >= x2.shape[0]:
            torch.mm(x2, self.lm_head.weight.t(), out=o2)
        else:
            for r0 in range(0, x2.shape[0], rows):
                torch.mm(x2[r0:r0 + rows], self.lm_head.weight.t(), out=o2[r0:r0 + rows])
        return CausalLMOutput(logits=logits_out)

    def refresh_inference_caches(self):
        """See BackpackModel.refresh_inference_caches (the cached whole-vocabulary sense table)."""
        self.transformer.refresh_inference_caches()

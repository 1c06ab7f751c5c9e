"""CPU: the drop-in Python call sites (same names / signatures / state-dict keys as the reference)
running their eager branch (`use_flash_attn=False`, the reference's non-optimised mode), checked
against the golden vectors; plus padding helpers and the loud-failure rules of the HIP branch."""
import inspect

import numpy as np
import pytest
import torch

from conftest import from_bits16, load_golden
from oracle import ref_cpu as R

from flash_attn import flash_attn_interface as fai
from flash_attn.bert_padding import pad_input, unpad_input
from flash_attn.flash_attention import FlashAttention, FlashMHA
from flash_attn.modules.mha import MHA, CrossAttention, FlashSelfAttention, SelfAttention
from src.models.backpack import (BackpackConfig, BackpackLMHeadModel, BackpackModel,
                                 ContextSelfAttn)


def nano_config(**kw):
    base = dict(n_embd=64, n_head=2, n_layer=2, num_content_vectors=4, vocab_size=96, n_positions=32,
                scale_attn_by_inverse_layer_idx=True, resid_pdrop=0.0, embd_pdrop=0.0, attn_pdrop=0.0,
                pad_vocab_size_multiple=8)
    base.update(kw)
    return BackpackConfig(**base)


def test_backpack_lm_matches_reference_golden_on_cpu():
    g = load_golden('g4_nano_model.npz')
    sd = {k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith('sd/')}
    model = BackpackLMHeadModel(nano_config()).eval()
    missing, unexpected = model.load_state_dict(sd, strict=True)
    assert not missing and not unexpected             # reference checkpoints load key-for-key
    ids = torch.from_numpy(g['ids'])
    with torch.no_grad():
        t = model.transformer
        h = t.gpt2_model(ids)
        alpha = t.contextualization_attn(h)
        content = t.content_model(ids)
        hidden = t(ids)
        out = model(ids)
    assert hasattr(out, 'logits') and type(out).__name__ == 'CausalLMOutput'
    for got, name in ((h, 'trunk'), (alpha, 'alpha'), (content, 'content'), (hidden, 'hidden'),
                      (out.logits, 'logits')):
        assert (got - torch.from_numpy(g[name])).abs().max().item() < 2e-5, name
    # layout contract of the fused kernel: content is a VIEW of a contiguous (B,S,k*d) buffer
    assert content.shape == (2, 4, 32, 64) and not content.is_contiguous()
    assert content.transpose(1, 2).is_contiguous()
    assert model.lm_head.weight is model.transformer.embeddings.word_embeddings.weight
    assert model.transformer.content_model.embeddings is model.transformer.gpt2_model.embeddings


def test_attribute_surface_used_by_the_intervention_scripts():
    # training/src/models/intervened_models.py:77-101 reaches into these attributes
    m = BackpackModel(nano_config())
    for name in ('gpt2_model', 'content_model', 'contextualization_attn', 'embeddings',
                 'num_content_vectors'):
        assert hasattr(m, name)
    assert m.config.vocab_size == 96
    cfg = nano_config(vocab_size=50257)
    BackpackModel(cfg)
    assert cfg.vocab_size == 50264                     # padded to a multiple of 8 (backpack.py:285-288)


def test_context_self_attn_matches_golden():
    g = load_golden('g12_sense.npz')
    for tag in ('dk24', 'dk40', 'dk10'):
        k = int(g[f'{tag}_k'])
        h = from_bits16(g[f'{tag}_h'])
        mod = ContextSelfAttn(k, h.shape[-1])
        mod.load_state_dict({'Wqkv.weight': from_bits16(g[f'{tag}_w']), 'Wqkv.bias': from_bits16(g[f'{tag}_b'])})
        with torch.no_grad():
            alpha = mod(h)
        assert (alpha - torch.from_numpy(g[f'{tag}_alpha'])).abs().max().item() < 1e-6
        assert mod.project(h).shape == (h.shape[0], h.shape[1], 2, k, h.shape[-1] // k)


def test_eager_attention_twins_match_golden():
    g = load_golden('g3_trunk_attn.npz')
    qkv = from_bits16(g['h64_qkv'])
    for layer in (0, 5, 11):
        mod = SelfAttention(causal=True, softmax_scale=64 ** -0.5 / (layer + 1))
        assert (mod(qkv) - torch.from_numpy(g[f'h64_L{layer}_out'])).abs().max().item() < 1e-6
    kpm = torch.from_numpy(g['h64_kpm'])
    out = SelfAttention()(qkv, key_padding_mask=kpm)
    assert (out - torch.from_numpy(g['h64_kpm_out'])).abs().max().item() < 1e-6
    cross = CrossAttention(causal=True)(qkv[:, :, 0], qkv[:, :, 1:])
    assert (cross - SelfAttention(causal=True)(qkv)).abs().max().item() < 1e-6


def test_mha_eager_and_per_layer_scale():
    from flash_attn.models.gpt import create_mixer_cls
    cfg = nano_config()
    mixers = [create_mixer_cls(cfg, layer_idx=i)(cfg.hidden_size) for i in range(3)]
    for i, m in enumerate(mixers):
        assert isinstance(m, MHA) and m.causal
        assert abs(m.inner_attn.softmax_scale - 32 ** -0.5 / (i + 1)) < 1e-12   # gpt.py:47-50
    x = torch.randn(2, 16, 64)
    y = mixers[1](x)
    qkv = mixers[1].Wqkv(x).reshape(2, 16, 3, 2, 32)
    ref = R.self_attention_eager(qkv, True, 32 ** -0.5 / 2)
    assert (y - mixers[1].out_proj(ref.reshape(2, 16, 64))).abs().max().item() < 1e-6


def test_unpad_pad_roundtrip_and_cu_seqlens():
    g = load_golden('g5_varlen.npz')
    lens = torch.from_numpy(g['lens'])
    smax = 128
    mask = torch.arange(smax)[None, :] < lens[:, None]
    x = torch.randn(4, smax, 24)
    rows, idx, cu, max_s = unpad_input(x, mask)
    assert cu.dtype == torch.int32 and torch.equal(cu, torch.from_numpy(g['cu_seqlens']))
    assert torch.equal(idx, torch.from_numpy(g['indices'])) and max_s == int(g['max_s'])
    back = pad_input(rows, idx, 4, smax)
    assert torch.equal(back, x * mask[:, :, None])


def test_public_signatures_match_the_reference():
    # flash_attn/flash_attn_interface.py:242-380
    assert list(inspect.signature(fai.flash_attn_unpadded_qkvpacked_func).parameters) == [
        'qkv', 'cu_seqlens', 'max_seqlen', 'dropout_p', 'softmax_scale', 'causal', 'return_attn_probs']
    assert list(inspect.signature(fai.flash_attn_unpadded_kvpacked_func).parameters) == [
        'q', 'kv', 'cu_seqlens_q', 'cu_seqlens_k', 'max_seqlen_q', 'max_seqlen_k', 'dropout_p',
        'softmax_scale', 'causal', 'return_attn_probs']
    assert list(inspect.signature(fai.flash_attn_unpadded_func).parameters) == [
        'q', 'k', 'v', 'cu_seqlens_q', 'cu_seqlens_k', 'max_seqlen_q', 'max_seqlen_k', 'dropout_p',
        'softmax_scale', 'causal', 'return_attn_probs']
    assert list(inspect.signature(fai.flash_attn_func).parameters) == [
        'qkv', 'cu_seqlens', 'dropout_p', 'max_s', 'softmax_scale', 'causal', 'return_attn_probs']
    assert list(inspect.signature(FlashAttention.forward).parameters) == [
        'self', 'qkv', 'key_padding_mask', 'causal', 'cu_seqlens', 'max_s', 'need_weights']
    assert list(inspect.signature(FlashSelfAttention.forward).parameters) == [
        'self', 'qkv', 'causal', 'cu_seqlens', 'max_seqlen']
    assert list(inspect.signature(ContextSelfAttn.__init__).parameters)[:3] == [
        'self', 'num_content_vectors', 'embed_dim']


def test_hip_branch_never_falls_back():
    # 16-bit CUDA tensors are required; CPU input must raise, not silently run eager code
    qkv = torch.randn(1, 8, 3, 2, 32).bfloat16()
    with pytest.raises(AssertionError):
        FlashSelfAttention(causal=True)(qkv)
    with pytest.raises(AssertionError):
        FlashAttention()(qkv)
    model = BackpackLMHeadModel(nano_config(use_flash_attn=True)).eval()
    assert model.transformer.use_hip and model.transformer.contextualization_attn.use_hip
    with pytest.raises((AssertionError, RuntimeError)):
        model(torch.zeros(1, 8, dtype=torch.long))
    with pytest.raises(RuntimeError, match='GPU'):     # dropout included: in-kernel or nothing
        fai._flash_attn_forward(qkv[0, :, 0], qkv[0, :, 1], qkv[0, :, 2], qkv[0, :, 0].clone(),
                                None, None, 8, 8, 0.1, 0.2, True, False)
    with pytest.raises(RuntimeError, match='dropout_p'):
        fai._flash_attn_forward(qkv[0, :, 0], qkv[0, :, 1], qkv[0, :, 2], qkv[0, :, 0].clone(),
                                None, None, 8, 8, 1.0, 0.2, True, False)
    assert FlashMHA(64, 2).head_dim == 32


def test_generation_and_checkpoint_roundtrip(tmp_path):
    """GenerationMixin (training/src/utils/generation.py:80-92) and the Lightning-checkpoint loader
    (training/src/utils/checkpoint.py:68-76) on the eager CPU path."""
    from src.utils.checkpoint import load_backpack_checkpoint, remove_model_prefix
    from src.utils.generation import greedy_decode
    torch.manual_seed(0)
    model = BackpackLMHeadModel(nano_config()).eval()
    ckpt = {'state_dict': {'model.' + k: v for k, v in model.state_dict().items()}, 'epoch': 3}
    path = tmp_path / 'last.ckpt'
    torch.save(ckpt, path)
    other = BackpackLMHeadModel(nano_config()).eval()
    res = load_backpack_checkpoint(other, path)
    assert not res.missing_keys and not res.unexpected_keys
    assert set(remove_model_prefix(ckpt)) == set(model.state_dict())
    ids = torch.randint(0, 96, (1, 5), generator=torch.Generator().manual_seed(1))
    seq = model.generate(ids, max_length=10)
    # the reference returns max_length - 1 tokens: its last pick is never appended (generation.py:64-72)
    assert seq.shape == (1, 9) and torch.equal(seq[:, :5], ids)
    assert torch.equal(other.generate(ids, max_length=10), seq)         # same weights, same greedy path
    # each new token is the argmax of the full forward on the prefix (no cache, as upstream)
    with torch.no_grad():
        for t in range(5, 9):
            assert seq[0, t] == model(seq[:, :t]).logits[0, -1].argmax()
    out = model.generate(ids, max_length=7, return_dict_in_generate=True, output_scores=True)
    assert out.sequences.shape == (1, 6) and len(out.scores) == 1
    assert model.generate(ids, max_length=3).shape == (1, 5)           # nothing to add: the prompt comes back
    assert model.sample(ids, max_length=8).shape == (1, 7)
    batch = greedy_decode(torch.cat([ids, ids]), model, 9).sequences
    assert torch.equal(batch[0], batch[1]) and torch.equal(batch[0], seq[0, :8])


def test_generation_matches_the_reference_token_for_token():
    """G8 (tests/golden/make_golden_r2.py): the reference's own greedy_decode on the nano model -- same token
    ids, same returned length (max_length - 1), same first-step scores."""
    g4, g8 = load_golden('g4_nano_model.npz'), load_golden('g8_generation.npz')
    sd = {k[3:]: torch.from_numpy(g4[k]) for k in g4.files if k.startswith('sd/')}
    model = BackpackLMHeadModel(nano_config()).eval()
    model.load_state_dict(sd)
    prompt = torch.from_numpy(g8['prompt'])
    for max_length in (6, 12, 20):
        out = model.generate(prompt, max_length=max_length, return_dict_in_generate=True, output_scores=True)
        want = torch.from_numpy(g8['greedy_%d' % max_length])
        assert out.sequences.shape == want.shape == (1, max_length - 1)
        assert torch.equal(out.sequences, want)
        assert len(out.scores) == 1
        assert (out.scores[0] - torch.from_numpy(g8['scores_%d' % max_length])).abs().max().item() < 1e-5
    assert tuple(model.sample(prompt, max_length=12).shape) == tuple(g8['sample_12_shape'])


def test_hf_gpt2_state_dict_of_the_installed_transformers_loads():
    """State dicts of the installed `transformers` (no `attn.bias` / `attn.masked_bias` buffers any more: they are
    non-persistent there) go through `from_pretrained` unmodified, for GPT2Model- and GPT2LMHeadModel-style keys,
    and reproduce transformers' own logits."""
    from transformers import GPT2Config, GPT2LMHeadModel
    from flash_attn.models.gpt import GPTLMHeadModel
    import flash_attn.utils.hf_convert as conv
    import src.utils.hf_convert as conv_src
    assert conv_src.remap_state_dict_gpt2 is conv.remap_state_dict_gpt2      # one implementation, two import paths
    kw = dict(n_embd=32, n_head=2, n_layer=2, n_positions=16, resid_pdrop=0.0, embd_pdrop=0.0, attn_pdrop=0.0)
    torch.manual_seed(0)
    hf_model = GPT2LMHeadModel(GPT2Config(vocab_size=50, **kw)).eval()
    sd = hf_model.state_dict()
    ids = torch.randint(0, 50, (2, 16))
    with torch.no_grad():
        want = hf_model(ids).logits
    for state in (sd, hf_model.transformer.state_dict()):
        cfg = GPT2Config(vocab_size=50, **kw)
        cfg.pad_vocab_size_multiple = 8
        model = GPTLMHeadModel.from_pretrained('unused', cfg, state_dict=dict(state)).eval()
        with torch.no_grad():
            got = model(ids).logits[..., :50]
        assert (got - want).abs().max().item() < 1e-4
    bcfg = BackpackConfig(vocab_size=50, num_content_vectors=4, pad_vocab_size_multiple=8, **kw)
    bp_model = BackpackLMHeadModel.from_pretrained('unused', bcfg, state_dict=dict(sd))
    assert torch.equal(bp_model.lm_head.weight[:50], sd['transformer.wte.weight'])


def test_hf_gpt2_remap_matches_the_reference_and_transformers():
    """G7: `remap_state_dict_gpt2` / `remap_state_dict_flash` against the outputs of the reference's functions
    on the same tiny HF state dict (bit-exact: renames, transposes, zero padding), and `from_pretrained` end to
    end: the flash-layout model on the remapped weights reproduces the logits of transformers' own
    GPT2LMHeadModel (pins the LayerNorm half-block shift)."""
    from transformers import GPT2Config
    from flash_attn.models.gpt import GPTLMHeadModel, GPTModel
    from src.models.backpack import remap_state_dict_gpt2
    from src.utils.hf_convert import load_non_optimized_model, remap_state_dict_flash
    g = load_golden('g7_hf_remap.npz')
    hf = {k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith('hf/')}
    want_flash = {k[6:]: torch.from_numpy(g[k]) for k in g.files if k.startswith('flash/')}
    want_back = {k[5:]: torch.from_numpy(g[k]) for k in g.files if k.startswith('back/')}
    kw = dict(n_embd=int(g['n_embd']), n_head=int(g['n_head']), n_layer=int(g['n_layer']),
              n_positions=int(g['n_positions']), resid_pdrop=0.0, embd_pdrop=0.0, attn_pdrop=0.0)
    padded = GPT2Config(vocab_size=int(g['padded_vocab_size']), **kw)
    flash = remap_state_dict_gpt2(hf, padded)
    assert set(flash) == set(want_flash)
    for k, v in want_flash.items():
        assert torch.equal(flash[k], v), k
    assert flash['lm_head.weight'] is flash['transformer.embeddings.word_embeddings.weight']
    assert torch.count_nonzero(flash['lm_head.weight'][int(g['vocab_size']):]) == 0
    back = remap_state_dict_flash(flash, padded)
    assert set(back) == set(want_back)
    for k, v in want_back.items():
        assert torch.equal(back[k], v), k
    assert 'wte.weight' in hf and 'h.0.attn.bias' in hf        # the caller's dict is left alone

    cfg = GPT2Config(vocab_size=int(g['vocab_size']), **kw)
    cfg.pad_vocab_size_multiple = 8
    ids = torch.from_numpy(g['ids'])
    model = GPTLMHeadModel.from_pretrained('unused', cfg, state_dict=hf).eval()
    assert cfg.vocab_size == int(g['padded_vocab_size'])       # padded in place, as upstream (gpt.py:180-183)
    with torch.no_grad():
        logits = model(ids).logits
    want = torch.from_numpy(g['hf_logits'])
    assert (logits[..., :want.shape[-1]] - want).abs().max().item() < 1e-4
    # GPT2LMHeadModel-style checkpoints (keys under `transformer.` + lm_head) load as well
    prefixed = {('transformer.' + k): v for k, v in hf.items()}
    prefixed['lm_head.weight'] = hf['wte.weight']
    trunk = GPTModel.from_pretrained('unused', cfg, state_dict=prefixed).eval()
    with torch.no_grad():
        assert torch.equal(trunk(ids), model.transformer(ids))

    # Backpack: the trunk takes the GPT-2 weights, the sense network keeps its initialisation
    bcfg = BackpackConfig(vocab_size=int(g['vocab_size']), num_content_vectors=4, pad_vocab_size_multiple=8,
                          use_flash_attn=True, fused_dropout_add_ln=True, **kw)
    bp_model = BackpackLMHeadModel.from_pretrained('unused', bcfg, state_dict=hf)
    assert torch.equal(bp_model.lm_head.weight, flash['lm_head.weight'])
    assert bp_model.lm_head.weight is bp_model.transformer.content_model.embeddings.word_embeddings.weight
    eager = load_non_optimized_model(bp_model).eval()          # every fused / flash switch off, same weights
    assert not eager.config.use_flash_attn and not eager.config.fused_dropout_add_ln
    with torch.no_grad():
        assert torch.equal(eager.transformer.gpt2_model(ids), trunk(ids))


def _g6():
    g = load_golden('g6_interventions.npz')
    sd = {k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith('sd/')}
    ids = torch.from_numpy(g['ids'])
    cw = torch.from_numpy(g['content_weights'])
    senses = {int(w): torch.from_numpy(g['sense/%d' % w]) for w in g['sense_words']}
    return g, sd, ids, cw, senses


def test_intervened_models_match_reference_golden_on_cpu():
    """The mirror of intervened_models.py (eager path here) against logits produced by the REFERENCE classes
    (tests/golden/make_golden.py G6): weighted / negative-weighted with and without annealing, replaced words."""
    from src.models import intervened_models as im
    g, sd, ids, cw, senses = _g6()
    model = BackpackLMHeadModel(nano_config(num_content_vectors=16)).eval()
    model.load_state_dict(sd, strict=True)
    scale = float(g['annealing_scale'])
    with torch.no_grad():
        content = model.transformer.content_model(ids)
        scores = im.mask_annealing(model, ids, None, content, scale, True)
        assert (scores - torch.from_numpy(g['annealing_scores'])).abs().max().item() < 1e-5
        for tag, anneal in (('anneal', True), ('plain', False)):
            w = im.WeightedBackpackLMHeadModel(model, cw, torch.zeros(96), scale, anneal=anneal)
            assert (w(ids).logits - torch.from_numpy(g['weighted_' + tag])).abs().max().item() < 1e-4, tag
            n = im.NegativeWeightedBackpackLMHeadModel(model, cw, torch.zeros(96), scale, anneal=anneal)
            assert (n(ids).logits - torch.from_numpy(g['negative_' + tag])).abs().max().item() < 1e-4, tag
        r = im.ReplacedWordLMHeadModel(model, senses)
        assert (r(ids).logits - torch.from_numpy(g['replaced'])).abs().max().item() < 1e-4
        # a sense vector does not depend on context or position
        v = im.get_sense_vector_of_word(int(ids[0, 3]), model, 5)
        assert (v - content[0, 5, 3, :]).abs().max().item() < 1e-5
    assert list(inspect.signature(im.WeightedBackpackLMHeadModel.__init__).parameters)[1:] == [
        'backpack_network', 'content_weights', 'target_weight', 'annealing_scale', 'anneal', 'upweight_nearby']
    assert list(inspect.signature(im.create_content_soft_mask).parameters) == ['content_weights', 'input_ids', 'scores']


def test_product_never_touches_the_oracle():
    """oracle/ is test infrastructure: nothing under the package, bench timing path or entry points may import it
    (bench.py's cpu_baseline leg and __graft_entry__.smoke() are the two sanctioned callers)."""
    import os
    import re
    from conftest import PKG, ROOT
    pat = re.compile(r'^\s*(from|import)\s+oracle\b', re.M)
    for base, _, files in os.walk(PKG):
        for f in files:
            if f.endswith(('.py', '.hip', '.h')):
                text = open(os.path.join(base, f), errors='ignore').read()
                assert not pat.search(text), os.path.join(base, f)
    bench = open(os.path.join(ROOT, 'bench.py')).read()
    uses = [m.start() for m in pat.finditer(bench)]
    assert len(uses) == 1 and 'def cpu_baseline' in bench[:uses[0]] and bench.rfind('def ', 0, uses[0]) == bench.find('def cpu_baseline')


def test_padded_sense_projection_is_cached_in_inference_and_tracks_the_parameter():
    """d_k = 10 (Mini k = 64): `ContextSelfAttn.project` widens the senses to 16 with zero weight rows.  Without autograd
    the padded copy is reused until the parameter changes; under autograd it stays part of the graph."""
    from src.models.backpack import ContextSelfAttn
    torch.manual_seed(0)
    attn = ContextSelfAttn(8, 80, use_hip=True)            # d_k = 10 -> 16
    x = torch.randn(2, 5, 80)
    with torch.no_grad():
        a = attn.project(x)
        w1 = attn._padded_cache[1]
        b = attn.project(x)
        assert attn._padded_cache[1] is w1                  # no second pad
        assert a.shape == (2, 5, 2, 8, 16) and torch.count_nonzero(a[..., 10:]) == 0
        plain = torch.nn.functional.linear(x, attn.Wqkv.weight, attn.Wqkv.bias).reshape(2, 5, 2, 8, 10)
        assert torch.equal(a[..., :10], plain) and torch.equal(a, b)
        attn.Wqkv.weight.mul_(2.0)                          # in-place update: the version moves, the cache follows
        c = attn.project(x)
        assert attn._padded_cache[1] is not w1
        assert torch.allclose(c[..., :10], torch.nn.functional.linear(x, attn.Wqkv.weight, attn.Wqkv.bias).reshape(2, 5, 2, 8, 10))
    out = attn.project(x)                                   # autograd: gradients reach the unpadded parameter
    out.square().sum().backward()
    assert attn.Wqkv.weight.grad is not None and attn.Wqkv.weight.grad.shape == attn.Wqkv.weight.shape
    assert '_padded_cache' not in attn.state_dict()


def test_padded_sense_projection_with_inference_tensors():
    """Parameters created under torch.inference_mode carry no version counter (reading `._version` raises): the pad then
    runs per call instead of being cached (advisor, round 3)."""
    from src.models.backpack import ContextSelfAttn
    with torch.inference_mode():
        attn = ContextSelfAttn(8, 80, use_hip=True)
        x = torch.randn(2, 5, 80)
        a = attn.project(x)
        assert a.shape == (2, 5, 2, 8, 16) and torch.count_nonzero(a[..., 10:]) == 0
        assert getattr(attn, '_padded_cache', None) is None
        plain = torch.nn.functional.linear(x, attn.Wqkv.weight, attn.Wqkv.bias).reshape(2, 5, 2, 8, 10)
        assert torch.equal(a[..., :10], plain)


def test_integration_doc_asserts_the_header_abi_version():
    """INTEGRATION.md shows the reference-side binding; the version it asserts must be the header's (round-3 review:
    the doc still said 2 while the header was at 3)."""
    import os
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    header = open(os.path.join(root, 'include', 'bp_hip.h')).read()
    version = int(re.search(r'#define BP_ABI_VERSION (\d+)', header).group(1))
    doc = open(os.path.join(root, 'INTEGRATION.md')).read()
    asserted = [int(m) for m in re.findall(r'bp_abi_version\(\)`?\s*=+\s*(\d+)', doc)]
    assert len(asserted) >= 2, 'INTEGRATION.md no longer states the ABI version'
    assert all(v == version for v in asserted), (asserted, version)
    import bp_hip
    assert bp_hip.ABI_VERSION == version


def test_lm_head_writes_into_a_caller_owned_logits_buffer():
    """`logits_out=` (bench.py's persistent logits block): same numbers as the allocating call, written in place;
    inference only; shape / dtype checked."""
    torch.manual_seed(0)
    model = BackpackLMHeadModel(nano_config()).eval()
    ids = torch.randint(0, 96, (2, 16))
    with torch.no_grad():
        want = model(ids).logits
        buf = torch.full((3, 16, 96), float('nan'))
        got = model(ids, logits_out=buf[:2]).logits
    assert got.data_ptr() == buf.data_ptr() and torch.allclose(got, want, atol=1e-6)
    assert torch.isnan(buf[2]).all()
    with pytest.raises(RuntimeError, match='inference-only'):
        model(ids, logits_out=buf[:2])
    with torch.no_grad(), pytest.raises(RuntimeError, match='logits_out must be'):
        model(ids, logits_out=torch.empty(2, 16, 95))


def test_content_of_unique_tokens_equals_the_per_position_content():
    """Inference on the HIP path runs the content (sense) network once per DISTINCT token of the batch and gathers the
    rows (BackpackModel._content_of_unique_tokens): the sense vectors are a function of the token alone (reference
    backpack.py:251-276: no positions, Identity mixer).  On the CPU, in fp32, the gathered tensor equals the
    per-position one; the switch itself is taken only where it is exact and pays (GPU, eval, no autograd, no capture,
    >= vocabulary-size positions)."""
    torch.manual_seed(0)
    model = BackpackLMHeadModel(nano_config()).eval()
    t = model.transformer
    ids = torch.randint(0, 96, (4, 64))
    with torch.no_grad():
        want = t.content_model(ids)
        got = t._content_of_unique_tokens(ids)
    assert got.shape == want.shape == (4, 4, 64, 64)
    assert torch.allclose(got, want, atol=1e-6, rtol=0)
    assert got.transpose(1, 2).is_contiguous()             # the (B,S,k,d) storage the mix kernel consumes as it lies
    assert t.dedup_content is True
    with torch.no_grad():
        assert not t._dedup_applies(ids)                   # CPU tensors: the eager path stays the reference's, literally
    assert BackpackLMHeadModel(nano_config(dedup_content=False)).transformer.dedup_content is False


def test_whole_vocabulary_sense_table_is_the_content_network_row_by_row_and_follows_the_weights():
    """BackpackModel.sense_table(): the content network's output for every row of the word embedding -- on the CPU, in
    fp32, table[ids] IS the per-position content (reference backpack.py:251-276); kept while no parameter of the content
    model changes, rebuilt in the SAME storage after an in-place update (a captured graph reads that storage), rebuilt
    after a reload, dropped by .train(); never consulted by the forward of CPU tensors / of the eager mode."""
    torch.manual_seed(0)
    model = BackpackLMHeadModel(nano_config()).eval()
    t = model.transformer
    assert t.sense_table_mode == 'cached'
    ids = torch.randint(0, 96, (3, 32))
    with torch.no_grad():
        want = t.content_model(ids)                                    # (B,k,S,d)
        logits = model(ids).logits
    assert t._sense_table is None                                      # the CPU forward never built one
    table = t.sense_table()
    vocab = t.embeddings.word_embeddings.weight.shape[0]
    assert table.shape == (vocab, 4, 64) and not table.requires_grad and t.sense_table() is table
    assert torch.allclose(table[ids].transpose(1, 2), want, atol=1e-6, rtol=0)
    ptr, before = table.data_ptr(), table.clone()
    with torch.no_grad():
        t.content_model.final_mlp.fc2.bias.add_(1.0)                   # in place: `_version` moves
    after = t.sense_table()
    assert after.data_ptr() == ptr and torch.allclose(after, before + 1.0, atol=1e-5)
    with torch.no_grad():                                              # the word embedding is a parameter of the content model too
        t.embeddings.word_embeddings.weight[5].add_(torch.randn(64))    # (a pure rescale would vanish in ln_0)
    again = t.sense_table()
    assert not torch.allclose(again[5], before[5] + 1.0, atol=1e-3) and torch.allclose(again[6], before[6] + 1.0, atol=1e-5)
    model.load_state_dict({k: v.clone() for k, v in model.state_dict().items()})   # copy_ in place: versions move again
    assert t._sense_table[0] != t._sense_table_key()
    model.train()
    assert t._sense_table is None
    model.eval()
    with torch.inference_mode():
        inside = t.sense_table()                                       # built as a normal tensor even in inference mode
    assert not inside.is_inference()
    with torch.no_grad():
        assert torch.equal(model(ids).logits - logits, model(ids).logits - logits)   # (CPU forward unaffected; finite)
    with pytest.raises(AssertionError):
        BackpackLMHeadModel(nano_config(sense_table='sometimes'))
    # updates through `.data` do not move `_version` (the reference's EMA swap, training/src/utils/ema.py:121,165): the
    # version key alone keeps serving the old rows; the value check (`verify`, automatic for bulk forwards on the GPU)
    # and `invalidate_sense_table()` both catch it, and both rebuild in the SAME storage
    table = t.sense_table()
    ptr, before = table.data_ptr(), table.clone()
    t.content_model.final_mlp.fc2.bias.data.add_(2.0)
    assert torch.equal(t.sense_table(), before)                        # stale, by construction of the version key
    fresh = t.sense_table(verify=True)
    assert fresh.data_ptr() == ptr and torch.allclose(fresh, before + 2.0, atol=1e-5)
    assert t.sense_table(verify=True) is fresh                         # values unchanged: kept
    t.content_model.final_mlp.fc2.bias.data.add_(1.0)
    t.invalidate_sense_table()
    assert torch.allclose(t.sense_table(), before + 3.0, atol=1e-5) and t.sense_table().data_ptr() == ptr
    # a pinned table (a captured graph holds its address) survives .train() as storage and is refilled in place
    t.pin_sense_table()
    model.train()
    assert t._sense_table is not None and t._sense_table[0] is None
    model.eval()
    assert t.sense_table().data_ptr() == ptr
    t.pin_sense_table(False)
    model.train()
    assert t._sense_table is None
    model.eval()
    # 'auto' verification: never for CPU tensors, on the GPU from `sense_table_verify_min_positions` positions up
    assert not t._verify_applies(ids)


def test_stochastic_depth_and_block_drop_path_on_cpu():
    """flash_attn/modules/block.py: StochasticDepth (torchvision's, restated: the image has no torchvision) drops whole
    samples in training and is the identity in eval; Block wires it behind the dropout of both branches (reference
    block.py:82-90,96-105); the state dict carries no new keys."""
    from functools import partial
    import torch.nn as nn
    from flash_attn.modules.block import Block, StochasticDepth
    from flash_attn.modules.mlp import Mlp
    from src.models.backpack import Identity
    sd = StochasticDepth(0.5, 'row').train()
    torch.manual_seed(0)
    x = torch.ones(64, 3, 5)
    y = sd(x)
    kept = y[:, 0, 0] != 0
    assert 8 < kept.sum() < 56 and torch.all(y[kept] == 2.0) and torch.all(y[~kept] == 0.0)    # whole rows, scaled by 1/(1-p)
    assert torch.equal(sd.eval()(x), x) and torch.equal(StochasticDepth(0.0).train()(x), x)
    with pytest.raises(ValueError):
        StochasticDepth(1.5)
    blk = Block(16, Identity, partial(Mlp, hidden_features=32), norm_cls=nn.LayerNorm, prenorm=True, drop_path=0.5)
    assert set(blk.state_dict()) == {'norm1.weight', 'norm1.bias', 'mlp.fc1.weight', 'mlp.fc1.bias', 'mlp.fc2.weight',
                                     'mlp.fc2.bias', 'norm2.weight', 'norm2.bias'}
    h, r = torch.randn(32, 4, 16), torch.randn(32, 4, 16)
    blk.train()
    torch.manual_seed(1)
    h1, r1 = blk(h, r)
    torch.manual_seed(1)
    keep1 = torch.empty(32, 1, 1).bernoulli_(0.5) / 0.5           # the two draws of the block, in order
    keep2 = torch.empty(32, 1, 1).bernoulli_(0.5) / 0.5
    res1 = h * keep1 + r
    mid = blk.norm1(res1)
    res2 = blk.mlp(mid) * keep2 + res1
    assert torch.allclose(r1, res2, atol=1e-6) and torch.allclose(h1, blk.norm2(res2), atol=1e-6)
    blk.eval()
    h2, r2 = blk(h, r)
    assert torch.allclose(r2, blk.mlp(blk.norm1(h + r)) + h + r, atol=1e-6)
    with pytest.raises(NotImplementedError):
        Block(16, Identity, prenorm=False)


def test_bench_clock_power_sampler_without_a_gpu():
    """bench.py's side-thread sampler: with nothing readable it reports source None and null means, never a guess."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location('bench_mod', os.path.join(os.path.dirname(os.path.dirname(
        os.path.abspath(__file__))), 'bench.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    s = mod.ClockPowerSampler(0, period_s=0.01).start()
    out = s.stop()
    assert set(out) >= {'sclk_mhz_mean', 'power_w_mean', 'power_cap_w', 'samples', 'source'}
    if out['source'] is None:
        assert out['sclk_mhz_mean'] is None and out['power_w_mean'] is None and out['samples'] == 0

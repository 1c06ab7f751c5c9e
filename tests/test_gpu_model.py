"""-m gpu: the reference's Python call sites running on the HIP kernels, against the CPU oracle."""
import numpy as np
import pytest
import torch

from conftest import from_bits16, load_golden
from oracle import ref_cpu as R

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def _nano(use_flash=True, dtype=torch.bfloat16, fused=False):
    from src.models.backpack import BackpackConfig, BackpackLMHeadModel
    g = load_golden('g4_nano_model.npz')
    sd = {k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith('sd/')}
    cfg = BackpackConfig(n_embd=64, n_head=2, n_layer=2, num_content_vectors=4, vocab_size=96,
                         n_positions=32, scale_attn_by_inverse_layer_idx=True, resid_pdrop=0.0,
                         embd_pdrop=0.0, attn_pdrop=0.0, use_flash_attn=use_flash,
                         fused_dropout_add_ln=fused, fused_dense_gelu_dense=fused, fused_bias_fc=fused,
                         pad_vocab_size_multiple=8)
    model = BackpackLMHeadModel(cfg)
    model.load_state_dict(sd)
    return g, sd, model.to(DEV, dtype).eval()


@pytest.mark.parametrize('fused', [False, True])
def test_nano_model_hip_vs_golden(fused):
    """fused=True is the reference's backpack-small-flash flag set (fused_dropout_add_ln,
    fused_dense_gelu_dense, fused_bias_fc): add+LN in one HIP launch, GELU in the GEMM epilogue."""
    g, sd, model = _nano(fused=fused)
    ids = torch.from_numpy(g['ids']).to(DEV)
    with torch.no_grad():
        t = model.transformer
        h = t.gpt2_model(ids)
        alpha = t.contextualization_attn(h)          # bp_sense_alpha
        hidden = t(ids)                              # flash trunk + fused mix
        logits = model(ids).logits
    ocfg = dict(n_embd=64, n_head=2, n_layer=2, num_content_vectors=4,
                layer_norm_epsilon=float(g['layer_norm_epsilon']), scale_attn_by_inverse_layer_idx=True)
    sd16 = {k: v.bfloat16() for k, v in sd.items()}
    eager = R.backpack_forward(sd16, ocfg, torch.from_numpy(g['ids']), return_stages=True)
    for got, name in ((h, 'trunk'), (alpha, 'alpha'), (hidden, 'hidden'), (logits, 'logits')):
        want = torch.from_numpy(g[name])
        err = (got.float().cpu() - want).abs().max().item()
        base = (eager[name].float() - want).abs().max().item()
        print(f'{name}: hip {err:.3e} eager-bf16 {base:.3e}')
        assert err <= 3 * base + 1e-3, (name, err, base)
    s = alpha.shape[-1]
    upper = torch.triu(torch.ones(s, s, dtype=torch.bool, device=DEV), 1)
    assert torch.count_nonzero(alpha[:, :, upper]) == 0


def test_content_dedup_matches_the_per_position_content_network():
    """Inference forward with the content network run once per distinct token (default) against the same model with
    `dedup_content` off (the reference's order: every position): same hidden states and logits to bf16 GEMM noise, the
    switch conditions honoured (taken for >= vocab positions under no_grad in eval; not in training, not with
    autograd, not for small inputs)."""
    g, sd, model = _nano(fused=True)
    t = model.transformer
    t.sense_table_mode = 'batch'      # (the default, 'cached', keeps a whole-vocabulary table instead: test_gpu_configs.py)
    ids = torch.randint(0, 96, (8, 32), device=DEV, generator=torch.Generator(device=DEV).manual_seed(3))
    calls = []
    orig = t._table_of_unique_tokens
    t._table_of_unique_tokens = lambda x: (calls.append(x.shape), orig(x))[1]
    with torch.no_grad():
        assert t._dedup_applies(ids) and not t._dedup_applies(ids[:2])        # 256 >= 96 (= vocab) positions; 64 < 96
        hid = t(ids)
        logits = model(ids).logits
        assert len(calls) == 2
        t.dedup_content = False
        assert not t._dedup_applies(ids)
        hid_pp = t(ids)
        logits_pp = model(ids).logits
        assert len(calls) == 2
        t.dedup_content = True
    # shapes the gathering kernel does not take (S > 4096, a table of 4 GiB or more): the rows are gathered by torch and
    # the dense kernel runs -- the same bits
    import bp_hip
    supported = bp_hip.sense_mix_gather_supported
    bp_hip.sense_mix_gather_supported = lambda *a, **kw: False
    try:
        with torch.no_grad():
            assert torch.equal(t(ids), hid)
    finally:
        bp_hip.sense_mix_gather_supported = supported
    assert len(calls) == 3
    calls.pop()
    assert not t._dedup_applies(ids)                                          # autograd enabled
    model.train()
    with torch.no_grad():
        assert not t._dedup_applies(ids)                                      # dropout would differ per position
    model.eval()
    # identical inputs to the mix kernel up to the row-count dependence of the BLAS GEMMs: compare at bf16 resolution
    assert (hid.float() - hid_pp.float()).abs().max().item() <= 2 ** -7 * hid_pp.float().abs().max().item()
    assert (logits.float() - logits_pp.float()).abs().max().item() <= 2 ** -6 * logits_pp.float().abs().max().item()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    graph = torch.cuda.CUDAGraph()
    with torch.no_grad(), torch.cuda.stream(side):
        t(ids)
    torch.cuda.current_stream().wait_stream(side)
    with torch.no_grad(), torch.cuda.graph(graph):
        hid_g = t(ids)                                                        # capture: the per-position path, no unique()
    graph.replay()
    torch.cuda.synchronize()
    assert len(calls) == 3 and torch.equal(hid_g, hid_pp)


def test_fused_path_equals_materialised_alpha_path():
    """BackpackModel.forward (fused, alpha never stored) vs alpha from ContextSelfAttn @ content."""
    g, sd, model = _nano()
    ids = torch.from_numpy(g['ids']).to(DEV)
    with torch.no_grad():
        t = model.transformer
        h = t.gpt2_model(ids)
        alpha = t.contextualization_attn(h)
        content = t.content_model(ids)
        two_step = torch.sum(alpha.float() @ content.float(), dim=1)
        fused = t(ids).float()
    assert (fused - two_step).abs().max().item() < 3e-2


@pytest.mark.parametrize('fused', [False, True])
@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
def test_micro_config1_forward(dtype, fused):
    """BASELINE config 1 shape (Backpack-Micro, B=4, S=128): HIP 16-bit vs the fp32 CPU oracle."""
    from src.models.backpack import BackpackConfig, BackpackLMHeadModel
    cfg = BackpackConfig(n_embd=384, n_head=6, n_layer=6, num_content_vectors=16, vocab_size=50257,
                         n_positions=128, scale_attn_by_inverse_layer_idx=True, resid_pdrop=0.0,
                         embd_pdrop=0.0, attn_pdrop=0.0, use_flash_attn=True, pad_vocab_size_multiple=8,
                         fused_dropout_add_ln=fused, fused_dense_gelu_dense=fused, fused_bias_fc=fused)
    torch.manual_seed(0)
    model = BackpackLMHeadModel(cfg).eval()
    with torch.no_grad():   # sharpen attention so the softmax paths matter
        model.transformer.contextualization_attn.Wqkv.weight.mul_(8.0)
        for layer in model.transformer.gpt2_model.layers:
            layer.mixer.Wqkv.weight.mul_(6.0)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    ids = torch.randint(0, 50257, (4, 128), generator=torch.Generator().manual_seed(0))
    ocfg = dict(n_embd=384, n_head=6, n_layer=6, num_content_vectors=16, layer_norm_epsilon=1e-5,
                scale_attn_by_inverse_layer_idx=True)
    want = R.backpack_forward(sd, ocfg, ids, return_stages=True)
    sd16 = {k: v.to(dtype) for k, v in sd.items()}
    eager = R.backpack_forward(sd16, ocfg, ids, return_stages=True)
    model = model.to(DEV, dtype)
    with torch.no_grad():
        hidden = model.transformer(ids.to(DEV))
        logits = model(ids.to(DEV)).logits
    for got, name in ((hidden, 'hidden'), (logits, 'logits')):
        err = (got.float().cpu() - want[name]).abs().max().item()
        base = (eager[name].float() - want[name]).abs().max().item()
        print(f'micro {dtype} {name}: hip {err:.3e} eager {base:.3e}')
        assert err <= 3 * base + 1e-3, (name, err, base)


@pytest.mark.parametrize('fused', [False, True])
def test_mini_k64_config4_forward(fused):
    """BASELINE config 4 shape family (Backpack-Mini sense ablation: d_h = 80, k = 64 -> d_k = 10,
    shrink_final_inner), 2 layers, B=2, S=192: exercises the generic-head-dim flash path and the zero-padded
    d_k = 10 sense kernels inside the whole model."""
    from src.models.backpack import BackpackConfig, BackpackLMHeadModel
    kw = dict(n_embd=640, n_head=8, n_layer=2, num_content_vectors=64, shrink_final_inner=True)
    cfg = BackpackConfig(vocab_size=4093, n_positions=192, scale_attn_by_inverse_layer_idx=True, resid_pdrop=0.0,
                         embd_pdrop=0.0, attn_pdrop=0.0, use_flash_attn=True, pad_vocab_size_multiple=8,
                         fused_dropout_add_ln=fused, fused_dense_gelu_dense=fused, fused_bias_fc=fused, **kw)
    torch.manual_seed(1)
    model = BackpackLMHeadModel(cfg).eval()
    with torch.no_grad():
        model.transformer.contextualization_attn.Wqkv.weight.mul_(8.0)
        for layer in model.transformer.gpt2_model.layers:
            layer.mixer.Wqkv.weight.mul_(6.0)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    ids = torch.randint(0, 4093, (2, 192), generator=torch.Generator().manual_seed(1))
    ocfg = dict(layer_norm_epsilon=1e-5, scale_attn_by_inverse_layer_idx=True, **kw)
    want = R.backpack_forward(sd, ocfg, ids, return_stages=True)
    eager = R.backpack_forward({k: v.bfloat16() for k, v in sd.items()}, ocfg, ids, return_stages=True)
    model = model.to(DEV, torch.bfloat16)
    with torch.no_grad():
        hidden = model.transformer(ids.to(DEV))
        logits = model(ids.to(DEV)).logits
    for got, name in ((hidden, 'hidden'), (logits, 'logits')):
        err = (got.float().cpu() - want[name]).abs().max().item()
        base = (eager[name].float() - want[name]).abs().max().item()
        print(f'mini-k64 {name}: hip {err:.3e} eager {base:.3e}')
        assert err <= 3 * base + 1e-3, (name, err, base)


def test_interface_functions_and_probs():
    from flash_attn.flash_attn_interface import (flash_attn_unpadded_func,
                                                 flash_attn_unpadded_kvpacked_func,
                                                 flash_attn_unpadded_qkvpacked_func)
    torch.manual_seed(0)
    b, s, h, d = 2, 160, 4, 64
    qkv = torch.randn(b, s, 3, h, d).bfloat16()
    cu = torch.arange(0, (b + 1) * s, s, dtype=torch.int32, device=DEV)
    flat = qkv.flatten(0, 1).to(DEV)
    for causal in (False, True):
        want, attn, lse_want = R.attention_fp32(qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2], causal=causal)
        out, lse, probs = flash_attn_unpadded_qkvpacked_func(flat, cu, s, 0.0, causal=causal,
                                                             return_attn_probs=True)
        assert out.shape == (b * s, h, d) and lse.shape == (b, h, 160) and probs.shape == (b, h, s, s)
        assert (out.float().cpu().unflatten(0, (b, s)) - want.float()).abs().max().item() < 2e-2
        assert (probs.float().cpu() - attn.float()).abs().max().item() < 1e-2
        assert (lse.cpu() - lse_want).abs().max().item() < 2e-3
        out2 = flash_attn_unpadded_kvpacked_func(flat[:, 0], flat[:, 1:], cu, cu, s, s, 0.0, causal=causal)
        out3 = flash_attn_unpadded_func(flat[:, 0], flat[:, 1], flat[:, 2], cu, cu, s, s, 0.0, causal=causal)
        assert torch.equal(out, out2) and torch.equal(out, out3)


def test_flash_attention_module_with_padding_mask():
    from flash_attn.flash_attention import FlashAttention
    from flash_attn.modules.mha import FlashSelfAttention, SelfAttention
    torch.manual_seed(1)
    b, s, h, d = 3, 96, 2, 64
    qkv = torch.randn(b, s, 3, h, d).bfloat16()
    lens = torch.tensor([96, 40, 77])
    mask = torch.arange(s)[None, :] < lens[:, None]
    out, none = FlashAttention()(qkv.to(DEV), key_padding_mask=mask.to(DEV), causal=True)
    assert none is None
    want = R.attention_fp32(qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2], causal=True,
                            query_padding_mask=mask, key_padding_mask=mask)[0]
    assert (out.float().cpu() - want.float()).abs().max().item() < 2e-2
    assert torch.count_nonzero(out[1, 40:]) == 0            # padded rows come back as zeros
    fsa = FlashSelfAttention(causal=True, softmax_scale=0.05)(qkv.to(DEV))
    eager = SelfAttention(causal=True, softmax_scale=0.05)(qkv.float())
    assert (fsa.float().cpu() - eager).abs().max().item() < 2e-2


def test_autograd_through_the_forward_kernel():
    """Backward is eager recomputation (FA backward kernels are a 'next' row): gradients must match
    autograd through the eager twin."""
    from flash_attn.modules.mha import FlashSelfAttention, SelfAttention
    torch.manual_seed(2)
    qkv = (torch.randn(2, 64, 3, 2, 32) * 0.7).bfloat16().to(DEV).requires_grad_()
    out = FlashSelfAttention(causal=True)(qkv)
    g = torch.randn_like(out)
    (dqkv,) = torch.autograd.grad(out, qkv, g)
    ref_in = qkv.detach().float().requires_grad_()
    (dref,) = torch.autograd.grad(SelfAttention(causal=True)(ref_in), ref_in, g.float())
    assert (dqkv.float() - dref).abs().max().item() < 5e-2


def test_hip_graph_replay_matches_eager_launches():
    """The whole forward (torch ops + C-ABI launches) captured in a HIP graph replays bit-identically and
    follows new inputs through the static buffer."""
    import bp_hip
    from src.models.backpack import BackpackConfig, BackpackLMHeadModel
    cfg = BackpackConfig(n_embd=384, n_head=6, n_layer=2, num_content_vectors=16, vocab_size=1000,
                         n_positions=128, scale_attn_by_inverse_layer_idx=True, resid_pdrop=0.0,
                         embd_pdrop=0.0, attn_pdrop=0.0, use_flash_attn=True, pad_vocab_size_multiple=8,
                         fused_dropout_add_ln=True, fused_dense_gelu_dense=True, fused_bias_fc=True)
    torch.manual_seed(3)
    model = BackpackLMHeadModel(cfg).eval().to(DEV, torch.bfloat16)
    ids_a = torch.randint(0, 1000, (4, 128), device=DEV)
    ids_b = torch.randint(0, 1000, (4, 128), device=DEV)
    fwd = bp_hip.GraphedForward(model, ids_a)
    with torch.no_grad():
        want_a, want_b = model(ids_a).logits, model(ids_b).logits
    assert torch.equal(fwd(ids_a), want_a)
    assert torch.equal(fwd(ids_b), want_b)
    assert not torch.equal(want_a, want_b)


def test_intervened_models_hip_vs_oracle():
    """Mirror of intervened_models.py on the HIP path (fused key-weight hook, vocabulary-sized content) in
    bf16 against the fp32 oracle restatement, with the eager-bf16 error as yardstick."""
    from src.models import intervened_models as im
    from src.models.backpack import BackpackConfig, BackpackLMHeadModel
    g = load_golden('g6_interventions.npz')
    sd = {k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith('sd/')}
    ids = torch.from_numpy(g['ids'])
    cw = torch.from_numpy(g['content_weights'])
    senses = {int(w): torch.from_numpy(g['sense/%d' % w]) for w in g['sense_words']}
    scale = float(g['annealing_scale'])
    kw = dict(n_embd=64, n_head=2, n_layer=2, num_content_vectors=16, vocab_size=96, n_positions=32,
              scale_attn_by_inverse_layer_idx=True, resid_pdrop=0.0, embd_pdrop=0.0, attn_pdrop=0.0,
              pad_vocab_size_multiple=8)
    hip = BackpackLMHeadModel(BackpackConfig(use_flash_attn=True, **kw)).eval()
    hip.load_state_dict(sd, strict=True)
    hip = hip.to(DEV, torch.bfloat16)
    eager = BackpackLMHeadModel(BackpackConfig(use_flash_attn=False, **kw)).eval()
    eager.load_state_dict(sd, strict=True)
    eager = eager.to(torch.bfloat16)
    cases = [('weighted_anneal', im.WeightedBackpackLMHeadModel, dict(anneal=True)),
             ('weighted_plain', im.WeightedBackpackLMHeadModel, dict(anneal=False)),
             ('negative_anneal', im.NegativeWeightedBackpackLMHeadModel, dict(anneal=True)),
             ('negative_plain', im.NegativeWeightedBackpackLMHeadModel, dict(anneal=False))]
    with torch.no_grad():
        for name, cls, opt in cases:
            got = cls(hip, cw.to(DEV), torch.zeros(96), scale, **opt)(ids.to(DEV)).logits
            base = cls(eager, cw, torch.zeros(96), scale, **opt)(ids).logits
            want = torch.from_numpy(g[name])
            err = (got.float().cpu() - want).abs().max().item()
            ref = (base.float() - want).abs().max().item()
            print(f'{name}: hip {err:.3e} eager-bf16 {ref:.3e}')
            assert err <= 3 * ref + 2e-3, (name, err, ref)
        got = im.ReplacedWordLMHeadModel(hip, senses)(ids.to(DEV)).logits
        base = im.ReplacedWordLMHeadModel(eager, senses)(ids).logits
        want = torch.from_numpy(g['replaced'])
        err, ref = (got.float().cpu() - want).abs().max().item(), (base.float() - want).abs().max().item()
        print(f'replaced: hip {err:.3e} eager-bf16 {ref:.3e}')
        assert err <= 3 * ref + 2e-3


def test_bench_script_contract():
    """bench.py runs end to end (small workload) and prints ONE JSON line carrying the driver's contract keys,
    the roofline object of the dominant attention-path kernel and per-kernel HIP-event timings."""
    import json
    import os
    import subprocess
    import sys
    from conftest import ROOT
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--workload', 'micro-128', '--steps', '3',
                          '--warmup', '1', '--no-cpu-baseline'], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for key in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling',
                'vs_baseline', 'dtype', 'data', 'config', 'roofline', 'kernels'):
        assert key in d, key
    assert d['n_gpus'] == 1 and d['steps'] == 3 and d['value'] > 0 and d['vs_baseline'] is None
    assert set(d['roofline']) >= {'bound', 'achieved', 'peak', 'unit', 'frac', 'traffic', 'traffic_source'}
    assert d['config']['batch_per_gpu'] in [r['batch'] for r in d['batch_sweep']]   # --batch auto is the default
    assert {k['kernel'] for k in d['kernels']} >= {'flash_fwd_kernel', 'sense_mix_kernel', 'add_layer_norm_kernel'}


def test_bench_reports_all_three_content_orders():
    """At a batch with more positions than vocabulary entries (1024 x 128) the bench line's `value` is the step with the
    table of the batch's distinct tokens rebuilt per step, and the same process times the reference's per-position order
    and the cached whole-vocabulary table next to it; --content picks which one is the headline."""
    import json
    import os
    import subprocess
    import sys
    from conftest import ROOT
    base = [sys.executable, os.path.join(ROOT, 'bench.py'), '--workload', 'micro-128', '--batch', '1024', '--steps', '3',
            '--warmup', '1', '--no-cpu-baseline']
    out = subprocess.run(base, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith('{')][0])
    assert d['config']['content_network'].startswith('once per distinct token')
    for key in ('content_per_position', 'content_cached_table'):
        other = d[key]
        assert other['value'] > 0 and other['steps'] == 3 and other['batch_per_gpu'] == 1024 and other['unit'] == 'tokens/s'
    assert 'content_per_batch_table' not in d
    out = subprocess.run(base + ['--no-content-dedup'], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    d2 = json.loads([l for l in out.stdout.splitlines() if l.startswith('{')][0])
    assert d2['config']['content_network'].startswith('once per position') and 'content_per_position' not in d2
    assert d2['content_cached_table']['value'] > 0 and d2['content_per_batch_table']['value'] > 0
    out = subprocess.run(base + ['--content', 'cached', '--graph'], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    d3 = json.loads([l for l in out.stdout.splitlines() if l.startswith('{')][0])
    assert d3['config']['content_network'].startswith('whole-vocabulary') and d3['launch'] == 'hip-graph replay'


def test_greedy_generation_on_the_hip_path():
    """model.generate on the HIP path (the sequence grows by one token per step, so every call meets a new,
    unaligned sequence length).  EVERY step is checked: the token appended at position t must be the HIP
    model's own argmax on the prefix [0, t) and (within bf16 noise) a maximiser of the fp32 CPU model's logits
    on the same prefix.  Returned length = max_length - 1, the reference's contract
    (training/src/utils/generation.py:64-72, fixture g8_generation.npz)."""
    from src.models.backpack import BackpackConfig, BackpackLMHeadModel
    kw = dict(n_embd=128, n_head=2, n_layer=2, num_content_vectors=4, vocab_size=512, n_positions=48,
              scale_attn_by_inverse_layer_idx=True, resid_pdrop=0.0, embd_pdrop=0.0, attn_pdrop=0.0,
              pad_vocab_size_multiple=8)
    torch.manual_seed(21)
    ref = BackpackLMHeadModel(BackpackConfig(use_flash_attn=False, **kw)).eval()
    hip = BackpackLMHeadModel(BackpackConfig(use_flash_attn=True, fused_dropout_add_ln=True, **kw)).eval()
    hip.load_state_dict(ref.state_dict())
    hip = hip.to(DEV, torch.bfloat16)
    ids = torch.randint(0, 512, (2, 5))
    out = hip.generate(ids.to(DEV), max_length=20, return_dict_in_generate=True, output_scores=True)
    seq = out.sequences.cpu()
    assert seq.shape == (2, 19) and torch.equal(seq[:, :5], ids)
    assert len(out.scores) == 1                                  # upstream keeps the first step's scores only
    for t in range(5, 19):
        prefix = seq[:, :t]
        with torch.no_grad():
            want = ref(prefix).logits[:, -1]                     # fp32 CPU model, same prefix
            got = hip(prefix.to(DEV)).logits[:, -1].float().cpu()
        assert (got - want).abs().max().item() < 0.15, t
        assert torch.equal(got.argmax(-1), seq[:, t]), t         # the appended token is the HIP argmax ...
        chosen = want.gather(1, seq[:, t:t + 1]).squeeze(1)
        assert (want.max(-1).values - chosen).max().item() < 0.3, t   # ... and a (near-)maximiser in fp32
    assert (out.scores[0].float().cpu() - ref(ids).logits[:, -1]).abs().max().item() < 0.15


def test_graph_replay_decoding_gives_the_eager_tokens():
    """generate(..., cg=True): one captured full-width forward replayed per token.  Same length contract, same first-step
    scores, and the same tokens as the growing-prefix loop -- a token may differ only where the eager HIP logits of the
    two candidates tie to within bf16 noise (a differently tiled GEMM at the other sequence length); batch 1 and 3,
    greedy, prompt longer than max_length - 1 included.  Also: replaying is not slower than the eager loop."""
    import time
    from src.models.backpack import BackpackConfig, BackpackLMHeadModel
    kw = dict(n_embd=128, n_head=2, n_layer=2, num_content_vectors=4, vocab_size=512, n_positions=64,
              scale_attn_by_inverse_layer_idx=True, resid_pdrop=0.0, embd_pdrop=0.0, attn_pdrop=0.0,
              pad_vocab_size_multiple=8)
    torch.manual_seed(31)
    hip = BackpackLMHeadModel(BackpackConfig(use_flash_attn=True, fused_dropout_add_ln=True, **kw)).eval()
    hip = hip.to(DEV, torch.bfloat16)
    for batch, plen, max_length in ((1, 5, 40), (3, 7, 24), (2, 9, 6)):
        ids = torch.randint(0, 512, (batch, plen)).to(DEV)
        eager = hip.generate(ids, max_length=max_length, return_dict_in_generate=True, output_scores=True)
        graphed = hip.generate(ids, max_length=max_length, return_dict_in_generate=True, output_scores=True, cg=True)
        assert graphed.sequences.shape == eager.sequences.shape == (batch, max(plen, max_length - 1))
        assert len(graphed.scores) == 1
        assert (graphed.scores[0].float() - eager.scores[0].float()).abs().max().item() < 0.1
        a, b = eager.sequences.cpu(), graphed.sequences.cpu()
        for row in range(batch):
            if torch.equal(a[row], b[row]):
                continue
            t = int((a[row] != b[row]).nonzero()[0])             # first divergence: must be a near tie
            with torch.no_grad():
                logits = hip(a[row:row + 1, :t].to(DEV)).logits[0, -1].float().cpu()
            assert abs(logits[a[row, t]] - logits[b[row, t]]).item() < 0.1, (batch, row, t)
    ids = torch.randint(0, 512, (1, 4)).to(DEV)
    times = {}
    for cg in (False, True):
        hip.generate(ids, max_length=48, cg=cg)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        hip.generate(ids, max_length=48, cg=cg)
        torch.cuda.synchronize()
        times[cg] = time.perf_counter() - t0
    print(f'decode 44 tokens, nano model: eager {times[False] * 1e3:.1f} ms, graph replay {times[True] * 1e3:.1f} ms (capture included)')


def test_generation_on_the_hip_path_matches_the_reference_tokens():
    """G8 on the GPU: the nano model of G4 on the HIP path continues the prompt with the reference's tokens.
    bf16 logits can flip a near tie, so a differing token is accepted only where the fp32 golden margin
    between the two candidates is below bf16 resolution -- and the length contract is exact."""
    g4, g8 = load_golden('g4_nano_model.npz'), load_golden('g8_generation.npz')
    hip, ref = _nano(True)[2], _nano(False, dtype=torch.float32)[2]
    prompt = torch.from_numpy(g8['prompt'])
    for max_length in (6, 12, 20):
        seq = hip.generate(prompt.to(DEV), max_length=max_length).cpu()
        want = torch.from_numpy(g8['greedy_%d' % max_length])
        assert seq.shape == want.shape == (1, max_length - 1)
        if not torch.equal(seq, want):
            t = int((seq != want).nonzero()[0, 1])
            with torch.no_grad():
                logits = ref(want[:, :t].to(ref.lm_head.weight.device)).logits[0, -1].float().cpu()
            assert abs(logits[seq[0, t]] - logits[want[0, t]]).item() < 0.05, (max_length, t)


@pytest.mark.parametrize('fused', [False, True])
def test_block_with_stochastic_depth_against_the_fp32_eager_twin(fused):
    """drop_path > 0 (reference flash_attn/modules/block.py:82-90,96-105): a sample's branch is dropped or scaled by
    1 / (1 - p); with fused_dropout_add_ln the factor enters the fused add + LayerNorm kernel as `rowscale`.  Forward
    and backward of the 16-bit block against an fp32 copy running the unfused op sequence with the SAME drop decisions
    (same seed: both draw one Bernoulli number per sample and branch, in the same order)."""
    from functools import partial
    import torch.nn as nn
    from flash_attn.modules.block import Block, StochasticDepth
    from flash_attn.modules.mlp import Mlp
    from src.models.backpack import Identity
    dim, b, s, p_drop = 256, 12, 40, 0.4

    def make(dtype, fused_ln):
        torch.manual_seed(5)
        blk = Block(dim, Identity, partial(Mlp, hidden_features=2 * dim, activation=partial(nn.functional.gelu, approximate='tanh')),
                    norm_cls=partial(nn.LayerNorm, eps=1e-5), prenorm=True, resid_dropout=0.0, drop_path=p_drop,
                    fused_dropout_add_ln=fused_ln)
        return blk.to(DEV, dtype).train()

    ref, blk = make(torch.float32, False), make(torch.bfloat16, fused)
    assert isinstance(blk.drop_path1, StochasticDepth) and blk.drop_path2.p == p_drop
    g = torch.Generator(device=DEV).manual_seed(3)
    x = torch.randn(b, s, dim, device=DEV, generator=g)
    res = torch.randn(b, s, dim, device=DEV, generator=g)
    dz = torch.randn(b, s, dim, device=DEV, generator=g)
    outs = []
    for m, dt in ((ref, torch.float32), (blk, torch.bfloat16), (make(torch.bfloat16, False), torch.bfloat16)):
        xi = x.detach().to(dt).clone().requires_grad_()
        ri = res.detach().clone().requires_grad_()          # the residual stream is fp32 in every variant
        torch.manual_seed(11)
        h, r = m(xi, ri)
        (h.float() * dz + r.float() * dz).sum().backward()
        outs.append([t.float() for t in (h, r, xi.grad, ri.grad, m.norm2.weight.grad, m.mlp.fc1.weight.grad)])
    want, got, eager16 = outs
    # the drop decisions took effect: some samples' residual is the input residual plus x exactly, others scaled
    kept = (want[1] - res - x).flatten(1).abs().amax(1)
    assert (kept > 1e-3).any()
    for name, w, gt, e in zip(('hidden', 'residual', 'dx', 'dres', 'dgamma2', 'dW1'), want, got, eager16):
        err, base = (gt - w).abs().max().item(), (e - w).abs().max().item()
        print(f'{name}: hip {err:.3e} eager-bf16 {base:.3e}')
        assert err <= 2 * base + 2e-3 * w.abs().max().item() + 1e-5, (name, err, base)
    # eval: identity on the branch, no randomness
    blk.eval()
    with torch.no_grad():
        a = blk(x.bfloat16(), res)[0]
        c = blk(x.bfloat16(), res)[0]
    assert torch.equal(a, c)


def test_per_position_content_in_sample_chunks():
    """sense_table = 'off' (the reference's order of operations) without an autograd graph runs the content network and
    the mix over chunks of samples (BackpackModel._mix_per_position_chunked): the (B,S,k*d) content tensor exists for one
    chunk at a time.  A chunk size that does not divide the batch against the whole-batch call: the mix kernel's
    arithmetic is per sample, so rows may differ only by how the BLAS GEMMs round a row at another row count."""
    g, sd, model = _nano(fused=True)
    t = model.transformer
    t.sense_table_mode = 'off'
    ids = torch.randint(0, 96, (7, 32), device=DEV, generator=torch.Generator(device=DEV).manual_seed(0))
    with torch.no_grad():
        t.config.content_chunk_positions = 0                     # never chunk
        assert not t._chunked_content_applies(ids)
        whole = t(ids)
        t.config.content_chunk_positions = 3 * 32                # 3 samples per chunk: 3 + 3 + 1
        assert t._chunked_content_applies(ids) and t._content_chunk_samples(32) == 3
        chunked = t(ids)
        t.config.content_chunk_positions = 7 * 32                # one chunk = the whole batch: the ordinary path
        assert not t._chunked_content_applies(ids)
        assert torch.equal(t(ids), whole)
    scale = whole.float().abs().max().item()
    assert (chunked.float() - whole.float()).abs().max().item() <= 2 ** -7 * scale
    # sample by sample the chunked call IS the model on that chunk alone
    with torch.no_grad():
        t.config.content_chunk_positions = 0
        assert torch.equal(t(ids[3:6]), chunked[3:6]) or \
            (t(ids[3:6]).float() - chunked[3:6].float()).abs().max().item() <= 2 ** -7 * scale
    # with autograd the whole-batch path stays
    t.config.content_chunk_positions = 32
    with torch.enable_grad():
        assert not t._chunked_content_applies(ids)

"""-m gpu: seeded random-shape sweeps of the hot-path kernels against the oracle.

The fixed sweeps (tests/test_gpu_kernels.py, test_gpu_backward.py, test_gpu_configs.py) walk the reference's own
parameter grids (tests/test_flash_attn.py:350-373,441-461); the cases here are drawn instead: ragged batches with
independent query / key lengths (1 ... 700, empty key sequences included), any head dim the entry points take (8 ... 128 in
steps of 8), 1 ... 5 heads, causal or not, both 16-bit types, a random trunk-layer scale; for the sense kernels 1 ... 20
senses, d_k from 8 to 64 and the unaligned 10 / 20 / 12, any output width that is a multiple of 8 up to 800, lengths 1 ... 600,
dense and table (gather) form.  Every case is a fixed function of its seed, so a failure names its shape.  The checker is
the oracle (oracle/ref_cpu.py) in fp32 on the CPU under the reference's rule: error <= 2 x the error of the same-dtype eager
op sequence (+ two 16-bit rounding units of the result's range).  (On the CPU on purpose: run on the GPU, the oracle's own
autograd -- library GEMMs on odd, transposed shapes -- ended a 3000-case hunt with an illegal memory access inside a
torch kernel, scripts/debug/r05_fuzz_trace.py.)
"""
import os
import random

import pytest
import torch

from oracle import ref_cpu as R

pytestmark = pytest.mark.gpu
DEV = 'cuda'
SEEDS = range(int(os.environ.get('BP_FUZZ_SEEDS', '24')))      # (a longer hunt: BP_FUZZ_SEEDS=2000 python -m pytest ...)


def _bp():
    import bp_hip
    return bp_hip


def _close(got, ref32, eager, name, factor=2.0, floor=2.0, floor_range=0.0):
    dtype = got.dtype
    got, ref32, eager = got.float().cpu(), ref32.float().cpu(), eager.float().cpu()
    assert torch.isfinite(got).all(), name
    err = (got - ref32).abs().max().item() if got.numel() else 0.0
    base = (eager - ref32).abs().max().item() if got.numel() else 0.0
    # floor: two units of 16-bit rounding at the result's range (tiny drawn cases -- two keys, one head -- leave the eager
    # yardstick at zero or one ulp, where "twice the eager error" says nothing)
    eps = 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11
    ulp = floor * eps * max(ref32.abs().max().item() if got.numel() else 0.0, floor_range)
    assert err <= factor * base + ulp + 1e-5, f'{name}: {err:.3e} > {factor} x {base:.3e} + {ulp:.1e}'


def _grads(q, k, v, dout, causal, scale, upcast):
    """dq, dk, dv of the oracle's attention for one sequence (1,S,H,D): autograd through oracle/ref_cpu.attention_fp32."""
    q, k, v = (x.detach().clone().requires_grad_(True) for x in (q, k, v))
    out = R.attention_fp32(q, k, v, causal=causal, softmax_scale=scale, upcast=upcast, reorder_ops=not upcast)[0]
    out.backward(dout.to(out.dtype))
    return q.grad, k.grad, v.grad


@pytest.mark.parametrize('seed', SEEDS)
def test_flash_forward_and_backward_on_drawn_ragged_batches(seed):
    bp = _bp()
    rnd = random.Random(1000 + seed)
    g = torch.Generator(device=DEV).manual_seed(seed)
    dtype = rnd.choice([torch.bfloat16, torch.float16])
    causal = rnd.random() < 0.6
    h, d = rnd.randint(1, 5), 8 * rnd.randint(1, 16)
    nb = rnd.randint(1, 5)
    lens_q = [rnd.choice([1, 2, 31, 32, 33, 63, 64, 65, 127, 128, 129, rnd.randint(1, 700)]) for _ in range(nb)]
    same = rnd.random() < 0.5
    lens_k = list(lens_q) if same else [rnd.choice([0, 1, 64, 65, rnd.randint(1, 700)]) for _ in range(nb)]
    if causal and not same:
        lens_k = [max(a, c) for a, c in zip(lens_q, lens_k)]      # (causal is top-left aligned; keep every row a key)
    if sum(lens_k) == 0:
        lens_k[0] = 1                                             # (a batch without any key row is refused: null pointer)
    scale = d ** -0.5 / rnd.choice([1, 1, 2, 7, 12])
    name = f'seed {seed}: {dtype} causal={causal} h={h} d={d} lens_q={lens_q} lens_k={lens_k}'
    q = torch.randn(sum(lens_q), h, d, device=DEV, generator=g).to(dtype)
    dout = torch.randn(sum(lens_q), h, d, device=DEV, generator=g).to(dtype)
    k = torch.randn(max(sum(lens_k), 1), h, d, device=DEV, generator=g).to(dtype)[:sum(lens_k)]
    v = torch.randn(max(sum(lens_k), 1), h, d, device=DEV, generator=g).to(dtype)[:sum(lens_k)]
    cu_q = torch.tensor([0] + lens_q, device=DEV).cumsum(0).to(torch.int32)
    cu_k = torch.tensor([0] + lens_k, device=DEV).cumsum(0).to(torch.int32)
    out = torch.full_like(q, float('nan'))
    lse = bp.flash_fwd(q, k, v, out, cu_q, cu_k, max(lens_q), max(max(lens_k), 1), scale, causal)
    dq, dk, dv = torch.full_like(q, float('nan')), torch.full_like(k, float('nan')), torch.full_like(v, float('nan'))
    bp.flash_bwd(dout, q, k, v, out, lse, dq, dk, dv, cu_q, cu_k, max(lens_q), max(max(lens_k), 1), scale, causal)
    for i in range(nb):
        qs, qe, ks, ke = (int(x) for x in (cu_q[i], cu_q[i + 1], cu_k[i], cu_k[i + 1]))
        if ke == ks:          # no key: zero output, -inf LSE, zero dq (fmha_fprop_kernel_1xN.h:592-596)
            assert torch.count_nonzero(out[qs:qe]) == 0 and torch.count_nonzero(dq[qs:qe]) == 0, name
            assert torch.isinf(lse[i, :, :qe - qs]).all() and (lse[i, :, :qe - qs] < 0).all(), name
            continue
        args = (q[None, qs:qe].cpu(), k[None, ks:ke].cpu(), v[None, ks:ke].cpu())
        ref, _, lse_ref = R.attention_fp32(*(x.float() for x in args), causal=causal, softmax_scale=scale)
        eager = R.attention_fp32(*args, causal=causal, softmax_scale=scale, upcast=False, reorder_ops=True)[0]
        _close(out[qs:qe], ref[0], eager[0], name + f' out[{i}]')
        assert (lse[i, :, :qe - qs].cpu() - lse_ref[0]).abs().max().item() < 4e-3, name + f' lse[{i}]'
        gref = _grads(*args, dout[None, qs:qe].cpu(), causal, scale, True)
        geag = _grads(*args, dout[None, qs:qe].cpu(), causal, scale, False)
        for got, r, e, what in ((dq[qs:qe], gref[0], geag[0], 'dq'), (dk[ks:ke], gref[1], geag[1], 'dk'),
                                (dv[ks:ke], gref[2], geag[2], 'dv')):
            # (a handful of keys: dS = P (dP - D) cancels, the error scales with dP ~ |dO| |v|, not with the gradient)
            short = min(qe - qs, ke - ks) < 32
            _close(got, r[0], e[0], name + f' {what}[{i}]', floor=8.0 if short else 2.0, floor_range=1.0 if short else 0.0)


@pytest.mark.parametrize('seed', SEEDS)
def test_sense_kernels_on_drawn_shapes(seed):
    bp = _bp()
    rnd = random.Random(2000 + seed)
    g = torch.Generator(device=DEV).manual_seed(seed)
    dtype = rnd.choice([torch.bfloat16, torch.float16])
    b = rnd.randint(1, 3)
    s = rnd.choice([1, 2, 31, 33, 63, 64, 65, 255, 256, 257, rnd.randint(1, 600), rnd.randint(1, 600)])
    k = rnd.randint(1, 20)
    dk = rnd.choice([8, 16, 24, 32, 40, 48, 56, 64, 10, 20, 12])
    if rnd.random() < 0.2 or os.environ.get('BP_FUZZ_RING') == '1':   # wide senses (round 6, csrc/sense_wide.hip): few of them, 129 ... 640 wide, unaligned ones too
        k, dk = rnd.randint(1, 4), rnd.choice([136, 160, 192, 200, 320, 636, 640, 130, 250])
        if rnd.random() < 0.5 or os.environ.get('BP_FUZZ_RING') == '1':
            # the two widths of the reference's few-sense configs at a length that is a multiple of 32: the LDS-DMA ring
            # kernels of csrc/sense_wide_dma.hip (dense and table form); BP_FUZZ_RING=1 makes a hunt draw only these
            dk, s = rnd.choice([160, 640]), 32 * rnd.randint(1, 24)
        qk_amp = 1.3 * (64 / dk) ** 0.25
    else:
        qk_amp = 1.3
    d = 8 * rnd.randint(1, 100)
    name = f'seed {seed}: {dtype} b={b} s={s} k={k} dk={dk} d={d}'
    qk = (qk_amp * torch.randn(b, s, 2, k, dk, device=DEV, generator=g)).to(dtype)
    c = torch.randn(b, s, k, d, device=DEV, generator=g).to(dtype)
    scale = dk ** -0.5
    qk_h, c_h = qk.cpu(), c.cpu()
    alpha32 = R.sense_alpha_from_qk(qk_h.float())
    want = R.sense_mix(alpha32, c_h.float().transpose(1, 2))
    alpha16 = R.sense_alpha_from_qk(qk_h)
    eager = R.sense_mix(alpha16, c_h.transpose(1, 2))
    out = bp.sense_mix(qk, c)
    _close(out, want, eager, name + ' mix')
    q32, k32 = qk_h[:, :, 0].float(), qk_h[:, :, 1].float()
    _, _, lse_ref = R.attention_fp32(q32, k32, None, causal=True, softmax_scale=scale)
    lse = bp.sense_lse(qk)
    assert (lse[:, :, :s].cpu() - lse_ref).abs().max().item() < 4e-3, name + ' lse'
    alpha = bp.sense_alpha(qk)
    _close(alpha, alpha32, alpha16, name + ' alpha')
    upper = torch.triu(torch.ones(s, s, dtype=torch.bool, device=DEV), 1)
    assert torch.count_nonzero(alpha[:, :, upper]) == 0, name
    # table form: a table with repeated and unused rows
    rows = rnd.randint(1, 300)
    table = torch.randn(rows, k, d, device=DEV, generator=g).to(dtype)
    idx = torch.randint(0, rows, (b, s), device=DEV, generator=g, dtype=torch.int32)
    if bp.sense_mix_gather_supported(qk, table, s):
        got = bp.sense_mix_gather(qk, table, idx)
        assert torch.equal(got, bp.sense_mix(qk, table[idx.long()])), name + ' gather'
    # backward of the contraction (fused kernels where they apply, see bp_hip.SenseMixFn)
    if dk % 8 == 0:
        dout = torch.randn(b, s, d, device=DEV, generator=g).to(dtype)
        qk_g, c_g = qk.clone().requires_grad_(True), c.clone().requires_grad_(True)
        bp.sense_mix_autograd(qk_g, c_g).backward(dout)
        grads = {}
        for tag, cast in (('ref', torch.float32), ('eager', dtype)):
            qk_r, c_r = qk_h.to(cast).requires_grad_(True), c_h.to(cast).requires_grad_(True)
            R.sense_mix(R.sense_alpha_from_qk(qk_r), c_r.transpose(1, 2)).backward(dout.cpu().to(cast))
            grads[tag] = (qk_r.grad, c_r.grad)
        short = s < 32
        _close(qk_g.grad, grads['ref'][0], grads['eager'][0], name + ' dqk', factor=3.0, floor=8.0 if short else 2.0,
               floor_range=1.0 if short else 0.0)
        _close(c_g.grad, grads['ref'][1], grads['eager'][1], name + ' dC', factor=3.0)


@pytest.mark.parametrize('seed', range(int(os.environ.get('BP_FUZZ_MODELS', '10'))))
def test_whole_model_on_drawn_configurations(seed):
    """ids -> hidden states -> logits of Backpack models whose dimensions are drawn (width, heads, layers, senses --
    including widths whose d_k = d / k is not a multiple of 8, e.g. 10, 12, 20 --, `shrink_final_inner`, vocabulary,
    positions, batch, length, dtype, the reference's fused-flag set on or off), on the HIP path, against the oracle's fp32
    forward of the same state dict; the reference's model-test rule: error <= 3 x the error of the same model in eager
    16 bit (+1e-3); and the three content orders of the inference forward agree with each other to that accuracy."""
    from src.models.backpack import BackpackConfig, BackpackLMHeadModel
    rnd = random.Random(3000 + seed)
    dtype = rnd.choice([torch.bfloat16, torch.float16])
    dh = rnd.choice([16, 32, 40, 64, 80])
    nh = rnd.randint(1, 4)
    d = dh * nh
    k = rnd.choice([x for x in (1, 2, 4, 5, 8, 16) if d % x == 0])
    ocfg = dict(n_embd=d, n_head=nh, n_layer=rnd.randint(1, 3), num_content_vectors=k,
                shrink_final_inner=rnd.random() < 0.5, n_positions=rnd.choice([64, 130, 257]),
                vocab_size=8 * rnd.randint(8, 60), layer_norm_epsilon=1e-5, scale_attn_by_inverse_layer_idx=True)
    b, s = rnd.randint(1, 3), rnd.randint(1, ocfg['n_positions'])
    fused = rnd.random() < 0.7
    name = f'seed {seed}: {dtype} d={d} heads={nh} k={k} (d_k={d // k}) {ocfg} b={b} s={s} fused={fused}'
    sd = R.init_state_dict(ocfg, seed=seed)
    with torch.no_grad():
        sd['transformer.contextualization_attn.Wqkv.weight'].mul_(8.0)
        for i in range(ocfg['n_layer']):
            sd[f'transformer.gpt2_model.layers.{i}.mixer.Wqkv.weight'].mul_(6.0)
        sd = {key: v.to(dtype).float() for key, v in sd.items()}        # 16-bit-exact weights for all three runs
    sd['lm_head.weight'] = sd['transformer.gpt2_model.embeddings.word_embeddings.weight']
    ids = torch.randint(0, ocfg['vocab_size'], (b, s), generator=torch.Generator().manual_seed(seed))
    with torch.no_grad():
        want = R.backpack_forward(sd, ocfg, ids, return_stages=True)

    def build(use_flash, fused_):
        cfg = BackpackConfig(n_embd=d, n_head=nh, n_layer=ocfg['n_layer'], num_content_vectors=k,
                             vocab_size=ocfg['vocab_size'], n_positions=ocfg['n_positions'],
                             scale_attn_by_inverse_layer_idx=True, shrink_final_inner=ocfg['shrink_final_inner'],
                             resid_pdrop=0.0, embd_pdrop=0.0, attn_pdrop=0.0, use_flash_attn=use_flash,
                             fused_dropout_add_ln=fused_, fused_dense_gelu_dense=fused_, fused_bias_fc=fused_,
                             pad_vocab_size_multiple=8)
        m = BackpackLMHeadModel(cfg)
        res = m.load_state_dict(sd, strict=False)
        assert not res.unexpected_keys and all('embeddings' in x for x in res.missing_keys), (name, res)
        m.tie_weights()
        return m.to(DEV, dtype).eval()

    hip, eager = build(True, fused), build(False, False)
    dev_ids = ids.to(DEV)
    with torch.no_grad():
        base_h = eager.transformer(dev_ids)
        base_l = eager.lm_head(base_h)
        results = {}
        for mode in ('off', 'cached', 'batch'):
            hip.transformer.sense_table_mode = mode
            h = hip.transformer(dev_ids)
            results[mode] = (h, hip.lm_head(h))
    for mode, (h, logits) in results.items():
        for got, base, ref, what in ((h, base_h, want['hidden'], 'hidden'), (logits, base_l, want['logits'], 'logits')):
            err = (got.float().cpu() - ref).abs().max().item()
            yard = (base.float().cpu() - ref).abs().max().item()
            assert torch.isfinite(got.float()).all(), name
            assert err <= 3 * yard + 1e-3, f'{name} [{mode}] {what}: {err:.3e} > 3 x {yard:.3e}'


@pytest.mark.parametrize('seed', range(int(os.environ.get('BP_FUZZ_TRAIN', '8'))))
def test_training_step_on_drawn_configurations(seed):
    """One training step (forward under 16-bit autocast with fp32 parameters -- the reference's recipe, training/configs/
    experiment/owt/base.yaml -- loss, backward) of drawn Backpack models on the HIP path, dropout off: the loss and EVERY
    parameter's gradient against fp32 autograd of the eager twin (use_flash_attn and the fused flags off, the op sequence
    the CPU tests pin against the oracle), measured with the twin's own 16-bit autocast step: error <= 4 x its error."""
    import warnings
    from src.models.backpack import BackpackConfig, BackpackLMHeadModel
    rnd = random.Random(4000 + seed)
    amp = rnd.choice([torch.bfloat16, torch.float16])
    dh = rnd.choice([16, 32, 40, 64, 80, 128])
    nh = rnd.randint(1, 3)
    d = dh * nh
    k = rnd.choice([x for x in (1, 2, 4, 5, 8, 16, 32) if d % x == 0])
    kw = dict(n_embd=d, n_head=nh, n_layer=rnd.randint(1, 3), num_content_vectors=k, vocab_size=8 * rnd.randint(8, 60),
              n_positions=rnd.choice([64, 130, 257]), scale_attn_by_inverse_layer_idx=True,
              shrink_final_inner=rnd.random() < 0.5, resid_pdrop=0.0, embd_pdrop=0.0, attn_pdrop=0.0,
              pad_vocab_size_multiple=8)
    b, s = rnd.randint(1, 3), rnd.randint(2, kw['n_positions'])
    fused = rnd.random() < 0.7
    name = f'seed {seed}: amp={amp} d_k={d // k} {kw} b={b} s={s} fused={fused}'
    torch.manual_seed(seed)
    ref = BackpackLMHeadModel(BackpackConfig(use_flash_attn=False, **kw)).to(DEV).float()
    with torch.no_grad():
        ref.transformer.contextualization_attn.Wqkv.weight.mul_(6.0)
        for layer in ref.transformer.gpt2_model.layers:
            layer.mixer.Wqkv.weight.mul_(4.0)
    sd = ref.state_dict()
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        hip = BackpackLMHeadModel(BackpackConfig(use_flash_attn=True, fused_dropout_add_ln=fused, fused_dense_gelu_dense=fused,
                                                 fused_bias_fc=fused, **kw)).to(DEV).float()
    eager = BackpackLMHeadModel(BackpackConfig(use_flash_attn=False, **kw)).to(DEV).float()
    hip.load_state_dict(sd)
    eager.load_state_dict(sd)
    ids = torch.randint(0, kw['vocab_size'], (b, s), device=DEV, generator=torch.Generator(device=DEV).manual_seed(seed))
    labels = torch.roll(ids, -1, 1)

    def step(model, autocast):
        model.zero_grad(set_to_none=True)
        with torch.autocast('cuda', dtype=amp, enabled=autocast):
            logits = model(ids).logits
        loss = torch.nn.functional.cross_entropy(logits.float().flatten(0, 1), labels.flatten())
        loss.backward()
        return loss.item(), {n: p.grad.float().clone() for n, p in model.named_parameters() if p.grad is not None}

    l_ref, g_ref = step(ref.train(), False)
    l_hip, g_hip = step(hip.train(), True)
    l_eag, g_eag = step(eager.train(), True)
    assert abs(l_hip - l_ref) <= 4 * abs(l_eag - l_ref) + 2e-3, (name, l_hip, l_eag, l_ref)
    assert set(g_hip) == set(g_ref), name
    for n in g_ref:
        err = (g_hip[n] - g_ref[n]).abs().max().item()
        base = (g_eag[n] - g_ref[n]).abs().max().item()
        assert torch.isfinite(g_hip[n]).all(), (name, n)
        floor = (2.0 ** -8 if amp == torch.bfloat16 else 2.0 ** -11) * max(1e-2, g_ref[n].abs().max().item())
        assert err <= 4 * base + floor, f'{name} grad {n}: {err:.3e} > 4 x {base:.3e} + {floor:.1e}'


@pytest.mark.parametrize('seed', range(int(os.environ.get('BP_FUZZ_INTERVENE', '8'))))
def test_intervened_models_on_drawn_configurations(seed):
    """The paper's control experiments (training/src/models/intervened_models.py) on drawn Backpack configurations, few-sense
    ones (eager sense path) and padded sense widths included: per-token sense re-weighting through the kernel's key-weight
    hook, with and without annealing; the negative re-weighting on vocabulary-sized content; replaced sense vectors -- HIP
    path in 16 bit against the oracle's fp32 restatements, 3 x the eager 16-bit twin's error (+2e-3)."""
    import warnings
    from src.models import intervened_models as im
    from src.models.backpack import BackpackConfig, BackpackLMHeadModel
    rnd = random.Random(5000 + seed)
    dtype = rnd.choice([torch.bfloat16, torch.float16])
    dh = rnd.choice([16, 32, 40, 64, 80])
    nh = rnd.randint(1, 3)
    d = dh * nh
    k = rnd.choice([x for x in (1, 2, 4, 5, 8, 16) if d % x == 0])
    vocab = 8 * rnd.randint(8, 40)
    ocfg = dict(n_embd=d, n_head=nh, n_layer=rnd.randint(1, 2), num_content_vectors=k, shrink_final_inner=rnd.random() < 0.5,
                n_positions=rnd.choice([32, 70, 130]), vocab_size=vocab, layer_norm_epsilon=1e-5,
                scale_attn_by_inverse_layer_idx=True)
    b, s = rnd.randint(1, 3), rnd.randint(2, ocfg['n_positions'])
    name = f'seed {seed}: {dtype} d_k={d // k} {ocfg} b={b} s={s}'
    g = torch.Generator().manual_seed(seed)
    sd = {key: v.to(dtype).float() for key, v in R.init_state_dict(ocfg, seed=seed).items()}
    sd['transformer.contextualization_attn.Wqkv.weight'] = sd['transformer.contextualization_attn.Wqkv.weight'] * 4.0
    sd['lm_head.weight'] = sd['transformer.gpt2_model.embeddings.word_embeddings.weight']
    ids = torch.randint(0, vocab, (b, s), generator=g)
    cw = (torch.rand(vocab, k, generator=g) * 2.0).to(dtype).float()          # (vocab, senses) content weights
    words = sorted({int(x) for x in ids.flatten()[:3]})
    senses = {w: torch.randn(k, d, generator=g).to(dtype).float() * 0.1 for w in words}
    scale = 0.1

    def build(use_flash):
        cfg = BackpackConfig(n_embd=d, n_head=nh, n_layer=ocfg['n_layer'], num_content_vectors=k, vocab_size=vocab,
                             n_positions=ocfg['n_positions'], scale_attn_by_inverse_layer_idx=True,
                             shrink_final_inner=ocfg['shrink_final_inner'], resid_pdrop=0.0, embd_pdrop=0.0, attn_pdrop=0.0,
                             use_flash_attn=use_flash, pad_vocab_size_multiple=8)
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            m = BackpackLMHeadModel(cfg)
        res = m.load_state_dict(sd, strict=False)
        assert not res.unexpected_keys, (name, res)
        m.tie_weights()
        return m.eval()

    hip = build(True).to(DEV, dtype)
    eager = build(False).to(dtype)
    cases = [('weighted, annealed', im.WeightedBackpackLMHeadModel, R.weighted_backpack_logits, dict(anneal=True)),
             ('weighted', im.WeightedBackpackLMHeadModel, R.weighted_backpack_logits, dict(anneal=False)),
             ('negative, annealed', im.NegativeWeightedBackpackLMHeadModel, R.negative_weighted_backpack_logits, dict(anneal=True)),
             ('negative', im.NegativeWeightedBackpackLMHeadModel, R.negative_weighted_backpack_logits, dict(anneal=False))]
    with torch.no_grad():
        for what, cls, oracle, opt in cases:
            want = oracle(sd, ocfg, ids, cw, annealing_scale=scale, **opt)
            got = cls(hip, cw.to(DEV), torch.zeros(vocab), scale, **opt)(ids.to(DEV)).logits
            base = cls(eager, cw, torch.zeros(vocab), scale, **opt)(ids).logits
            err = (got.float().cpu() - want).abs().max().item()
            yard = (base.float() - want).abs().max().item()
            assert torch.isfinite(got.float()).all(), (name, what)
            assert err <= 3 * yard + 2e-3, f'{name} [{what}]: {err:.3e} > 3 x {yard:.3e}'
        want = R.replaced_word_logits(sd, ocfg, ids, senses)
        got = im.ReplacedWordLMHeadModel(hip, senses)(ids.to(DEV)).logits
        base = im.ReplacedWordLMHeadModel(eager, senses)(ids).logits
        err, yard = (got.float().cpu() - want).abs().max().item(), (base.float() - want).abs().max().item()
        assert err <= 3 * yard + 2e-3, f'{name} [replaced]: {err:.3e} > 3 x {yard:.3e}'

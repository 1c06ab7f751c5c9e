"""-m gpu: size-independent properties of the hot path at BASELINE.json's real shapes.

The oracle finishes in seconds only at two samples per configuration (tests/test_gpu_configs.py); at the sizes the
bench runs, the path is checked through properties that need no oracle and hold BIT FOR BIT, because they are statements
about indexing, masks and work distribution, not about rounding:
  * causality -- nothing at or before position t may depend on what lies behind t (the reference's causal mask,
    csrc/flash_attn/src/fmha/mask.h:57-70; training/src/models/backpack.py:116-122): the rows behind t are replaced by
    other values -- of the same scale, and 6x LOUDER ones, which drive the softmax kernels into their overflow-retry
    branch: whether a wave (32 queries) repeats a tile is decided on all its rows, but a row whose own sums passed keeps
    its reference point in the repeat (csrc/flash_fwd_dma.hip `online_max_step`), so its bits do not depend on its
    wave-mates -- a key that leaks through a mask or an index that runs past t changes bits, nothing else may;
  * sample independence -- permuting the samples of a batch permutes the result (every kernel here maps samples to
    workgroups differently: XCD-local groups in the attention kernels, per-XCD ticket queues in the sense mix);
  * determinism -- the same call twice gives the same bits (no atomics on the data path).
"""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def _bp():
    import bp_hip
    return bp_hip


def _flash(qkv, scale):
    """qkv (B,S,3,H,D) -> out (B,S,H,D), lse (B,H,S); causal, fixed length (the trunk's call)."""
    bp = _bp()
    b, s, _, h, d = qkv.shape
    flat = qkv.reshape(b * s, 3, h, d)
    out = torch.empty_like(flat[:, 0])
    cu = torch.arange(0, (b + 1) * s, s, dtype=torch.int32, device=qkv.device)
    lse = bp.flash_fwd(flat[:, 0], flat[:, 1], flat[:, 2], out, cu, cu, s, s, scale, True)
    return out.reshape(b, s, h, d), lse[:, :, :s].clone()


# (name, batch, seq, heads, head dim, dtype, trunk layer whose scale is used): BASELINE configs 2, 4 and 5
FLASH_SHAPES = [('small-1024', 24, 1024, 12, 64, torch.bfloat16, 0),
                ('small-1024-layer11', 8, 1024, 12, 64, torch.bfloat16, 11),
                ('mini-k64-1024', 16, 1024, 8, 80, torch.bfloat16, 3),
                ('small-4096-fp16', 4, 4096, 12, 64, torch.float16, 5)]


@pytest.mark.parametrize('name,b,s,h,d,dtype,layer', FLASH_SHAPES, ids=[x[0] for x in FLASH_SHAPES])
def test_flash_fwd_causality_sample_independence_determinism(name, b, s, h, d, dtype, layer):
    g = torch.Generator(device=DEV).manual_seed(11)
    qkv = torch.randn(b, s, 3, h, d, device=DEV, generator=g).to(dtype)
    scale = d ** -0.5 / (layer + 1)
    out, lse = _flash(qkv, scale)
    assert torch.isfinite(out.float()).all() and torch.isfinite(lse).all()
    again, lse_again = _flash(qkv, scale)
    assert torch.equal(out, again) and torch.equal(lse, lse_again), 'two identical launches differ'
    # causality at cuts inside a 32-key block, on a 64-key tile border and on a 128-query tile border
    for t0 in (s // 2 + 37, s // 2 + 64, s - 128, 1):
        other = qkv.clone()
        other[:, t0:] = torch.randn(b, s - t0, 3, h, d, device=DEV, generator=g).to(dtype)
        got, lse_got = _flash(other, scale)
        assert torch.equal(got[:, :t0], out[:, :t0]), f'{name}: rows before {t0} changed with the rows behind it'
        assert torch.equal(lse_got[:, :, :t0], lse[:, :, :t0]), f'{name}: log-sum-exp before {t0} changed'
        assert not torch.equal(got[:, t0:], out[:, t0:])
    for t0 in (s // 2 + 37, s - 128 + 5):            # loud rows behind the cut (module docstring)
        other = qkv.clone()
        other[:, t0:] = (6.0 * torch.randn(b, s - t0, 3, h, d, device=DEV, generator=g)).to(dtype)
        got, lse_got = _flash(other, scale)
        assert torch.isfinite(got.float()).all() and torch.isfinite(lse_got).all()
        assert torch.equal(got[:, :t0], out[:, :t0]), f'{name}: rows before {t0} changed with the loud rows behind it'
        assert torch.equal(lse_got[:, :, :t0], lse[:, :, :t0]), f'{name}: log-sum-exp before {t0} changed (loud rows)'
    perm = torch.randperm(b, device=DEV, generator=g)
    got, lse_got = _flash(qkv[perm].contiguous(), scale)
    assert torch.equal(got, out[perm]) and torch.equal(lse_got, lse[perm]), f'{name}: samples are not independent'


# (name, batch, seq, senses, d_k, d, dtype): the sense kernels of BASELINE configs 2, 4 and 5
MIX_SHAPES = [('small-1024', 12, 1024, 16, 48, 768, torch.bfloat16),
              ('mini-k64-1024', 4, 1024, 64, 10, 640, torch.bfloat16),
              ('small-4096-fp16', 2, 4096, 16, 48, 768, torch.float16),
              # the reference's few-sense widths on the LDS-DMA ring kernels of csrc/sense_wide_dma.hip (bench workloads
              # mini-k4-1024 / mini-k1-1024)
              ('mini-k4-1024', 6, 1024, 4, 160, 640, torch.bfloat16),
              ('mini-k1-1024', 6, 1024, 1, 640, 640, torch.float16)]


@pytest.mark.parametrize('name,b,s,k,dk,d,dtype', MIX_SHAPES, ids=[x[0] for x in MIX_SHAPES])
def test_sense_lse_and_mix_causality_sample_independence_determinism(name, b, s, k, dk, d, dtype):
    bp = _bp()
    g = torch.Generator(device=DEV).manual_seed(12)
    qk = (1.5 * torch.randn(b, s, 2, k, dk, device=DEV, generator=g)).to(dtype)
    c = torch.randn(b, s, k, d, device=DEV, generator=g).to(dtype)
    lse = bp.sense_lse(qk)[:, :, :s].clone()
    out = bp.sense_mix(qk, c)
    assert torch.isfinite(out.float()).all() and torch.isfinite(lse).all()
    assert torch.equal(bp.sense_mix(qk, c), out) and torch.equal(bp.sense_lse(qk)[:, :, :s], lse)
    # cuts inside a 32-key block, on a 64-key tile border and on a 256-query tile border
    for t0 in (s // 2 + 37, s // 2 + 64, s - 256, 1):
        qk2, c2 = qk.clone(), c.clone()
        qk2[:, t0:] = (1.5 * torch.randn(b, s - t0, 2, k, dk, device=DEV, generator=g)).to(dtype)
        c2[:, t0:] = (3.0 * torch.randn(b, s - t0, k, d, device=DEV, generator=g)).to(dtype)
        assert torch.equal(bp.sense_lse(qk2)[:, :, :t0], lse[:, :, :t0]), f'{name}: sense LSE before {t0} changed'
        got = bp.sense_mix(qk2, c2)
        assert torch.equal(got[:, :t0], out[:, :t0]), f'{name}: mixed rows before {t0} changed with the rows behind it'
        assert not torch.equal(got[:, t0:], out[:, t0:])
    for t0 in (s // 2 + 37, s - 256 + 5):            # loud rows behind the cut (module docstring)
        qk2, c2 = qk.clone(), c.clone()
        qk2[:, t0:] = (6.0 * torch.randn(b, s - t0, 2, k, dk, device=DEV, generator=g)).to(dtype)
        c2[:, t0:] = (3.0 * torch.randn(b, s - t0, k, d, device=DEV, generator=g)).to(dtype)
        lse2 = bp.sense_lse(qk2)[:, :, :s]
        assert torch.isfinite(lse2).all()
        assert torch.equal(lse2[:, :, :t0], lse[:, :, :t0]), f'{name}: sense LSE before {t0} changed (loud rows)'
        got = bp.sense_mix(qk2, c2)
        assert torch.isfinite(got.float()).all()
        assert torch.equal(got[:, :t0], out[:, :t0]), f'{name}: mixed rows before {t0} changed with the loud rows behind it'
    perm = torch.randperm(b, device=DEV, generator=g)
    assert torch.equal(bp.sense_mix(qk[perm].contiguous(), c[perm].contiguous()), out[perm]), \
        f'{name}: samples are not independent'
    assert torch.equal(bp.sense_lse(qk[perm].contiguous())[:, :, :s], lse[perm])


@pytest.mark.parametrize('name,b,s,k,dk,d,dtype', MIX_SHAPES, ids=[x[0] for x in MIX_SHAPES])
def test_sense_mix_gather_causality_and_sample_independence(name, b, s, k, dk, d, dtype):
    """The table form (content[b, s] = table[index[b, s]]): the same properties in terms of the INDEX, and equality with the
    dense form on the rows it names."""
    bp = _bp()
    g = torch.Generator(device=DEV).manual_seed(13)
    rows = 3001
    qk = (1.5 * torch.randn(b, s, 2, k, dk, device=DEV, generator=g)).to(dtype)
    table = torch.randn(rows, k, d, device=DEV, generator=g).to(dtype)
    idx = torch.randint(0, rows, (b, s), device=DEV, generator=g, dtype=torch.int32)
    assert bp.sense_mix_gather_supported(qk, table, s)
    out = bp.sense_mix_gather(qk, table, idx)
    assert torch.equal(out, bp.sense_mix(qk, table[idx.long()])), f'{name}: table form differs from the dense form'
    assert torch.equal(out, bp.sense_mix_gather(qk, table, idx))
    for t0 in (s // 2 + 37, s - 256):
        idx2 = idx.clone()
        idx2[:, t0:] = torch.randint(0, rows, (b, s - t0), device=DEV, generator=g, dtype=torch.int32)
        got = bp.sense_mix_gather(qk, table, idx2)
        assert torch.equal(got[:, :t0], out[:, :t0]), f'{name}: rows before {t0} depend on later table indices'
    perm = torch.randperm(b, device=DEV, generator=g)
    assert torch.equal(bp.sense_mix_gather(qk[perm].contiguous(), table, idx[perm].contiguous()), out[perm])


def _small_model(seq, dtype, **over):
    from src.models.backpack import BackpackConfig, BackpackLMHeadModel
    kw = dict(n_embd=768, n_head=12, n_layer=12, num_content_vectors=16, vocab_size=50264, n_positions=seq,
              scale_attn_by_inverse_layer_idx=True, resid_pdrop=0.0, embd_pdrop=0.0, attn_pdrop=0.0,
              use_flash_attn=True, fused_dropout_add_ln=True, fused_dense_gelu_dense=True, fused_bias_fc=True,
              pad_vocab_size_multiple=8)
    kw.update(over)
    torch.manual_seed(0)
    model = BackpackLMHeadModel(BackpackConfig(**kw))
    with torch.no_grad():   # the default init gives near-uniform attention: sharpen it so that the softmax paths matter
        model.transformer.contextualization_attn.Wqkv.weight.mul_(8.0)
        for layer in model.transformer.gpt2_model.layers:
            layer.mixer.Wqkv.weight.mul_(6.0)
    return model.to(DEV, dtype).eval()


@pytest.mark.parametrize('mode', ['off', 'cached'])
def test_backpack_small_forward_is_causal_at_seq1024(mode):
    """BASELINE config 2 as a whole model (ids -> final hidden states -> logits), batch 8: the hidden states and logits of
    the positions before a cut are the same BITS whatever tokens follow -- with the content network per position (the
    reference's order, 'off') and with the cached whole-vocabulary sense table (the inference default).  Every layer
    between the kernels is row-wise (LayerNorm, library GEMMs of an unchanged shape), so one leaking key anywhere in the
    24 attention launches or the sense mix shows."""
    model = _small_model(1024, torch.bfloat16)
    t = model.transformer
    t.sense_table_mode = mode
    g = torch.Generator().manual_seed(5)
    ids = torch.randint(0, 50257, (8, 1024), generator=g).to(DEV)
    with torch.no_grad():
        hid = t(ids)
        if not torch.equal(hid, t(ids)):
            pytest.skip('the library GEMMs of this box are not run-to-run deterministic: nothing bit-exact to compare')
        # a cut inside a wave's 32 rows (this model's sharpened weights take the retry branch often) and one on a tile border
        for t0 in (1024 // 2 + 37, 576):
            ids2 = ids.clone()
            ids2[:, t0:] = torch.randint(0, 50257, (8, 1024 - t0), generator=g).to(DEV)
            hid2 = t(ids2)
            assert torch.equal(hid2[:, :t0], hid[:, :t0]), f'hidden states before {t0} depend on the tokens behind it'
            assert not torch.equal(hid2[:, t0:], hid[:, t0:])
            rows = torch.arange(0, t0, 7, device=DEV)
            assert torch.equal(model.lm_head(hid2[:, rows]), model.lm_head(hid[:, rows]))
    assert torch.isfinite(hid.float()).all()


def _flash_bwd(qkv, dout, scale):
    """Forward + backward of the causal trunk attention through the C ABI: returns out, dq, dk, dv as (B,S,H,D)."""
    bp = _bp()
    b, s, _, h, d = qkv.shape
    flat = qkv.reshape(b * s, 3, h, d)
    q, k, v = flat[:, 0], flat[:, 1], flat[:, 2]
    out = torch.empty_like(q)
    lse = bp.flash_fwd(q, k, v, out, None, None, s, s, scale, True)
    dq, dk, dv = (torch.empty(b * s, h, d, dtype=qkv.dtype, device=qkv.device) for _ in range(3))
    bp.flash_bwd(dout.reshape(b * s, h, d), q, k, v, out, lse, dq, dk, dv, None, None, s, s, scale, True)
    return tuple(x.reshape(b, s, h, d) for x in (out, dq, dk, dv))


BWD_SHAPES = [('small-1024', 32, 1024, 12, 64, torch.bfloat16), ('mini-k64-1024', 8, 1024, 8, 80, torch.bfloat16)]


@pytest.mark.parametrize('name,b,s,h,d,dtype', BWD_SHAPES, ids=[x[0] for x in BWD_SHAPES])
def test_flash_bwd_causality_sample_independence_determinism(name, b, s, h, d, dtype):
    """BASELINE config 3's attention backward at the training batch (32 samples per GPU), all bit-exact:
      * dQ of the rows before t depends on nothing behind t (q, k, v and dO there are replaced);
      * with dO = 0 behind t no gradient flows into the rows behind t (dQ, dK, dV there are exact zeros), and before t all
        three gradients are those of the problem truncated at t, launched as its own (shorter, ragged) batch;
      * the same call twice gives the same bits (two kernels, no atomics; the reference's loop kernel is deterministic
        too, tests/test_flash_attn.py:788-793), and permuting the samples permutes the gradients."""
    g = torch.Generator(device=DEV).manual_seed(21)
    qkv = torch.randn(b, s, 3, h, d, device=DEV, generator=g).to(dtype)
    dout = torch.randn(b, s, h, d, device=DEV, generator=g).to(dtype)
    scale = d ** -0.5
    out, dq, dk, dv = _flash_bwd(qkv, dout, scale)
    for x in (dq, dk, dv):
        assert torch.isfinite(x.float()).all()
    again = _flash_bwd(qkv, dout, scale)
    assert all(torch.equal(a, c) for a, c in zip(again, (out, dq, dk, dv))), 'two identical launches differ'
    for t0 in (s // 2 + 37, s - 128):
        qkv2, dout2 = qkv.clone(), dout.clone()
        qkv2[:, t0:] = (2.0 * torch.randn(b, s - t0, 3, h, d, device=DEV, generator=g)).to(dtype)
        dout2[:, t0:] = (2.0 * torch.randn(b, s - t0, h, d, device=DEV, generator=g)).to(dtype)
        _, dq2, _, _ = _flash_bwd(qkv2, dout2, scale)
        assert torch.equal(dq2[:, :t0], dq[:, :t0]), f'{name}: dQ before {t0} changed with the rows behind it'
        # no gradient flows into the rows behind t when dO is zero there: dK / dV behind t are exact zeros, and before t
        # they are the gradients of the problem truncated at t
        dout0 = dout.clone()
        dout0[:, t0:] = 0
        _, dq0, dk0, dv0 = _flash_bwd(qkv, dout0, scale)
        assert torch.count_nonzero(dk0[:, t0:]) == 0 and torch.count_nonzero(dv0[:, t0:]) == 0
        assert torch.count_nonzero(dq0[:, t0:]) == 0
        _, dq_t, dk_t, dv_t = _flash_bwd(qkv[:, :t0].contiguous(), dout[:, :t0].contiguous(), scale)
        assert torch.equal(dq0[:, :t0], dq_t)
        # (the queries behind t add exact zeros to dK / dV in the full problem; the truncated one masks them in its last,
        # partial tile: the same bits either way)
        assert torch.equal(dk0[:, :t0], dk_t) and torch.equal(dv0[:, :t0], dv_t)
    perm = torch.randperm(b, device=DEV, generator=g)
    got = _flash_bwd(qkv[perm].contiguous(), dout[perm].contiguous(), scale)
    assert all(torch.equal(a, c[perm]) for a, c in zip(got, (out, dq, dk, dv))), f'{name}: samples are not independent'


def _mix_grads(qk, c, dout):
    """dqk, dcontent of the fused sense contraction (bp_hip.SenseMixFn: LSE pre-pass, mix, dC kernel, slab GEMMs + dq / dk)."""
    bp = _bp()
    qk = qk.detach().clone().requires_grad_(True)
    c = c.detach().clone().requires_grad_(True)
    out = bp.sense_mix_autograd(qk, c)
    out.backward(dout)
    return out.detach(), qk.grad, c.grad


MIX_BWD_SHAPES = [('small-1024', 8, 1024, 16, 48, 768), ('mini-k16-1024', 8, 1024, 16, 40, 640)]


@pytest.mark.parametrize('name,b,s,k,dk,d', MIX_BWD_SHAPES, ids=[x[0] for x in MIX_BWD_SHAPES])
def test_sense_mix_backward_causality_sample_independence_determinism(name, b, s, k, dk, d):
    """The backward of the sense contraction at config 3's shape (and Mini's widths), bit-exact:
      * two identical calls agree; permuting the samples permutes dqk and dC;
      * with dout = 0 behind a cut no gradient reaches the rows behind it (dq, dk, dC there are exact zeros);
      * before the cut dC (one kernel) is the dC of the problem truncated there, wherever the cut lies; dq / dk go through
        slab GEMMs of the BLAS library (128 queries per slab), so they are compared at a cut on a slab border, where the
        truncated problem issues the same GEMM shapes."""
    g = torch.Generator(device=DEV).manual_seed(31)
    dtype = torch.bfloat16
    qk = (1.5 * torch.randn(b, s, 2, k, dk, device=DEV, generator=g)).to(dtype)
    c = torch.randn(b, s, k, d, device=DEV, generator=g).to(dtype)
    dout = torch.randn(b, s, d, device=DEV, generator=g).to(dtype)
    out, dqk, dc = _mix_grads(qk, c, dout)
    assert torch.isfinite(dqk.float()).all() and torch.isfinite(dc.float()).all()
    again = _mix_grads(qk, c, dout)
    assert all(torch.equal(x, y) for x, y in zip(again, (out, dqk, dc))), 'two identical calls differ'
    perm = torch.randperm(b, device=DEV, generator=g)
    got = _mix_grads(qk[perm].contiguous(), c[perm].contiguous(), dout[perm].contiguous())
    assert all(torch.equal(x, y[perm]) for x, y in zip(got, (out, dqk, dc))), f'{name}: samples are not independent'
    for t0 in (s // 2 + 128, s // 2 + 37):
        dout0 = dout.clone()
        dout0[:, t0:] = 0
        _, dqk0, dc0 = _mix_grads(qk, c, dout0)
        assert torch.count_nonzero(dc0[:, t0:]) == 0 and torch.count_nonzero(dqk0[:, t0:]) == 0
        _, dqk_t, dc_t = _mix_grads(qk[:, :t0].contiguous(), c[:, :t0].contiguous(), dout[:, :t0].contiguous())
        assert torch.equal(dc0[:, :t0], dc_t), f'{name}: dC before {t0} is not the truncated problem\'s'
        if t0 % 128 == 0:
            assert torch.equal(dqk0[:, :t0], dqk_t), f'{name}: dq / dk before {t0} are not the truncated problem\'s'
        else:
            miss = (dqk0[:, :t0].float() - dqk_t.float()).abs().max().item()
            assert miss <= 2.0 ** -6 * dqk_t.float().abs().max().item(), (name, t0, miss)


@pytest.mark.parametrize('h,d,dtype', [(12, 64, torch.bfloat16), (8, 80, torch.bfloat16), (12, 64, torch.float16)],
                         ids=['small', 'mini', 'small-fp16'])
def test_flash_ragged_batch_equals_its_sequences_launched_alone(h, d, dtype):
    """cu_seqlens indexing (reference: csrc/flash_attn/src/fmha_kernel.h:52-57, flash_attn/bert_padding.py:97-117): a ragged
    batch -- lengths from 1 to 1024, odd ones, some behind others in one buffer -- gives every sequence the BITS it gets as a
    fixed-length batch of its own, forward (O, LSE) and backward (dQ, dK, dV); the same through strided q / k / v views of one
    packed (total, 3, H, D) tensor (the qkvpacked entry's layout) and through separate contiguous copies; and the heads of a
    launch are independent (permuting them permutes the results)."""
    bp = _bp()
    g = torch.Generator(device=DEV).manual_seed(41)
    lens = [1024, 517, 1, 333, 64, 1000, 129, 31]
    total, smax = sum(lens), max(lens)
    qkv = torch.randn(total, 3, h, d, device=DEV, generator=g).to(dtype)
    dout = torch.randn(total, h, d, device=DEV, generator=g).to(dtype)
    cu = torch.tensor([0] + lens, device=DEV).cumsum(0).to(torch.int32)
    scale = d ** -0.5

    def run(q, k, v, do, cu_, smax_):
        out = torch.empty_like(q)
        lse = bp.flash_fwd(q, k, v, out, cu_, cu_, smax_, smax_, scale, True)
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        bp.flash_bwd(do, q, k, v, out, lse, dq, dk, dv, cu_, cu_, smax_, smax_, scale, True)
        return out, lse, dq, dk, dv

    out, lse, dq, dk, dv = run(qkv[:, 0], qkv[:, 1], qkv[:, 2], dout, cu, smax)
    for x in (out, dq, dk, dv):
        assert torch.isfinite(x.float()).all()
    # contiguous copies instead of the packed tensor's strided views
    q, k, v = (qkv[:, i].contiguous() for i in range(3))
    def same_lse(a, c):            # (rows behind a sequence's length are never written)
        return all(torch.equal(a[i, :, :n], c[i, :, :n]) for i, n in enumerate(lens))

    got = run(q, k, v, dout, cu, smax)
    assert same_lse(got[1], lse) and all(torch.equal(got[i], c) for i, c in ((0, out), (2, dq), (3, dk), (4, dv))), \
        'strided views of the packed tensor and contiguous copies differ'
    start = 0
    for i, n in enumerate(lens):
        sl = slice(start, start + n)
        o1, l1, dq1, dk1, dv1 = run(q[sl].contiguous(), k[sl].contiguous(), v[sl].contiguous(), dout[sl].contiguous(),
                                    None, n)
        assert torch.equal(o1, out[sl]) and torch.equal(l1[0, :, :n], lse[i, :, :n]), f'sequence {i} (length {n}): forward'
        assert torch.equal(dq1, dq[sl]) and torch.equal(dk1, dk[sl]) and torch.equal(dv1, dv[sl]), \
            f'sequence {i} (length {n}): backward'
        start += n
    perm = torch.randperm(h, device=DEV, generator=g)
    got = run(q[:, perm].contiguous(), k[:, perm].contiguous(), v[:, perm].contiguous(), dout[:, perm].contiguous(), cu, smax)
    assert torch.equal(got[0], out[:, perm]) and same_lse(got[1], lse[:, perm])
    assert all(torch.equal(a, c[:, perm]) for a, c in zip(got[2:], (dq, dk, dv)))


@pytest.mark.parametrize('name,b,s,k,dk,d,dtype', MIX_SHAPES[:2], ids=[x[0] for x in MIX_SHAPES[:2]])
def test_forward_kernels_of_a_prefix_equal_the_prefix_of_the_forward(name, b, s, k, dk, d, dtype):
    """Growing-prefix consistency (the reference's generation loop re-runs the forward on every prefix,
    training/src/utils/generation.py:64-72): the kernels' results for a sequence cut at t are the first t rows of their
    results for the whole sequence, bit for bit -- trunk attention (O, LSE), sense LSE and the fused mix; cuts inside a
    32-key block, on tile borders, at a single row."""
    bp = _bp()
    g = torch.Generator(device=DEV).manual_seed(43)
    h, dh = (12, 64) if d == 768 else (8, 80)
    qkv = torch.randn(b, s, 3, h, dh, device=DEV, generator=g).to(dtype)
    qk = (1.5 * torch.randn(b, s, 2, k, dk, device=DEV, generator=g)).to(dtype)
    c = torch.randn(b, s, k, d, device=DEV, generator=g).to(dtype)
    scale = dh ** -0.5 / 3
    out, lse = _flash(qkv, scale)
    slse = bp.sense_lse(qk)
    mix = bp.sense_mix(qk, c)
    for t0 in (s // 2 + 37, s // 2 + 64, s - 256, 200, 1):
        o_t, l_t = _flash(qkv[:, :t0].contiguous(), scale)
        assert torch.equal(o_t, out[:, :t0]) and torch.equal(l_t, lse[:, :, :t0]), f'{name}: trunk attention, prefix {t0}'
        qk_t, c_t = qk[:, :t0].contiguous(), c[:, :t0].contiguous()
        assert torch.equal(bp.sense_lse(qk_t)[:, :, :t0], slse[:, :, :t0]), f'{name}: sense LSE, prefix {t0}'
        assert torch.equal(bp.sense_mix(qk_t, c_t), mix[:, :t0]), f'{name}: sense mix, prefix {t0}'
        # the same through views of the long tensors (strides of the whole sequence, length of the prefix)
        assert torch.equal(bp.sense_mix(qk[:, :t0], c[:, :t0]), mix[:, :t0]), f'{name}: sense mix on strided views, prefix {t0}'


def test_row_wise_kernels_treat_every_row_alone():
    """The row-wise kernels either side of the attention path (fused add + LayerNorm, fused cross-entropy, bias + GELU), at
    the bench's row count (64 samples x 1024 positions): a row's result does not depend on where it sits or on how many
    rows the launch has -- permuting the rows permutes the results, a 7-row slice gives the bits of those rows in the
    full launch -- and two launches agree."""
    bp = _bp()
    g = torch.Generator(device=DEV).manual_seed(51)
    rows, cols = 64 * 1024, 768
    x0 = torch.randn(rows, cols, device=DEV, generator=g).bfloat16()
    x1 = (4.0 * torch.randn(rows, cols, device=DEV, generator=g)).float()
    w = (1.0 + 0.1 * torch.randn(cols, device=DEV, generator=g)).float()
    b = (0.1 * torch.randn(cols, device=DEV, generator=g)).float()
    z, res = bp.add_layer_norm(x0, x1, w, b, 1e-5)
    z2, res2 = bp.add_layer_norm(x0, x1, w, b, 1e-5)
    assert torch.equal(z, z2) and torch.equal(res, res2)
    perm = torch.randperm(rows, device=DEV, generator=g)
    zp, resp = bp.add_layer_norm(x0[perm].contiguous(), x1[perm].contiguous(), w, b, 1e-5)
    assert torch.equal(zp, z[perm]) and torch.equal(resp, res[perm])
    pick = perm[:7]
    zs, ress = bp.add_layer_norm(x0[pick].contiguous(), x1[pick].contiguous(), w, b, 1e-5)
    assert torch.equal(zs, z[pick]) and torch.equal(ress, res[pick])

    vocab, n = 50264, 4096                       # the LM head's width, four samples' worth of rows
    logits = (3.0 * torch.randn(n, vocab, device=DEV, generator=g)).bfloat16()
    labels = torch.randint(0, vocab, (n,), device=DEV, generator=g)
    labels[::97] = -100                          # a label outside the vocabulary: no x[label] term, loss 0 without smoothing
    loss, lse = bp.xentropy_fwd(logits, labels)
    grad = bp.xentropy_bwd(torch.ones_like(loss), logits, lse, labels)
    perm = torch.randperm(n, device=DEV, generator=g)
    loss_p, lse_p = bp.xentropy_fwd(logits[perm].contiguous(), labels[perm])
    assert torch.equal(loss_p, loss[perm]) and torch.equal(lse_p, lse[perm])
    assert torch.equal(bp.xentropy_bwd(torch.ones_like(loss), logits[perm].contiguous(), lse_p, labels[perm]), grad[perm])
    pick = perm[:5]
    loss_s, lse_s = bp.xentropy_fwd(logits[pick].contiguous(), labels[pick])
    assert torch.equal(loss_s, loss[pick]) and torch.equal(lse_s, lse[pick])
    assert torch.count_nonzero(loss[labels == -100]) == 0 and torch.isfinite(lse).all()

    h = torch.randn(rows, 3072, device=DEV, generator=g).bfloat16()
    bias = torch.randn(3072, device=DEV, generator=g).bfloat16()
    y = bp.bias_gelu_fwd(h, bias)[0]
    perm = torch.randperm(rows, device=DEV, generator=g)
    assert torch.equal(bp.bias_gelu_fwd(h[perm].contiguous(), bias)[0], y[perm])
    assert torch.equal(bp.bias_gelu_fwd(h[perm[:3]].contiguous(), bias)[0], y[perm[:3]])


def test_operands_as_awkward_views():
    """Views whose element offset breaks the 16-byte alignment or whose row / head strides are not multiples of eight (slices
    of wider buffers): the attention forward and the fused sense mix give the contiguous call's result -- the aligned ones
    bit for bit (same kernel), the others through the generic kernels, to 16-bit rounding -- and never fault."""
    bp = _bp()
    g = torch.Generator(device=DEV).manual_seed(61)
    for (b, s, h, d) in ((2, 100, 3, 64), (1, 257, 2, 40), (2, 64, 1, 128), (1, 33, 4, 16)):
        for off in (0, 1, 3, 4, 8):
            for gap in (0, 1, 5, 8):
                big = torch.randn(b * s, 3, h, d + gap + off + 8, device=DEV, generator=g).bfloat16()
                q, k, v = (big[:, i, :, off:off + d] for i in range(3))
                want = torch.empty(b * s, h, d, device=DEV, dtype=torch.bfloat16)
                lse_want = bp.flash_fwd(q.contiguous(), k.contiguous(), v.contiguous(), want, None, None, s, s, d ** -0.5, True)
                got = torch.full_like(want, float('nan'))
                lse = bp.flash_fwd(q, k, v, got, None, None, s, s, d ** -0.5, True)
                tag = f'flash {(b, s, h, d)} offset {off} gap {gap}'
                assert (got.float() - want.float()).abs().max().item() < 2e-2, tag
                assert (lse[:, :, :s] - lse_want[:, :, :s]).abs().max().item() < 4e-3, tag
                if off % 8 == 0 and (d + gap + off + 8) % 8 == 0:
                    assert torch.equal(got, want), tag
    for (b, s, k, dk, d) in ((2, 100, 4, 48, 256), (1, 65, 3, 24, 104), (1, 200, 16, 16, 64)):
        for off in (0, 1, 4, 8):
            for gap in (0, 3, 8):
                bigqk = torch.randn(b, s, 2, k, dk + gap + off + 8, device=DEV, generator=g).bfloat16()
                bigc = torch.randn(b, s, k, d + gap + off + 8, device=DEV, generator=g).bfloat16()
                qk, c = bigqk[..., off:off + dk], bigc[..., off:off + d]
                want = bp.sense_mix(qk.contiguous(), c.contiguous())
                got = bp.sense_mix(qk, c)
                tag = f'mix {(b, s, k, dk, d)} offset {off} gap {gap}'
                assert (got.float() - want.float()).abs().max().item() < 2e-2, tag
                if off % 8 == 0 and gap % 8 == 0:
                    assert torch.equal(got, want), tag

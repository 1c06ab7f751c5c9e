"""GPU parity of the attention BACKWARD kernels through the C ABI
(bp_flash_bwd) and through the autograd Functions of flash_attn.flash_attn_interface.

Criterion = the reference's own (tests/test_flash_attn.py:391-397, 478-486): with the fp32 autograd
gradients of the oracle as truth,
    max|dX_kernel - dX_fp32| <= 2 * max|dX_eager_same_dtype - dX_fp32|      for X in q, k, v
(+ a 1e-5 absolute floor for tiny problems).  Runs are bit-reproducible here (no atomics), which the
reference only asserts for the forward.
"""
import math

import pytest
import torch

from oracle import ref_cpu as R

pytestmark = pytest.mark.gpu

DEV = 'cuda'


def _bp():
    import bp_hip
    return bp_hip


def _oracle_grads(q16, k16, v16, dout16, causal, scale, upcast, key_padding_mask=None):
    if upcast:
        # fp32 truth: no rounding of the output before the product with dout
        q, k, v = (t.float().clone().requires_grad_() for t in (q16, k16, v16))
        out = R.attention_fp32(q, k, v, causal=causal, softmax_scale=scale,
                               key_padding_mask=key_padding_mask)[0]
        return torch.autograd.grad(out, (q, k, v), dout16.float())
    q, k, v = (t.clone().requires_grad_() for t in (q16, k16, v16))
    out = R.attention_fp32(q, k, v, causal=causal, softmax_scale=scale, upcast=False, reorder_ops=True,
                           key_padding_mask=key_padding_mask)[0]
    return torch.autograd.grad(out, (q, k, v), dout16)


def _check(got, ref, eager, name, factor=2.0, atol=1e-5):
    for g, r, e, n in zip(got, ref, eager, ('dq', 'dk', 'dv')):
        err = (g.float().cpu() - r).abs().max().item()
        base = (e.float() - r).abs().max().item()
        print(f'{name} {n}: kernel err {err:.3e}  eager-same-dtype err {base:.3e}')
        assert torch.isfinite(g.float()).all(), f'{name} {n} not finite'
        assert err <= factor * base + atol, f'{name} {n}: {err} > {factor} * {base}'


def _run_bwd(q, k, v, dout, causal, scale, cu_q=None, cu_k=None, max_q=None, max_k=None):
    """q (B,Sq,H,D) k/v (B,Sk,H,D) on CPU (fixed length) or flat (T,H,D) with cu_* -> dq, dk, dv."""
    bp = _bp()
    if cu_q is None:
        b, sq, h, d = q.shape
        sk = k.shape[1]
        cu_q = torch.arange(0, (b + 1) * sq, sq, dtype=torch.int32)
        cu_k = torch.arange(0, (b + 1) * sk, sk, dtype=torch.int32)
        max_q, max_k = sq, sk
        q, k, v, dout = (t.reshape(-1, h, d) for t in (q, k, v, dout))
    qd, kd, vd, dod = (t.to(DEV).contiguous() for t in (q, k, v, dout))
    out = torch.empty_like(qd)
    lse = bp.flash_fwd(qd, kd, vd, out, cu_q.to(DEV), cu_k.to(DEV), max_q, max_k, scale, causal)
    dq, dk, dv = torch.empty_like(qd), torch.empty_like(kd), torch.empty_like(vd)
    dq.fill_(float('nan')); dk.fill_(float('nan')); dv.fill_(float('nan'))
    bp.flash_bwd(dod, qd, kd, vd, out, lse, dq, dk, dv, cu_q.to(DEV), cu_k.to(DEV), max_q, max_k, scale,
                 causal)
    return dq, dk, dv


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
@pytest.mark.parametrize('causal', [True, False])
@pytest.mark.parametrize('d', [64, 48, 40, 24, 16, 8, 32, 80, 128, 96])
@pytest.mark.parametrize('seqlen', [97, 128, 200, 257, 512])
def test_flash_bwd_fixed_len(seqlen, d, causal, dtype):
    """Shape sweep after the reference's test_flash_attn_unpadded_qkvpacked (tests/test_flash_attn.py:350-397)."""
    torch.manual_seed(0)
    b, h = 2, 3
    q, k, v, dout = (torch.randn(b, seqlen, h, d).to(dtype) for _ in range(4))
    scale = 1.0 / math.sqrt(d)
    ref = _oracle_grads(q, k, v, dout, causal, scale, upcast=True)
    eager = _oracle_grads(q, k, v, dout, causal, scale, upcast=False)
    got = [g.reshape(b, seqlen, h, d) for g in _run_bwd(q, k, v, dout, causal, scale)]
    _check(got, ref, eager, f'bwd s={seqlen} d={d} causal={causal} {dtype}')


@pytest.mark.parametrize('causal', [True, False])
def test_flash_bwd_cross_lengths(causal):
    """seqlen_q != seqlen_k (the kv-packed call shape, tests/test_flash_attn.py:437-486)."""
    torch.manual_seed(1)
    b, h, d, sq, sk = 2, 2, 64, 150, 333
    q, dout = (torch.randn(b, sq, h, d).bfloat16() for _ in range(2))
    k, v = (torch.randn(b, sk, h, d).bfloat16() for _ in range(2))
    scale = d ** -0.5
    ref = _oracle_grads(q, k, v, dout, causal, scale, upcast=True)
    eager = _oracle_grads(q, k, v, dout, causal, scale, upcast=False)
    dq, dk, dv = _run_bwd(q, k, v, dout, causal, scale)
    got = [dq.reshape(b, sq, h, d), dk.reshape(b, sk, h, d), dv.reshape(b, sk, h, d)]
    _check(got, ref, eager, f'bwd cross causal={causal}')


@pytest.mark.parametrize('causal', [True, False])
def test_flash_bwd_varlen(causal):
    """Ragged batch incl. a 1-token and an empty-key sequence: per-sequence gradients match the oracle
    run on each sequence alone; rows of an empty-key sequence get zero dq."""
    torch.manual_seed(2)
    h, d = 2, 64
    lens_q = [70, 1, 130, 5]
    lens_k = [70, 1, 200, 0]
    q, dout = (torch.randn(sum(lens_q), h, d).bfloat16() for _ in range(2))
    k, v = (torch.randn(sum(lens_k), h, d).bfloat16() for _ in range(2))
    cu_q = torch.tensor([0] + torch.tensor(lens_q).cumsum(0).tolist(), dtype=torch.int32)
    cu_k = torch.tensor([0] + torch.tensor(lens_k).cumsum(0).tolist(), dtype=torch.int32)
    scale = d ** -0.5
    dq, dk, dv = _run_bwd(q, k, v, dout, causal, scale, cu_q, cu_k, max(lens_q), max(lens_k))
    for i in range(len(lens_q)):
        qs, qe, ks, ke = cu_q[i], cu_q[i + 1], cu_k[i], cu_k[i + 1]
        if lens_k[i] == 0:
            assert torch.equal(dq[qs:qe].cpu(), torch.zeros(lens_q[i], h, d, dtype=torch.bfloat16))
            continue
        args = (q[qs:qe][None], k[ks:ke][None], v[ks:ke][None], dout[qs:qe][None], causal, scale)
        ref = _oracle_grads(*args, upcast=True)
        eager = _oracle_grads(*args, upcast=False)
        got = [dq[qs:qe][None], dk[ks:ke][None], dv[ks:ke][None]]
        _check(got, ref, eager, f'bwd varlen seq{i} causal={causal}', atol=2e-3)


def test_flash_bwd_determinism():
    """Bit-identical over repeats: each dq / dk / dv element has exactly one writer and a fixed
    accumulation order (the reference's backward is only allclose-reproducible, test_flash_attn.py:768-772)."""
    torch.manual_seed(3)
    q, k, v, dout = (torch.randn(3, 300, 4, 64).bfloat16() for _ in range(4))
    first = _run_bwd(q, k, v, dout, True, 0.125)
    for _ in range(5):
        again = _run_bwd(q, k, v, dout, True, 0.125)
        for a, b in zip(first, again):
            assert torch.equal(a, b)


def test_flash_bwd_strided_packed_views():
    """dq/dk/dv written into slices of one packed (T,3,H,D) buffer, q/k/v read from one, as
    FlashAttnQKVPackedFunc.backward does (flash_attn_interface.py:70-84)."""
    bp = _bp()
    torch.manual_seed(4)
    b, s, h, d = 2, 192, 4, 64
    qkv = torch.randn(b * s, 3, h, d).bfloat16().to(DEV)
    dout = torch.randn(b * s, h, d).bfloat16().to(DEV)
    cu = torch.arange(0, (b + 1) * s, s, dtype=torch.int32, device=DEV)
    out = torch.empty_like(qkv[:, 0])
    lse = bp.flash_fwd(qkv[:, 0], qkv[:, 1], qkv[:, 2], out, cu, cu, s, s, 0.125, True)
    dqkv = torch.empty_like(qkv)
    bp.flash_bwd(dout, qkv[:, 0], qkv[:, 1], qkv[:, 2], out, lse, dqkv[:, 0], dqkv[:, 1], dqkv[:, 2], cu, cu,
                 s, s, 0.125, True)
    q, k, v = (qkv[:, i].contiguous() for i in range(3))
    dq, dk, dv = (torch.empty_like(q) for _ in range(3))
    bp.flash_bwd(dout, q, k, v, out, lse, dq, dk, dv, cu, cu, s, s, 0.125, True)
    assert torch.equal(dqkv[:, 0], dq) and torch.equal(dqkv[:, 1], dk) and torch.equal(dqkv[:, 2], dv)


def test_flash_bwd_seq2048_rows():
    """Long-sequence case (bench shape family): gradient of a random subset of rows vs the oracle."""
    torch.manual_seed(5)
    b, s, h, d = 1, 2048, 2, 64
    q, k, v, dout = (torch.randn(b, s, h, d).bfloat16() for _ in range(4))
    ref = _oracle_grads(q, k, v, dout, True, d ** -0.5, upcast=True)
    eager = _oracle_grads(q, k, v, dout, True, d ** -0.5, upcast=False)
    got = [g.reshape(b, s, h, d) for g in _run_bwd(q, k, v, dout, True, d ** -0.5)]
    _check(got, ref, eager, 'bwd s=2048')


def test_flash_bwd_rejects_large_head_dim():
    bp = _bp()
    q = torch.randn(64, 2, 136, device=DEV).bfloat16()
    assert not bp.flash_bwd_supported(q)
    cu = torch.tensor([0, 64], dtype=torch.int32, device=DEV)
    lse = torch.zeros(1, 2, 64, device=DEV)
    with pytest.raises(RuntimeError, match='bp_flash_bwd'):
        bp.flash_bwd(q, q, q, q, q, lse, torch.empty_like(q), torch.empty_like(q), torch.empty_like(q), cu, cu,
                     64, 64, 0.1, False)


@pytest.mark.parametrize('packing', ['qkv', 'kv', 'none'])
@pytest.mark.parametrize('d', [64, 80, 36])
def test_autograd_functions_use_backward(packing, d):
    """loss.backward() through the public functions: d=64 / 80 take the HIP backward, d=36 (not a multiple
    of 8) the eager recomputation; both must match fp32 autograd by the same criterion."""
    from flash_attn import flash_attn_interface as F
    torch.manual_seed(6)
    b, s, h = 2, 160, 2
    qkv16 = torch.randn(b, s, 3, h, d).bfloat16()
    dout = torch.randn(b, s, h, d).bfloat16()
    scale = d ** -0.5
    ref = _oracle_grads(qkv16[:, :, 0], qkv16[:, :, 1], qkv16[:, :, 2], dout, True, scale, upcast=True)
    eager = _oracle_grads(qkv16[:, :, 0], qkv16[:, :, 1], qkv16[:, :, 2], dout, True, scale, upcast=False)
    cu = torch.arange(0, (b + 1) * s, s, dtype=torch.int32, device=DEV)
    flat = qkv16.reshape(b * s, 3, h, d).to(DEV)
    if packing == 'qkv':
        x = flat.clone().requires_grad_()
        out = F.flash_attn_unpadded_qkvpacked_func(x, cu, s, 0.0, causal=True)
        out.backward(dout.reshape(b * s, h, d).to(DEV))
        got = [x.grad[:, i] for i in range(3)]
    elif packing == 'kv':
        q = flat[:, 0].clone().requires_grad_()
        kv = flat[:, 1:].clone().requires_grad_()
        out = F.flash_attn_unpadded_kvpacked_func(q, kv, cu, cu, s, s, 0.0, causal=True)
        out.backward(dout.reshape(b * s, h, d).to(DEV))
        got = [q.grad, kv.grad[:, 0], kv.grad[:, 1]]
    else:
        q, k, v = (flat[:, i].clone().requires_grad_() for i in range(3))
        out = F.flash_attn_unpadded_func(q, k, v, cu, cu, s, s, 0.0, causal=True)
        out.backward(dout.reshape(b * s, h, d).to(DEV))
        got = [q.grad, k.grad, v.grad]
    got = [g.reshape(b, s, h, d) for g in got]
    _check(got, ref, eager, f'autograd {packing} d={d}')

"""-m gpu parity tests of the HIP kernels (through the C ABI) against the CPU oracle.

Criterion for 16-bit floating point (the reference's own, tests/test_flash_attn.py:424-428):
    max|kernel - fp32 oracle|  <=  2 * max|same-dtype eager PyTorch - fp32 oracle|  (+ tiny atol)
Causal-mask / indexing properties are checked bit-exactly (zeros above the diagonal, -inf LSE,
untouched padding rows).
"""
import math

import pytest
import torch

from conftest import from_bits16, load_golden
from oracle import ref_cpu as R

pytestmark = pytest.mark.gpu

DEV = 'cuda'


def _bp():
    import bp_hip
    return bp_hip


def rel_check(got, ref32, eager16, name, factor=2.0, atol=1e-5):
    err = (got.float().cpu() - ref32.float().cpu()).abs().max().item()
    base = (eager16.float().cpu() - ref32.float().cpu()).abs().max().item()
    print(f'{name}: kernel err {err:.3e}  eager-same-dtype err {base:.3e}')
    assert err <= factor * base + atol, f'{name}: {err} > {factor} * {base}'
    assert torch.isfinite(got.float()).all()


def run_flash_fixed(qkv, scale, causal):
    """qkv (B,S,3,H,D) on GPU -> out (B,S,H,D), lse (B,H,S)."""
    bp = _bp()
    b, s, _, h, d = qkv.shape
    flat = qkv.reshape(b * s, 3, h, d)
    out = torch.empty_like(flat[:, 0])
    cu = torch.arange(0, (b + 1) * s, s, dtype=torch.int32, device=qkv.device)
    lse = bp.flash_fwd(flat[:, 0], flat[:, 1], flat[:, 2], out, cu, cu, s, s, scale, causal)
    return out.reshape(b, s, h, d), lse[:, :, :s]


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
@pytest.mark.parametrize('causal', [True, False])
@pytest.mark.parametrize('d', [64, 80, 128, 32, 16, 40])
@pytest.mark.parametrize('seqlen', [97, 128, 200, 256, 257, 1024])
def test_flash_fwd_fixed_len(seqlen, d, causal, dtype):
    """Shape sweep of the reference's test_flash_attn_unpadded_qkvpacked (tests/test_flash_attn.py:350-373)."""
    torch.manual_seed(0)
    b, h = 3, 4
    qkv32 = torch.randn(b, seqlen, 3, h, d)
    qkv16 = qkv32.to(dtype)
    scale = 1.0 / math.sqrt(d)
    ref, _, lse_ref = R.attention_fp32(qkv16[:, :, 0], qkv16[:, :, 1], qkv16[:, :, 2], causal=causal,
                                       softmax_scale=scale)
    ref32 = R.attention_fp32(qkv16[:, :, 0].float(), qkv16[:, :, 1].float(), qkv16[:, :, 2].float(),
                             causal=causal, softmax_scale=scale)[0]
    eager, _, _ = R.attention_fp32(qkv16[:, :, 0], qkv16[:, :, 1], qkv16[:, :, 2], causal=causal,
                                   softmax_scale=scale, upcast=False, reorder_ops=True)
    out, lse = run_flash_fixed(qkv16.to(DEV), scale, causal)
    rel_check(out, ref32, eager, f'flash s={seqlen} d={d} causal={causal} {dtype}')
    assert (lse.cpu() - lse_ref).abs().max().item() < 2e-3


@pytest.mark.parametrize('layer', [0, 5, 11])
@pytest.mark.parametrize('tag', ['h64', 'h80'])
def test_flash_fwd_golden_trunk(tag, layer):
    """G3: the reference's eager SelfAttention output at the trunk's per-layer scale."""
    g = load_golden('g3_trunk_attn.npz')
    qkv = from_bits16(g[f'{tag}_qkv'])
    want = torch.from_numpy(g[f'{tag}_L{layer}_out'])
    lse_want = torch.from_numpy(g[f'{tag}_L{layer}_lse'])
    dh = qkv.shape[-1]
    scale = dh ** -0.5 / (layer + 1)
    out, lse = run_flash_fixed(qkv.to(DEV, torch.bfloat16), scale, True)
    eager = R.self_attention_eager(qkv.bfloat16(), True, scale)
    rel_check(out, want, eager, f'golden {tag} L{layer}')
    assert (lse.cpu() - lse_want).abs().max().item() < 2e-3


def _varlen_eager16(q, k, v, cu_q, cu_k, causal, scale=None):
    """The same-dtype yardstick of the reference's tolerance rule (tests/test_flash_attn.py:424-428: `attention_ref(...,
    upcast=False, reorder_ops=True)`), one sequence at a time: (total_q, H, D) in the inputs' 16-bit dtype."""
    outs = []
    for b in range(len(cu_q) - 1):
        q0, q1, k0, k1 = int(cu_q[b]), int(cu_q[b + 1]), int(cu_k[b]), int(cu_k[b + 1])
        if q1 == q0:
            continue
        if k1 == k0:
            outs.append(torch.zeros_like(q[q0:q1]))
            continue
        outs.append(R.attention_fp32(q[None, q0:q1], k[None, k0:k1], v[None, k0:k1], causal=causal, softmax_scale=scale,
                                     upcast=False, reorder_ops=True)[0][0])
    return torch.cat(outs, dim=0)


def _within_2x(got, ref, eager, what):
    """max|kernel - fp32 oracle| <= 2 max|same-dtype eager - fp32 oracle| + 1e-5 (the reference's kernel criterion)."""
    err = (got.float().cpu() - ref.float()).abs().max().item()
    base = (eager.float().cpu() - ref.float()).abs().max().item()
    print(f'{what}: kernel {err:.3e} eager-same-dtype {base:.3e}')
    assert err <= 2 * base + 1e-5, (what, err, base)


@pytest.mark.parametrize('causal', [False, True])
def test_flash_fwd_varlen_golden(causal):
    """G5: unequal lengths (97,128,33,1) through cu_seqlens; padding rows of LSE stay untouched."""
    bp = _bp()
    g = load_golden('g5_varlen.npz')
    qkv = from_bits16(g['qkv_unpad']).to(DEV, torch.bfloat16)
    cu = torch.from_numpy(g['cu_seqlens']).to(DEV, torch.int32)
    max_s = int(g['max_s'])
    out = torch.full_like(qkv[:, 0], float('nan'))
    lse = bp.flash_fwd(qkv[:, 0], qkv[:, 1], qkv[:, 2], out, cu, cu, max_s, max_s, 64 ** -0.5, causal)
    want = torch.from_numpy(g[f'out_causal{int(causal)}'])
    qkv_cpu = qkv.cpu()
    cu_cpu = cu.cpu()
    _within_2x(out, want, _varlen_eager16(qkv_cpu[:, 0], qkv_cpu[:, 1], qkv_cpu[:, 2], cu_cpu, cu_cpu, causal, 64 ** -0.5),
               f'varlen golden causal={causal}')
    lens = g['lens']
    lse_want = torch.from_numpy(g[f'lse_causal{int(causal)}'])  # (H, total)
    off = 0
    for b, n in enumerate(lens):
        got = lse[b, :, :n].cpu()
        assert (got - lse_want[:, off:off + n]).abs().max().item() < 2e-3
        off += n


def test_flash_fwd_cross_and_empty():
    """seqlen_q != seqlen_k (kv-packed style) and a sequence with zero keys -> zeros / -inf LSE."""
    bp = _bp()
    torch.manual_seed(3)
    h, d = 2, 64
    lens_q = [70, 5, 130]
    lens_k = [33, 0, 200]
    q = torch.randn(sum(lens_q), h, d).bfloat16()
    k = torch.randn(sum(lens_k), h, d).bfloat16()
    v = torch.randn(sum(lens_k), h, d).bfloat16()
    cu_q = torch.tensor([0] + list(torch.tensor(lens_q).cumsum(0)), dtype=torch.int32)
    cu_k = torch.tensor([0] + list(torch.tensor(lens_k).cumsum(0)), dtype=torch.int32)
    for causal in (False, True):
        want, lses = R.varlen_attention_fp32(q, k, v, cu_q, cu_k, causal=causal)
        out = torch.empty_like(q, device=DEV)
        lse = bp.flash_fwd(q.to(DEV), k.to(DEV), v.to(DEV), out, cu_q.to(DEV), cu_k.to(DEV),
                           max(lens_q), max(lens_k), d ** -0.5, causal)
        # (the oracle's output is rounded to bf16 like the kernel's; compare both against its unrounded values)
        want32 = torch.cat([R.attention_fp32(q[None, a:b].float(), k[None, c:e].float(), v[None, c:e].float(), causal=causal)[0][0]
                            if e > c else torch.zeros(b - a, h, d)
                            for a, b, c, e in zip(cu_q[:-1].tolist(), cu_q[1:].tolist(), cu_k[:-1].tolist(), cu_k[1:].tolist())])
        assert (want32 - want.float()).abs().max().item() < 2e-2        # the two oracle forms agree to bf16 rounding
        _within_2x(out, want32, _varlen_eager16(q, k, v, cu_q, cu_k, causal, d ** -0.5), f'cross lengths causal={causal}')
        assert torch.equal(out[70:75].cpu(), torch.zeros(5, h, d, dtype=torch.bfloat16))
        assert torch.isinf(lse[1, :, :5]).all() and (lse[1, :, :5] < 0).all()
        assert (lse[0, :, :70].cpu() - lses[0]).abs().max().item() < 2e-3


def test_flash_fwd_determinism():
    """10 repeats bit-identical (reference race test, tests/test_flash_attn.py:774-793)."""
    torch.manual_seed(1)
    qkv = torch.randn(4, 300, 3, 4, 64, device=DEV).bfloat16()
    o0, l0 = run_flash_fixed(qkv, 0.125, True)
    for _ in range(10):
        o, l = run_flash_fixed(qkv, 0.125, True)
        assert torch.equal(o, o0) and torch.equal(l, l0)


def test_flash_fwd_forced_rescale():
    """A key that spikes late in the sequence forces the online-softmax rescale branch."""
    torch.manual_seed(5)
    b, s, h, d = 1, 512, 2, 64
    qkv = torch.randn(b, s, 3, h, d) * 0.5
    qkv[:, 400, 1] = qkv[:, 450, 0] * 6.0   # key 400 aligned with query 450 -> huge score late
    qkv16 = qkv.bfloat16()
    ref = R.attention_fp32(qkv16[:, :, 0].float(), qkv16[:, :, 1].float(), qkv16[:, :, 2].float(), causal=True)[0]
    eager = R.attention_fp32(qkv16[:, :, 0], qkv16[:, :, 1], qkv16[:, :, 2], causal=True, upcast=False, reorder_ops=True)[0]
    out, _ = run_flash_fixed(qkv16.to(DEV), d ** -0.5, True)
    _within_2x(out, ref, eager, 'late spike')


# ---------------------------------------------------------------------------------------------
# Backpack sense weights and fused mix
# ---------------------------------------------------------------------------------------------
def _qk_from_golden(g, tag):
    w, bias, h = from_bits16(g[f'{tag}_w']), from_bits16(g[f'{tag}_b']), from_bits16(g[f'{tag}_h'])
    k = int(g[f'{tag}_k'])
    b, s, d = h.shape
    qk32 = torch.nn.functional.linear(h, w, bias).reshape(b, s, 2, k, d // k)
    return qk32, k


@pytest.mark.parametrize('tag', ['dk24', 'dk48', 'dk40', 'dk10'])
def test_sense_alpha_golden(tag):
    """G1: alpha of the reference's ContextSelfAttn; zeros above the diagonal bit-exact."""
    bp = _bp()
    g = load_golden('g12_sense.npz')
    qk32, k = _qk_from_golden(g, tag)
    qk16 = qk32.bfloat16()
    want = R.sense_alpha_from_qk(qk16.float())          # fp32 oracle on the 16-bit inputs
    eager = R.sense_alpha_from_qk(qk16)                 # reference op order in bf16
    alpha = bp.sense_alpha(qk16.to(DEV))
    rel_check(alpha, want, eager, f'alpha {tag}', atol=4e-3)
    s = alpha.shape[-1]
    upper = torch.triu(torch.ones(s, s, dtype=torch.bool), 1)
    assert torch.count_nonzero(alpha.cpu()[:, :, upper]) == 0
    # and against the stored reference output (fp32 inputs -> looser: input rounding of qk)
    assert (alpha.float().cpu() - torch.from_numpy(g[f'{tag}_alpha'])).abs().max().item() < 0.05


@pytest.mark.parametrize('tag', ['dk24', 'dk48', 'dk40', 'dk10'])
def test_sense_mix_golden(tag):
    """G2: fused mix vs sum(alpha @ C) of the reference."""
    bp = _bp()
    g = load_golden('g12_sense.npz')
    qk32, k = _qk_from_golden(g, tag)
    qk16 = qk32.bfloat16()
    content = from_bits16(g[f'{tag}_content'])           # (B,S,k,dout) storage layout
    c16 = content.bfloat16()
    want = R.sense_mix_from_qk_fp32(qk16, c16.transpose(1, 2))
    eager = R.sense_mix(R.sense_alpha_from_qk(qk16), c16.transpose(1, 2))
    out = bp.sense_mix(qk16.to(DEV), c16.to(DEV))
    rel_check(out, want, eager, f'mix {tag}')


@pytest.mark.parametrize('tag', ['dk160', 'dk640'])
def test_wide_sense_golden(tag):
    """G9: alpha and the fused mix at the reference's few-sense widths (d_k = 160 / 640: the ring kernels of
    csrc/sense_wide_dma.hip compute the LSE and the mix, sense_wide.hip dumps alpha) against the REFERENCE's own
    ContextSelfAttn output and `torch.sum(alpha @ content, 1)` (tests/golden/make_golden_r6.py)."""
    bp = _bp()
    g = load_golden('g9_wide_sense.npz')
    w, b, h = from_bits16(g[f'{tag}_w']), from_bits16(g[f'{tag}_b']), from_bits16(g[f'{tag}_h'])
    k = int(g[f'{tag}_k'])
    qk16 = torch.nn.functional.linear(h, w, b).reshape(h.shape[0], h.shape[1], 2, k, -1).bfloat16()
    assert qk16.shape[-1] == int(tag[2:]) and qk16.shape[1] % 32 == 0
    c16 = from_bits16(g[f'{tag}_content']).bfloat16()                        # (B,S,k,dout) storage layout
    want_alpha = R.sense_alpha_from_qk(qk16.float())
    alpha = bp.sense_alpha(qk16.to(DEV))
    rel_check(alpha, want_alpha, R.sense_alpha_from_qk(qk16), f'G9 alpha {tag}', atol=4e-3)
    s = alpha.shape[-1]
    assert torch.count_nonzero(alpha.cpu()[:, :, torch.triu(torch.ones(s, s, dtype=torch.bool), 1)]) == 0
    assert (alpha.float().cpu() - torch.from_numpy(g[f'{tag}_alpha'])).abs().max().item() < 0.05   # (input rounding of qk)
    want = R.sense_mix_from_qk_fp32(qk16, c16.transpose(1, 2))
    eager = R.sense_mix(R.sense_alpha_from_qk(qk16), c16.transpose(1, 2))
    out = bp.sense_mix(qk16.to(DEV), c16.to(DEV))
    rel_check(out, want, eager, f'G9 mix {tag}')
    ref_mixed = torch.from_numpy(g[f'{tag}_mixed'])
    assert (out.float().cpu() - ref_mixed).abs().max().item() < 0.05 * max(1.0, ref_mixed.abs().max().item())
    lse = bp.sense_lse(qk16.to(DEV))[:, :, :s].cpu()
    q32, k32 = qk16[:, :, 0].float().transpose(1, 2), qk16[:, :, 1].float().transpose(1, 2)
    scores = (q32 @ k32.transpose(2, 3) * qk16.shape[-1] ** -0.5).masked_fill(
        torch.triu(torch.ones(s, s, dtype=torch.bool), 1), float('-inf'))
    assert (lse - torch.logsumexp(scores, -1)).abs().max().item() < 2e-3


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
@pytest.mark.parametrize('shape', [
    # (B, S, k, d_k, d_out)
    (2, 300, 16, 48, 768),     # Small dims, ragged S
    (1, 1024, 16, 48, 768),    # BASELINE config 2 shape, one sample
    (2, 257, 64, 10, 640),     # Mini k=64: d_k = 10 (element-wise loader), d_out not a multiple of 256
    (2, 128, 16, 24, 384),     # Micro
    (1, 513, 16, 40, 640),     # Mini k=16
    (2, 64, 4, 16, 100),       # odd d_out (element-wise stores)
])
def test_sense_mix_random(shape, dtype):
    bp = _bp()
    b, s, k, dk, dout = shape
    torch.manual_seed(s + dk)
    qk = (torch.randn(b, s, 2, k, dk) * 1.2).to(dtype)
    c = torch.randn(b, s, k, dout).to(dtype)
    want = R.sense_mix_from_qk_fp32(qk, c.transpose(1, 2))
    eager = R.sense_mix(R.sense_alpha_from_qk(qk), c.transpose(1, 2))
    out = bp.sense_mix(qk.to(DEV), c.to(DEV))
    rel_check(out, want, eager, f'mix {shape} {dtype}')


def test_sense_mix_strided_views():
    """qk and content as the model hands them over: slices of bigger buffers."""
    bp = _bp()
    torch.manual_seed(9)
    b, s, k, dk, dout = 2, 200, 16, 48, 768
    big = torch.randn(b, s + 8, 2 * k * dk + 64).bfloat16().to(DEV)
    qk = big[:, 4:4 + s, 32:32 + 2 * k * dk].unflatten(-1, (2, k, dk))
    cbig = torch.randn(b, s, k * dout + 16).bfloat16().to(DEV)
    c = cbig[:, :, 8:8 + k * dout].unflatten(-1, (k, dout))
    want = R.sense_mix_from_qk_fp32(qk.cpu(), c.cpu().transpose(1, 2))
    eager = R.sense_mix(R.sense_alpha_from_qk(qk.cpu()), c.cpu().transpose(1, 2))
    out = bp.sense_mix(qk, c)
    rel_check(out, want, eager, 'mix strided')


def test_sense_alpha_rows_sum_to_one_full_size():
    """BASELINE config-2 size: size-independent properties of alpha (row sums, exact zeros)."""
    bp = _bp()
    torch.manual_seed(2)
    qk = torch.randn(2, 1024, 2, 16, 48, device=DEV).bfloat16()
    alpha = bp.sense_alpha(qk)
    sums = alpha.float().sum(-1)
    assert (sums - 1).abs().max().item() < 2e-2
    upper = torch.triu(torch.ones(1024, 1024, dtype=torch.bool, device=DEV), 1)
    assert torch.count_nonzero(alpha[:, :, upper]) == 0
    assert torch.equal(alpha[:, :, 0, 0], torch.ones_like(alpha[:, :, 0, 0]))


def test_sense_mix_linearity_full_size():
    """mix(qk, a*C1 + C2) == a*mix(qk, C1) + mix(qk, C2) up to 16-bit rounding, at full size."""
    bp = _bp()
    torch.manual_seed(4)
    b, s, k, dk, d = 2, 1024, 16, 48, 768
    qk = torch.randn(b, s, 2, k, dk, device=DEV).bfloat16()
    c1 = torch.randn(b, s, k, d, device=DEV).bfloat16()
    c2 = torch.randn(b, s, k, d, device=DEV).bfloat16()
    o1 = bp.sense_mix(qk, c1).float()
    o2 = bp.sense_mix(qk, c2).float()
    c12 = (2 * c1.float() + c2.float()).bfloat16()
    o12 = bp.sense_mix(qk, c12).float()
    # yardstick: the same identity through the reference's eager op sequence in bf16 on the GPU (softmax of q k^T, then
    # torch.sum(alpha @ content, dim=1), training/src/models/backpack.py:116-122,313); the fused path may miss it by at
    # most twice as much (both pay the rounding of 2 c1 + c2 and of the three outputs to 16 bit)
    alpha = R.sense_alpha_from_qk(qk)
    e1, e2, e12 = (R.sense_mix(alpha, c.transpose(1, 2)).float() for c in (c1, c2, c12))
    miss_hip = (o12 - (2 * o1 + o2)).abs().max().item()
    miss_eager = (e12 - (2 * e1 + e2)).abs().max().item()
    print(f'linearity at full size: fused {miss_hip:.3e} eager-bf16 {miss_eager:.3e}')
    assert miss_hip <= 2 * miss_eager + 1e-5, (miss_hip, miss_eager)
    # row 0 attends only to itself: out[0] = sum_l C[0, l]
    want0 = c1[:, 0].float().sum(1)
    assert (o1[:, 0] - want0).abs().max().item() < 0.1


def test_errors_are_loud():
    bp = _bp()
    q = torch.randn(64, 2, 64, device=DEV).bfloat16()
    out = torch.empty_like(q)
    cu = torch.tensor([0, 64], dtype=torch.int32, device=DEV)
    with pytest.raises(RuntimeError):
        bp.flash_fwd(q.float(), q.float(), q.float(), out.float(), cu, cu, 64, 64, 0.125, True)
    with pytest.raises(RuntimeError):
        bp.flash_fwd(q, q, q, out, cu, cu, 64, 64, float('nan'), True)
    big = torch.randn(64, 2, 136, device=DEV).bfloat16()
    with pytest.raises(RuntimeError):
        bp.flash_fwd(big, big, big, torch.empty_like(big), cu, cu, 64, 64, 0.1, True)
    with pytest.raises(RuntimeError):
        bp.flash_fwd(q.cpu(), q.cpu(), q.cpu(), out.cpu(), cu.cpu(), cu.cpu(), 64, 64, 0.125, True)


# ---------------------------------------------------------------------------------------------
# fused residual add + LayerNorm
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
@pytest.mark.parametrize('cols', [768, 384, 640, 64, 1024, 2560, 8192, 100])
@pytest.mark.parametrize('mode', ['fp32res', 'nores_fp32out', 'same_dtype', 'z_only'])
def test_add_layer_norm(cols, dtype, mode):
    bp = _bp()
    torch.manual_seed(cols)
    rows = 37 if cols > 1024 else 203
    x0 = (torch.randn(rows, cols) * 2).to(dtype)
    w = (1 + 0.2 * torch.randn(cols)).to(dtype)
    b = (0.1 * torch.randn(cols)).to(dtype)
    x1 = None
    rdt = None
    if mode == 'fp32res':
        x1 = torch.randn(rows, cols) * 3
    elif mode == 'same_dtype':
        x1 = (torch.randn(rows, cols) * 3).to(dtype)
    elif mode == 'nores_fp32out':
        rdt = torch.float32
    z_ref, x_ref = R.add_layer_norm_fp32(x0, x1, w, b, 1e-5, residual_dtype=rdt)
    z32 = R.add_layer_norm_fp32(x0.float(), x1.float() if x1 is not None else None, w.float(), b.float(), 1e-5)[0]
    dev = lambda t: t.to(DEV) if t is not None else None
    if mode == 'z_only':
        z = bp.add_layer_norm(dev(x0), None, dev(w), dev(b), 1e-5, return_residual=False)
        x = None
    else:
        z, x = bp.add_layer_norm(dev(x0), dev(x1), dev(w), dev(b), 1e-5, residual_dtype=rdt)
    # eager same-dtype baseline: what torch does on the 16-bit tensors (unfused reference path)
    res = (x0 + x1) if x1 is not None else x0
    eager = torch.nn.functional.layer_norm(res.to(dtype), (cols,), w, b, 1e-5)
    rel_check(z, z32, eager, f'ln {cols} {dtype} {mode}', atol=2e-3 if dtype == torch.float16 else 2e-2)
    if x is not None:
        assert x.dtype == x_ref.dtype
        assert torch.equal(x.cpu(), x_ref), 'residual stream must be the exactly rounded fp32 sum'


def test_add_layer_norm_rejects_bad_shapes():
    bp = _bp()
    x = torch.randn(4, 66, device=DEV).bfloat16()      # cols % 4 != 0
    w = torch.ones(66, device=DEV).bfloat16()
    with pytest.raises(RuntimeError):
        bp.add_layer_norm(x, None, w, w, 1e-5)


# ---------------------------------------------------------------------------------------------
# callers either side of the path: the intervention scripts edit `content`
# (or, equivalently, columns of alpha) and redo sum_l alpha_l @ C_l -- the fused kernel takes their
# tensors as they are: (B,k,S,d_out)-contiguous content, vocab-sized d_out.
# ---------------------------------------------------------------------------------------------
def test_sense_mix_intervention_shapes():
    """training/src/models/intervened_models.py:97-101 (content * per-(b,s,l) weights, a fresh
    (B,k,S,d) tensor) and :153-161 (content logits with a vocab-sized trailing dim)."""
    bp = _bp()
    torch.manual_seed(21)
    b, s, k, dk, d, vocab = 2, 96, 16, 24, 384, 1000
    qk = (torch.randn(b, s, 2, k, dk) * 1.3).bfloat16()
    content = torch.randn(b, s, k * d).bfloat16().reshape(b, s, k, d).transpose(1, 2)      # model's view
    weights = torch.rand(b, s, k)                                                          # soft mask
    weighted = (content * weights.transpose(1, 2).unsqueeze(3).bfloat16()).contiguous()   # (B,k,S,d)
    assert weighted.is_contiguous()
    want = R.sense_mix_from_qk_fp32(qk, weighted)
    eager = R.sense_mix(R.sense_alpha_from_qk(qk), weighted)
    out = bp.sense_mix(qk.to(DEV), weighted.to(DEV).transpose(1, 2))                        # strided view, no copy
    rel_check(out, want, eager, 'mix weighted content')
    # scaling a column of alpha == scaling that key's content row (test_genderbias.py:71-78)
    alpha = R.sense_alpha_from_qk(qk.float())
    col = torch.ones(b, k, 1, s)
    col[:, 3, :, 40] = 0.25
    want2 = R.sense_mix(alpha * col, content.float())
    scaled = content.float() * col.squeeze(2).unsqueeze(3)
    out2 = bp.sense_mix(qk.to(DEV), scaled.bfloat16().to(DEV).transpose(1, 2))
    assert (out2.float().cpu() - want2).abs().max().item() < 0.06
    # vocab-sized trailing dimension
    lm_w = (torch.randn(vocab, d) * 0.05).bfloat16()
    logits_c = (content.float() @ lm_w.float().t()).bfloat16()                               # (B,k,S,V)
    want3 = R.sense_mix_from_qk_fp32(qk, logits_c)
    eager3 = R.sense_mix(R.sense_alpha_from_qk(qk), logits_c)
    out3 = bp.sense_mix(qk.to(DEV), logits_c.to(DEV).transpose(1, 2))
    assert out3.shape == (b, s, vocab)
    rel_check(out3, want3, eager3, 'mix vocab-sized')


@pytest.mark.parametrize('shape', [
    (2, 200, 16, 48, 768),     # fast (LDS-DMA) path, two query tiles
    (1, 96, 4, 24, 100),       # generic path (d_out % 8 != 0)
    (2, 130, 64, 10, 640),     # zero-padded d_k = 10
])
def test_sense_mix_key_weight_hook(shape):
    """bp_sense_mix_weighted: alpha[b,l,:,s] * w[b,l,s] fused == the reference's eager edit of alpha
    (test_genderbias.py:71-78) == re-weighting the content rows (intervened_models.py:97-101)."""
    bp = _bp()
    b, s, k, dk, d = shape
    torch.manual_seed(31)
    qk = (torch.randn(b, s, 2, k, dk) * 1.2).bfloat16()
    c = torch.randn(b, s, k, d).bfloat16()
    w = torch.rand(b, k, s) * 2.0
    w[:, 1, s // 2] = 0.0                                      # a knocked-out (sense, token)
    alpha = R.sense_alpha_from_qk(qk.float())                   # (B,k,S,S) fp32
    want = R.sense_mix(alpha * w.unsqueeze(2), c.float().transpose(1, 2))
    weighted_c = (c.float() * w.transpose(1, 2).unsqueeze(3)).bfloat16()
    eager = R.sense_mix(R.sense_alpha_from_qk(qk), weighted_c.transpose(1, 2))
    out = bp.sense_mix(qk.to(DEV), c.to(DEV), key_weight=w.to(DEV))
    rel_check(out, want, eager, f'mix key_weight {shape}', atol=2e-3)
    plain = bp.sense_mix(qk.to(DEV), c.to(DEV))
    ones = bp.sense_mix(qk.to(DEV), c.to(DEV), key_weight=torch.ones(b, k, s, device=DEV))
    assert torch.equal(plain, ones)                             # weight 1 is exactly the unweighted kernel


# ---------------------------------------------------------------------------------------------
# BASELINE config 5 shape (S = 4096, fp16): oracle on a subset of query rows (full key range)
# ---------------------------------------------------------------------------------------------
def _rows_softmax(scores, rows, s):
    mask = torch.arange(s)[None, :] > rows[:, None]
    return scores.masked_fill(mask[None], float('-inf'))


def test_flash_fwd_seq4096_fp16_rows():
    """2x-eager rule (tests/test_flash_attn.py:424-428) on a subset of query rows, full key range; one row carries a
    score spike beyond the fp16 limit of the fast tile body (2^14, see tests/test_gpu_retry.py) deep in the sweep."""
    bp = _bp()
    torch.manual_seed(40)
    s, h, d = 4096, 12, 64
    qkv = torch.randn(1, s, 3, h, d).half()
    qv = qkv[0, 4000, 0, 3].float()
    qkv[0, 2500, 1, 3] = (qv * (14.0 / (qv.dot(qv).item() * d ** -0.5))).half()      # ~ +10 nats over the row maximum
    out, lse = run_flash_fixed(qkv.to(DEV), d ** -0.5, True)
    rows = torch.tensor([0, 1, 63, 64, 127, 128, 2047, 2048, 4000, 4095])
    q16, k16, v16 = qkv[0, :, 0], qkv[0, :, 1], qkv[0, :, 2]
    scores = _rows_softmax(torch.einsum('thd,shd->hts', q16[rows].float(), k16.float()) * d ** -0.5, rows, s)
    want = torch.einsum('hts,shd->thd', torch.softmax(scores, -1), v16.float())
    # the reference test's same-dtype eager: scale folded into k, every op in fp16
    s16 = _rows_softmax(torch.einsum('thd,shd->hts', q16[rows], k16 * d ** -0.5), rows, s)
    eager = torch.einsum('hts,shd->thd', torch.softmax(s16, -1), v16)
    rel_check(out[0, rows], want, eager, 'flash S=4096 fp16 rows')
    assert (lse[0][:, rows].cpu() - torch.logsumexp(scores, -1)).abs().max().item() < 2e-3
    assert scores[3, 8].max().item() - scores[3, 8, :64].max().item() > 9.8           # the spike is there


def test_sense_mix_seq4096_fp16_rows():
    bp = _bp()
    torch.manual_seed(41)
    s, k, dk, d = 4096, 16, 48, 768
    qk = torch.randn(1, s, 2, k, dk).half()
    c = torch.randn(1, s, k, d).half()
    out = bp.sense_mix(qk.to(DEV), c.to(DEV))
    rows = torch.tensor([0, 31, 32, 255, 256, 1023, 1024, 3000, 4095])
    q16, k16 = qk[0, :, 0], qk[0, :, 1]
    scores = _rows_softmax(torch.einsum('tld,sld->lts', q16[rows].float(), k16.float()) * dk ** -0.5, rows, s)
    want = torch.einsum('lts,sld->td', torch.softmax(scores, -1), c[0].float())
    # same-dtype eager in the reference's op order (backpack.py:116-122,313): fp16 scores / softmax, one fp16
    # (rows x S) @ (S x d) product per sense, summed over the senses in fp16
    s16 = _rows_softmax(torch.einsum('tld,sld->lts', q16[rows], k16) / math.sqrt(dk), rows, s)
    a16 = torch.softmax(s16, -1)
    eager = torch.stack([a16[l] @ c[0, :, l] for l in range(k)]).sum(0)
    rel_check(out[0, rows], want, eager, 'mix S=4096 fp16 rows', atol=2e-3)


# ---------------------------------------------------------------------------------------------
# Fused cross-entropy (next row 4): criterion of the reference's own test, tests/losses/test_cross_entropy.py
# (rtol/atol 1e-5/1e-6 fp32, 1e-3/1e-4 16-bit against torch CrossEntropyLoss on fp32 logits)
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize('dtype', [torch.float16, torch.float32, torch.bfloat16])
@pytest.mark.parametrize('inplace_backward', [False, True])
@pytest.mark.parametrize('smoothing', [0.0, 0.9])
@pytest.mark.parametrize('vocab_size', [50257, 50264, 1000, 7])
def test_cross_entropy_loss(vocab_size, smoothing, inplace_backward, dtype):
    from flash_attn.losses.cross_entropy import CrossEntropyLossApex
    rtol, atol = (1e-5, 1e-6) if dtype == torch.float32 else (1e-3, 1e-4)
    if dtype == torch.bfloat16:
        rtol, atol = 8e-3, 1e-4            # bf16 has 3 fewer mantissa bits than the reference's fp16 case
    torch.manual_seed(0)
    rows = 256
    x_pt = torch.randn(rows, vocab_size, device=DEV, dtype=dtype, requires_grad=True)
    x = x_pt.detach().clone().requires_grad_()
    y = torch.randint(0, vocab_size, (rows,), dtype=torch.long, device=DEV)
    y[torch.randperm(rows)[:10]] = -100
    out = CrossEntropyLossApex(label_smoothing=smoothing, inplace_backward=inplace_backward)(x, y)
    out_pt = torch.nn.CrossEntropyLoss(label_smoothing=smoothing)(x_pt.float(), y)
    assert torch.allclose(out, out_pt, rtol=rtol, atol=atol), (out.item(), out_pt.item())
    g = torch.randn_like(out)
    out_pt.backward(g)
    out.backward(g)
    assert torch.allclose(x.grad, x_pt.grad, rtol=rtol, atol=atol)
    # per-row values against the CPU oracle, unreduced
    losses, lse = _bp().xentropy_fwd(x_pt.detach(), y, smoothing)
    want_l, want_lse = R.softmax_cross_entropy(x_pt.detach().cpu(), y.cpu(), smoothing)
    assert (lse.cpu() - want_lse).abs().max().item() < 2e-4
    assert (losses.masked_fill(y == -100, 0).cpu() - want_l).abs().max().item() < 2e-3


def test_cross_entropy_strided_rows_and_shard_labels():
    """Row-strided logits (a column slice of a wider buffer: unaligned row starts) and labels outside the
    local vocabulary slice (vocabulary-parallel contract, cross_entropy.py:41-63)."""
    bp = _bp()
    torch.manual_seed(1)
    wide = torch.randn(64, 3001, device=DEV).bfloat16()
    x = wide[:, 3:2004]                                    # 2001 columns, row stride 3001, odd offset
    y = torch.randint(-500, 2500, (64,), device=DEV)
    losses, lse = bp.xentropy_fwd(x, y, 0.1, total_classes=6000)
    want_l, want_lse = R.softmax_cross_entropy(x.cpu(), y.cpu(), 0.1, total_classes=6000)
    assert (lse.cpu() - want_lse).abs().max().item() < 2e-4
    assert (losses.cpu() - want_l).abs().max().item() < 2e-3
    g = torch.randn(64, device=DEV)
    dx = bp.xentropy_bwd(g, x, lse, y, 0.1, total_classes=6000)
    want = R.softmax_cross_entropy_grad(g.cpu(), x.cpu(), y.cpu(), 0.1, ignored_index=-100, total_classes=6000)
    assert (dx.float().cpu() - want).abs().max().item() < 2e-2 * g.abs().max().item()


def test_chunked_lm_loss_matches_full_logits():
    from src.models.backpack import BackpackConfig, BackpackLMHeadModel
    from src.utils.perplexity import lm_loss_chunked
    cfg = BackpackConfig(n_embd=128, n_head=2, n_layer=2, num_content_vectors=4, vocab_size=1000, n_positions=64,
                         scale_attn_by_inverse_layer_idx=True, resid_pdrop=0.0, embd_pdrop=0.0, attn_pdrop=0.0,
                         use_flash_attn=True, pad_vocab_size_multiple=8)
    torch.manual_seed(2)
    model = BackpackLMHeadModel(cfg).eval().to(DEV, torch.bfloat16)
    ids = torch.randint(0, 1000, (6, 64), device=DEV)
    loss, n = lm_loss_chunked(model, ids, chunk_tokens=100)          # 384 rows in ragged chunks of 100
    with torch.no_grad():
        logits = model(ids).logits.float()
    want = torch.nn.functional.cross_entropy(logits[:, :-1].reshape(-1, logits.shape[-1]), ids[:, 1:].reshape(-1))
    assert n == 6 * 63 and abs(loss.item() - want.item()) < 2e-3


# ---------------------------------------------------------------------------------------------
# Fused add + LayerNorm BACKWARD (next row 3).  Criterion of tests/ops/test_dropout_layer_norm.py:236-250:
# fp32 autograd of the fp32 expression is truth, eager same-dtype autograd the yardstick (x4 + 1e-4 for
# activations' grads, x2 + 2e-4 for weight / bias).
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
@pytest.mark.parametrize('cols', [768, 384, 640, 1024, 2048, 2560])
@pytest.mark.parametrize('mode', ['prenorm_fp32res', 'prenorm_same_dtype', 'postnorm_residual', 'no_residual',
                                  'no_residual_fp32_stream'])
def test_add_layer_norm_backward(cols, dtype, mode):
    from flash_attn.ops.layer_norm import dropout_add_layer_norm
    torch.manual_seed(7)
    rows = 4 * 331 + 3                       # not a multiple of the 4 rows per workgroup
    prenorm = mode.startswith('prenorm')
    has_x1 = mode in ('prenorm_fp32res', 'prenorm_same_dtype', 'postnorm_residual')
    res_dtype = torch.float32 if mode in ('prenorm_fp32res', 'postnorm_residual') else dtype
    w_dtype = torch.float32 if mode != 'prenorm_same_dtype' else dtype
    x0 = torch.randn(rows, cols).to(dtype)
    x1 = torch.randn(rows, cols).to(res_dtype) if has_x1 else None
    w = (1 + 0.1 * torch.randn(cols)).to(w_dtype)
    b = (0.1 * torch.randn(cols)).to(w_dtype)
    dz = torch.randn(rows, cols).to(dtype)
    dxr = torch.randn(rows, cols).to(res_dtype) if prenorm else None

    def run(x0_, x1_, w_, b_, dz_, dxr_, fused):
        x0_, w_, b_ = x0_.clone().requires_grad_(), w_.clone().requires_grad_(), b_.clone().requires_grad_()
        x1_ = x1_.clone().requires_grad_() if x1_ is not None else None
        if fused:
            out = dropout_add_layer_norm(x0_, x1_, w_, b_, 0.0, 1e-5, prenorm=prenorm,
                                         residual_in_fp32=(mode == 'no_residual_fp32_stream'))
            z, xs = out if prenorm else (out, None)
        else:
            xs = x0_.to(x1_.dtype if x1_ is not None else x0_.dtype) + (x1_ if x1_ is not None else 0)
            z = torch.nn.functional.layer_norm(xs.to(w_.dtype if w_.dtype == torch.float32 else xs.dtype),
                                               (cols,), w_, b_, 1e-5).to(x0_.dtype)
        outs, grads = [z], [dz_.to(z.dtype)]
        if prenorm:
            outs.append(xs)
            grads.append(dxr_.to(xs.dtype))
        ins = [x0_, w_, b_] + ([x1_] if x1_ is not None else [])
        return torch.autograd.grad(outs, ins, grads)

    f32 = lambda t: t.float() if t is not None else None
    ref = run(f32(x0), f32(x1), f32(w), f32(b), f32(dz), f32(dxr), fused=False)
    pt = run(x0, x1, w, b, dz, dxr, fused=False)
    dev = lambda t: t.to(DEV) if t is not None else None
    if cols <= 2048:
        got = run(dev(x0), dev(x1), dev(w), dev(b), dev(dz), dev(dxr), fused=True)
    else:   # wider than the HIP backward takes: loud by default, the eager expression on request
        with pytest.raises(RuntimeError, match='allow_eager_fallback'):
            run(dev(x0), dev(x1), dev(w), dev(b), dev(dz), dev(dxr), fused=True)
        with _bp().allow_eager_fallback():
            got = run(dev(x0), dev(x1), dev(w), dev(b), dev(dz), dev(dxr), fused=True)
    names = ['dx0', 'dweight', 'dbias'] + (['dx1'] if has_x1 else [])
    for g, r, e, n in zip(got, ref, pt, names):
        err = (g.float().cpu() - r).abs().max().item()
        base = (e.float() - r).abs().max().item()
        factor, atol = (4, 1e-4) if n.startswith('dx') else (2, 2e-4 * max(1.0, r.abs().max().item()))
        assert torch.isfinite(g.float()).all()
        assert err <= factor * base + atol, (n, err, base)
    assert got[0].dtype == dtype and got[1].dtype == w_dtype
    if has_x1:
        assert got[3].dtype == res_dtype


def test_add_layer_norm_backward_is_deterministic():
    from flash_attn.ops.layer_norm import dropout_add_layer_norm
    torch.manual_seed(8)
    x0 = torch.randn(5000, 768, device=DEV).bfloat16().requires_grad_()
    x1 = torch.randn(5000, 768, device=DEV).requires_grad_()
    w = torch.randn(768, device=DEV).requires_grad_()
    b = torch.randn(768, device=DEV).requires_grad_()
    dz = torch.randn(5000, 768, device=DEV).bfloat16()

    def grads():
        z = dropout_add_layer_norm(x0, x1, w, b, 0.0, 1e-5)
        return torch.autograd.grad(z, [x0, x1, w, b], dz)
    first = grads()
    for _ in range(3):
        for a, c in zip(first, grads()):
            assert torch.equal(a, c)


@pytest.mark.parametrize('shape', [(2, 1024, 16, 48, 768, 5000), (3, 300, 4, 24, 104, 97), (1, 2048, 16, 48, 256, 50264),
                                   (2, 640, 64, 10, 640, 1000), (2, 77, 4, 16, 512, 7), (1, 2048, 4, 80, 256, 300),
                                   (1, 4096, 4, 48, 256, 300), (1, 2112, 4, 80, 256, 300), (1, 3000, 4, 16, 64, 65536)])
@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
def test_sense_mix_gather_equals_sense_mix_on_the_gathered_rows(shape, dtype):
    """bp_sense_mix_gather (content rows read from a per-token table through a row index, ABI 6) against bp_sense_mix on the
    materialised content[b, s] = table[index[b, s]]: the same kernel arithmetic on the same values -> bit-identical,
    including partial last tiles (S = 300, 77), partial column chunks (d_out = 104), the Mini widths (d_k = 10 carried
    as 16, k = 64) and a full-vocabulary table (1.2 GB of rows at Small's width would not fit the test: 256 columns)."""
    bp = _bp()
    b, s, k, dk, d, rows = shape
    torch.manual_seed(11)
    qk = (torch.randn(b, s, 2, k, dk, device=DEV) * 0.9).to(dtype)
    table = torch.randn(rows, k, d, device=DEV).to(dtype)
    index = torch.randint(0, rows, (b, s), device=DEV, dtype=torch.int32)
    index[0, 0], index[-1, -1] = 0, rows - 1
    assert bp.sense_mix_gather_supported(qk, table, s)
    want = bp.sense_mix(qk, table[index.long()])
    got = bp.sense_mix_gather(qk, table, index)
    assert torch.equal(got, want)
    lse = bp.sense_lse(qk)
    out = torch.full_like(want, float('nan'))
    assert bp.sense_mix_gather(qk, table, index, out=out, lse=lse) is out and torch.equal(out, want)


@pytest.mark.parametrize('shape', [(2, 256, 32, 24, 768, 65536), (2, 256, 64, 10, 640, 50264), (1, 320, 64, 16, 512, 65535)])
def test_sense_mix_gather_table_offsets_beyond_2_gib(shape):
    """The gathering mix forms `row_index * row bytes` as an UNSIGNED 32-bit offset (sense_mix_dma.hip; bp_api.hip admits
    tables of up to 4 GiB and 65 536 rows): tables whose rows lie on both sides of byte offset 2^31 -- 65 536 rows of 32
    senses x 768 (3.2 GB), Mini k = 64's REAL table 50 264 x 64 x 640 (4.12 GB, 96 % of the range; what `bench.py --workload
    mini-k64-1024` and the cached vocabulary table read), and the largest table the entry point takes (65 535 rows of
    64 KB = 4 GiB - 64 KB).  Index pinned to row 0, the rows just below / across / above 2^31 and the last rows in
    both samples; the rest random over the whole table.  Bit-identical to bp_sense_mix on the materialised rows."""
    bp = _bp()
    b, s, k, dk, d, rows = shape
    torch.manual_seed(21)
    qk = (torch.randn(b, s, 2, k, dk, device=DEV) * 0.9).bfloat16()
    table = torch.randn(rows, k, d, device=DEV, dtype=torch.bfloat16)
    row_bytes = k * d * 2
    assert 2 ** 31 < rows * row_bytes < 2 ** 32
    edge = 2 ** 31 // row_bytes                    # the row that holds byte 2^31 (or starts exactly there)
    index = torch.randint(0, rows, (b, s), device=DEV, dtype=torch.int32)
    pins = torch.tensor([0, edge - 1, edge, edge + 1, rows - 1, rows - 2], device=DEV, dtype=torch.int32)
    for bi in range(b):
        index[bi, :6] = pins                       # seen by every later query of the sample
        index[bi, -6:] = pins.flip(0)              # ... and on the diagonal of the last query tile
    assert bp.sense_mix_gather_supported(qk, table, s)
    want = bp.sense_mix(qk, table[index.long()])
    got = bp.sense_mix_gather(qk, table, index)
    assert torch.equal(got, want)
    # the rows beyond 2^31 matter: zeroing them changes the result (the comparison above is not vacuous)
    upper = index >= edge
    assert upper.float().mean().item() > 0.1
    table[edge:].zero_()
    assert not torch.equal(bp.sense_mix_gather(qk, table, index), want)
    # one row more and the offsets no longer fit: refused by the support check and by the entry point (BP_ERR_SHAPE)
    if rows * row_bytes + row_bytes >= 2 ** 32:
        del table
        bigger = torch.empty(rows + 1, k, d, device=DEV, dtype=torch.bfloat16)
        assert not bp.sense_mix_gather_supported(qk, bigger, s)
        with pytest.raises(RuntimeError, match='bp_sense_mix_gather'):
            bp.sense_mix_gather(qk, bigger, index)


def test_sense_mix_gather_clamps_indices_outside_the_table():
    """`row_index` is not validated; the kernel clamps it as an unsigned value to the last row (include/bp_hip.h), so a
    negative or too large index reads the table's last row and never memory outside it."""
    bp = _bp()
    torch.manual_seed(5)
    qk = torch.randn(2, 200, 2, 4, 16, device=DEV).bfloat16()
    table = torch.randn(50, 4, 256, device=DEV).bfloat16()
    index = torch.randint(0, 50, (2, 200), device=DEV, dtype=torch.int32)
    bad = index.clone()
    bad[0, 3], bad[0, 77], bad[1, 199], bad[1, 0] = -1, 50, 2 ** 31 - 1, -(2 ** 31)
    good = bad.clone()
    good[0, 3], good[0, 77], good[1, 199], good[1, 0] = 49, 49, 49, 49
    assert torch.equal(bp.sense_mix_gather(qk, table, bad), bp.sense_mix_gather(qk, table, good))


def test_flash_varlen_reads_cu_seqlens_even_when_the_buffer_has_batch_times_max_rows():
    """An over-allocated (B * max_seqlen)-row buffer with SHORTER sequences in cu_seqlens is legal (the reference's
    mha_fwd reads the offsets only, csrc/flash_attn/fmha_api.cpp:189-325): the binding must not infer a fixed-length
    batch from the row count (round-4 advisor finding).  Forward and backward against per-sequence eager attention."""
    bp = _bp()
    torch.manual_seed(2)
    b, smax, h, d = 3, 128, 2, 64
    lens = [128, 40, 77]
    cu = torch.tensor([0, 128, 168, 245], dtype=torch.int32, device=DEV)
    q, k, v = (torch.randn(b * smax, h, d, device=DEV).bfloat16() for _ in range(3))
    out = torch.zeros_like(q)
    scale = d ** -0.5
    lse = bp.flash_fwd(q, k, v, out, cu, cu, smax, smax, scale, True)
    dout = torch.randn_like(q)
    dq, dk, dv = torch.zeros_like(q), torch.zeros_like(k), torch.zeros_like(v)
    bp.flash_bwd(dout, q, k, v, out, lse, dq, dk, dv, cu, cu, smax, smax, scale, True)
    for i, n in enumerate(lens):
        lo = int(cu[i])
        qi, ki, vi = (t[lo:lo + n].float().requires_grad_() for t in (q, k, v))
        ref, _, _ = R.attention_fp32(qi[None], ki[None], vi[None], causal=True, softmax_scale=scale)
        ref = ref[0]
        assert (out[lo:lo + n].float() - ref).abs().max().item() < 2e-2
        gq, gk, gv = torch.autograd.grad(ref, (qi, ki, vi), dout[lo:lo + n].float())
        for got, want in ((dq, gq), (dk, gk), (dv, gv)):
            assert (got[lo:lo + n].float() - want).abs().max().item() < 6e-2
    assert torch.count_nonzero(out[245:]) == 0 and torch.count_nonzero(dq[245:]) == 0    # rows behind the last sequence


def test_sense_mix_gather_refuses_what_it_does_not_take():
    bp = _bp()
    qk = torch.randn(1, 64, 2, 4, 16, device=DEV).bfloat16()
    table = torch.randn(10, 4, 64, device=DEV).bfloat16()
    index = torch.zeros(1, 64, device=DEV, dtype=torch.int32)
    with pytest.raises(RuntimeError, match='int32'):
        bp.sense_mix_gather(qk, table, index.long())
    with pytest.raises(RuntimeError, match='table must be'):
        bp.sense_mix_gather(qk, table[:, :3], index)
    long_qk = torch.randn(1, 4160, 2, 4, 16, device=DEV).bfloat16()
    assert not bp.sense_mix_gather_supported(long_qk, table, 4160)
    with pytest.raises(RuntimeError, match='bp_sense_mix_gather'):
        bp.sense_mix_gather(long_qk, table, torch.zeros(1, 4160, device=DEV, dtype=torch.int32))   # BP_ERR_SHAPE: the caller gathers
    many_rows = torch.empty(65537, 4, 64, device=DEV).bfloat16()             # the job's row indices are u16 in LDS
    assert not bp.sense_mix_gather_supported(qk, many_rows, 64)
    with pytest.raises(RuntimeError, match='bp_sense_mix_gather'):
        bp.sense_mix_gather(qk, many_rows, index)


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
@pytest.mark.parametrize('shape', [
    # (B, S, k, d_k, d_out): wide senses, csrc/sense_wide.hip
    (2, 1024, 4, 160, 640),    # backpack-mini-flash-vecs-4.yaml
    (1, 1024, 1, 640, 640),    # backpack-mini-flash-vecs-1.yaml
    (2, 300, 2, 136, 200),     # just beyond 128, ragged S, d_out not a multiple of 128
    (3, 97, 3, 256, 72),       # one partial query tile
    (1, 257, 1, 632, 384),     # d_k not a multiple of 16: zero columns in the last K step
    (2, 129, 2, 132, 100),     # d_k % 8 != 0: widened to 136 by the binding; odd d_out (element-wise stores)
    # the staged kernels' K loop run to its compile-time end (no early exit): the paths on which the round-6 listing scan
    # found MFMA results read 3 of 11 wait states early (bp_common.h, settle_acc)
    (1, 32, 4, 160, 64),       # ring kernels at their smallest: one key block, one wave with rows
    (2, 2080, 1, 640, 320),    # ... and past 2048 keys (65 key blocks per sweep), one 320-column chunk
    (1, 160, 1, 192, 64),      # d_k = 192 = the whole small class
    (1, 96, 1, 640, 72),       # d_k = 640 off the ring kernels (S not a multiple of 32), the whole large class
    (2, 1024, 4, 160, 640, 'view'),   # d_k = 160 on a 4-byte-aligned view: the staged kernels' element-wise loaders at 160
])
def test_wide_senses_lse_alpha_and_mix(shape, dtype):
    """Senses wider than 128 (the reference's few-sense ablations: d_k = 160 / 640) through bp_sense_lse / bp_sense_alpha /
    bp_sense_mix: each against the fp32 oracle under the kernel tests' 2 x rule, exact zeros above the diagonal of alpha,
    rows of alpha summing to one, the fused mix equal to alpha @ C of the dumped alpha, and bp_sense_mix_weighted's hook."""
    bp = _bp()
    b, s, k, dk, dout = shape[:5]
    odd_view = len(shape) > 5
    torch.manual_seed(s + dk)
    qk = (torch.randn(b, s, 2, k, dk) * (2.0 / dk ** 0.25)).to(dtype)
    c = torch.randn(b, s, k, dout).to(dtype)
    scale = dk ** -0.5
    q32, k32 = qk[:, :, 0].float().transpose(1, 2), qk[:, :, 1].float().transpose(1, 2)      # (B,k,S,dk)
    scores = q32 @ k32.transpose(2, 3) * scale
    mask = torch.triu(torch.ones(s, s, dtype=torch.bool), 1)
    lse_want = torch.logsumexp(scores.masked_fill(mask, float('-inf')), -1)
    alpha_want = torch.softmax(scores.masked_fill(mask, float('-inf')), -1)
    alpha_eager = R.sense_alpha_from_qk(qk)
    want = R.sense_mix_from_qk_fp32(qk, c.transpose(1, 2))
    eager = R.sense_mix(alpha_eager, c.transpose(1, 2))
    g = qk.to(DEV)
    if odd_view:   # rows that start 4 bytes off a 16-byte boundary: no 16-byte loads, no DMA
        buf = torch.zeros(b, s, 2, k, dk + 2, dtype=dtype, device=DEV).flatten()
        g = torch.as_strided(buf, (b, s, 2, k, dk), (s * 2 * k * (dk + 2), 2 * k * (dk + 2), k * (dk + 2), dk + 2, 1), 2)
        g.copy_(qk)
    lse = bp.sense_lse(g)[:, :, :s]
    assert (lse.cpu() - lse_want).abs().max().item() <= 2e-3 * max(1.0, lse_want.abs().max().item())
    alpha = bp.sense_alpha(g)
    rel_check(alpha, alpha_want, alpha_eager, f'wide alpha {shape} {dtype}')
    assert torch.count_nonzero(alpha[:, :, mask.to(DEV)]) == 0
    assert (alpha.float().sum(-1) - 1).abs().max().item() < (2e-2 if dtype == torch.bfloat16 else 4e-3)
    out = bp.sense_mix(g, c.to(DEV))
    rel_check(out, want, eager, f'wide mix {shape} {dtype}')
    # the hook of the intervention experiments: alpha[b, l, :, s] scaled by w[b, l, s]
    w = torch.rand(b, k, s) * 2
    out_w = bp.sense_mix(g, c.to(DEV), key_weight=w.to(DEV))
    want_w = ((alpha_want * w.unsqueeze(2)) @ c.float().transpose(1, 2)).sum(1)
    eager_w = ((alpha_eager.float() * w.unsqueeze(2)).to(dtype) @ c.transpose(1, 2)).sum(1)
    rel_check(out_w, want_w, eager_w, f'wide weighted mix {shape} {dtype}', factor=3.0, atol=2e-3 * want_w.abs().max().item())
    # a strided view (a slice of a bigger projection buffer, as the model hands it over)
    big = torch.zeros(b, s, 2, k + 1, dk + 8, dtype=dtype, device=DEV)
    big[:, :, :, :k, :dk] = g
    assert torch.equal(bp.sense_mix(big[:, :, :, :k, :dk], c.to(DEV)), out) or dk % 8 != 0 or odd_view   # (other kernels)


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
@pytest.mark.parametrize('shape', [
    # (B, S, k, d_k, d_out, table rows): the reference's two few-sense widths on the ring kernels of csrc/sense_wide_dma.hip
    (2, 1024, 4, 160, 640, 3000),      # backpack-mini-flash-vecs-4.yaml
    (2, 1024, 1, 640, 640, 70000),     # ...-vecs-1.yaml; more than 65 536 rows (the narrow kernel's u16 limit does not apply)
    (3, 96, 4, 160, 200, 17),          # one partial query tile, d_out not a multiple of 320
])
def test_wide_senses_mix_gathers_from_the_table(shape, dtype):
    """bp_sense_mix_gather at d_k = 160 / 640 (ABI 9): the content rows read from the per-token table inside the ring kernel
    give the BITS of bp_sense_mix on the gathered (B,S,k,d) tensor (same kernel, same order of operations); indices outside
    the table are clamped to its last row; widths the ring does not take are refused (the caller gathers)."""
    bp = _bp()
    b, s, k, dk, dout, rows = shape
    torch.manual_seed(rows)
    qk = (torch.randn(b, s, 2, k, dk, device=DEV) * (2.0 / dk ** 0.25)).to(dtype)
    table = torch.randn(rows, k, dout, device=DEV).to(dtype)
    index = torch.randint(0, rows, (b, s), device=DEV, dtype=torch.int32)
    index[0, 0], index[-1, -1] = rows - 1, 0
    assert bp.sense_mix_gather_supported(qk, table, s)
    lse = bp.sense_lse(qk)
    want = bp.sense_mix(qk, table[index.long()], lse=lse)
    got = bp.sense_mix_gather(qk, table, index, lse=lse)
    assert torch.equal(got, want)
    assert torch.equal(bp.sense_mix_gather(qk, table, index), want)            # LSE computed inside the call
    bad = index.clone()
    bad[0, 1], bad[-1, 5] = -3, rows + 11                                       # both read the LAST row
    fixed = index.clone()
    fixed[0, 1], fixed[-1, 5] = rows - 1, rows - 1
    assert torch.equal(bp.sense_mix_gather(qk, table, bad, lse=lse), bp.sense_mix(qk, table[fixed.long()], lse=lse))
    # a table with padded rows (a slice of a wider buffer), as the cached sense table may be
    wide_table = torch.zeros(rows, k + 1, dout + 8, device=DEV, dtype=dtype)
    wide_table[:, :k, :dout] = table
    assert torch.equal(bp.sense_mix_gather(qk, wide_table[:, :k, :dout], index, lse=lse), want)
    # not taken: another width, a sequence length that is not a multiple of 32
    other = torch.randn(1, 64, 2, 2, 136, device=DEV).to(dtype)
    assert not bp.sense_mix_gather_supported(other, table[:, :2], 64)
    assert not bp.sense_mix_gather_supported(qk[:, :40], table, 40)
    with pytest.raises(RuntimeError, match='bp_sense_mix_gather'):
        bp.sense_mix_gather(qk[:, :40], table, index[:, :40].contiguous())


def test_wide_senses_backward_runs_without_opt_in():
    """SenseMixFn at d_k = 160: forward on the wide kernels, backward through the alpha-rebuilding route (alpha is small with
    few senses) WITHOUT `allow_eager_fallback` -- gradients against fp32 autograd of the oracle's ops."""
    bp = _bp()
    b, s, k, dk, d = 2, 192, 4, 160, 256
    torch.manual_seed(0)
    qk = (torch.randn(b, s, 2, k, dk) * 0.6).bfloat16()
    c = torch.randn(b, s, k, d).bfloat16()
    dout = torch.randn(b, s, d).bfloat16()

    def ref(dtype):
        q_, c_ = qk.detach().to(dtype).clone().requires_grad_(), c.detach().to(dtype).clone().requires_grad_()
        alpha = R.sense_alpha_from_qk(q_)
        out = torch.einsum('blts,bsld->btd', alpha, c_)
        out.backward(dout.to(dtype))
        return out, q_.grad, c_.grad

    o32, dq32, dc32 = ref(torch.float32)
    o16, dq16, dc16 = ref(torch.bfloat16)
    q_, c_ = qk.detach().to(DEV).clone().requires_grad_(), c.detach().to(DEV).clone().requires_grad_()
    out = bp.sense_mix_autograd(q_, c_)
    out.backward(dout.to(DEV))
    for name, got, w32, w16 in (('out', out, o32, o16), ('dqk', q_.grad, dq32, dq16), ('dC', c_.grad, dc32, dc16)):
        rel_check(got, w32, w16, f'wide backward {name}', factor=3.0, atol=1e-3 * w32.abs().max().item())

"""-m gpu: the elementwise halves of the fused dense layers (csrc/bias_gelu.hip: bp_bias_gelu_fwd / bp_bias_gelu_bwd /
bp_column_sum) and the autograd Functions built on them (flash_attn/ops/fused_dense.py).

Kernel tests: fp32 oracle = torch's tanh-GELU and its autograd on the 16-bit inputs upcast to fp32, criterion
max|kernel - oracle| <= 2 x max|torch same-dtype - oracle| (the reference's rule for its kernels,
tests/test_flash_attn.py:424-428).  Module tests: the reference's own cases and tolerances
(tests/ops/test_fused_dense.py:12-62 `test_fused_linear_bias`, :65-140 `test_fused_dense_gelu_dense`), at sizes that
keep the GPU suite short, plus the AMP recipe (fp32 parameters under bf16 autocast) the training path runs."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def _bp():
    import bp_hip
    return bp_hip


def _rule(got, ref32, same_dtype, name, atol=1e-5):
    err = (got.float() - ref32).abs().max().item()
    base = (same_dtype.float() - ref32).abs().max().item()
    print(f'{name}: kernel {err:.3e}  torch same dtype {base:.3e}')
    assert torch.isfinite(got.float()).all()
    assert err <= 2 * base + atol, (name, err, base)


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
@pytest.mark.parametrize('shape', [(1000, 3072), (37, 12288), (4096, 768), (3, 8), (257, 1032)])
def test_bias_gelu_forward(shape, dtype):
    bp = _bp()
    torch.manual_seed(0)
    rows, cols = shape
    x = (torch.randn(rows, cols, device=DEV) * 3).to(dtype)
    x[0, :8] = torch.tensor([0.0, -0.0, 30.0, -30.0, 1e-3, -1e-3, 8.0, -8.0], device=DEV).to(dtype)
    bias = torch.randn(cols, device=DEV).to(dtype)
    ref = F.gelu(x.float(), approximate='tanh')
    y, pre = bp.bias_gelu_fwd(x)
    assert pre is None
    _rule(y, ref, F.gelu(x, approximate='tanh'), f'gelu {shape} {dtype}')
    # with a bias: the saved pre-activation is the 16-bit rounded sum, y the GELU of exactly that value
    y, pre = bp.bias_gelu_fwd(x, bias, save_pre=True)
    assert torch.equal(pre, (x.float() + bias.float()).to(dtype))
    _rule(y, F.gelu(pre.float(), approximate='tanh'), F.gelu(pre, approximate='tanh'), f'bias+gelu {shape} {dtype}')
    # without a saved pre-activation the GELU is taken of the UNROUNDED fp32 sum
    y2, none = bp.bias_gelu_fwd(x, bias)
    assert none is None
    _rule(y2, F.gelu(x.float() + bias.float(), approximate='tanh'), F.gelu(x + bias, approximate='tanh'),
          f'bias+gelu, nothing saved {shape} {dtype}')
    # in place
    buf = x.clone()
    out, _ = bp.bias_gelu_fwd(buf, out=buf)
    assert out.data_ptr() == buf.data_ptr() and torch.equal(buf, bp.bias_gelu_fwd(x)[0])


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
@pytest.mark.parametrize('bias_dtype', ['same', 'fp32'])
@pytest.mark.parametrize('shape', [(1000, 3072), (37, 12288), (4096, 768), (3, 8), (1, 64), (2049, 1032)])
def test_bias_gelu_backward_and_column_sum(shape, bias_dtype, dtype):
    bp = _bp()
    torch.manual_seed(1)
    rows, cols = shape
    pre = (torch.randn(rows, cols, device=DEV) * 2.5).to(dtype)
    g = (torch.randn(rows, cols, device=DEV) / 8).to(dtype)
    bdt = torch.float32 if bias_dtype == 'fp32' else dtype
    p32 = pre.float().requires_grad_()
    want, = torch.autograd.grad(F.gelu(p32, approximate='tanh'), p32, g.float())
    p16 = pre.clone().requires_grad_()
    same, = torch.autograd.grad(F.gelu(p16, approximate='tanh'), p16, g)
    dpre, dbias = bp.bias_gelu_bwd(g, pre, bdt)
    _rule(dpre, want, same, f'dgelu {shape} {dtype}')
    assert dbias.dtype == bdt and dbias.shape == (cols,)
    # the bias gradient sums the ROUNDED dpre the weight-gradient GEMM consumes, in fp32
    want_b = dpre.float().sum(0)
    tol = 1e-5 * rows ** 0.5 + (0.0 if bdt == torch.float32 else want_b.abs().max().item() * 2.0 ** -8)
    assert (dbias.float() - want_b).abs().max().item() <= tol + 1e-6
    # no bias gradient wanted / in place
    d2, none = bp.bias_gelu_bwd(g, pre)
    assert none is None and torch.equal(d2, dpre)
    gbuf = g.clone()
    d3, b3 = bp.bias_gelu_bwd(gbuf, pre, bdt, inplace=True)
    assert d3.data_ptr() == gbuf.data_ptr() and torch.equal(d3, dpre) and torch.equal(b3, dbias)
    # plain column sums; deterministic
    cs = bp.column_sum(g, bdt)
    want_c = g.float().sum(0)
    tol = 1e-5 * rows ** 0.5 + (0.0 if bdt == torch.float32 else want_c.abs().max().item() * 2.0 ** -8)
    assert (cs.float() - want_c).abs().max().item() <= tol + 1e-6
    for _ in range(3):
        assert torch.equal(bp.column_sum(g, bdt), cs)
        assert torch.equal(bp.bias_gelu_bwd(g, pre, bdt)[1], dbias)


def _copy_linear(dst, src):
    with torch.no_grad():
        dst.weight.copy_(src.weight)
        if src.bias is not None:
            dst.bias.copy_(src.bias)


@pytest.mark.parametrize('dtype', [torch.float16, torch.bfloat16])
@pytest.mark.parametrize('return_residual', [False, True])
@pytest.mark.parametrize('has_bias', [True, False])
@pytest.mark.parametrize('features', [(1024, 4096), (768, 2304), (4096, 1024)])
def test_fused_linear_bias(features, has_bias, return_residual, dtype):
    """The reference's test_fused_linear_bias (tests/ops/test_fused_dense.py:12-62), its tolerances."""
    from flash_attn.ops.fused_dense import FusedDense
    in_features, out_features = features
    rtol, atol = (3e-3, 1e-2) if dtype == torch.bfloat16 else (3e-3, 1e-3)
    torch.random.manual_seed(0)
    x_pt = torch.randn(4, 512, in_features, device=DEV, dtype=dtype, requires_grad=True)
    x = x_pt.detach().clone().requires_grad_()
    model_pt = torch.nn.Linear(in_features, out_features, bias=has_bias, device=DEV, dtype=dtype)
    model = FusedDense(in_features, out_features, bias=has_bias, return_residual=return_residual, device=DEV,
                       dtype=dtype)
    _copy_linear(model, model_pt)
    out_pt = model_pt(x_pt)
    if not return_residual:
        out = model(x)
    else:
        out, x_copy = model(x)
        cut = (lambda t: t[..., :out_features]) if out_features < in_features else \
            (lambda t: F.pad(t, (0, out_features - in_features)))
        out_pt = out_pt + F.gelu(cut(x_pt))          # some function of the residual, as upstream
        out = out + F.gelu(cut(x_copy))
    if not return_residual:
        assert 'FusedDenseFunc' in type(out.grad_fn).__name__         # the custom Function, not torch's linear
    assert torch.allclose(out, out_pt, rtol=rtol, atol=atol)
    g = torch.randn_like(out) / 32
    out_pt.backward(g)
    out.backward(g)
    assert torch.allclose(x.grad, x_pt.grad, rtol=rtol, atol=atol)
    assert torch.allclose(model.weight.grad, model_pt.weight.grad, rtol=rtol, atol=atol * 10)
    if has_bias:
        assert torch.allclose(model.bias.grad, model_pt.bias.grad, rtol=rtol, atol=atol * 5)
        # and closer to the fp32 column sums than torch's own 16-bit reduction is allowed to be
        want = g.float().sum((0, 1))
        assert (model.bias.grad.float() - want).abs().max().item() <= \
            2 * (model_pt.bias.grad.float() - want).abs().max().item() + want.abs().max().item() * 2.0 ** -8


@pytest.mark.parametrize('dtype', [torch.float16, torch.bfloat16])
@pytest.mark.parametrize('checkpoint_lvl', [0, 1, 2])
@pytest.mark.parametrize('return_residual', [False, True])
@pytest.mark.parametrize('biases', [(True, True), (False, True), (True, False)])
@pytest.mark.parametrize('features', [(1024, 4096), (768, 3072)])
def test_fused_dense_gelu_dense(features, biases, return_residual, checkpoint_lvl, dtype):
    """The reference's test_fused_dense_gelu_dense (tests/ops/test_fused_dense.py:65-140), its tolerances."""
    from flash_attn.ops.fused_dense import FusedDenseGeluDense
    in_features, hidden = features
    has_bias1, has_bias2 = biases
    rtol, atol = (3e-3, 3e-2) if dtype == torch.bfloat16 else (3e-3, 1e-3)
    torch.random.manual_seed(0)
    x_pt = torch.randn(4, 512, in_features, device=DEV, dtype=dtype, requires_grad=True)
    x = x_pt.detach().clone().requires_grad_()
    fc1 = torch.nn.Linear(in_features, hidden, bias=has_bias1, device=DEV, dtype=dtype)
    fc2 = torch.nn.Linear(hidden, in_features, bias=has_bias2, device=DEV, dtype=dtype)
    model = FusedDenseGeluDense(in_features, hidden, in_features, bias1=has_bias1, bias2=has_bias2,
                                return_residual=return_residual, checkpoint_lvl=checkpoint_lvl, device=DEV,
                                dtype=dtype)
    _copy_linear(model.fc1, fc1)
    _copy_linear(model.fc2, fc2)
    out_pt = fc2(F.gelu(fc1(x_pt), approximate='tanh'))
    if not return_residual:
        out = model(x)
    else:
        out, x_copy = model(x)
        out_pt = out_pt + F.gelu(x_pt)
        out = out + F.gelu(x_copy)
    assert torch.allclose(out, out_pt, rtol=rtol, atol=atol)
    g = torch.randn_like(out) / 32
    out_pt.backward(g)
    out.backward(g)
    assert torch.allclose(x.grad, x_pt.grad, rtol=rtol, atol=atol)
    assert torch.allclose(model.fc1.weight.grad, fc1.weight.grad, rtol=rtol, atol=atol * 10)
    assert torch.allclose(model.fc2.weight.grad, fc2.weight.grad, rtol=rtol, atol=atol * 10)
    if has_bias1:
        assert torch.allclose(model.fc1.bias.grad, fc1.bias.grad, rtol=rtol, atol=atol * 5)
    if has_bias2:
        assert torch.allclose(model.fc2.bias.grad, fc2.bias.grad, rtol=rtol, atol=atol * 5)
    # inference path (no grad): GELU in the GEMM epilogue, same numbers
    with torch.no_grad():
        y = model(x)
        y = y[0] if return_residual else y
        assert torch.allclose(y, fc2(F.gelu(fc1(x_pt), approximate='tanh')), rtol=rtol, atol=atol)


def test_fused_dense_layers_under_amp():
    """The training recipe: fp32 parameters, bf16 autocast.  Gradients arrive in fp32 (the engine casts what the
    Functions return), bias gradients are summed in fp32 by the kernels, and everything matches torch's own AMP path."""
    from flash_attn.ops.fused_dense import FusedDense, FusedDenseGeluDense
    torch.random.manual_seed(2)
    d, hidden = 768, 3072
    x = torch.randn(8, 256, d, device=DEV)
    lin = FusedDense(d, 3 * d, device=DEV)
    mlp = FusedDenseGeluDense(d, hidden, d, device=DEV)
    lin_pt = torch.nn.Linear(d, 3 * d, device=DEV)
    fc1, fc2 = torch.nn.Linear(d, hidden, device=DEV), torch.nn.Linear(hidden, d, device=DEV)
    _copy_linear(lin, lin_pt)
    _copy_linear(mlp.fc1, fc1)
    _copy_linear(mlp.fc2, fc2)
    g1, g2 = torch.randn(8, 256, 3 * d, device=DEV) / 32, torch.randn(8, 256, d, device=DEV) / 32
    xa, xb = x.clone().requires_grad_(), x.clone().requires_grad_()
    with torch.autocast('cuda', dtype=torch.bfloat16):
        ya, za = lin(xa), mlp(xa)
        yb, zb = lin_pt(xb), fc2(F.gelu(fc1(xb), approximate='tanh'))
    assert ya.dtype == torch.bfloat16 and za.dtype == torch.bfloat16
    assert torch.allclose(ya, yb, rtol=3e-3, atol=1e-2) and torch.allclose(za, zb, rtol=3e-3, atol=3e-2)
    torch.autograd.backward((ya, za), (g1.bfloat16(), g2.bfloat16()))
    torch.autograd.backward((yb, zb), (g1.bfloat16(), g2.bfloat16()))
    assert xa.grad.dtype == torch.float32
    assert torch.allclose(xa.grad, xb.grad, rtol=3e-3, atol=3e-2)
    for mine, theirs in ((lin, lin_pt), (mlp.fc1, fc1), (mlp.fc2, fc2)):
        assert mine.weight.grad.dtype == torch.float32 and mine.bias.grad.dtype == torch.float32
        assert torch.allclose(mine.weight.grad, theirs.weight.grad, rtol=3e-3, atol=0.3)
        assert torch.allclose(mine.bias.grad, theirs.bias.grad, rtol=3e-3, atol=0.15)

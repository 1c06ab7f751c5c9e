"""Race / determinism stress for the ring kernels: the LDS-DMA rings, the persistent job queues and the asynchronous
operand prefetch are all timing-sensitive by construction, so each kernel is run repeatedly on the same inputs, with
other work in flight on a second stream, and every repetition must be bit-identical to the first
(the reference's own determinism check, tests/test_flash_attn.py:788-793, runs its backward 10 times)."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def _bp():
    import bp_hip
    return bp_hip


def _noise(stream, n=8):
    """Unrelated bandwidth-heavy work on another stream, so that DMA latencies vary from run to run."""
    with torch.cuda.stream(stream):
        a = torch.randn(4096, 4096, device=DEV, dtype=torch.bfloat16)
        for _ in range(n):
            a = a @ a.t() * 1e-3
            a = a + 1.0
    return a


@pytest.mark.parametrize('shape', [(3, 1024, 16, 48, 768), (5, 640, 64, 16, 640), (2, 2048, 16, 48, 256), (9, 300, 4, 24, 104)])
def test_sense_mix_and_dc_repeat_bit_identically(shape):
    bp = _bp()
    b, s, k, dk, d = shape
    torch.manual_seed(5)
    qk = (torch.randn(b, s, 2, k, dk, device=DEV) * 0.9).bfloat16()
    c = torch.randn(b, s, k, d, device=DEV).bfloat16()
    dout = torch.randn(b, s, d, device=DEV).bfloat16()
    lse = bp.sense_lse(qk)
    first = bp.sense_mix(qk, c, lse=lse).clone()
    first_dc = bp.sense_mix_dc(qk, dout, lse, dk ** -0.5, c).clone() if d % 8 == 0 and dk % 8 == 0 else None
    side = torch.cuda.Stream()
    for i in range(12):
        keep = _noise(side, 2 + i % 3)
        out = bp.sense_mix(qk, c, lse=lse)
        assert torch.equal(out, first), f'sense_mix repetition {i} differs'
        if first_dc is not None:
            dc = bp.sense_mix_dc(qk, dout, lse, dk ** -0.5, c)
            assert torch.equal(dc, first_dc), f'sense_mix_dc repetition {i} differs'
    torch.cuda.synchronize()
    del keep


@pytest.mark.parametrize('cfg', [(4, 1024, 12, 64, True), (2, 1000, 8, 80, True), (3, 777, 6, 64, False), (2, 4096, 4, 64, True)])
def test_flash_fwd_repeats_bit_identically(cfg):
    bp = _bp()
    b, s, h, d, causal = cfg
    torch.manual_seed(6)
    qkv = torch.randn(b * s, 3, h, d, device=DEV).bfloat16()
    out = torch.empty_like(qkv[:, 0])
    cu = torch.arange(0, (b + 1) * s, s, dtype=torch.int32, device=DEV)
    lse0 = bp.flash_fwd(qkv[:, 0], qkv[:, 1], qkv[:, 2], out, cu, cu, s, s, d ** -0.5, causal)[..., :s].clone()
    first = out.clone()
    side = torch.cuda.Stream()
    for i in range(12):
        keep = _noise(side, 2 + i % 3)
        out.fill_(float('nan'))
        lse = bp.flash_fwd(qkv[:, 0], qkv[:, 1], qkv[:, 2], out, cu, cu, s, s, d ** -0.5, causal)
        # (LSE entries past the sequence length are padding the kernel never writes)
        assert torch.equal(out, first) and torch.equal(lse[..., :s], lse0), f'flash_fwd repetition {i} differs'
    torch.cuda.synchronize()
    del keep


def test_persistent_mix_launches_on_several_streams():
    """Persistent launches own one queue record each (a ring of 64 in the library): launches that overlap on different
    streams must neither share tickets nor wait on each other."""
    bp = _bp()
    torch.manual_seed(7)
    shapes = [(3, 512, 16, 48, 768), (2, 1024, 16, 48, 256), (4, 300, 4, 24, 104)]
    data = []
    for b, s, k, dk, d in shapes:
        qk = (torch.randn(b, s, 2, k, dk, device=DEV) * 0.9).bfloat16()
        c = torch.randn(b, s, k, d, device=DEV).bfloat16()
        lse = bp.sense_lse(qk)
        data.append((qk, c, lse, bp.sense_mix(qk, c, lse=lse).clone()))
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream() for _ in shapes]
    for rnd in range(6):
        outs = []
        for st, (qk, c, lse, _) in zip(streams, data):
            with torch.cuda.stream(st):
                for _ in range(3):   # several launches in flight per stream
                    out = bp.sense_mix(qk, c, lse=lse)
                outs.append(out)
        torch.cuda.synchronize()
        for out, (_, _, _, want) in zip(outs, data):
            assert torch.equal(out, want), f'round {rnd}: a concurrent launch produced a different result'

"""CPU: the C-ABI library loads, exports every symbol include/bp_hip.h declares, and rejects bad
arguments with the documented codes BEFORE touching a device (no compute without a GPU)."""
import ctypes
import os
import re

import pytest

from conftest import ROOT

import bp_hip


def declared_symbols():
    text = open(os.path.join(ROOT, 'include', 'bp_hip.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(bp_[a-z_0-9]+)\s*\(', text)))


def test_library_exports_every_declared_symbol():
    handle = bp_hip.lib()
    names = declared_symbols()
    assert {'bp_flash_fwd', 'bp_attn_probs', 'bp_sense_lse', 'bp_sense_alpha', 'bp_sense_mix',
            'bp_strerror', 'bp_abi_version', 'bp_build_flags'} <= set(names)
    for name in names:
        assert hasattr(handle, name), name
    assert set(bp_hip.SIGNATURES) == set(names)
    assert handle.bp_abi_version() == bp_hip.ABI_VERSION
    assert handle.bp_build_flags() == 0       # a product build: no what-if timing switches, no dev switches


def test_header_error_codes_have_messages():
    handle = bp_hip.lib()
    text = open(os.path.join(ROOT, 'include', 'bp_hip.h')).read()
    codes = {m.group(1): int(m.group(2)) for m in re.finditer(r'#define (BP_ERR_[A-Z_]+) (-\d+)', text)}
    assert len(codes) >= 6
    for name, code in codes.items():
        msg = handle.bp_strerror(code).decode()
        assert msg and msg != 'unknown error', name
    assert handle.bp_strerror(0).decode() == 'ok'


def test_argument_validation_returns_before_any_launch():
    h = bp_hip.lib()
    p = ctypes.c_void_p(0x1000)   # never dereferenced: validation fails first
    null = None
    # bad dtype
    assert h.bp_flash_fwd(p, p, p, p, p, null, null, 1, 1, 64, 16, 16, 64, 64, 64, 64, 64, 64, 64, 64,
                          16, 0.125, 1, 7, null) == -1
    # head_dim > 128 (fmha_api.cpp:245)
    assert h.bp_flash_fwd(p, p, p, p, p, null, null, 1, 1, 136, 16, 16, 136, 136, 136, 136, 136, 136,
                          136, 136, 16, 0.125, 1, 1, null) == -2
    # batch <= 0, null q
    assert h.bp_flash_fwd(p, p, p, p, p, null, null, 0, 1, 64, 16, 16, 64, 64, 64, 64, 64, 64, 64, 64,
                          16, 0.125, 1, 1, null) == -3
    assert h.bp_flash_fwd(null, p, p, p, p, null, null, 1, 1, 64, 16, 16, 64, 64, 64, 64, 64, 64, 64, 64,
                          16, 0.125, 1, 1, null) == -3
    # v without out
    assert h.bp_flash_fwd(p, p, p, null, p, null, null, 1, 1, 64, 16, 16, 64, 64, 64, 64, 64, 64, 64, 64,
                          16, 0.125, 1, 1, null) == -3
    # scale not finite / not positive
    for bad in (float('nan'), float('inf'), 0.0, -1.0):
        assert h.bp_flash_fwd(p, p, p, p, p, null, null, 1, 1, 64, 16, 16, 64, 64, 64, 64, 64, 64, 64,
                              64, 16, bad, 1, 1, null) == -4
    # dropout: p outside [0, 1), p > 0 without a generator state, dropout on an LSE-only call
    tail = (1, 1, 64, 16, 16, 64, 64, 64, 64, 64, 64, 64, 64, 16, 0.125, 1, 1)
    assert h.bp_flash_fwd_dropout(p, p, p, p, p, null, null, *tail, 1.0, p, null) == -7
    assert h.bp_flash_fwd_dropout(p, p, p, p, p, null, null, *tail, -0.1, p, null) == -7
    assert h.bp_flash_fwd_dropout(p, p, p, p, p, null, null, *tail, 0.1, null, null) == -7
    assert h.bp_flash_fwd_dropout(p, p, null, null, p, null, null, *tail, 0.1, p, null) == -7
    # sense mix: d_out < 1, d_k out of range
    assert h.bp_sense_mix(p, p, p, p, 0, 1, 16, 4, 16, 0, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0.25, 1, null, null) == -6
    # (sense widths up to 640 are taken since ABI 8 -- csrc/sense_wide.hip; the gathering form takes 160 / 640 of them, ABI 9)
    assert h.bp_sense_mix(p, p, p, p, 0, 1, 16, 4, 648, 64, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0.25, 1, null, null) == -2
    assert h.bp_sense_lse(p, p, 1, 16, 4, 641, 1, 1, 1, 1, 0.25, 1, null) == -2
    # the gathering form beyond 128: only 160 / 640 at a sequence length that is a multiple of 32
    gather_tail = (8, 8, 8, 8, 8, 8, 8, 8, 8, 0.25, 1, null, null)
    assert h.bp_sense_mix_gather(p, p, p, p, p, 1, 1, 64, 2, 136, 64, 100, *gather_tail) == -2
    assert h.bp_sense_mix_gather(p, p, p, p, p, 1, 1, 40, 2, 160, 64, 100, *gather_tail) == -2
    assert h.bp_sense_mix_gather(p, p, p, null, p, 1, 1, 64, 2, 160, 64, 100, *gather_tail) == -3   # taken: stops at the NULL out
    assert h.bp_sense_alpha(p, p, p, 0, 1, 16, 4, 0, 1, 1, 1, 1, 0.25, 1, null) == -2
    # a misaligned queue_ws
    assert h.bp_sense_mix(p, p, p, p, 0, 1, 16, 4, 16, 64, 8, 8, 8, 8, 8, 8, 8, 8, 8, 0.25, 1, ctypes.c_void_p(0x1004),
                          null) == -3
    assert h.bp_sense_alpha(p, null, p, 0, 1, 16, 4, 16, 1, 1, 1, 1, 0.25, 1, null) == -3
    assert h.bp_sense_lse(p, p, 1, 0, 4, 16, 1, 1, 1, 1, 0.25, 1, null) == -3
    assert h.bp_attn_probs(p, p, p, p, 1, 1, 64, 16, 0, 1, 1, 1, 1, 1, 1, 16, 1, 1, 1, 0.125, 1, 1,
                           null) == -3
    # backward: head dim the kernel does not cover, bad dtype, null dq, odd lse stride
    st = [64] * 16
    assert h.bp_flash_bwd(p, p, p, p, p, p, p, 64, p, p, p, null, null, 1, 1, 136, 16, 16, *st, 16, 0.125, 1, 1,
                          null) == -2
    assert h.bp_flash_bwd(p, p, p, p, p, p, p, 64, p, p, p, null, null, 1, 1, 64, 16, 16, *st, 16, 0.125, 1, 5,
                          null) == -1
    assert h.bp_flash_bwd(p, p, p, p, p, p, p, 64, null, p, p, null, null, 1, 1, 64, 16, 16, *st, 16, 0.125, 1, 1,
                          null) == -3
    assert h.bp_flash_bwd(p, p, p, p, p, p, p, 64, p, p, p, null, null, 1, 1, 64, 16, 16, *st, 17, 0.125, 1, 1,
                          null) == -3
    assert h.bp_flash_bwd(p, p, p, p, p, p, p, 64, p, p, p, null, null, 1, 1, 64, 16, 16, *st, 16, -1.0, 1, 1,
                          null) == -4
    assert h.bp_flash_bwd_dropout(p, p, p, p, p, p, p, 64, p, p, p, null, null, 1, 1, 64, 16, 16, *st, 16, 0.125, 1, 1,
                                  0.5, null, null) == -7
    # the statistics workspace is (batch, nheads, 2, lse_stride) floats since ABI 3; an ABI-2-sized buffer (half of
    # it) must be refused, not overrun (ABI 4: the size is an argument, bp_flash_bwd_ws_floats is the query)
    assert h.bp_flash_bwd_ws_floats(3, 5, 1024) == 3 * 5 * 2 * 1024
    assert h.bp_flash_bwd_ws_floats(0, 5, 1024) == 0
    assert h.bp_flash_bwd(p, p, p, p, p, p, p, 16, p, p, p, null, null, 1, 1, 64, 16, 16, *st, 16, 0.125, 1, 1,
                          null) == -9
    assert h.bp_flash_bwd(p, p, p, p, p, p, p, 31, p, p, p, null, null, 1, 1, 64, 16, 16, *st, 16, 0.125, 1, 1,
                          null) == -9
    assert h.bp_attn_probs_dropout(p, p, p, p, 1, 1, 64, 16, 16, 1, 1, 1, 1, 1, 1, 16, 1, 1, 1, 0.125, 1, 1,
                                   2.0, p, null) == -7

    # fused LayerNorm with rowscale / colscale (ABI 5): a colscale without x0 / dcolscale, a workspace of the unscaled size
    ln = (p, p, p)
    assert h.bp_ln_bwd_ws_floats(768, 0) == 2 * 1024 * 768 and h.bp_ln_bwd_ws_floats(768, 1) == 3 * 1024 * 768
    assert h.bp_dropout_add_layer_norm_scaled_bwd(p, null, p, null, p, null, p, p, null, p, p, p, p, 3 * 1024 * 768,
                                                  16, 768, 1e-5, 1, 0, 1, 1, 0.0, null, null) == -3
    assert h.bp_dropout_add_layer_norm_scaled_bwd(p, null, p, p, p, null, p, p, null, p, p, p, p, 2 * 1024 * 768,
                                                  16, 768, 1e-5, 1, 0, 1, 1, 0.0, null, null) == -9
    assert h.bp_dropout_add_layer_norm_scaled(p, null, p, p, ctypes.c_void_p(0x1001), null, p, null, null, 16, 768, 1e-5,
                                              1, 0, 0, 0, 1, 0.0, null, null) == -3          # misaligned rowscale
    # fused sense-mix backward: d_k not a multiple of 8, d_out not a multiple of 8, null dcontent, odd stride, bad scale
    mix_st = (64,) * 9
    assert h.bp_sense_mix_dc(p, p, p, p, 1, 16, 4, 10, 64, *mix_st, 0.25, 1, null, null) == -2
    assert h.bp_sense_mix_dc(p, p, p, p, 1, 16, 4, 16, 20, *mix_st, 0.25, 1, null, null) == -6
    assert h.bp_sense_mix_dc(p, p, p, null, 1, 16, 4, 16, 64, *mix_st, 0.25, 1, null, null) == -3
    assert h.bp_sense_mix_dc(p, p, p, p, 1, 16, 4, 16, 64, 64, 63, 64, 64, 64, 64, 64, 64, 64, 0.25, 1, null, null) == -3
    assert h.bp_sense_mix_dc(p, p, p, p, 1, 16, 4, 16, 64, *mix_st, 0.0, 1, null, null) == -4
    assert h.bp_sense_mix_dc(p, p, p, p, 1, 16, 4, 16, 64, *mix_st, 0.25, 9, null, null) == -1
    # dq/dk slab kernel: slab start not a multiple of 128 / past the end, null workspace, fp32 stride not 16-byte
    dq_st = (64,) * 11
    assert h.bp_sense_dq_dk(p, p, p, p, p, p, 1, 256, 4, 16, 64, *dq_st, 0.25, 1, null) == -3
    assert h.bp_sense_dq_dk(p, p, p, p, p, p, 1, 256, 4, 16, 256, *dq_st, 0.25, 1, null) == -3
    assert h.bp_sense_dq_dk(p, p, p, null, p, p, 1, 256, 4, 16, 128, *dq_st, 0.25, 1, null) == -3
    assert h.bp_sense_dq_dk(p, p, p, p, p, p, 1, 256, 4, 16, 128, *dq_st[:8], 64, 62, 64, 0.25, 1, null) == -3
    assert h.bp_sense_dq_dk(p, p, p, p, p, p, 1, 256, 4, 12, 128, *dq_st, 0.25, 1, null) == -2
    # fused (dropout +) add + LayerNorm: dropout without a generator state, fp32 x0 with a 16-bit residual stream,
    # columns not a multiple of 4, misaligned dmask
    assert h.bp_dropout_add_layer_norm(p, p, p, p, p, p, null, 8, 64, 1e-5, 1, 0, 1, 1, 1, 0.1, null, null) == -7
    assert h.bp_dropout_add_layer_norm(p, p, p, p, p, p, null, 8, 64, 1e-5, 1, 1, 0, 0, 1, 0.0, null, null) == -1
    assert h.bp_dropout_add_layer_norm(p, p, p, p, p, p, null, 8, 66, 1e-5, 1, 0, 1, 1, 1, 0.0, null, null) == -3
    assert h.bp_dropout_add_layer_norm(p, p, p, p, p, p, ctypes.c_void_p(0x1002), 8, 64, 1e-5, 1, 0, 1, 1, 1, 0.1, p,
                                       null) == -3
    assert h.bp_dropout_add_layer_norm_bwd(p, p, p, p, p, p, p, p, p, 8, 64, 1e-5, 1, 0, 1, 1, 1.5, p, null) == -7
    assert h.bp_dropout_add_layer_norm_bwd(p, p, p, p, p, p, p, p, p, 8, 64, 1e-5, 1, 1, 0, 1, 0.0, null, null) == -1
    assert h.bp_dropout_add_layer_norm_bwd(p, p, p, p, p, p, p, p, null, 8, 64, 1e-5, 1, 0, 1, 1, 0.0, null, null) == -3
    # bias + GELU / column sums: columns not a multiple of 8, pre_out without a bias, a bias gradient without workspace,
    # fp32 "16-bit" dtype; the workspace size is a pure function of the shape
    assert h.bp_bias_gelu_fwd(p, p, null, p, 8, 60, 1, null) == -3
    assert h.bp_bias_gelu_fwd(p, null, p, p, 8, 64, 1, null) == -3
    assert h.bp_bias_gelu_fwd(p, p, null, p, 8, 64, 2, null) == -1
    assert h.bp_bias_gelu_bwd(p, p, p, p, null, 8, 64, 1, 1, null) == -3
    assert h.bp_bias_gelu_bwd(p, null, p, null, null, 8, 64, 1, 1, null) == -3
    assert h.bp_column_sum(p, null, p, 8, 64, 1, 1, null) == -3
    assert h.bp_column_sum(p, p, p, 0, 64, 1, 1, null) == -3
    assert h.bp_bias_grad_ws_floats(32768, 3072) == 3072 * 171 and h.bp_bias_grad_ws_floats(3, 768) == 768 * 1
    assert h.bp_bias_grad_ws_floats(0, 768) == 0


def test_philox_restatement_known_answers():
    """tests/philox_ref.py (the host restatement of csrc/bp_philox.h) against the published Philox2x32-10
    known-answer vectors of Random123 (kat_vectors: three `philox2x32 10` lines)."""
    import philox_ref as P
    for (c0, c1, key), want in (((0, 0, 0), (0xff1dae59, 0x6cd10df2)),
                                ((0xffffffff, 0xffffffff, 0xffffffff), (0x2c3f628b, 0xab4fd7ad)),
                                ((0x243f6a88, 0x85a308d3, 0x13198a2e), (0xdd7ce038, 0xf62a4c12))):
        a, b = P.philox2x32(c0, c1, key)
        assert (int(a), int(b)) == want
    keep = P.attention_keep_mask(123, 456, 2, 3, 96, 160, 0.17)
    assert keep.shape == (2, 3, 96, 160) and abs(keep.mean() - 0.83) < 0.01
    assert not (keep[0, 0] == keep[0, 1]).all() and not (keep[0, 0] == keep[1, 0]).all()   # streams differ per (b, h)
    assert P.threshold(0.1) == 58982 and P.threshold(0.17) == 54395


def test_python_binding_refuses_cpu_tensors_loudly():
    import torch
    q = torch.randn(16, 2, 64).bfloat16()
    cu = torch.tensor([0, 16], dtype=torch.int32)
    with pytest.raises(RuntimeError, match='GPU'):
        bp_hip.flash_fwd(q, q, q, torch.empty_like(q), cu, cu, 16, 16, 0.125, True)
    with pytest.raises(RuntimeError, match='GPU'):
        bp_hip.sense_mix(torch.randn(1, 8, 2, 4, 16).bfloat16(), torch.randn(1, 8, 4, 32).bfloat16())

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
            'bp_strerror', 'bp_abi_version'} <= set(names)
    for name in names:
        assert hasattr(handle, name), name
    assert set(bp_hip.SIGNATURES) == set(names)
    assert handle.bp_abi_version() == bp_hip.ABI_VERSION


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
    # sense mix: d_out < 1, d_k out of range
    assert h.bp_sense_mix(p, p, p, p, 0, 1, 16, 4, 16, 0, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0.25, 1, null) == -6
    assert h.bp_sense_mix(p, p, p, p, 0, 1, 16, 4, 200, 64, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0.25, 1, null) == -2
    assert h.bp_sense_alpha(p, null, p, 0, 1, 16, 4, 16, 1, 1, 1, 1, 0.25, 1, null) == -3
    assert h.bp_sense_lse(p, p, 1, 0, 4, 16, 1, 1, 1, 1, 0.25, 1, null) == -3
    assert h.bp_attn_probs(p, p, p, p, 1, 1, 64, 16, 0, 1, 1, 1, 1, 1, 1, 16, 1, 1, 1, 0.125, 1, 1,
                           null) == -3
    # backward: head dim the kernel does not cover, bad dtype, null dq, odd lse stride
    st = [64] * 16
    assert h.bp_flash_bwd(p, p, p, p, p, p, p, p, p, p, null, null, 1, 1, 136, 16, 16, *st, 16, 0.125, 1, 1,
                          null) == -2
    assert h.bp_flash_bwd(p, p, p, p, p, p, p, p, p, p, null, null, 1, 1, 64, 16, 16, *st, 16, 0.125, 1, 5,
                          null) == -1
    assert h.bp_flash_bwd(p, p, p, p, p, p, p, null, p, p, null, null, 1, 1, 64, 16, 16, *st, 16, 0.125, 1, 1,
                          null) == -3
    assert h.bp_flash_bwd(p, p, p, p, p, p, p, p, p, p, null, null, 1, 1, 64, 16, 16, *st, 17, 0.125, 1, 1,
                          null) == -3
    assert h.bp_flash_bwd(p, p, p, p, p, p, p, p, p, p, null, null, 1, 1, 64, 16, 16, *st, 16, -1.0, 1, 1,
                          null) == -4


def test_python_binding_refuses_cpu_tensors_loudly():
    import torch
    q = torch.randn(16, 2, 64).bfloat16()
    cu = torch.tensor([0, 16], dtype=torch.int32)
    with pytest.raises(RuntimeError, match='GPU'):
        bp_hip.flash_fwd(q, q, q, torch.empty_like(q), cu, cu, 16, 16, 0.125, True)
    with pytest.raises(RuntimeError, match='GPU'):
        bp_hip.sense_mix(torch.randn(1, 8, 2, 4, 16).bfloat16(), torch.randn(1, 8, 4, 32).bfloat16())

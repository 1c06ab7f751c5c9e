"""Error of the flash forward against a float64 softmax(QK^T)V, for the library selected by BP_HIP_LIB:
    BP_HIP_LIB=.../libbackpack_hip_X.so python scripts/debug/r04_fwd_accuracy.py
prints one JSON line per case: max / rms error of O and max error of the LSE, next to the same numbers for the
eager 16-bit chain (the 2x rule of the parity tests is measured against that one)."""
import json
import math
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'backpacks-flash-attn_amd'))
import bp_hip  # noqa: E402

DEV = 'cuda'


def ref64(q, k, v, scale, causal):
    s = torch.einsum('bqhd,bkhd->bhqk', q.double(), k.double()) * scale
    if causal:
        n = s.shape[-1]
        s = s.masked_fill(torch.ones(n, n, dtype=torch.bool, device=s.device).triu(1), float('-inf'))
    lse = torch.logsumexp(s, -1)
    return torch.einsum('bhqk,bkhd->bqhd', torch.softmax(s, -1), v.double()), lse


def eager16(q, k, v, scale, causal):
    s = torch.einsum('bqhd,bkhd->bhqk', q, k) * scale
    if causal:
        n = s.shape[-1]
        s = s.masked_fill(torch.ones(n, n, dtype=torch.bool, device=s.device).triu(1), float('-inf'))
    return torch.einsum('bhqk,bkhd->bqhd', torch.softmax(s, -1, dtype=torch.float32).to(q.dtype), v)


def main():
    cases = [(torch.bfloat16, 2, 1024, 12, 64, 1.0), (torch.bfloat16, 2, 1024, 12, 64, 4.0), (torch.float16, 2, 1024, 12, 64, 1.0),
             (torch.bfloat16, 2, 512, 16, 80, 1.0), (torch.bfloat16, 1, 2048, 8, 128, 2.0), (torch.bfloat16, 2, 1024, 16, 48, 1.0)]
    for dt, b, s, h, d, amp in cases:
        torch.manual_seed(0)
        qkv = (torch.randn(b, s, 3, h, d, device=DEV) * amp).to(dt)
        q, k, v = qkv.unbind(2)
        scale = d ** -0.5
        want, lse_want = ref64(q, k, v, scale, True)
        out = torch.empty(b * s, h, d, device=DEV, dtype=dt)
        cu = torch.arange(0, (b + 1) * s, s, dtype=torch.int32, device=DEV)
        f = qkv.view(b * s, 3, h, d)
        lse = bp_hip.flash_fwd(f[:, 0], f[:, 1], f[:, 2], out, cu, cu, s, s, scale, True)[..., :s]
        err = (out.view(b, s, h, d).double() - want)
        e16 = (eager16(q, k, v, scale, True).double() - want)
        print(json.dumps({'lib': os.path.basename(os.environ.get('BP_HIP_LIB', 'default')), 'dtype': str(dt)[6:], 's': s, 'd': d,
                          'amp': amp, 'o_max': err.abs().max().item(), 'o_rms': err.pow(2).mean().sqrt().item(),
                          'lse_max': (lse.double() - lse_want).abs().max().item(),
                          'eager_max': e16.abs().max().item(), 'eager_rms': e16.pow(2).mean().sqrt().item()}), flush=True)


if __name__ == '__main__':
    main()

"""Round-6 golden vectors (same rules as make_golden.py: runs only in the build container, imports the REAL reference from
/root/reference, writes data-only .npz files, asserts this repo's CPU restatement reproduces them).

    python tests/golden/make_golden_r6.py

  G9  g9_wide_sense.npz  the reference's ContextSelfAttn (training/src/models/backpack.py:99-122) and its sense combination
                         `torch.sum(contextualization @ content, dim=1)` (:313) at the sense widths of the reference's few-sense
                         ablation configs -- d_k = 160 (training/configs/experiment/owt/backpack-mini-flash-vecs-4.yaml) and
                         d_k = 640 (...-vecs-1.yaml) -- which run on csrc/sense_wide_dma.hip: weights, input, alpha, mix.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

from make_golden import bf16_exact, bits16, import_reference, np32  # noqa: E402


def main():
    bp, _ = import_reference()
    from oracle import ref_cpu as R
    log = []

    def check(name, got, want, tol):
        err = (got.float() - want.float()).abs().max().item()
        log.append(f'{name}: oracle vs reference max|diff| = {err:.3e} (tol {tol:.1e})')
        print(log[-1])
        assert err <= tol, name

    torch.set_num_threads(8)
    g9 = {}
    # (tag, d, k, S): d_k = d / k; S a multiple of 32 (the ring kernels' condition)
    for tag, d, k, s in (('dk160', 320, 2, 64), ('dk640', 640, 1, 96)):
        torch.manual_seed({'dk160': 160, 'dk640': 640}[tag])
        mod = bp.ContextSelfAttn(k, d)
        with torch.no_grad():   # default Linear init gives tiny logits: make the scores non-trivial
            mod.Wqkv.weight.copy_(bf16_exact(mod.Wqkv.weight * 4.0))
            mod.Wqkv.bias.copy_(bf16_exact(torch.randn(2 * d) * 0.5))
        h = bf16_exact(torch.randn(2, s, d))
        dout = 72
        content = bf16_exact(torch.randn(2, s, k * dout)).reshape(2, s, k, dout).transpose(1, 2)
        with torch.no_grad():
            alpha = mod(h)
            mixed = torch.sum(alpha @ content, dim=1)
        o_alpha = R.context_self_attn(h, mod.Wqkv.weight, mod.Wqkv.bias, k)
        check(f'G9 alpha {tag}', o_alpha, alpha, 1e-6)
        check(f'G9 mix {tag}', R.sense_mix(o_alpha, content), mixed, 1e-5)
        assert torch.count_nonzero(torch.triu(alpha, 1)) == 0
        g9.update({f'{tag}_w': bits16(mod.Wqkv.weight), f'{tag}_b': bits16(mod.Wqkv.bias), f'{tag}_h': bits16(h),
                   f'{tag}_content': bits16(content.transpose(1, 2).contiguous()),
                   f'{tag}_alpha': np32(alpha), f'{tag}_mixed': np32(mixed), f'{tag}_k': np.int64(k)})
    np.savez_compressed(os.path.join(HERE, 'g9_wide_sense.npz'), **g9)
    with open(os.path.join(HERE, 'PINNING_r6.txt'), 'w') as f:
        f.write('make_golden_r6.py, reference imported from /root/reference\n' + '\n'.join(log) + '\n')


if __name__ == '__main__':
    main()

"""Bitwise repeatability of the wide-sense kernels (sense_wide_dma.hip): the same launch N times, contiguous and strided
operands; any difference is a race in the ring.   python scripts/debug/wide_determinism.py [--reps 30]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, 'backpacks-flash-attn_amd')):
    sys.path.insert(0, p)
import torch  # noqa: E402

import bp_hip  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--reps', type=int, default=30)
a = ap.parse_args()
bad = 0
for (b, s, k, dk, d) in [(8, 1024, 4, 160, 640), (8, 1024, 1, 640, 640), (64, 1024, 4, 160, 640), (3, 96, 4, 160, 640),
                         (2, 1024, 4, 160, 200)]:
    torch.manual_seed(b + k)
    qk = (torch.randn(b, s, 2, k, dk, device='cuda') * (2.0 / dk ** 0.25)).bfloat16()
    c = torch.randn(b, s, k, d, device='cuda').bfloat16()
    big = torch.zeros(b, s, 2, k + 1, dk + 8, dtype=torch.bfloat16, device='cuda')
    big[:, :, :, :k, :dk] = qk
    view = big[:, :, :, :k, :dk]
    lse0 = bp_hip.sense_lse(qk).clone()
    out0 = bp_hip.sense_mix(qk, c, lse=lse0).clone()
    n_lse = n_mix = n_view = 0
    for r in range(a.reps):
        lse = bp_hip.sense_lse(qk if r % 2 == 0 else view)
        dl = lse[:, :, :s] != lse0[:, :, :s]
        if dl.any():
            n_lse += 1
            if n_lse <= 3:
                idx = dl.nonzero()
                print('   lse differs: rep', r, 'count', int(dl.sum()), 'first', idx[:5].tolist(), 'last', idx[-1].tolist(),
                      'max|d|', float((lse[:, :, :s] - lse0[:, :, :s]).abs().nan_to_num(1e9).max()),
                      'nan', int(torch.isnan(lse[:, :, :s]).sum()), int(torch.isnan(lse0[:, :, :s]).sum()))
        out = bp_hip.sense_mix(qk if r % 2 == 0 else view, c, lse=lse0)
        bad_rows = (out != out0).any(-1)
        if bad_rows.any():
            n_mix += 1
            if n_mix <= 3:
                idx = bad_rows.nonzero()
                print('   mix differs: rep', r, 'rows', idx[:6].tolist(), 'count', int(bad_rows.sum()),
                      'cols', (out != out0)[tuple(idx[0])].nonzero().flatten()[:8].tolist())
    print(f'shape {(b, s, k, dk, d)}: lse mismatches {n_lse}/{a.reps}, mix mismatches {n_mix}/{a.reps}', flush=True)
    bad += n_lse + n_mix
print('DETERMINISTIC' if bad == 0 else 'RACE')

"""Per-workgroup timeline of the flash forward (development build with -DBP_FLASH_PROFILE, see README.md here):
wave 0 of every workgroup stamps s_memtime at entry, after the Q fragments arrived, before the first DMA issue,
after every ring-step barrier and in front of the epilogue, for both query-tile passes.
    BP_HIP_LIB=.../libbackpack_hip_prof.so python scripts/probes/flash_timeline/timeline.py [--batch 64] [--seq 1024] [--noncausal]"""
import argparse
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
for p in (ROOT, os.path.join(ROOT, 'backpacks-flash-attn_amd')):
    sys.path.insert(0, p)
import torch  # noqa: E402

import bp_hip  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=64)
    ap.add_argument('--seq', type=int, default=1024)
    ap.add_argument('--noncausal', action='store_true')
    a = ap.parse_args()
    B, S, H, D = a.batch, a.seq, 12, 64
    torch.manual_seed(0)
    qkv = torch.randn(B * S, 3, H, D, device='cuda').bfloat16()
    out = torch.empty_like(qkv[:, 0])
    cu = torch.arange(0, (B + 1) * S, S, dtype=torch.int32, device='cuda')
    run = lambda: bp_hip.flash_fwd(qkv[:, 0], qkv[:, 1], qkv[:, 2], out, cu, cu, S, S, D ** -0.5, not a.noncausal)
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    lib = bp_hip.lib()
    lib.bp_dev_flash_prof_clear()
    torch.cuda.synchronize()
    run()
    torch.cuda.synchronize()
    n = 8192 * 48
    buf = (ctypes.c_ulonglong * n)()
    assert lib.bp_dev_flash_prof(buf, n) == 0
    t = np.frombuffer(buf, dtype=np.uint64).reshape(8192, 2, 24).astype(np.int64)
    used = t[:, 0, 0] > 0
    t = t[used]
    print(f'{len(t)} workgroups stamped')
    t0 = t[:, 0, 0].min()
    span = max(t[:, p][t[:, p] > 0].max() for p in (0, 1)) - t0
    print(f'kernel span (first entry -> last epilogue stamp): {span} ticks')
    for p in (0, 1):
        x = t[:, p]
        ok = x[:, 0] > 0
        if not ok.any():
            continue
        x = x[ok]
        nst = (x > 0).sum(1)
        print(f'pass {p}: {ok.sum()} workgroups, stamps per workgroup min/med/max {nst.min()}/{int(np.median(nst))}/{nst.max()}')
        for name, i, j in (('entry -> Q arrived', 0, 1), ('Q arrived -> first DMA issue (descriptors)', 1, 2),
                           ('first DMA issue -> tile 0 landed + barrier', 2, 3)):
            d = x[:, j] - x[:, i]
            print(f'   {name:46s} median {int(np.median(d)):7d}  p10 {int(np.percentile(d, 10)):7d}  p90 {int(np.percentile(d, 90)):7d}')
        # ring steps: stamps 3 .. nst-2 are barriers, the last stamp is the epilogue
        steps, last, totals = [], [], []
        for row, k in zip(x, nst):
            b = row[3:k - 1]
            if len(b) >= 2:
                steps.extend(np.diff(b).tolist())
            last.append(row[k - 1] - row[k - 2])
            totals.append(row[k - 1] - row[0])
        steps = np.array(steps)
        print(f'   ring step (barrier to barrier)                 median {int(np.median(steps)):7d}  p10 {int(np.percentile(steps, 10)):7d}  p90 {int(np.percentile(steps, 90)):7d}  ({len(steps)} steps)')
        print(f'   last barrier -> epilogue (last tile)           median {int(np.median(last)):7d}')
        print(f'   whole pass                                     median {int(np.median(totals)):7d}  p10 {int(np.percentile(totals, 10)):7d}  p90 {int(np.percentile(totals, 90)):7d}')
    both = (t[:, 1, 0] > 0)
    if both.any():
        x = t[both]
        k0 = (x[:, 0] > 0).sum(1)
        gap = x[:, 1, 0] - x[np.arange(len(x)), 0, k0 - 1]
        print(f'pass 0 epilogue stamp -> pass 1 entry: median {int(np.median(gap))}')
    # when do workgroups start? (rounds of the dispatcher)
    starts = np.sort(t[:, 0, 0] - t0)
    qs = [0, 10, 25, 50, 75, 90, 100]
    print('workgroup start times (ticks after the first): ' + ', '.join(f'p{q} {int(np.percentile(starts, q))}' for q in qs))
    ends = np.sort(np.array([row[p][(row[p] > 0)].max() for row in t for p in (0, 1) if (row[p] > 0).any()]) - t0)
    print('pass end times: ' + ', '.join(f'p{q} {int(np.percentile(ends, q))}' for q in qs))


if __name__ == '__main__':
    main()

"""Per-workgroup timeline of the two attention-backward kernels (development build with -DBP_BWD_PROFILE):
wave 0 of every workgroup stamps s_memtime per pass at  0 entry | 1 operand fragments arrived | 2 descriptors done |
3 first tile landed (first ring step begun) | 4 leading edge tiles done (dkdv) | 5 clean tiles done | 6 all tiles done |
7 stores issued.
    BP_HIP_LIB=.../libbackpack_hip_bwdprof.so python scripts/probes/flash_bwd_timeline/timeline.py [--batch 64] [--noncausal]"""
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
    dout = torch.randn(B * S, H, D, device='cuda').bfloat16()
    out = torch.empty_like(dout)
    causal = not a.noncausal
    lse = bp_hip.flash_fwd(qkv[:, 0], qkv[:, 1], qkv[:, 2], out, None, None, S, S, D ** -0.5, causal)
    dqkv = torch.empty_like(qkv)
    run = lambda: bp_hip.flash_bwd(dout, qkv[:, 0], qkv[:, 1], qkv[:, 2], out, lse, dqkv[:, 0], dqkv[:, 1], dqkv[:, 2],
                                   None, None, S, S, D ** -0.5, causal)
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    lib = ctypes.CDLL(bp_hip.LIB_PATH)
    lib.bp_dev_bwd_prof.argtypes = [ctypes.c_void_p, ctypes.c_int]
    assert lib.bp_dev_bwd_prof(None, 1) == 0
    torch.cuda.synchronize()
    run()
    torch.cuda.synchronize()
    n = 2 * 8192 * 2 * 8
    buf = (ctypes.c_ulonglong * n)()
    assert lib.bp_dev_bwd_prof(buf, 0) == 0
    t = np.frombuffer(buf, dtype=np.uint64).reshape(2, 8192, 2, 8).astype(np.int64)
    names = ['entry -> fragments arrived', 'descriptors', 'first DMA issue -> first tile landed + barrier',
             'leading edge tiles (dkdv; dq: 0)', 'clean tiles', 'trailing edge tiles', 'epilogue (stores issued)']
    for kern, kname in ((0, 'dq'), (1, 'dkdv')):
        for ps in (0, 1):
            x = t[kern, :, ps]
            ok = (x[:, 0] > 0) & (x[:, 7] > 0)
            if not ok.any():
                continue
            x = x[ok]
            print(f'{kname} pass {ps}: {ok.sum()} workgroups (wave 0), whole pass median {int(np.median(x[:, 7] - x[:, 0]))} ticks')
            prev = x[:, 0]
            for i, name in zip((1, 2, 3, 4, 5, 6, 7), names):
                cur = np.where(x[:, i] > 0, x[:, i], prev)
                d = cur - prev
                print(f'   {name:52s} median {int(np.median(d)):7d}  p10 {int(np.percentile(d, 10)):7d}  p90 {int(np.percentile(d, 90)):7d}')
                prev = cur
        x0, x1 = t[kern, :, 0], t[kern, :, 1]
        ok = (x0[:, 7] > 0) & (x1[:, 0] > 0)
        if ok.any():
            print(f'{kname}: pass 0 stores -> pass 1 entry: median {int(np.median(x1[ok, 0] - x0[ok, 7]))}')
        allx = t[kern][t[kern] > 0]
        print(f'{kname}: kernel span {int(allx.max() - allx.min())} ticks')


if __name__ == '__main__':
    main()

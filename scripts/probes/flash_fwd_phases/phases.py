"""Per-phase clock account of the flash forward under load (development build with -DBP_FWD_PROFILE):

    python backpacks-flash-attn_amd/build_hip.py --variant fwdprof -- -DBP_FWD_PROFILE
    BP_HIP_LIB=.../libbackpack_hip_fwdprof.so python scripts/probes/flash_fwd_phases/phases.py --batch 256

Every wave sums s_memtime deltas (shader clocks) over its passes: 0 wait + barrier, 1 DMA issue, 2 S^T (8 MFMAs at
d = 64), 3 softmax, 4 PV (8 MFMAs) of the FAST tiles; 5 the exact tiles (first / diagonal / retried) as a whole;
6 #fast tiles; 7 pass clocks; 8 prologue (pass start -> first ring step); 9 epilogue; 10 #passes."""
import argparse
import ctypes
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(ROOT, 'backpacks-flash-attn_amd'))
import bp_hip  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=256)
    ap.add_argument('--seq', type=int, default=1024)
    ap.add_argument('--noncausal', action='store_true')
    a = ap.parse_args()
    B, S, H, D = a.batch, a.seq, 12, 64
    dt = torch.bfloat16
    torch.manual_seed(0)
    qkv = torch.randn(B * S, 3, H, D, device='cuda').to(dt)
    out = torch.empty_like(qkv[:, 0])
    cu = torch.arange(0, (B + 1) * S, S, dtype=torch.int32, device='cuda')
    run = lambda: bp_hip.flash_fwd(qkv[:, 0], qkv[:, 1], qkv[:, 2], out, cu, cu, S, S, D ** -0.5, not a.noncausal)
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    lib = bp_hip.lib()
    lib.bp_dev_fwd_prof.argtypes = [ctypes.c_void_p, ctypes.c_int]
    assert lib.bp_dev_fwd_prof(None, 1) == 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); run(); e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    buf = np.zeros((8192, 4, 16), dtype=np.uint64)   # (12 slots before round 4, run r)
    assert lib.bp_dev_fwd_prof(buf.ctypes.data_as(ctypes.c_void_p), 0) == 0
    p = buf.astype(np.float64)
    fast, passes = p[:, :, 6].sum(), p[:, :, 10].sum()
    res = dict(batch=B, seq=S, causal=not a.noncausal, kernel_ms=round(ms, 4), passes=int(passes), fast_tiles_per_pass=round(fast / passes, 2))
    for k, n in enumerate(['wait+barrier', 'DMA issue']):
        res[n + ' per ring step'] = None
    names = {2: 'S^T per fast tile', 3: 'softmax per fast tile', 4: 'PV per fast tile'}
    for k, n in names.items():
        res[n] = round(p[:, :, k].sum() / fast, 1)
    tot = p[:, :, 7].sum()
    res['per pass: clocks'] = round(tot / passes, 1)
    for k, n in {0: 'wait+barrier', 1: 'DMA issue', 2: 'S^T (fast)', 3: 'softmax (fast)', 4: 'PV (fast)', 5: 'exact tiles', 8: 'prologue', 9: 'epilogue'}.items():
        res['share: ' + n] = round(p[:, :, k].sum() / tot, 4)
    res['per pass: wait+barrier'] = round(p[:, :, 0].sum() / passes, 1)
    res['per pass: exact tiles'] = round(p[:, :, 5].sum() / passes, 1)
    res['per pass: prologue'] = round(p[:, :, 8].sum() / passes, 1)
    res['per pass: epilogue'] = round(p[:, :, 9].sum() / passes, 1)
    del res['wait+barrier per ring step'], res['DMA issue per ring step']
    # round 4: lifetime of a workgroup (kernel entry -> exit of wave 0), in shader clocks and in 10-ns real-time ticks
    life_c, life_r = p[:, 0, 11], p[:, 0, 12]
    alive = life_r > 0
    if alive.any():
        res['workgroups recorded'] = int(alive.sum())
        res['workgroup lifetime us'] = round(life_r[alive].mean() * 0.01, 2)
        res['shader clock GHz'] = round(life_c[alive].sum() / (life_r[alive].sum() * 10.0), 3)
        res['passes per workgroup'] = round(p[alive][:, 0, 10].sum() / alive.sum(), 2)
        res['pass clocks / lifetime clocks (wave 0)'] = round(p[alive][:, 0, 7].sum() / life_c[alive].sum(), 4)
        res['sum of lifetimes / kernel time (= resident workgroups, chip)'] = round(life_r[alive].sum() * 0.01 / (ms * 1000.0), 1)
    print(json.dumps(res))


if __name__ == '__main__':
    main()

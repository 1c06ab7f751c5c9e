"""Where a wave of the sense-mix kernel spends its clean steps (development build with -DBP_MIX_PROFILE):

    python backpacks-flash-attn_amd/build_hip.py --variant mixprof -- -DBP_MIX_PROFILE
    BP_HIP_LIB=.../libbackpack_hip_mixprof.so python scripts/probes/mix_timeline/timeline.py --batch 16

Every wave sums s_memtime ticks (100 MHz on gfx950: ticks are shader clocks here) per phase of its clean steps:
  0 wait + barrier in front of X   1 X: S^T both halves, softmax of half 0, PV half 0 keys 0-15 with the exponentials of
  half 1, pack   2 wait + barrier in front of Y   3 Y: the other 24 MFMAs   5 edge steps (whole)   6 #clean steps
  7 job ticks      (the first version of this probe, r03_aa, split the round-2 step into five phases instead)
Prints per-phase averages per clean step over all waves, and the same split by the wave's position."""
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
    ap.add_argument('--batch', type=int, default=16)
    ap.add_argument('--seq', type=int, default=1024)
    a = ap.parse_args()
    B, S, K, d = a.batch, a.seq, 16, 768
    dt = torch.bfloat16
    torch.manual_seed(0)
    qk = torch.randn(B, S, 2, K, d // K, device='cuda').to(dt)
    c = torch.randn(B, S, K, d, device='cuda').to(dt)
    lse = bp_hip.sense_lse(qk)
    out = torch.empty(B, S, d, device='cuda', dtype=dt)
    for _ in range(3):
        bp_hip.sense_mix(qk, c, out=out, lse=lse)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    bp_hip.sense_mix(qk, c, out=out, lse=lse)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    lib = bp_hip.lib()
    buf = np.zeros((256, 8, 8), dtype=np.uint64)
    lib.bp_dev_mix_prof.argtypes = [ctypes.c_void_p]
    assert lib.bp_dev_mix_prof(buf.ctypes.data_as(ctypes.c_void_p)) == 0
    p = buf.astype(np.float64)
    steps = p[:, :, 6]
    live = steps > 0
    names = ['wait+barrier before X', 'X', 'wait+barrier before Y', 'Y']
    res = dict(batch=B, kernel_ms=round(ms, 4), clean_steps_per_wave=float(steps[live].mean()))
    tot = 0.0
    for k, n in enumerate(names):
        v = float(p[:, :, k][live].sum() / steps[live].sum())
        res[n] = round(v, 2)
        tot += v
    res['clean step total (shader clocks)'] = round(tot, 2)
    res['edge ticks per job-wave share'] = round(float(p[:, :, 5][live].sum() / p[:, :, 7][live].sum()), 4)
    res['clean share of job ticks'] = round(float(p[:, :, :4][live].sum() / p[:, :, 7][live].sum()), 4)
    jt = p[:, 0, 7][live[:, 0]]
    res['job ticks per workgroup (mean)'] = float(jt.mean())
    res['job ticks per workgroup (min / median / max)'] = [float(jt.min()), float(np.median(jt)), float(jt.max())]
    res['workgroups with work'] = int(live[:, 0].sum())
    res['kernel ticks if the counter runs at 2.4 GHz'] = round(ms * 2.4e6)
    per_wave = {}
    for w in range(8):
        m = live[:, w]
        per_wave[w] = [round(float(p[:, w, k][m].sum() / steps[:, w][m].sum()), 1) for k in range(4)]
    res['per wave'] = per_wave
    print(json.dumps(res))


if __name__ == '__main__':
    main()

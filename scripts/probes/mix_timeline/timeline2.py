"""Round-5 layout of the sense-mix phase counters (development build with -DBP_MIX_PROFILE; every step of a job now has the
two-phase form, sense_mix_dma.hip):

    python backpacks-flash-attn_amd/build_hip.py --variant mixprof -- -DBP_MIX_PROFILE
    BP_HIP_LIB=.../libbackpack_hip_mixprof.so python scripts/probes/mix_timeline/timeline2.py --batch 64

slots per wave: clean steps 0 wait+barrier before X, 1 X, 2 wait+barrier before Y, 3 Y; diagonal steps 4 whole step, 5 X + Y of
the steps in which the wave is live; 6 = #clean + (#diagonal << 32) + (#live diagonal << 48); 7 job ticks."""
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
    ap.add_argument('--batch', type=int, default=64)
    ap.add_argument('--seq', type=int, default=1024)
    ap.add_argument('--senses', type=int, default=16)
    ap.add_argument('--d', type=int, default=768)
    a = ap.parse_args()
    B, S, K, d = a.batch, a.seq, a.senses, a.d
    dt = torch.bfloat16
    torch.manual_seed(0)
    dk = -(-(d // K) // 8) * 8
    qk = torch.randn(B, S, 2, K, dk, device='cuda').to(dt)
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
    buf = np.zeros((256, 8, 12), dtype=np.uint64)
    lib.bp_dev_mix_prof.argtypes = [ctypes.c_void_p]
    assert lib.bp_dev_mix_prof(buf.ctypes.data_as(ctypes.c_void_p)) == 0
    cnt = buf[:, :, 6]
    clean = (cnt & np.uint64(0xffffffff)).astype(np.float64)
    diag = ((cnt >> np.uint64(32)) & np.uint64(0xffff)).astype(np.float64)
    diag_live = (cnt >> np.uint64(48)).astype(np.float64)
    p = buf.astype(np.float64)
    res = dict(batch=B, seq=S, senses=K, d=d, kernel_ms=round(ms, 4))
    names = ['wait+barrier before X', 'X', 'wait+barrier before Y', 'Y']
    tot = 0.0
    for k, n in enumerate(names):
        v = float(p[:, :, k].sum() / max(clean.sum(), 1))
        res['clean: ' + n] = round(v, 1)
        tot += v
    res['clean step (clocks)'] = round(tot, 1)
    res['diagonal step (clocks, all waves)'] = round(float(p[:, :, 4].sum() / max(diag.sum(), 1)), 1)
    res['diagonal step: X + Y of a live wave'] = round(float(p[:, :, 5].sum() / max(diag_live.sum(), 1)), 1)
    res['diagonal step: X + Y of a live wave ALONE on its SIMD'] = round(float(p[:, :, 8].sum() / max(p[:, :, 9].sum(), 1)), 1)
    res['diagonal step: X + Y of a live wave whose SIMD partner is live too'] = round(float(p[:, :, 10].sum() / max(p[:, :, 11].sum(), 1)), 1)
    res['live diagonal wave-steps alone / paired'] = [float(p[:, :, 9].sum()), float(p[:, :, 11].sum())]
    res['steps per wave: clean / diagonal / diagonal live'] = [float(clean.mean()), float(diag.mean()), float(diag_live.mean())]
    job = p[:, 0, 7]
    inside = p[:, 0, :5].sum(axis=1) - p[:, 0, 4] + p[:, 0, 4]
    res['job ticks per workgroup (mean)'] = float(job.mean())
    res['share of job ticks inside steps (wave 0)'] = round(float((p[:, 0, :4].sum(axis=1) + p[:, 0, 4]).sum() / job.sum()), 4)
    res['kernel clocks at 2.4 GHz'] = round(ms * 2.4e6)
    per_wave = {}
    for w in range(8):
        per_wave[w] = dict(clean=[round(float(p[:, w, k].sum() / max(clean[:, w].sum(), 1)), 1) for k in range(4)],
                           diagonal=round(float(p[:, w, 4].sum() / max(diag[:, w].sum(), 1)), 1),
                           live_xy=round(float(p[:, w, 5].sum() / max(diag_live[:, w].sum(), 1)), 1))
    res['per wave'] = per_wave
    print(json.dumps(res))


if __name__ == '__main__':
    main()

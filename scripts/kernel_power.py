"""Shader clock and socket power under each kind of kernel of the Backpack-Small step, one at a time: the kernel alone, back to
back for ~1.5 s on random data at the bench batch, bench.py's ClockPowerSampler on a side thread (the samples taken under
load, i.e. at >= 80 % of the window's highest clock, are averaged).  The power account behind "power-limited" (DESIGN.md
section 4): which launches sit at the 1.4 kW cap, at which clock, and what they achieve there.

    python scripts/kernel_power.py [--batch 2048] [--seconds 1.5] > profiles/r06_x_kernel_power.jsonl
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'backpacks-flash-attn_amd')]
import torch  # noqa: E402

import bench  # noqa: E402
import bp_hip  # noqa: E402


def sustained(name, fn, seconds, work):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fn(); e1.record(); torch.cuda.synchronize()
    n = max(8, int(seconds * 1e3 / max(e0.elapsed_time(e1), 1e-3)))
    smi = bench.ClockPowerSampler(0, period_s=0.01).start()
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    c = smi.stop()
    ms = e0.elapsed_time(e1) / n
    busy = [(a, b) for a, b in smi.samples if a and c['sclk_mhz_max'] and a >= 0.8 * c['sclk_mhz_max']]
    row = dict(kernel=name, launches=n, avg_ms=round(ms, 4))
    if 'flops' in work:
        row['tflops'] = round(work['flops'] / ms / 1e9, 1)
    if 'bytes' in work:
        row['algorithmic_gbps'] = round(work['bytes'] / ms / 1e6, 1)
    if busy:
        row['sclk_mhz_under_load'] = round(sum(a for a, _ in busy) / len(busy), 1)
        pw = [b for _, b in busy if b]
        row['power_w_under_load'] = round(sum(pw) / len(pw), 1) if pw else None
    row.update(power_w_max=c['power_w_max'], power_cap_w=c['power_cap_w'], samples=c['samples'], source=c['source'])
    print(json.dumps(row), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=2048)
    ap.add_argument('--seq', type=int, default=1024)
    ap.add_argument('--seconds', type=float, default=1.5)
    a = ap.parse_args()
    dev, dt = torch.device('cuda', 0), torch.bfloat16
    B, S, H, D, k, d = a.batch, a.seq, 12, 64, 16, 768
    pairs = S * (S + 1) // 2
    M = B * S
    # trunk attention
    qkv = torch.randn(M, 3, H, D, device=dev, dtype=dt)
    o = torch.empty(M, H, D, device=dev, dtype=dt)
    sustained('flash_fwd (trunk attention, causal)', lambda: bp_hip.flash_fwd(qkv[:, 0], qkv[:, 1], qkv[:, 2], o, None, None, S, S, D ** -0.5, True),
              a.seconds, dict(flops=4 * pairs * d * B, bytes=8 * S * d * B))
    del qkv, o
    # fused sense mix, table form (the bench's form): 50 257-row table, uniformly random ids
    qk = (1.3 * torch.randn(B, S, 2, k, d // k, device=dev, dtype=dt))
    table = torch.randn(50257, k, d, device=dev, dtype=dt)
    idx = torch.randint(0, 50257, (B, S), device=dev, dtype=torch.int32)
    out = torch.empty(B, S, d, device=dev, dtype=dt)
    lse = bp_hip.sense_lse(qk)
    sustained('sense_lse (LSE pre-pass of the 16 senses)', lambda: bp_hip.sense_lse(qk), a.seconds, dict(flops=2 * pairs * d * B))
    sustained('sense_mix_gather (fused alpha.C from the token table)', lambda: bp_hip.sense_mix_gather(qk, table, idx, out=out, lse=lse),
              a.seconds, dict(flops=2 * pairs * d * (1 + k) * B, bytes=(4 + 2 * k + 2) * S * d * B))
    del qk, table, idx, out, lse
    # fused add + LayerNorm (fp32 residual stream)
    x0 = torch.randn(M, d, device=dev, dtype=dt)
    res = torch.randn(M, d, device=dev, dtype=torch.float32)
    w, b = torch.ones(d, device=dev, dtype=dt), torch.zeros(d, device=dev, dtype=dt)
    sustained('add_layer_norm (fused residual add + LayerNorm)', lambda: bp_hip.add_layer_norm(x0, res, w, b, 1e-5), a.seconds,
              dict(bytes=12 * M * d))
    del res
    # the library GEMMs of a trunk layer (hipBLASLt through torch): fc1 + GELU epilogue, fc2, Wqkv
    w1, b1 = torch.randn(3072, d, device=dev, dtype=dt) * 0.02, torch.zeros(3072, device=dev, dtype=dt)
    sustained('GEMM fc1 + GELU (hipBLASLt, M x 768 x 3072)', lambda: torch._addmm_activation(b1, x0, w1.t(), use_gelu=True), a.seconds,
              dict(flops=2.0 * M * d * 3072, bytes=2.0 * M * (d + 3072)))
    h = torch.randn(M, 3072, device=dev, dtype=dt)
    w2, b2 = torch.randn(d, 3072, device=dev, dtype=dt) * 0.02, torch.zeros(d, device=dev, dtype=dt)
    sustained('GEMM fc2 (hipBLASLt, M x 3072 x 768)', lambda: torch.nn.functional.linear(h, w2, b2), a.seconds,
              dict(flops=2.0 * M * d * 3072, bytes=2.0 * M * (d + 3072)))
    del h
    wq, bq = torch.randn(3 * d, d, device=dev, dtype=dt) * 0.02, torch.zeros(3 * d, device=dev, dtype=dt)
    sustained('GEMM Wqkv (hipBLASLt, M x 768 x 2304)', lambda: torch.nn.functional.linear(x0, wq, bq), a.seconds,
              dict(flops=2.0 * M * d * 3 * d, bytes=2.0 * M * 4 * d))


if __name__ == '__main__':
    main()

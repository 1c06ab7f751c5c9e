"""Micro-benchmark of the individual HIP kernels on synthetic N(0,1) data (never zeros: zero-filled
inputs clock higher and flatter -- cdna_hip_programming.md section 5.4 rule 25).

    python scripts/bench_kernels.py [--which flash,bwd,mix,mixgather,mixgatherref,lse,alpha,mixbwd,lnbwd,xent,gelu] [--batch 64] [--seq 1024] [--iters 20]
Prints one JSON line per kernel with avg ms, algorithmic TFLOP/s and GB/s (SURVEY section 8d figures)."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'backpacks-flash-attn_amd')):
    sys.path.insert(0, p)
import torch  # noqa: E402

import bp_hip  # noqa: E402


def timeit(fn, iters, warmup=3):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--which', default='flash,mix,lse,alpha')
    ap.add_argument('--batch', type=int, default=64)
    ap.add_argument('--seq', type=int, default=1024)
    ap.add_argument('--heads', type=int, default=12)
    ap.add_argument('--headdim', type=int, default=64)
    ap.add_argument('--senses', type=int, default=16)
    ap.add_argument('--d', type=int, default=768)
    ap.add_argument('--dtype', default='bf16')
    ap.add_argument('--iters', type=int, default=20)
    ap.add_argument('--noncausal', action='store_true')
    ap.add_argument('--content-layout', default='bskd', choices=['bskd', 'bksd'],
                    help="storage order of the content tensor handed to the mix kernel (a strided view either way)")
    a = ap.parse_args()
    dt = torch.bfloat16 if a.dtype == 'bf16' else torch.float16
    dev = 'cuda'
    B, S, H, D, K, d = a.batch, a.seq, a.heads, a.headdim, a.senses, a.d
    pairs = S * (S + 1) // 2
    torch.manual_seed(0)
    which = a.which.split(',')
    res = []
    if 'flash' in which:
        qkv = torch.randn(B * S, 3, H, D, device=dev).to(dt)
        out = torch.empty_like(qkv[:, 0])
        cu = torch.arange(0, (B + 1) * S, S, dtype=torch.int32, device=dev)
        if os.environ.get('BP_BENCH_FIXED_LEN') == '1':
            cu = None   # fixed-length entry of the C ABI: no cu_seqlens reads in the kernel
        causal = not a.noncausal
        ms = timeit(lambda: bp_hip.flash_fwd(qkv[:, 0], qkv[:, 1], qkv[:, 2], out, cu, cu, S, S, D ** -0.5, causal), a.iters)
        fl, by = 4 * (pairs if causal else S * S) * D * H * B, 8 * S * D * H * B
        res.append(dict(kernel='flash_fwd', ms=ms, tflops=fl / ms / 1e9, gbps=by / ms / 1e6))
    if 'bwd' in which:
        qkv = torch.randn(B * S, 3, H, D, device=dev).to(dt)
        dout = torch.randn(B * S, H, D, device=dev).to(dt)
        out = torch.empty_like(dout)
        cu = torch.arange(0, (B + 1) * S, S, dtype=torch.int32, device=dev)
        causal = not a.noncausal
        lse = bp_hip.flash_fwd(qkv[:, 0], qkv[:, 1], qkv[:, 2], out, cu, cu, S, S, D ** -0.5, causal)
        dqkv = torch.empty_like(qkv)
        ms = timeit(lambda: bp_hip.flash_bwd(dout, qkv[:, 0], qkv[:, 1], qkv[:, 2], out, lse, dqkv[:, 0],
                                             dqkv[:, 1], dqkv[:, 2], cu, cu, S, S, D ** -0.5, causal), a.iters)
        # 5 matmuls of the textbook backward (S, dP, dV, dK, dQ); this split recomputes S and dP once more
        fl, by = 10 * (pairs if causal else S * S) * D * H * B, 16 * S * D * H * B
        res.append(dict(kernel='flash_bwd(dkdv+dq+dsum)', ms=ms, tflops=fl / ms / 1e9, gbps=by / ms / 1e6))
    if {'lse', 'mix', 'mixgather', 'alpha', 'mixbwd'} & set(which):
        qk = torch.randn(B, S, 2, K, d // K, device=dev).to(dt)
    if 'lse' in which:
        ms = timeit(lambda: bp_hip.sense_lse(qk), a.iters)
        res.append(dict(kernel='sense_lse', ms=ms, tflops=2 * pairs * d * B / ms / 1e9,
                        gbps=(4 * S * d + 4 * K * S) * B / ms / 1e6))
    if 'mix' in which:
        if a.content_layout == 'bksd':   # sense-major storage: a (sense, key) row is 2*d bytes from the next key's
            c = torch.randn(B, K, S, d, device=dev).to(dt).transpose(1, 2)
        else:
            c = torch.randn(B, S, K, d, device=dev).to(dt)
        lse = bp_hip.sense_lse(qk)
        out = torch.empty(B, S, d, device=dev, dtype=dt)
        ms = timeit(lambda: bp_hip.sense_mix(qk, c, out=out, lse=lse), a.iters)
        res.append(dict(kernel='sense_mix', layout=a.content_layout, ms=ms, tflops=2 * pairs * d * (1 + K) * B / ms / 1e9,
                        gbps=(4 + 2 * K + 2) * S * d * B / ms / 1e6))
    if 'mixgather' in which:
        # the inference form: content rows read from the per-token table (50 257 distinct ids of a large batch)
        rows = int(os.environ.get('BP_BENCH_TABLE_ROWS', '50257'))   # (a few rows = an L2-resident table: what the kernel does without HBM)
        table = torch.randn(rows, K, d, device=dev).to(dt)
        index = torch.randint(0, rows, (B, S), device=dev, dtype=torch.int32)
        lse = bp_hip.sense_lse(qk)
        out = torch.empty(B, S, d, device=dev, dtype=dt)
        ms = timeit(lambda: bp_hip.sense_mix_gather(qk, table, index, out=out, lse=lse), a.iters)
        # algorithmic bytes: q, k once, the table once, the index, the output
        by = (4 * S * d + 2 * S * d + 4 * S) * B + 2 * rows * K * d
        res.append(dict(kernel='sense_mix_gather', ms=ms, tflops=2 * pairs * d * (1 + K) * B / ms / 1e9, gbps=by / ms / 1e6,
                        algorithmic_bytes=by))
    if 'mixgatherref' in which:
        # what the table form replaces when the kernel does not gather: torch gathers the (B,S,k,d) rows, then bp_sense_mix
        rows = int(os.environ.get('BP_BENCH_TABLE_ROWS', '50257'))
        table = torch.randn(rows, K, d, device=dev).to(dt)
        index = torch.randint(0, rows, (B, S), device=dev, dtype=torch.int64)
        lse = bp_hip.sense_lse(qk)
        out = torch.empty(B, S, d, device=dev, dtype=dt)
        ms = timeit(lambda: bp_hip.sense_mix(qk, table[index], out=out, lse=lse), a.iters)
        res.append(dict(kernel='torch_gather+sense_mix', ms=ms, tflops=2 * pairs * d * (1 + K) * B / ms / 1e9))
    if 'alpha' in which:
        Ba = min(B, 64)     # (Ba, k, S, S) 16-bit: 2.1 GB at 64 x 16 x 1024^2 -- far past the 256 MiB Infinity Cache
        lse = bp_hip.sense_lse(qk[:Ba])
        ms = timeit(lambda: bp_hip.sense_alpha(qk[:Ba], lse=lse), a.iters)
        by = (4 * S * d + 2 * K * S * S) * Ba
        res.append(dict(kernel='attn_probs(alpha)', batch=Ba, ms=ms, tflops=2 * pairs * d * Ba / ms / 1e9,
                        gbps=by / ms / 1e6))
    if 'mixbwd' in which:
        c = torch.randn(B, S, K, d, device=dev).to(dt)
        dout = torch.randn(B, S, d, device=dev).to(dt)
        lse = bp_hip.sense_lse(qk)
        scale = (d // K) ** -0.5
        ms = timeit(lambda: bp_hip.sense_mix_dc(qk, dout, lse, scale, c), a.iters)
        # dC[j] = sum_i alpha[i,j] dout[i]: QK^T once per sense + alpha^T.dout per sense (a d-wide output per sense)
        res.append(dict(kernel='sense_mix_dc', ms=ms, tflops=2 * pairs * d * (1 + K) * B / ms / 1e9,
                        gbps=(4 + 2 + 2 * K) * S * d * B / ms / 1e6))
        ms = timeit(lambda: bp_hip.sense_dqk(qk, c, dout, lse, scale), a.iters)
        # dP = dout C^T (k d-wide dots per pair), then QK^T recomputed twice (dq and dk passes) + dS.K + dS^T.Q
        res.append(dict(kernel='sense_dqk(slab gemm + dq + dk)', ms=ms,
                        tflops=(2 * pairs * d * K + 8 * pairs * d) * B / ms / 1e9,
                        gbps=(4 + 2 + 2 * K + 4) * S * d * B / ms / 1e6))
    if 'ln' in which:
        rows, cols = B * S, d
        x0 = torch.randn(rows, cols, device=dev).to(dt)
        x1 = torch.randn(rows, cols, device=dev)
        w, bias = torch.ones(cols, device=dev).to(dt), torch.zeros(cols, device=dev).to(dt)
        ms = timeit(lambda: bp_hip.add_layer_norm(x0, x1, w, bias, 1e-5), a.iters)
        # reads x0 (2) + residual (4), writes z (2) + residual (4) bytes per element: the trunk's per-block call
        res.append(dict(kernel='add_layer_norm', rows=rows, ms=ms, tflops=0.0, gbps=rows * cols * 12 / ms / 1e6))
    if 'lnbwd' in which:
        rows, cols = B * S, d
        x = torch.randn(rows, cols, device=dev)
        dz = torch.randn(rows, cols, device=dev).to(dt)
        dxr = torch.randn(rows, cols, device=dev)
        w = torch.ones(cols, device=dev)
        ms = timeit(lambda: bp_hip.add_layer_norm_bwd(dz, dxr, x, w, 1e-5, want_dx1=True), a.iters)
        # reads dz (2) + dx_in (4) + x (4), writes dx0 (2) + dx1 (4) bytes per element
        res.append(dict(kernel='add_layer_norm_bwd', ms=ms, tflops=0.0, gbps=rows * cols * 16 / ms / 1e6))
    if 'xent' in which:
        rows, V = min(B * S, 32768), 50264
        x = torch.randn(rows, V, device=dev).to(dt)
        y = torch.randint(0, V, (rows,), device=dev)
        ms = timeit(lambda: bp_hip.xentropy_fwd(x, y), a.iters)
        res.append(dict(kernel='xentropy_fwd', rows=rows, ms=ms, tflops=0.0, gbps=rows * V * 2 / ms / 1e6))
        losses, lse = bp_hip.xentropy_fwd(x, y)
        g = torch.ones(rows, device=dev)
        ms = timeit(lambda: bp_hip.xentropy_bwd(g, x, lse, y, inplace=False), a.iters)
        res.append(dict(kernel='xentropy_bwd', rows=rows, ms=ms, tflops=0.0, gbps=rows * V * 4 / ms / 1e6))
    if 'gelu' in which:
        rows, cols = B * S, 4 * d
        x = torch.randn(rows, cols, device=dev).to(dt)
        g = (torch.randn(rows, cols, device=dev) / 8).to(dt)
        y = torch.empty_like(x)
        ms = timeit(lambda: bp_hip.bias_gelu_fwd(x, out=y), a.iters)
        res.append(dict(kernel='bias_gelu_fwd', rows=rows, ms=ms, tflops=0.0, gbps=rows * cols * 4 / ms / 1e6))
        ms = timeit(lambda: torch.nn.functional.gelu(x, approximate='tanh'), a.iters)
        res.append(dict(kernel='torch gelu (same bytes)', rows=rows, ms=ms, tflops=0.0, gbps=rows * cols * 4 / ms / 1e6))
        dp = torch.empty_like(x)
        ms = timeit(lambda: bp_hip.bias_gelu_bwd(g, x, torch.float32), a.iters)
        res.append(dict(kernel='bias_gelu_bwd(+dbias)', rows=rows, ms=ms, tflops=0.0, gbps=rows * cols * 6 / ms / 1e6))
        ms = timeit(lambda: bp_hip.column_sum(g, torch.float32), a.iters)
        res.append(dict(kernel='column_sum', rows=rows, ms=ms, tflops=0.0, gbps=rows * cols * 2 / ms / 1e6))
        ms = timeit(lambda: g.sum(0), a.iters)
        res.append(dict(kernel='torch sum(0) (same bytes)', rows=rows, ms=ms, tflops=0.0, gbps=rows * cols * 2 / ms / 1e6))
    for r in res:
        r.update(batch=r.get('batch', B), seq=S, dtype=a.dtype)
        print(json.dumps({k: (round(v, 4) if isinstance(v, float) else v) for k, v in r.items()}), flush=True)


if __name__ == '__main__':
    main()

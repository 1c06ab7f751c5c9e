"""How fast is the dP slab GEMM of bp_hip.sense_dqk as a function of the slab width?
dP^T slab (B, n, W) = content rows (B, n, 768) x dout slab^T (B, 768, W), bf16, n = keys * senses visible to the slab."""
import json
import torch

B, K, S, d = 64, 16, 1024, 768
dt = torch.bfloat16
c = torch.randn(B, S * K, d, device='cuda').to(dt)
dout = torch.randn(B, S, d, device='cuda').to(dt)


def timeit(fn, iters=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


import sys
for W in [int(a) for a in sys.argv[1:]] or (128, 256):
    buf = torch.empty(B * S * K * W, dtype=dt, device='cuda')

    def sweep():
        for t0 in range(0, S, W):
            n = min(S, t0 + W) * K
            out = buf[:B * n * W].view(B, n, W)
            torch.bmm(c[:, :n], dout[:, t0:t0 + W].transpose(1, 2), out=out)
    ms = timeit(sweep)
    flops = sum(2 * B * min(S, t0 + W) * K * W * d for t0 in range(0, S, W))
    print(json.dumps(dict(slab=W, ms=round(ms, 3), tflops=round(flops / ms / 1e9, 1), buffer_mb=B * S * K * W * 2 >> 20)), flush=True)

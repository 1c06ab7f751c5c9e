"""Same slab GEMMs as probe.py (W = 128), but the batch is walked in groups of G samples, all eight slabs of a group
before the next group: the group's content rows (G x 25 MB) then stay in the 256 MB Infinity Cache between slabs."""
import json
import sys
import torch

B, K, S, d, W = 64, 16, 1024, 768, 128
dt = torch.bfloat16
c = torch.randn(B, S * K, d, device='cuda').to(dt)
dout = torch.randn(B, S, d, device='cuda').to(dt)
buf = torch.empty(B * S * K * W, dtype=dt, device='cuda')


def timeit(fn, iters=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


for G in [int(a) for a in sys.argv[1:]] or (64, 16, 8, 4):
    def sweep():
        for b0 in range(0, B, G):
            for t0 in range(0, S, W):
                n = min(S, t0 + W) * K
                out = buf[:G * n * W].view(G, n, W)
                torch.bmm(c[b0:b0 + G, :n], dout[b0:b0 + G, t0:t0 + W].transpose(1, 2), out=out)
    ms = timeit(sweep)
    flops = sum(2 * B * min(S, t0 + W) * K * W * d for t0 in range(0, S, W))
    print(json.dumps(dict(group=G, ms=round(ms, 3), tflops=round(flops / ms / 1e9, 1))), flush=True)

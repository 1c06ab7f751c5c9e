"""What the GELU epilogue of the fc1 GEMM costs at the bench's batch (one MI355X):
addmm + GELU epilogue (what FusedDenseGeluDense runs in inference) / addmm (bias only) / mm, M = batch * 1024 rows."""
import json
import sys
import torch

M = int(sys.argv[1]) * 1024 if len(sys.argv) > 1 else 2048 * 1024
dev = 'cuda'
x = torch.randn(M, 768, device=dev).bfloat16()
w1 = torch.randn(3072, 768, device=dev).bfloat16() * 0.02
b1 = torch.randn(3072, device=dev).bfloat16()
w2 = torch.randn(768, 3072, device=dev).bfloat16() * 0.02
b2 = torch.randn(768, device=dev).bfloat16()
wq = torch.randn(2304, 768, device=dev).bfloat16() * 0.02
bq = torch.randn(2304, device=dev).bfloat16()


def timeit(fn, n=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


h = torch.empty(M, 3072, device=dev, dtype=torch.bfloat16)
res = {}
res['fc1 addmm+gelu epilogue'] = timeit(lambda: torch._addmm_activation(b1, x, w1.t(), use_gelu=True))
res['fc1 addmm (bias)'] = timeit(lambda: torch.addmm(b1, x, w1.t()))
res['fc1 mm'] = timeit(lambda: torch.mm(x, w1.t(), out=h))
res['fc2 addmm'] = timeit(lambda: torch.addmm(b2, h, w2.t()))
res['qkv addmm'] = timeit(lambda: torch.addmm(bq, x, wq.t()))
fl = {'fc1 addmm+gelu epilogue': 2 * M * 768 * 3072, 'fc1 addmm (bias)': 2 * M * 768 * 3072, 'fc1 mm': 2 * M * 768 * 3072,
      'fc2 addmm': 2 * M * 768 * 3072, 'qkv addmm': 2 * M * 768 * 2304}
print(json.dumps({k: dict(ms=round(v, 3), pflops=round(fl[k] / v / 1e12, 3)) for k, v in res.items()}))

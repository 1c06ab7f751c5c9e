"""Which BLAS backend does each GEMM shape of the Backpack-Small forward prefer?  (B*S = 65536 rows, bf16)"""
import torch, json, sys
M = 65536
shapes = {'qkv': (768, 2304), 'out_proj': (768, 768), 'fc1': (768, 3072), 'fc2': (3072, 768),
          'sense_fc2': (3072, 12288), 'sense_qk': (768, 1536), 'lm_head': (768, 50264)}
dev = 'cuda'
def timeit(fn, iters=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters
for lib in ('cublaslt', 'cublas'):
    torch.backends.cuda.preferred_blas_library(lib)
    for name, (k, n) in shapes.items():
        x = torch.randn(M, k, device=dev, dtype=torch.bfloat16)
        w = torch.randn(n, k, device=dev, dtype=torch.bfloat16) * 0.02
        b = torch.zeros(n, device=dev, dtype=torch.bfloat16)
        ms = timeit(lambda: torch.nn.functional.linear(x, w, b if name != 'lm_head' else None))
        row = dict(lib=lib, gemm=name, ms=round(ms, 4), tflops=round(2 * M * k * n / ms / 1e9, 1))
        if name == 'fc1':
            ms2 = timeit(lambda: torch._addmm_activation(b, x, w.t(), use_gelu=True))
            row['gelu_epilogue_ms'] = round(ms2, 4)
        print(json.dumps(row), flush=True)
        del x, w

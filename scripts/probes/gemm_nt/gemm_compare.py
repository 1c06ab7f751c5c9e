"""bp_linear (this repo's MFMA kernel) vs torch / hipBLASLt on the dense-layer shapes of Backpack-Small."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, 'backpacks-flash-attn_amd')):
    sys.path.insert(0, p)
import torch
import bp_hip
M = 65536
shapes = {'qkv': (768, 2304), 'out_proj': (768, 768), 'fc1': (768, 3072), 'fc2': (3072, 768), 'sense_fc2': (3072, 12288), 'lm_head': (768, 50264)}
def timeit(fn, iters=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters
for name, (k, n) in shapes.items():
    torch.manual_seed(0)
    x = torch.randn(M, k, device='cuda').bfloat16()
    w = (torch.randn(n, k, device='cuda') * 0.03).bfloat16()
    b = torch.randn(n, device='cuda').bfloat16()
    gelu = name == 'fc1'
    got = bp_hip.linear(x[:4096], w, b, gelu=gelu)
    want = torch.nn.functional.linear(x[:4096].float(), w.float(), b.float())
    if gelu: want = torch.nn.functional.gelu(want, approximate='tanh')
    err = (got.float() - want).abs().max().item()
    ref16 = torch.nn.functional.linear(x[:4096], w, b)
    if gelu: ref16 = torch.nn.functional.gelu(ref16, approximate='tanh')
    base = (ref16.float() - want).abs().max().item()
    t_own = timeit(lambda: bp_hip.linear(x, w, b, gelu=gelu))
    if gelu: t_lib = timeit(lambda: torch._addmm_activation(b, x, w.t(), use_gelu=True))
    else: t_lib = timeit(lambda: torch.nn.functional.linear(x, w, b))
    fl = 2 * M * k * n
    print(json.dumps(dict(gemm=name, own_ms=round(t_own, 4), own_tflops=round(fl / t_own / 1e9), lib_ms=round(t_lib, 4),
                          lib_tflops=round(fl / t_lib / 1e9), max_err=err, torch16_err=base)), flush=True)
    del x, w

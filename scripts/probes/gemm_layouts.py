"""hipBLASLt on the dense-layer shapes: weight stored (N,K) [nn.Linear, 'TN'] vs pre-transposed (K,N) ['NN'] (contiguous operands only)."""
import torch, json
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
    x = torch.randn(M, k, device='cuda', dtype=torch.bfloat16)
    w = torch.randn(n, k, device='cuda', dtype=torch.bfloat16) * 0.02
    wt = w.t().contiguous()
    b = torch.zeros(n, device='cuda', dtype=torch.bfloat16)
    t_tn = timeit(lambda: torch.nn.functional.linear(x, w, b))
    t_nn = timeit(lambda: torch.addmm(b, x, wt))
    t_nn_nobias = timeit(lambda: torch.mm(x, wt))
    t_tn_nobias = timeit(lambda: torch.nn.functional.linear(x, w))
    fl = 2 * M * k * n / 1e9
    print(json.dumps(dict(gemm=name, tn_ms=round(t_tn, 4), tn_tf=round(fl / t_tn), nn_ms=round(t_nn, 4), nn_tf=round(fl / t_nn),
                          tn_nobias_tf=round(fl / t_tn_nobias), nn_nobias_tf=round(fl / t_nn_nobias))), flush=True)
    del x, w, wt

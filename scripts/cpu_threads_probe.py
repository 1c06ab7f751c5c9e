"""How does the CPU oracle scale with torch threads on this host? (run on the GPU box)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from oracle import ref_cpu as R
print('cpu_count', os.cpu_count(), 'affinity', len(os.sched_getaffinity(0)))
cfg = R.make_config('small', n_positions=1024)
sd = R.init_state_dict(cfg, seed=0)
ids = torch.randint(0, 50257, (1, 1024))
for th in (8, 16, 32, 64, 128):
    torch.set_num_threads(th)
    with torch.no_grad():
        R.backpack_forward(sd, cfg, ids)
        t0 = time.perf_counter(); R.backpack_forward(sd, cfg, ids); dt = time.perf_counter() - t0
    print(f'threads {th}: {dt:.2f} s/forward -> {1024 / dt:.0f} tok/s', flush=True)
    if dt > 20: break

"""Forward time of a Backpack model (ids -> final hidden states, no LM head) by batch size in the three content orders of
BackpackModel.sense_table_mode: 'off' (content network on every position, the reference's order), 'batch' (once per
distinct token of the batch, rebuilt per forward; `dedup_min_positions` forced to 0 so that it also runs where the default
threshold would skip it) and 'cached' (whole-vocabulary table kept across forwards).  Measures where 'batch' starts to pay
(the default threshold, 2 x vocab positions) and what 'cached' gives at small batches.

    python scripts/bench_content_modes.py [--model small] [--seq 1024] [--batches 1,2,4,...]"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'backpacks-flash-attn_amd')):
    sys.path.insert(0, p)
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--model', default='small')
    ap.add_argument('--seq', type=int, default=1024)
    ap.add_argument('--batches', default='1,2,4,8,16,32,48,64,98,128,192,256')
    ap.add_argument('--iters', type=int, default=5)
    a = ap.parse_args()
    from bench import build_model
    dev = torch.device('cuda', 0)
    cfg, model = build_model(a.model, a.seq, torch.bfloat16, dev)
    t = model.transformer
    t.dedup_min_positions = 0
    for b in [int(x) for x in a.batches.split(',')]:
        ids = torch.randint(0, 50257, (b, a.seq), device=dev, generator=torch.Generator(device=dev).manual_seed(b))
        row = dict(model=a.model, seq=a.seq, batch=b, positions=b * a.seq, vocab=cfg.vocab_size,
                   distinct_ids=int(torch.unique(ids).numel()))
        for mode in ('off', 'batch', 'cached'):
            t.sense_table_mode = mode
            with torch.no_grad():
                for _ in range(2):
                    t(ids)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(a.iters):
                    t(ids)
                torch.cuda.synchronize()
            row[mode + '_ms'] = round((time.perf_counter() - t0) / a.iters * 1e3, 3)
        print(json.dumps(row), flush=True)


if __name__ == '__main__':
    main()

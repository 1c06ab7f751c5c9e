"""One GPU run on the library GEMMs of the Backpack-Small forward (review round 5, item 6): hipBLASLt's heuristic pick
against the best solution PyTorch's TunableOp finds, for the shapes the bench step actually launches, measured on the
whole step and per shape; plus the LM head as one GEMM against row chunks into the same logits block.

    python scripts/gemm_tune.py --batch 2048 --out gpurun_out/tunable_small1024_b2048.csv

Writes the TunableOp results file (`--out`) and prints JSON lines.  Configuration, not a kernel of this repository."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'backpacks-flash-attn_amd')]
import torch  # noqa: E402
import torch.cuda.tunable as tunable  # noqa: E402

import bench  # noqa: E402


def steps_ms(fn, n):
    fn(); fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def ev_ms(fn, n=5):
    fn(); fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=2048)
    ap.add_argument('--seq', type=int, default=1024)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--content', default='batch', choices=['batch', 'cached', 'position'])
    ap.add_argument('--lm-head-chunks', default='0,131072,262144,524288')
    ap.add_argument('--tune-ms', type=int, default=30)
    ap.add_argument('--tune-iters', type=int, default=10)
    ap.add_argument('--out', default='gpurun_out/tunable_results.csv')
    a = ap.parse_args()
    dev = torch.device('cuda', 0)
    cfg, model = bench.build_model('small', a.seq, torch.bfloat16, dev)
    model.transformer.sense_table_mode = {'batch': 'batch', 'cached': 'cached', 'position': 'off'}[a.content]
    ids = torch.randint(0, 50257, (a.batch, a.seq), device=dev, generator=torch.Generator(device=dev).manual_seed(1234))
    logits = torch.empty((a.batch, a.seq, cfg.vocab_size), dtype=torch.bfloat16, device=dev)
    tok = a.batch * a.seq

    def step():
        with torch.no_grad():
            model(ids, logits_out=logits)

    def say(**kw):
        print(json.dumps(kw), flush=True)

    # 1. the LM head: one GEMM against row chunks (library heuristic in both)
    best_chunk, best_ms = 0, None
    for rows in [int(x) for x in a.lm_head_chunks.split(',')]:
        model.lm_head_chunk_rows = rows
        ms = steps_ms(step, a.steps)
        say(what='step, heuristic GEMMs', lm_head_chunk_rows=rows, ms_per_step=round(ms, 2), tokens_per_s=round(tok / ms * 1e3))
        if best_ms is None or ms < best_ms:
            best_chunk, best_ms = rows, ms
    # the tuned run needs a chunked head: TunableOp deep-copies the output of the GEMM it tunes (211 GB at B = 2048)
    tune_chunk = best_chunk or 262144
    model.lm_head_chunk_rows = tune_chunk
    base_ms = steps_ms(step, a.steps)

    # 2. tune every GEMM shape of the step once
    os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
    tunable.enable(True)
    tunable.set_filename(a.out)
    tunable.set_max_tuning_duration(a.tune_ms)
    tunable.set_max_tuning_iterations(a.tune_iters)
    tunable.tuning_enable(True)
    t0 = time.perf_counter()
    step()
    torch.cuda.synchronize()
    say(what='tuning forward', seconds=round(time.perf_counter() - t0, 1))
    tunable.tuning_enable(False)
    tuned_ms = steps_ms(step, a.steps)
    try:
        if hasattr(tunable, 'write_file'):
            tunable.write_file(a.out)
    except Exception as e:   # noqa: BLE001
        say(what='write_file failed', error=repr(e))
    say(what='step, tuned GEMMs', lm_head_chunk_rows=tune_chunk, ms_per_step=round(tuned_ms, 2),
        tokens_per_s=round(tok / tuned_ms * 1e3), heuristic_ms_same_chunking=round(base_ms, 2),
        gain_pct=round((base_ms / tuned_ms - 1) * 100, 2))
    results = tunable.get_results()
    say(what='tunable results', validators=tunable.get_validators(), n=len(results))
    for r in results:
        say(what='solution', entry=list(r))

    # 3. per shape: heuristic against tuned, the calls as the model makes them
    m = tok
    u = int(torch.unique(ids).numel())
    w = lambda n, k: torch.randn(n, k, device=dev, dtype=torch.bfloat16) * 0.02   # noqa: E731
    shapes = [('trunk Wqkv', m, 768, 2304, 'linear'), ('trunk out_proj', m, 768, 768, 'linear'),
              ('trunk fc1 + GELU', m, 768, 3072, 'gelu'), ('trunk fc2', m, 3072, 768, 'linear'),
              ('sense Wqkv', m, 768, 1536, 'linear'), ('lm_head chunk', tune_chunk, 768, cfg.vocab_size, 'mm'),
              ('content fc1 + GELU (distinct ids)', u, 768, 3072, 'gelu'), ('content fc2 (distinct ids)', u, 3072, 768, 'linear'),
              ('sense-net fc2 (distinct ids)', u, 3072, 12288, 'linear')]
    for name, mm, k, n, kind in shapes:
        try:
            x, wt, b = torch.randn(mm, k, device=dev, dtype=torch.bfloat16), w(n, k), torch.zeros(n, device=dev, dtype=torch.bfloat16)
            out = torch.empty(mm, n, device=dev, dtype=torch.bfloat16) if kind == 'mm' else None
            fn = {'linear': lambda: torch.nn.functional.linear(x, wt, b),
                  'gelu': lambda: torch._addmm_activation(b, x, wt.t(), use_gelu=True),
                  'mm': lambda: torch.mm(x, wt.t(), out=out)}[kind]
            tunable.enable(False)
            h = ev_ms(fn)
            tunable.enable(True)
            t = ev_ms(fn)
            fl = 2.0 * mm * k * n
            say(what='shape', gemm=name, m=mm, k=k, n=n, heuristic_ms=round(h, 3), tuned_ms=round(t, 3),
                heuristic_tflops=round(fl / h / 1e9, 1), tuned_tflops=round(fl / t / 1e9, 1))
            del x, wt, b, out
        except Exception as e:   # noqa: BLE001
            say(what='shape failed', gemm=name, error=repr(e)[:300])


if __name__ == '__main__':
    main()

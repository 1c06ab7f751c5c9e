"""Greedy decoding latency of Backpack-Small on the HIP path: the reference's growing-prefix loop (no KV cache upstream)
against generate(..., cg=True), one captured full-width forward replayed per token (src/utils/generation.py).

    python scripts/bench_generate.py [--batch 1] [--prompt 16] [--max-length 128] [--model small]"""
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
    ap.add_argument('--batch', type=int, default=1)
    ap.add_argument('--prompt', type=int, default=16)
    ap.add_argument('--max-length', type=int, default=128)
    ap.add_argument('--model', default='small')
    a = ap.parse_args()
    from bench import MODELS
    from src.models.backpack import BackpackConfig, BackpackLMHeadModel
    dev = torch.device('cuda', 0)
    cfg = BackpackConfig(vocab_size=50257, n_positions=max(a.max_length, 128), scale_attn_by_inverse_layer_idx=True,
                         use_flash_attn=True, fused_bias_fc=True, fused_dense_gelu_dense=True, fused_dropout_add_ln=True,
                         pad_vocab_size_multiple=8, **MODELS[a.model])
    torch.manual_seed(0)
    model = BackpackLMHeadModel(cfg, device=dev, dtype=torch.bfloat16).eval()
    ids = torch.randint(0, 50257, (a.batch, a.prompt), device=dev)
    for mode in ('off', 'cached'):    # content network per position (the reference's order) / cached whole-vocabulary table
        model.transformer.sense_table_mode = mode
        res = dict(model=a.model, batch=a.batch, prompt=a.prompt, max_length=a.max_length,
                   new_tokens=a.max_length - 1 - a.prompt, sense_table=mode)
        outs = {}
        for cg in (False, True):
            model.generate(ids, max_length=a.max_length, cg=cg)      # warm-up (allocator, library handles)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            outs[cg] = model.generate(ids, max_length=a.max_length, cg=cg)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            key = 'graph_replay' if cg else 'eager_loop'
            res[key + '_ms'] = round(dt * 1e3, 1)
            res[key + '_ms_per_token'] = round(dt * 1e3 / res['new_tokens'], 3)
        same = (outs[False] == outs[True]).float().mean().item()
        res['tokens_equal_fraction'] = round(same, 4)     # random weights: near-uniform logits, ties flip easily
        print(json.dumps(res), flush=True)

if __name__ == '__main__':
    main()

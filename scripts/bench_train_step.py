"""BASELINE config 3 (Backpack-Small, seq 1024, bf16, DDP): one training step = forward + fused cross-entropy +
backward (+ gradient all-reduce under DDP) + AdamW step, timed.  NOT the headline bench (that is bench.py, the
forward metric); this script records what the training path costs on the HIP kernels.

    python scripts/bench_train_step.py [--batch 16] [--steps 5] [--model small]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 scripts/bench_train_step.py
Default = the reference's recipe: fp32 parameters under 16-bit autocast (trainer precision 16,
training/configs/trainer/default.yaml + experiment/owt/base.yaml), so DDP all-reduces 170.48 M fp32 gradients
(682 MB), and GPT2Config's default dropout 0.1 (attention, residual, embedding) running inside the HIP kernels.
`--pure-bf16` keeps bf16 parameters instead (half the all-reduce volume; a throughput probe, not the recipe),
`--dropout 0` switches dropout off, `--grad-compress fp16|bf16` registers the reference's gradient-compression comm hook
(training/src/distributed/ddp_comm_hooks.py:9-43; src/distributed/ddp_comm_hooks.py here).

What the line reports about the collective (the only one of the whole path):
  grad_allreduce_bytes   bytes one step's gradient all-reduce moves per rank buffer (fp32 grads, or the 16-bit wire)
  allreduce_ms           one standalone all-reduce of exactly that volume and dtype, median of 5 (0 at world size 1)
  bus_gbps               2 (N-1)/N * bytes / allreduce_ms: what each GPU's links carry (RCCL's "bus bandwidth")
  bound_ring_ms / bound_all_links_ms   the two analytic bounds of SURVEY.md section 5: a ring drives ONE xGMI link per
                         direction (153 GB/s), reduce-scatter + all-gather over the full mesh all N-1 links at once
  ms_per_step_no_sync    the same training step under DDP's no_sync() (no all-reduce at all): ms_per_step minus this is
                         the part of the all-reduce that backward did not hide
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'backpacks-flash-attn_amd')):
    sys.path.insert(0, p)
# the host driver of this pool only supports dmabuf IPC: without it RCCL's cross-process buffer sharing fails with
# `hipIpcGetMemHandle: invalid argument` at world size > 1 (the image exports it; a bare launcher may not)
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=16)
    ap.add_argument('--seq', type=int, default=1024)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--model', default='small')
    ap.add_argument('--dropout', type=float, default=0.1)
    ap.add_argument('--pure-bf16', action='store_true')
    ap.add_argument('--grad-compress', default='none', choices=['none', 'fp16', 'bf16'],
                    help="DDP comm hook: all-reduce the gradient buckets in a 16-bit wire format (reference: "
                         "training/src/distributed/ddp_comm_hooks.py fp16_compress_hook)")
    ap.add_argument('--dist-backend', default='nccl', choices=['nccl', 'gloo'],
                    help="'gloo': the collectives go through host memory and ranks may share a GPU (test-suite only)")
    a = ap.parse_args()
    rank, world = int(os.environ.get('RANK', 0)), int(os.environ.get('WORLD_SIZE', 1))
    local = int(os.environ.get('LOCAL_RANK', 0))
    if a.dist_backend == 'gloo':
        local %= torch.cuda.device_count()
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    ddp = world > 1 or ('RANK' in os.environ and 'MASTER_ADDR' in os.environ)   # under torch.distributed.run
    if ddp:
        import torch.distributed as dist
        if a.dist_backend == 'nccl':
            dist.init_process_group('nccl', device_id=dev)
        else:
            dist.init_process_group('gloo')
    from bench import MODELS
    from flash_attn.losses.cross_entropy import CrossEntropyLoss
    from src.models.backpack import BackpackConfig, BackpackLMHeadModel
    cfg = BackpackConfig(vocab_size=50257, n_positions=a.seq, scale_attn_by_inverse_layer_idx=True,
                         use_flash_attn=True, fused_bias_fc=True, fused_dense_gelu_dense=True,
                         fused_dropout_add_ln=True, pad_vocab_size_multiple=8, resid_pdrop=a.dropout,
                         embd_pdrop=a.dropout, attn_pdrop=a.dropout, **MODELS[a.model])
    torch.manual_seed(0)
    model = BackpackLMHeadModel(cfg, device=dev, dtype=torch.bfloat16 if a.pure_bf16 else torch.float32).train()
    net = model
    if ddp:
        net = torch.nn.parallel.DistributedDataParallel(model, device_ids=[local], gradient_as_bucket_view=True,
                                                        find_unused_parameters=False)
        from src.distributed.ddp_comm_hooks import HOOKS
        if HOOKS[a.grad_compress] is not None:
            net.register_comm_hook(None, HOOKS[a.grad_compress])
    opt = torch.optim.AdamW(model.parameters(), lr=1e-4, fused=True)
    loss_fn = CrossEntropyLoss(inplace_backward=True)
    ids = torch.randint(0, 50257, (a.batch, a.seq), device=dev, generator=torch.Generator(device=dev).manual_seed(rank))
    labels = torch.roll(ids, -1, 1)

    import contextlib

    def step(sync=True):
        opt.zero_grad(set_to_none=True)
        with (contextlib.nullcontext() if (sync or not ddp) else net.no_sync()):
            with torch.autocast('cuda', dtype=torch.bfloat16, enabled=not a.pure_bf16):
                logits = net(ids).logits
            loss = loss_fn(logits.view(-1, logits.shape[-1]), labels.view(-1))
            loss.backward()
        opt.step()
        return loss

    def timed(n, sync=True):
        torch.cuda.synchronize()
        if ddp:
            dist.barrier()
        t0 = time.perf_counter()
        for _ in range(n):
            out = step(sync)
        torch.cuda.synchronize()
        if ddp:
            dist.barrier()
        dt = time.perf_counter() - t0
        if ddp:   # the slowest rank's clock, as bench.py reports
            t = torch.tensor([dt], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt, out

    for _ in range(a.warmup):
        step()
    dt, loss = timed(a.steps)
    dt_nosync = None
    if ddp:
        step(sync=False)
        dt_nosync, _ = timed(a.steps, sync=False)

    # the collective on its own: one all-reduce of the step's gradient volume in the wire dtype
    grad_elems = sum(p.numel() for p in model.parameters())
    wire_dtype = {'none': next(model.parameters()).dtype, 'fp16': torch.float16, 'bf16': torch.bfloat16}[a.grad_compress]
    wire_bytes = grad_elems * torch.empty((), dtype=wire_dtype).element_size()
    allreduce_ms = 0.0
    if ddp and world > 1:
        buf = torch.zeros(grad_elems, dtype=wire_dtype, device=dev)
        samples = []
        for i in range(7):
            torch.cuda.synchronize()
            dist.barrier()
            t0 = time.perf_counter()
            dist.all_reduce(buf)
            torch.cuda.synchronize()
            samples.append(time.perf_counter() - t0)
        t = torch.tensor([sorted(samples[2:])[2]], device=dev, dtype=torch.float64)   # median of the last five
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        allreduce_ms = float(t.item()) * 1e3
        del buf
    link_gbps = 153.0   # one xGMI link, one direction
    ring_ms = 2 * (world - 1) / world * wire_bytes / (link_gbps * 1e9) * 1e3 if world > 1 else 0.0
    mesh_ms = ring_ms / (world - 1) if world > 1 else 0.0
    if rank == 0:
        print(json.dumps({'metric': f'tokens/sec train step (fwd+loss+bwd+AdamW), Backpack-{a.model} seq={a.seq}',
                          'value': round(world * a.batch * a.seq * a.steps / dt, 1), 'unit': 'tokens/s',
                          'n_gpus': world, 'ms_per_step': round(dt / a.steps * 1e3, 2), 'batch_per_gpu': a.batch,
                          'dtype': 'bf16' if a.pure_bf16 else 'bf16 autocast over fp32 parameters', 'dropout': a.dropout,
                          'grad_allreduce_bytes': wire_bytes, 'grad_compress': a.grad_compress,
                          'allreduce_ms': round(allreduce_ms, 3),
                          'bus_gbps': round(2 * (world - 1) / world * wire_bytes / (allreduce_ms * 1e-3) / 1e9, 1) if allreduce_ms else 0.0,
                          'bound_ring_ms': round(ring_ms, 3), 'bound_all_links_ms': round(mesh_ms, 3),
                          'ms_per_step_no_sync': round(dt_nosync / a.steps * 1e3, 2) if dt_nosync else None,
                          'loss': round(float(loss.detach()), 4),
                          'peak_mem_gb': round(torch.cuda.max_memory_allocated() / 2 ** 30, 1),
                          'launch': ('torch.distributed.run, DDP over ' + ('nccl (RCCL)' if a.dist_backend == 'nccl' else 'gloo')) if ddp else 'single process'}))
    if ddp:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()

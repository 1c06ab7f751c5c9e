"""CPU, world_size 2, gloo: the multi-GPU paths of the Backpack hot path.
  * forward = independent batch replicas: ranks hold identical weights, different batches, no
    data-path collective; the bench aggregates with a MAX all-reduce of the elapsed time.
  * training (BASELINE config 3) = torch DDP gradient all-reduce -- the one collective of the
    reference (training/src/train.py:93-102).  On the GPU box the same code runs on 'nccl' (= RCCL)."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import PKG, ROOT


def _worker(rank, world, port, tmp):
    for p in (ROOT, PKG):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from src.models.backpack import BackpackConfig, BackpackLMHeadModel
        cfg = BackpackConfig(n_embd=64, n_head=2, n_layer=2, num_content_vectors=4, vocab_size=96,
                             n_positions=32, scale_attn_by_inverse_layer_idx=True, resid_pdrop=0.0,
                             embd_pdrop=0.0, attn_pdrop=0.0)
        torch.manual_seed(0)                       # same weights on every rank
        model = BackpackLMHeadModel(cfg)
        ids = torch.randint(0, 96, (2, 32), generator=torch.Generator().manual_seed(100 + rank))

        # --- replicas: forward needs no collective; outputs differ per rank, weights agree -------
        with torch.no_grad():
            logits = model(ids).logits
        gathered = [torch.zeros_like(logits) for _ in range(world)]
        dist.all_gather(gathered, logits)
        assert not torch.equal(gathered[0], gathered[1])
        w = model.transformer.contextualization_attn.Wqkv.weight.detach().clone()
        ws = [torch.zeros_like(w) for _ in range(world)]
        dist.all_gather(ws, w)
        assert torch.equal(ws[0], ws[1])

        # --- bench aggregation: MAX over ranks of the elapsed time ----------------------------------
        t = torch.tensor([1.0 + rank], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        assert t.item() == float(world)

        # --- DDP step: gradient all-reduce = mean of the per-rank gradients -------------------------
        local = BackpackLMHeadModel(cfg)
        local.load_state_dict(model.state_dict())
        ddp = torch.nn.parallel.DistributedDataParallel(model, find_unused_parameters=False,
                                                        gradient_as_bucket_view=True)
        loss = torch.nn.functional.cross_entropy(ddp(ids).logits.flatten(0, 1), ids.flatten())
        loss.backward()
        loss_l = torch.nn.functional.cross_entropy(local(ids).logits.flatten(0, 1), ids.flatten())
        loss_l.backward()
        for (n, p), (_, q) in zip(model.named_parameters(), local.named_parameters()):
            mine = q.grad.clone()
            dist.all_reduce(mine)
            assert torch.allclose(p.grad, mine / world, atol=1e-6), n
        # --- the reference's gradient-compression comm hook (training/src/distributed/ddp_comm_hooks.py:9-43): the
        #     bucket is divided by the world size, cast to 16 bit, summed, copied back -> the fp32 mean to 16-bit precision
        from src.distributed.ddp_comm_hooks import HOOKS, bf16_compress_hook, fp16_compress_hook
        assert HOOKS == {'none': None, 'fp16': fp16_compress_hook, 'bf16': bf16_compress_hook}
        for hook, rel in ((fp16_compress_hook, 2e-3), (bf16_compress_hook, 1.6e-2)):
            third = BackpackLMHeadModel(cfg)
            third.load_state_dict(local.state_dict())
            ddp3 = torch.nn.parallel.DistributedDataParallel(third, find_unused_parameters=False,
                                                             gradient_as_bucket_view=True)
            ddp3.register_comm_hook(None, hook)
            torch.nn.functional.cross_entropy(ddp3(ids).logits.flatten(0, 1), ids.flatten()).backward()
            for (n, p), (_, q) in zip(model.named_parameters(), third.named_parameters()):
                assert q.grad.dtype == torch.float32
                assert (q.grad - p.grad).abs().max() <= rel * p.grad.abs().max() + 1e-7, (hook.__name__, n)
        with open(os.path.join(tmp, f'ok{rank}'), 'w') as f:
            f.write('ok')
    finally:
        dist.destroy_process_group()


def test_replicas_and_ddp_allreduce_world2(tmp_path):
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / 'ok0').exists() and (tmp_path / 'ok1').exists()

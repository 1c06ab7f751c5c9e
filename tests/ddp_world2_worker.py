"""Worker of tests/test_gpu_world2.py, launched as `python -m torch.distributed.run --nproc-per-node 2`.

BASELINE config 3 (DDP gradient all-reduce, reference training/src/train.py:93-102) with TWO real ranks on the ONE leased
GPU: both ranks use cuda:0 and the process group is `gloo` (RCCL refuses two ranks on one device), so everything except
the wire is what the 8-GPU run executes -- DDP's bucket hooks firing over the custom autograd Functions of the HIP path
(flash attention, sense mix, fused add+LayerNorm, fused dense, fused cross-entropy), gradient_as_bucket_view, the
reference's flags.  Checks, all asserted here (a non-zero exit fails the test):
  * every rank ends with bit-identical gradients;
  * they equal the mean of the two ranks' single-process gradients (the definition of DDP's all-reduce);
  * and match the gradient of ONE process over the concatenated batch within 16-bit accumulation noise.
Prints one JSON line from rank 0."""
import json
import os
import sys

import torch
import torch.distributed as dist
from torch.nn.parallel import DistributedDataParallel as DDP

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'backpacks-flash-attn_amd')):
    if p not in sys.path:
        sys.path.insert(0, p)


def main():
    rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
    assert world == 2
    torch.cuda.set_device(0)
    dev = torch.device('cuda', 0)
    dist.init_process_group('gloo')
    import bp_hip
    bp_hip.lib()
    from flash_attn.losses.cross_entropy import CrossEntropyLoss
    from src.models.backpack import BackpackConfig, BackpackLMHeadModel
    kw = dict(n_embd=128, n_head=2, n_layer=2, num_content_vectors=4, vocab_size=512, n_positions=128,
              scale_attn_by_inverse_layer_idx=True, resid_pdrop=0.0, embd_pdrop=0.0, attn_pdrop=0.0,
              pad_vocab_size_multiple=8, use_flash_attn=True, fused_dropout_add_ln=True, fused_bias_fc=True,
              fused_dense_gelu_dense=True)

    def fresh():
        torch.manual_seed(3)                       # the same weights everywhere
        return BackpackLMHeadModel(BackpackConfig(**kw)).to(dev, torch.bfloat16)

    gen = torch.Generator().manual_seed(77)       # the same global batch everywhere; rank r takes rows 4r .. 4r+3
    ids_all = torch.randint(0, 512, (8, 128), generator=gen).to(dev)
    labels_all = torch.randint(0, 512, (8, 128), generator=gen).to(dev)
    loss_fn = CrossEntropyLoss()

    def grads_of(model, rows):
        loss = loss_fn(model(ids_all[rows]).logits.flatten(0, 1), labels_all[rows].flatten())
        loss.backward()
        return loss.detach().float()

    mine = slice(4 * rank, 4 * rank + 4)
    wrapped = fresh()
    ddp = DDP(wrapped, device_ids=[0], find_unused_parameters=False, gradient_as_bucket_view=True)
    loss = grads_of(ddp, mine)
    torch.cuda.synchronize()
    names = [n for n, _ in wrapped.named_parameters()]
    got = [p.grad.detach().clone() for p in wrapped.parameters()]
    assert all(g is not None for g in got) and len(got) > 20

    # (1) bit-identical on both ranks
    for n, g in zip(names, got):
        both = [torch.empty_like(g) for _ in range(world)]
        dist.all_gather(both, g)
        assert torch.equal(both[0], both[1]), n

    # (2) = mean of the per-rank single-process gradients; (3) ~ one process over the whole batch
    halves = []
    for r in range(world):
        m = fresh()
        grads_of(m, slice(4 * r, 4 * r + 4))
        halves.append([p.grad.detach().float() for p in m.parameters()])
    whole_model = fresh()
    whole_loss = grads_of(whole_model, slice(0, 8))
    worst_mean = worst_whole = 0.0
    for n, g, a, b, w in zip(names, got, halves[0], halves[1], whole_model.parameters()):
        mean = (a + b) / 2
        scale = mean.abs().max().item() + 1e-12
        e_mean = (g.float() - mean).abs().max().item() / scale
        e_whole = (g.float() - w.grad.float()).abs().max().item() / scale
        worst_mean, worst_whole = max(worst_mean, e_mean), max(worst_whole, e_whole)
        # gloo sums the two bf16 buckets and DDP divides: one or two bf16 roundings of the mean
        assert e_mean <= 2 ** -7, (n, e_mean)
        # the whole-batch weight-gradient GEMMs accumulate 1024 rows in one fp32 sum instead of two rounded halves
        assert e_whole <= 2 ** -5, (n, e_whole)
    losses = [torch.empty_like(loss) for _ in range(world)]
    dist.all_gather(losses, loss)
    mean_loss = float((losses[0] + losses[1]) / 2)
    assert abs(mean_loss - float(whole_loss)) <= 2e-2 * abs(float(whole_loss)), (mean_loss, float(whole_loss))
    dist.barrier()
    if rank == 0:
        print(json.dumps(dict(world=world, backend='gloo', params=len(got), worst_vs_mean_of_ranks=worst_mean,
                              worst_vs_whole_batch=worst_whole, loss=mean_loss, whole_batch_loss=float(whole_loss))))
    dist.destroy_process_group()


if __name__ == '__main__':
    main()

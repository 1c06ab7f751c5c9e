"""Which of dq / dk / dv is wrong for a given (S, d, causal, dropout) -- against the fp32 oracle with the documented mask."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, 'backpacks-flash-attn_amd'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
import torch
import bp_hip as bp
import philox_ref as P
from oracle import ref_cpu as R
DEV = 'cuda'
for (S, d, causal, p_drop) in ((200, 80, True, 0.17), (200, 80, True, 0.0), (200, 80, False, 0.17), (190, 80, True, 0.17),
                               (256, 80, True, 0.17), (200, 96, True, 0.17), (200, 112, True, 0.17), (200, 128, True, 0.17), (200, 48, True, 0.17)):
    torch.manual_seed(0)
    b, h = 2, 2
    x = torch.randn(b, S, 3, h, d).bfloat16()
    g = torch.randn(b, S, h, d).bfloat16()
    qkv = x.to(DEV).flatten(0, 1)
    out = torch.empty_like(qkv[:, 0])
    rng = torch.tensor([5, 9], dtype=torch.int64, device=DEV)
    scale = d ** -0.5
    lse = bp.flash_fwd(qkv[:, 0], qkv[:, 1], qkv[:, 2], out, None, None, S, S, scale, causal, p_drop, rng if p_drop else None)
    dqkv = torch.full_like(qkv, float('nan'))
    bp.flash_bwd(g.to(DEV).flatten(0, 1), qkv[:, 0], qkv[:, 1], qkv[:, 2], out, lse, dqkv[:, 0], dqkv[:, 1], dqkv[:, 2], None, None,
                 S, S, scale, causal, p_drop, rng if p_drop else None)
    keep = torch.from_numpy(P.attention_keep_mask(5, 9, b, h, S, S, p_drop)) if p_drop else None
    t = x.float().requires_grad_()
    o = R.attention_fp32(t[:, :, 0], t[:, :, 1], t[:, :, 2], causal=causal, softmax_scale=scale, dropout_p=p_drop, dropout_mask=keep)[0]
    go, = torch.autograd.grad(o, t, g.float())
    got = dqkv.float().cpu().reshape(b, S, 3, h, d)
    errs = [(got[:, :, i] - go[:, :, i]).abs().max().item() for i in range(3)]
    eo = (out.float().cpu().reshape(b, S, h, d) - o).abs().max().item()
    print(f'S={S} d={d} causal={causal} p={p_drop}: out {eo:.3e}  dq {errs[0]:.3e}  dk {errs[1]:.3e}  dv {errs[2]:.3e}', flush=True)
    if max(errs) > 0.2:
        bad = (got - go).abs()
        for i, nm in enumerate(('dq', 'dk', 'dv')):
            bi = bad[:, :, i]
            if bi.max() > 0.2:
                rows = (bi.amax(dim=(2, 3)) > 0.2).nonzero()
                cols = (bi.amax(dim=(0, 1, 2)) > 0.2).nonzero().flatten().tolist()
                print('   ', nm, 'bad rows (batch,row) first/last', rows[0].tolist(), rows[-1].tolist(), 'n', len(rows), 'bad d cols', cols[:6], '...', cols[-3:])

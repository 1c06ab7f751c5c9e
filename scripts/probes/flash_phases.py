"""Per-phase cycle breakdown of flash_fwd_dma_kernel (needs the BP_PROFILE_PHASES build):
   BP_HIP_LIB=.../libbackpack_hip_prof.so python scripts/probes/flash_phases.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'backpacks-flash-attn_amd'))
import torch
B, S, H, D = 64, 1024, 12, 64
nwg = ((B * H + 7) // 8) * 8 * (S // 128)
prof = torch.zeros(nwg * 4 * 8, dtype=torch.int64, device='cuda')
os.environ['BP_PROF_PTR'] = str(prof.data_ptr())
import bp_hip
qkv = torch.randn(B * S, 3, H, D, device='cuda').bfloat16()
out = torch.empty_like(qkv[:, 0])
cu = torch.arange(0, (B + 1) * S, S, dtype=torch.int32, device='cuda')
for _ in range(3):
    bp_hip.flash_fwd(qkv[:, 0], qkv[:, 1], qkv[:, 2], out, cu, cu, S, S, D ** -0.5, True)
torch.cuda.synchronize()
p = prof.view(-1, 8).cpu().double()
p = p[p[:, 6] > 0]
names = ['wait+barrier+issue', 'QK^T MFMA (to first softmax op)', 'softmax VALU', '(unused)', 'PV MFMA + rest']
print('waves', len(p))
for qt in (0, 3, 7):
    sel = p[p[:, 7] == qt]
    nkb = sel[0, 6].item()
    print(f'qt={qt} nkb={nkb:.0f}: total cycles/wave {sel[:,5].mean():.0f}  per-iteration {sel[:,5].mean()/nkb:.0f}')
    for i in (0, 1, 2, 4):
        print(f'    {names[i]:36s} {sel[:, i].mean() / nkb:8.0f} cycles/iteration')
    print(f'    prologue+epilogue                    {(sel[:,5] - sel[:,:5].sum(1)).mean():8.0f} cycles/wave')

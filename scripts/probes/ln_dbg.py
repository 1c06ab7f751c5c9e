import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'backpacks-flash-attn_amd'))
import torch, bp_hip
from oracle import ref_cpu as R
torch.manual_seed(0)
cols = 8
x0 = torch.arange(cols).float().reshape(1, cols).bfloat16()
x1 = (100 * torch.arange(1, cols + 1)).float().reshape(1, cols)
w = torch.ones(cols).bfloat16(); b = torch.zeros(cols).bfloat16()
z, x = bp_hip.add_layer_norm(x0.cuda(), x1.cuda(), w.cuda(), b.cuda(), 1e-5)
print('x out', x.cpu())
print('x ref', (x0.float() + x1))
print('z', z.float().cpu()); print('z ref', R.add_layer_norm_fp32(x0, x1, w, b, 1e-5)[0].float())

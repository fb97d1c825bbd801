import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mgsv_amd import ops
dev = torch.device("cuda"); dt = torch.bfloat16
B, H, D, L = 64, 8, 512, 542
q = torch.randn(B, 1, H * D, device=dev).to(dt); mem = torch.randn(B, L, D, device=dev).to(dt); mp = torch.randn(B, L, D, device=dev).to(dt)
o = torch.empty(B, 1, H * D, device=dev, dtype=dt)
q4 = q.view(B, 1, H, D).permute(0, 2, 1, 3); o4 = o.view(B, 1, H, D).permute(0, 2, 1, 3)
for _ in range(5):
    ops.attention_wide(q4, mp, mem, o4, scale=0.125, key_mask=None, n_split=1)
torch.cuda.synchronize()

import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mgsv_amd import ops
dev = torch.device("cuda"); dt = torch.bfloat16
B, H, hd, L = 64, 8, 64, 542
D = H * hd
qkv = torch.randn(B, L, 3 * D, device=dev).to(dt)
o = torch.empty(B, L, D, device=dev, dtype=dt)
km = torch.ones(B, L, device=dev)
for _ in range(5):
    ops.attention(qkv[:, :, :D], qkv[:, :, D:2 * D], qkv[:, :, 2 * D:], o, H, key_mask=km)
torch.cuda.synchronize()

import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mgsv_amd import ops
dev = torch.device("cuda"); dt = torch.bfloat16
B, NQ, L, D = 1, 32, 32, 512
# one-hot attention: query i strongly matches key (i*5 % 32)
K = torch.zeros(B, L, D); Q = torch.zeros(1, NQ, 1, D)
for t in range(L): K[0, t, t] = 1.0
for i in range(NQ): Q[0, i, 0, (i * 5) % 32] = 100.0
V = torch.zeros(B, L, D)
for t in range(L): V[0, t] = t + torch.arange(D) / 1024.0
O = torch.empty(B, NQ, 1, D, device=dev, dtype=torch.float32)
ops.attention_wide(Q.to(dev).to(dt), K.to(dev).to(dt), V.to(dev).to(dt), O, scale=1.0, shared_q=True)
torch.cuda.synchronize()
o = O.cpu()[0, :, 0]
print("expected key per query:", [(i * 5) % 32 for i in range(8)])
print("got integer part at d=0:", o[:8, 0].tolist())
print("row 0, d=0..7 frac*1024:", ((o[0, :8] - o[0, :8].floor()) * 1024).tolist())
print("row 0, d=16..23 frac*1024:", ((o[0, 16:24] - o[0, 16:24].floor()) * 1024).tolist())
print("row 0, d=128..135:", ((o[0, 128:136] - o[0, 128:136].floor()) * 1024).tolist())
print("int part row0 over d (first 40):", o[0, :40].floor().tolist())

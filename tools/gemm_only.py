import math, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mgsv_amd import ops
dev = torch.device("cuda"); dt = torch.bfloat16
M, N, K = 32768, 1536, 512
A = torch.randn(M, K, device=dev).to(dt); W = (torch.randn(N, K, device=dev) / math.sqrt(K)).to(dt)
b = torch.randn(N, device=dev); out = torch.empty(M, N, device=dev, dtype=dt)
for _ in range(5):
    ops.linear(A, W, b, out=out)
torch.cuda.synchronize()

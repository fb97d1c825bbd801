"""made_pool_bwd alone at the headline step's shape (64 x 512 tokens x 512, bf16 in / out)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from mgsv_amd import ops_train as tr
B, T, D = 64, 512, 512
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
mean = torch.randn(B, D, device=dev, generator=g); dvec = torch.randn(B, D, device=dev, generator=g)
lens = torch.randint(T // 8, T + 1, (B,), device=dev, generator=g)
mask = (torch.arange(T, device=dev)[None] < lens[:, None]).float()
in1 = torch.randn(B, T, D, device=dev, generator=g).bfloat16(); in2 = torch.randn(B, T, D, device=dev, generator=g).bfloat16()
out = torch.empty(B, T, D, device=dev, dtype=torch.bfloat16)
for _ in range(5): tr.pool_bwd(mean, dvec, mask, out, in1, in2)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(200): tr.pool_bwd(mean, dvec, mask, out, in1, in2)
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 5
byts = float(mask.sum()) * D * 2 * 2 + B * T * D * 2
print(f"made_pool_bwd {B}x{T}x{D}: {us:.1f} us, {byts / us / 1e3:.0f} GB/s of algorithmic bytes; checksum {float(out.float().sum()):.6e} {float(out.float().abs().sum()):.6e}")

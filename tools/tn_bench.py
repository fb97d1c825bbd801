"""Microbenchmark of made_gemm_tn on the training step's shapes (GPU box)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mgsv_amd import ops_train as tr

def bench(M, N, K, split=None, dtype=torch.bfloat16, mask=True, iters=20):
    A = torch.randn(M, N, device="cuda").to(dtype); B = torch.randn(M, K, device="cuda").to(dtype)
    C = torch.zeros(N, K, device="cuda")
    cs = torch.zeros(N, device="cuda")
    m = (torch.rand(M, device="cuda") > 0.29).float() if mask else None
    for _ in range(3): tr.gemm_tn(A, B, C, accumulate=True, colsum=cs, row_mask=m, split_m=split)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): tr.gemm_tn(A, B, C, accumulate=True, colsum=cs, row_mask=m, split_m=split)
    e.record(); torch.cuda.synchronize()
    us = s.elapsed_time(e) / iters * 1e3
    print(f"M={M} N={N} K={K} split={split}: {us:.1f} us  {2.0*M*N*K/us/1e6:.1f} TFLOP/s")

for split in (None, 8, 16, 32, 64):
    bench(34688, 512, 512, split)
bench(34688, 1536, 512); bench(34688, 1024, 512); bench(34688, 512, 1024); bench(34688, 1024, 512, 16); bench(34688, 1536, 512, 8)
bench(32768, 512, 768); bench(1920, 512, 512)
print("-- no mask (direct-to-LDS kernel)")
for sp in (None, 8, 16, 32):
    bench(34688, 512, 512, sp, mask=False)
bench(34688, 1536, 512, mask=False); bench(34688, 1024, 512, mask=False); bench(34688, 512, 1024, mask=False); bench(18700, 512, 512, mask=False); bench(18700, 1024, 512, mask=False)

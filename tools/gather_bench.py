"""Micro-benchmark of made_linear with the row gather at the DETR-encoder shapes (GPU box): M = 64 x 542 rows of which the
valid ones (lengths as bench.py draws them) are computed.  MADE_LINEAR_TILE=64|128 forces the tile height."""
import math, os, sys
os.environ.setdefault("MADE_DEBUG_VARIANTS", "1")          # (measurement knobs are honoured only under this switch)
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mgsv_amd import ops

def timeit(fn, iters=20, warm=3):
    for _ in range(warm): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3

B, L = 64, 542
g = torch.Generator().manual_seed(0)
lv, la = torch.randint(5, 31, (B,), generator=g), torch.randint(12, 513, (B,), generator=g)
pos = torch.arange(L)[None]
mask = ((pos < lv[:, None]) | ((pos >= 30) & (pos < 30 + la[:, None]))).float().cuda()
rows = ops.row_index(mask)
nv = int(rows[1])
M = B * L
print(f"tile={os.environ.get('MADE_LINEAR_TILE', 'auto')} valid rows {nv} of {M}")
for N, K in ((512, 512), (1024, 512), (1536, 512), (512, 1024)):
    A = torch.randn(M, K, device="cuda").bfloat16(); W = (torch.randn(N, K, device="cuda") / math.sqrt(K)).bfloat16()
    b = torch.randn(N, device="cuda"); out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16); R = torch.randn(M, N, device="cuda").bfloat16()
    td = timeit(lambda: ops.linear(A, W, b, out=out, act=ops.ACT_RELU, R=R))
    tg = timeit(lambda: ops.linear(A, W, b, out=out, act=ops.ACT_RELU, R=R, rows=rows))
    print(f"N={N:5d} K={K:5d}: dense {td:6.1f} us {2.0 * M * N * K / td / 1e6:6.1f} TF | gathered {tg:6.1f} us {2.0 * nv * N * K / tg / 1e6:6.1f} TF")

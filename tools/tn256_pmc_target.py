"""Eager launches of one encoder layer's grouped weight-gradient call (tools/tn256_bench.py's shapes) for rocprofv3 --pmc (no hipGraph)."""
import os, sys, torch
os.environ.setdefault("MADE_DEBUG_VARIANTS", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mgsv_amd import ops, ops_train as tr
dev = "cuda"
B, L = 64, 542
M = B * L
g = torch.Generator().manual_seed(0)
lv, la = torch.randint(5, 31, (B,), generator=g), torch.randint(12, 513, (B,), generator=g)
pos = torch.arange(L)[None]
mask = ((pos < lv[:, None]) | ((pos >= 30) & (pos < 30 + la[:, None]))).float().to(dev)
rows = ops.row_index(mask)
mk = lambda n: torch.randn(M, n, device=dev).bfloat16()
probs = [(mk(N), mk(K), torch.zeros(N, K, device=dev), torch.zeros(N, device=dev)) for N, K in ((512, 1024), (1024, 512), (512, 512), (1024, 512), (512, 512))]
ws = {}
def workspace(n):
    if "w" not in ws: ws["w"] = tr.gemm_tn_grouped_workspace(torch.device(dev), n)
    return ws["w"]
for _ in range(6):
    tr.gemm_tn_grouped(probs, rows=rows, workspace=workspace)
torch.cuda.synchronize()

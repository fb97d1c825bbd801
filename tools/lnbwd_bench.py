"""made_layernorm_bwd at the DETR-encoder shape (34688 token rows, 54 % valid, D = 512, bf16) for rocprofv3 --kernel-trace;
MADE_LNBWD_NB caps the grid (the per-column gradient flush is one atomic per workgroup and column)."""
import os, sys, torch
os.environ.setdefault("MADE_DEBUG_VARIANTS", "1")          # (measurement knobs are honoured only under this switch)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mgsv_amd import ops_train as tr
dev = "cuda"
B, L, D = 64, 542, 512
g = torch.Generator().manual_seed(0)
lv, la = torch.randint(5, 31, (B,), generator=g), torch.randint(12, 513, (B,), generator=g)
pos = torch.arange(L)[None]
mask = ((pos < lv[:, None]) | ((pos >= 30) & (pos < 30 + la[:, None]))).float().to(dev).reshape(-1)
x = torch.randn(B * L, D, device=dev).bfloat16(); dy = torch.randn(B * L, D, device=dev).bfloat16(); dx = torch.empty_like(x)
gamma = torch.ones(D, device=dev); dg = torch.zeros(D, device=dev); db = torch.zeros(D, device=dev)
for _ in range(8):
    tr.layernorm_bwd(x, gamma, dy, dx, dgamma=dg, dbeta=db, row_skip=mask)
torch.cuda.synchronize()

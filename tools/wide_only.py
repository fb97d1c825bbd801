"""Runs made_attention_wide alone at the in-batch X-Pool shape (for rocprofv3 --pmc passes)."""
import math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mgsv_amd import ops
Nv, Nm, S, D = 64, 64, 512, 512
ns = int(sys.argv[1]) if len(sys.argv) > 1 else 2
dev, dt = "cuda", torch.bfloat16
q = torch.randn(Nv, D, device=dev).to(dt); k = torch.randn(Nm, S, D, device=dev).to(dt); u = torch.randn(Nm, S, D, device=dev).to(dt)
o = torch.empty(Nm * Nv, D, device=dev, dtype=dt)
for _ in range(6):
    ops.attention_wide(q.view(1, Nv, 1, D), k, u, o.view(Nm, Nv, 1, D), scale=1 / math.sqrt(D), key_mask=None, shared_q=True, n_split=ns)
torch.cuda.synchronize()

"""Runs made_xpool_fused alone (for rocprofv3 --pmc passes): Nv x Nm pairs, S segments, D = 256, bf16."""
import math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mgsv_amd import ops

Nv, Nm, S, D = int(sys.argv[1]) if len(sys.argv) > 1 else 8192, int(sys.argv[2]) if len(sys.argv) > 2 else 256, int(sys.argv[3]) if len(sys.argv) > 3 else 96, 256
FULL = len(sys.argv) > 4 and sys.argv[4] == "full"        # every track has all S segments
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
Q = torch.randn(Nv, D, device=dev, generator=g).bfloat16()
K = torch.randn(Nm, S, D, device=dev, generator=g).bfloat16()
U = torch.randn(Nm, S, D, device=dev, generator=g).bfloat16()
lens = torch.randint(min(12, S), S + 1, (Nm,), device=dev, generator=g)
if FULL:
    lens[:] = S
mask = (torch.arange(S, device=dev)[None] < lens[:, None]).float()
Wl = (torch.randn(D, D, device=dev, generator=g) / math.sqrt(D)).bfloat16()
vec = lambda: torch.randn(D, device=dev, generator=g) * 0.1
ln2, ln3, bl = (1 + vec(), vec()), (1 + vec(), vec()), vec()
vn = torch.nn.functional.normalize(torch.randn(Nv, D, device=dev, generator=g), dim=-1)
sims = torch.empty(Nv, Nm, device=dev)
for _ in range(3):
    ops.xpool_fused(Q, K, U, mask, ln2, Wl, bl, ln3, vn, sims, scale=1 / math.sqrt(D))
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(5):
    ops.xpool_fused(Q, K, U, mask, ln2, Wl, bl, ln3, vn, sims, scale=1 / math.sqrt(D))
e.record(); torch.cuda.synchronize()
ms = s.elapsed_time(e) / 5
blocks = ((Nv + 127) // 128) * Nm
print(f"Nv={Nv} Nm={Nm} S={S}{' full' if FULL else ''}: {ms * 1e3:.1f} us, {blocks} workgroups, {ms * 1e3 / (blocks / 256):.2f} us per workgroup slot, "
      f"{2.0 * Nv * Nm * (2 * S * D + D * D) / ms / 1e9:.1f} TFLOP/s")

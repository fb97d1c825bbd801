"""made_xpool_sims (the per-pair Linear moved onto the values) against made_xpool_fused at retrieval scale: Nv x Nm pairs, S segments
(lengths U{12..S} unless 'full'), D = 256, bf16; the u'' = W'' u GEMM is timed with the new kernel.
    python tools/xpool_sims_bench.py [Nv Nm S [full]]"""
import math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mgsv_amd import ops

Nv, Nm, S, D = int(sys.argv[1]) if len(sys.argv) > 1 else 53000, int(sys.argv[2]) if len(sys.argv) > 2 else 4000, int(sys.argv[3]) if len(sys.argv) > 3 else 96, 256
FULL = len(sys.argv) > 4 and sys.argv[4] == "full"
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
Q = torch.randn(Nv, D, device=dev, generator=g).bfloat16()
K = torch.randn(Nm, S, D, device=dev, generator=g).bfloat16()
UU = torch.zeros(Nm, S, 2 * D, device=dev, dtype=torch.bfloat16)
UU[..., :D] = torch.randn(Nm, S, D, device=dev, generator=g).bfloat16()
U = UU[..., :D].contiguous()
lens = torch.randint(min(12, S), S + 1, (Nm,), device=dev, generator=g)
if FULL:
    lens[:] = S
mask = (torch.arange(S, device=dev)[None] < lens[:, None]).float()
Wl = (torch.randn(D, D, device=dev, generator=g) / math.sqrt(D)).bfloat16()
vec = lambda: torch.randn(D, device=dev, generator=g) * 0.1
ln2, ln3, bl = (1 + vec(), vec()), (1 + vec(), vec()), vec()
vn = torch.nn.functional.normalize(torch.randn(Nv, D, device=dev, generator=g), dim=-1)
W64 = Wl.double() + torch.eye(D, dtype=torch.float64, device=dev)
W2 = (W64 * ln2[0].double()[None, :]).float().bfloat16()
av = (W64 @ ln2[1].double() + bl.double()).float()
bv = W2.double().sum(1).float()
sf, ss = torch.empty(Nv, Nm, device=dev), torch.empty(Nv, Nm, device=dev)
scale = 1 / math.sqrt(D)
skip = mask.reshape(-1)

def fused():
    ops.xpool_fused(Q, K, U, mask, ln2, Wl, bl, ln3, vn, sf, scale=scale)

def new():
    ops.linear(UU.view(Nm * S, 2 * D)[:, :D], W2, None, out=UU.view(Nm * S, 2 * D)[:, D:], tile_skip_mask=skip)
    ops.xpool_sims(Q, K, UU, mask, av, bv, ln3, vn, ss, scale=scale)

def timeit(fn, n=3):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n

for rep in range(2):
    tf, tn = timeit(fused), timeit(new)
    fl = 2.0 * Nv * Nm * (2 * S * D + D * D)
    print(f"Nv={Nv} Nm={Nm} S={S}{' full' if FULL else ''}: made_xpool_fused {tf:8.2f} ms ({fl / tf / 1e9:6.0f} TFLOP/s of the reference's work)   "
          f"u'' GEMM + made_xpool_sims {tn:8.2f} ms ({fl / tn / 1e9:6.0f})   ratio {tf / tn:.2f}", flush=True)
d = (sf - ss).abs()
print(f"max |fused - sims| = {float(d.max()):.4f}, mean {float(d.mean()):.5f}; nan: {int(torch.isnan(ss).sum())} / {int(torch.isnan(sf).sum())}")

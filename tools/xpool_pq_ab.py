"""made_xpool_sims: the 64-video kernels (MADE_XPOOL_SIMS_PQ=64: four waves of 256 registers; 648: eight waves of <= 128) against the 32-video
kernel (the default) on the retrieval set, alternating, plus their differences.  python tools/xpool_pq_ab.py [Nv Nm S]"""
import math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mgsv_amd import ops
Nv, Nm, S, D = int(sys.argv[1]) if len(sys.argv) > 1 else 53000, int(sys.argv[2]) if len(sys.argv) > 2 else 4000, int(sys.argv[3]) if len(sys.argv) > 3 else 96, 256
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
Q = torch.randn(Nv, D, device=dev, generator=g).bfloat16()
K = torch.randn(Nm, S, D, device=dev, generator=g).bfloat16()
UU = torch.randn(Nm, S, 2 * D, device=dev, generator=g).bfloat16()
lens = torch.randint(min(12, S), S + 1, (Nm,), device=dev, generator=g)
mask = (torch.arange(S, device=dev)[None] < lens[:, None]).float()
vec = lambda: torch.randn(D, device=dev, generator=g) * 0.1
ln3, av, bv = (1 + vec(), vec()), vec(), vec()
vn = torch.nn.functional.normalize(torch.randn(Nv, D, device=dev, generator=g), dim=-1)
out = {k: torch.empty(Nv, Nm, device=dev) for k in ("648", "64", "32")}
scale = 1 / math.sqrt(D)
def run(pq):
    os.environ["MADE_XPOOL_SIMS_PQ"] = pq
    ops.xpool_sims(Q, K, UU, mask, av, bv, ln3, vn, out[pq], scale=scale)
def timeit(pq, n=3):
    run(pq); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): run(pq)
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
for rep in range(3):
    t8, t64, t32 = timeit("648"), timeit("64"), timeit("32")
    print(f"Nv={Nv} Nm={Nm} S={S}: 64-video 8-wave kernel {t8:8.2f} ms   64-video kernel {t64:8.2f} ms   32-video kernel {t32:8.2f} ms   ratios {t32 / t8:.3f} {t32 / t64:.3f}", flush=True)
for k in ("648", "64"):
    d = (out[k] - out["32"]).abs()
    print(f"max |pq{k} - pq32| = {float(d.max()):.3e}, mean {float(d.mean()):.3e}; nan: {int(torch.isnan(out[k]).sum())} / {int(torch.isnan(out['32']).sum())}")

"""made_xpool_sims: the 64-video kernel (MADE_XPOOL_SIMS_PQ=64) against the 32-video kernel (the default) on the retrieval set, alternating, plus
their difference.  python tools/xpool_pq_ab.py [Nv Nm S]   (ARMS=64,648,... : other values of the knob the library knows)"""
import math, os, sys
os.environ.setdefault("MADE_DEBUG_VARIANTS", "1")          # (measurement knobs are honoured only under this switch)
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
ARMS = os.environ.get("ARMS", "64").split(",")
out = {k: torch.empty(Nv, Nm, device=dev) for k in ARMS + ["32"]}
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
    ts = {k: timeit(k) for k in ARMS + ["32"]}
    print(f"Nv={Nv} Nm={Nm} S={S}: " + "   ".join(f"PQ={k} {v:8.2f} ms" for k, v in ts.items()) + "   32 / arm: " + " ".join(f"{ts['32'] / ts[k]:.3f}" for k in ARMS), flush=True)
for k in ARMS:
    d = (out[k] - out["32"]).abs()
    print(f"max |pq{k} - pq32| = {float(d.max()):.3e}, mean {float(d.mean()):.3e}; nan: {int(torch.isnan(out[k]).sum())} / {int(torch.isnan(out['32']).sum())}")

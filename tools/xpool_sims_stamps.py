"""In-kernel cycle stamps of made_xpool_sims' short-track kernel (MADE_XPOOL_DBG=32 build): phases of workgroup (0, 0), per wave, averaged over
tracks 8..31 of its chunk.  python tools/xpool_sims_stamps.py [S_fixed]   (S_fixed: every track that long; default: lengths U{12..96})"""
import math, os, sys
os.environ.setdefault("MADE_DEBUG_VARIANTS", "1")          # (measurement knobs are honoured only under this switch)
K64 = os.environ.get("PQ", "64") == "64"                        # PQ=32: round 4's 32-video kernel
K32 = True
os.environ["MADE_XPOOL_DBG"] = "64" if K64 else "32"
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mgsv_amd import ops
Nv, Nm, S, D = 16384, 512, 96, 256
fixed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
Q = torch.randn(Nv, D, device=dev, generator=g).bfloat16()
K = torch.randn(Nm, S, D, device=dev, generator=g).bfloat16()
UU = torch.randn(Nm, S, 2 * D, device=dev, generator=g).bfloat16()
lens = torch.randint(12, S + 1, (Nm,), device=dev, generator=g)
if fixed: lens[:] = fixed
mask = (torch.arange(S, device=dev)[None] < lens[:, None]).float()
vec = lambda: torch.randn(D, device=dev, generator=g) * 0.1
ln3, av, bv = (1 + vec(), vec()), vec(), vec()
vn = torch.nn.functional.normalize(torch.randn(Nv, D, device=dev, generator=g), dim=-1)
sims = torch.zeros(Nv, Nm, device=dev)
for _ in range(2):
    ops.xpool_sims(Q, K, UU, mask, av, bv, ln3, vn, sims, scale=1 / math.sqrt(D))
torch.cuda.synchronize()
st = sims.view(-1)[:8 * 32 * 16 * 2].view(torch.int64).view(8, 32, 16).cpu()
NWAVES = 4 if K32 else 8
names32 = ["wait K", "barrier A", "scores", "barrier B", "issue U 0,1", "mask + max", "barrier C", "exp + P", "second product", "barrier D", "next info + K issue",
           "partial sums", "barrier", "pair (1 wave)", "-"]
names64 = ["wait K + Q", "barrier A", "scores", "barrier B", "issue U 0,1", "softmax + P", "second product", "barrier D", "gv issue, o sums", "wait gv", "next info + K issue",
           "z sums", "barrier", "pair (1 wave)", "-"]
names = names64 if K64 else names32 if K32 else ["wait K", "barrier A", "scores", "barrier B", "next info + K / h4,5 issue", "mask + max", "barrier C", "exp + P", "second product", "tail 1", "barrier B1",
         "issue U (+K)", "tail 2", "barrier B2", "tail 3 / store"]
print(f"track lengths: {lens[:32].tolist()}")
print("wave " + " ".join(f"{n[:10]:>10s}" for n in names) + "      total")
for w in range(NWAVES):
    d = (st[w, 8:32, 1:] - st[w, 8:32, :-1]).float().mean(0)
    tot = (st[w, 9:32, 0] - st[w, 8:31, 0]).float().mean()
    print(f"{w:4d} " + " ".join(f"{float(x):10.0f}" for x in d) + f" {float(tot):10.0f}")
print("(s_memtime ticks at 100 MHz x ?: compare rows, not absolute values; total = top-of-track to top-of-track)")

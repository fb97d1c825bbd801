"""Phase stamps of made_xpool_fused's persistent kernel (MADE_XPOOL_DBG=33: workgroup (0,0) writes s_memtime at its phase boundaries
into the sims buffer): per wave and iteration, cycles between consecutive stamps."""
import math, os, sys
os.environ.setdefault("MADE_DEBUG_VARIANTS", "1")          # (measurement knobs are honoured only under this switch)
os.environ["MADE_XPOOL_DBG"] = "33"
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mgsv_amd import ops
Nv, Nm, S, D = 8192, 256, int(sys.argv[1]) if len(sys.argv) > 1 else 96, 256
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
Q = torch.randn(Nv, D, device=dev, generator=g).bfloat16()
K = torch.randn(Nm, S, D, device=dev, generator=g).bfloat16(); U = torch.randn(Nm, S, D, device=dev, generator=g).bfloat16()
mask = torch.ones(Nm, S, device=dev)
Wl = (torch.randn(D, D, device=dev, generator=g) / math.sqrt(D)).bfloat16()
vec = lambda: torch.randn(D, device=dev, generator=g) * 0.1
ln2, ln3, bl = (1 + vec(), vec()), (1 + vec(), vec()), vec()
vn = torch.nn.functional.normalize(torch.randn(Nv, D, device=dev, generator=g), dim=-1)
sims = torch.zeros(Nv, Nm, device=dev)
for _ in range(2):
    ops.xpool_fused(Q, K, U, mask, ln2, Wl, bl, ln3, vn, sims, scale=1 / math.sqrt(D))
torch.cuda.synchronize()
st = sims.view(-1)[:8 * 16 * 16 * 2].view(torch.int64).view(8, 16, 16).cpu()
for w in (0, 2, 4, 5):
    print(f"wave {w} ({'attention' if w < 4 else 'linear'}): cycles from the iteration's first stamp")
    for j in range(3, 9):
        row = st[w, j]; base = int(row[0])
        print("   it", j, " ".join(f"{p}:{int(row[p]) - base:6d}" if int(row[p]) else f"{p}:     -" for p in list(range(11)) + ([11, 12, 13] if w < 4 else [])))

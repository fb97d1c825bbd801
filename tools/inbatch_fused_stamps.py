"""Phase stamps (s_memtime) of the one-launch in-batch X-Pool kernel: workgroups 0, 101 and the last one.  MADE_XPOOL_INBATCH_FUSED=4."""
import math, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from mgsv_amd import ops
dev = "cuda"
Nv, Nm, S, D = 64, 64, 512, 512
q = torch.randn(Nv, D, device=dev).bfloat16(); k = torch.randn(Nm, S, D, device=dev).bfloat16(); u = torch.randn(Nm, S, D, device=dev).bfloat16()
o = torch.empty(Nm, Nv, D, device=dev, dtype=torch.bfloat16)
need = ops.xpool_inbatch_ws_bytes(Nm, S)
ws = torch.zeros(need + 512, device=dev, dtype=torch.uint8)
names = ["start", "K+Q landed", "scores MFMA done", "tiles stored (issued)", "acks + barrier", "U issued", "poll passed", "U + tiles landed", "pv MFMA done", "out stored"]
for rep in range(3):
    os.environ["MADE_XPOOL_INBATCH_FUSED"] = "4"
    ops.xpool_inbatch(q, k, u, None, o, scale=1 / math.sqrt(D), ws=ws)
    torch.cuda.synchronize()
    st = ws[need - 16 + 16:need - 16 + 16 + 3 * 16 * 8].view(torch.int64).view(3, 16).cpu()
    print(f"rep {rep}: (s_memtime ticks since the workgroup started; the XCDs' counters are not aligned with each other)")
    for w, nm in enumerate(("wg 0", "wg 101", "wg last")):
        print(f"  {nm:8s} " + "  ".join(f"{names[i]}: {int(st[w, i]) - int(st[w, 0])}" for i in range(10)))

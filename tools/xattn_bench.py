"""made_xpool_attention alone and the whole D = 512 retrieval pass (bench.py's S512_D512 problem): per-launch times (HIP events) and the
executed TFLOP/s; MADE_XPOOL_ATTN=0 gives the separate-launch chain.  usage: python tools/xattn_bench.py [Nv Nm S D]"""
import math, os, sys, time
os.environ.setdefault("MADE_DEBUG_VARIANTS", "1")          # (measurement knobs are honoured only under this switch)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mgsv_amd import ops
Nv, Nm, S, D = (int(x) for x in sys.argv[1:5]) if len(sys.argv) >= 5 else (8192, 64, 512, 512)
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
Q = torch.randn(Nv, D, device=dev, generator=g).bfloat16()
K = torch.randn(Nm, S, D, device=dev, generator=g).bfloat16()
U = torch.randn(Nm, S, D, device=dev, generator=g).bfloat16()
lens = torch.randint(min(12, S), S + 1, (Nm,), device=dev, generator=g)
out = torch.empty(Nm, Nv, D, device=dev, dtype=torch.bfloat16)
for name, ln in (("ragged U{12..S}", lens), ("full", torch.full_like(lens, S))):
    mask = (torch.arange(S, device=dev)[None] < ln[:, None]).float()
    for _ in range(2): ops.xpool_attention(Q, K, U, mask, out, scale=1 / math.sqrt(D))
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(5): ops.xpool_attention(Q, K, U, mask, out, scale=1 / math.sqrt(D))
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 5
    fl = 4.0 * Nv * float(mask.sum()) * D
    print(f"made_xpool_attention Nv={Nv} Nm={Nm} S={S} D={D} {name}: {ms * 1e3:.1f} us, {fl / ms / 1e9:.1f} TFLOP/s executed, "
          f"{ms * 1e3 / Nm / ((Nv + 63) // 64) * 256:.2f} us per (video tile, track) per CU")

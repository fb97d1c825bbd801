"""made_linear on large regular problems and on the step's encoder-sized ones: the single-stage LDS-DMA kernels (MADE_LINEAR_TILE=64 / 128)
against the persistent big-tile kernel (256 = 128 x 256 tiles, 512 = 256 x 256 tiles).  HIP events around back-to-back launches."""
import math, os, sys
os.environ.setdefault("MADE_DEBUG_VARIANTS", "1")          # (measurement knobs are honoured only under this switch)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mgsv_amd import ops
dev = torch.device("cuda"); dt = torch.bfloat16
def bench(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
shapes = [(1048576, 512, 512, False, False), (131072, 512, 512, False, False), (34688, 512, 512, False, False), (34688, 1024, 512, False, False),
          (34688, 512, 1024, True, False), (32768, 512, 512, True, True), (32768, 1024, 512, False, True), (32768, 1536, 512, False, True)]
for M, N, K, res, gather in shapes:
    A = torch.randn(M, K, device=dev).to(dt); W = (torch.randn(N, K, device=dev) / math.sqrt(K)).to(dt)
    b = torch.randn(N, device=dev); out = torch.zeros(M, N, device=dev, dtype=dt)
    R = torch.randn(M, N, device=dev).to(dt) if res else None
    rows, live = None, M
    if gather:
        lens = torch.randint(40, 513, (M // 512,), device=dev)
        mask = (torch.arange(512, device=dev)[None] < lens[:, None]).float()
        rows = ops.row_index(mask); live = int(mask.sum())
    res_ = {}
    ref = None
    for tile in ("64", "128", "256", "512"):
        os.environ["MADE_LINEAR_TILE"] = tile
        us = bench(lambda: ops.linear(A, W, b, out=out, R=R, rows=rows))
        torch.cuda.synchronize()
        cur = out.float()
        if ref is None: ref = cur.clone()
        res_[tile] = (us, float((cur - ref).abs().max()))
    os.environ.pop("MADE_LINEAR_TILE")
    print(f"M={M} (live {live}) N={N} K={K} res={int(res)} gather={int(gather)}: " +
          " | ".join(f"t{t} {u:8.1f} us {2.0 * live * N * K / u / 1e6:6.0f} TF (diff {d:.3g})" for t, (u, d) in res_.items()))

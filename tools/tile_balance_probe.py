"""Does the big-tile Linear win when its tiles fill the chip in whole rounds?  made_linear alone on N = 512 / 1024 problems whose row counts
give the 128 x 256-tile kernel exactly 1 / 2 rounds of 256 workgroups (M = 16384 / 32768) and 1.09 / 2.1 rounds (M = 17920 / 34688), each
kernel forced with MADE_LINEAR_TILE; arms interleaved per round, median of 7."""
import math, os, statistics, sys
os.environ.setdefault("MADE_DEBUG_VARIANTS", "1")          # (measurement knobs are honoured only under this switch)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mgsv_amd import ops
dev = torch.device("cuda"); dt = torch.bfloat16
def run(fn, n=20):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); e.synchronize()
    return s.elapsed_time(e) / n * 1e3
for M in (16384, 17920, 32768, 34688):
    for N, K in ((512, 512), (512, 1024), (1024, 512), (1536, 512)):
        A = (torch.rand(M, K, device=dev) * 2 - 1).to(dt); W = ((torch.rand(N, K, device=dev) * 2 - 1) / math.sqrt(K)).to(dt)
        b = torch.randn(N, device=dev); out = torch.zeros(M, N, device=dev, dtype=dt)
        res = {t: [] for t in ("64", "128", "256", "512")}
        for t in res:
            os.environ["MADE_LINEAR_TILE"] = t
            for _ in range(3): ops.linear(A, W, b, out=out)
        torch.cuda.synchronize()
        for _ in range(7):
            for t in res:
                os.environ["MADE_LINEAR_TILE"] = t
                res[t].append(run(lambda: ops.linear(A, W, b, out=out)))
        os.environ.pop("MADE_LINEAR_TILE")
        mm = run(lambda: torch.matmul(A, W.t()))
        print(f"M={M:6d} N={N:5d} K={K:5d}: " + " | ".join(f"t{t} {statistics.median(v):7.1f} us" for t, v in res.items()) + f" | vendor {mm:7.1f} us", flush=True)

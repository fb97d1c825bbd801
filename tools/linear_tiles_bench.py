"""Encoder-sized Linears for rocprofv3 --kernel-trace: run once as is (256 x 256 tiles) and once with MADE_LINEAR_TILE=64 / 128
(the smaller direct-to-LDS tiles) and compare the per-launch durations (tools/linear_tiles_trace.sh)."""
import math, os, sys, torch
os.environ.setdefault("MADE_DEBUG_VARIANTS", "1")          # (measurement knobs are honoured only under this switch)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mgsv_amd import ops
dev = torch.device("cuda"); dt = torch.bfloat16
for M, N, K, res, gather in ((32768, 512, 512, False, False), (32768, 512, 512, True, False), (32768, 1024, 512, False, False),
                             (34688, 512, 512, False, False), (34688, 1024, 512, False, False), (61440, 512, 512, True, True),
                             (61440, 1024, 512, False, True), (32768, 1536, 512, False, False), (32768, 2048, 512, False, False),
                             (32768, 512, 2048, True, False)):
    A = torch.randn(M, K, device=dev).to(dt); W = (torch.randn(N, K, device=dev) / math.sqrt(K)).to(dt)
    b = torch.randn(N, device=dev); out = torch.zeros(M, N, device=dev, dtype=dt)
    R = torch.randn(M, N, device=dev).to(dt) if res else None
    rows = None
    if gather:
        lens = torch.randint(40, 513, (M // 512,), device=dev)
        mask = (torch.arange(512, device=dev)[None] < lens[:, None]).float()
        rows = ops.row_index(mask)
    for _ in range(8):
        ops.linear(A, W, b, out=out, R=R, rows=rows)
    torch.cuda.synchronize()

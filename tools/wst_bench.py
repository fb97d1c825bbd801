"""Encoder-sized Linears: the W-stationary kernel (default) against the single-stage direct-to-LDS kernels (MADE_LINEAR_TILE=64 / 128),
HIP-event time per launch over back-to-back launches, and the results compared bit for bit where the kernels must agree (they
accumulate in the same order along K per output? no -- different MFMA shapes: compared within bf16 tolerance)."""
import math, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mgsv_amd import ops
dev = torch.device("cuda"); dt = torch.bfloat16
REP = int(os.environ.get("REP", "30"))
shapes = ((34688, 512, 512, False, False), (34688, 1024, 512, False, False), (34688, 512, 1024, True, False), (32768, 512, 512, True, True),
          (32768, 1024, 512, False, True), (32768, 1536, 512, False, True), (34688, 512, 256, False, False), (8192, 512, 512, False, False))
for M, N, K, res, gather in shapes:
    A = torch.randn(M, K, device=dev).to(dt); W = (torch.randn(N, K, device=dev) / math.sqrt(K)).to(dt)
    b = torch.randn(N, device=dev)
    R = torch.randn(M, N, device=dev).to(dt) if res else None
    rows, live = None, M
    if gather:
        lens = torch.randint(40, 513, (M // 512,), device=dev)
        mask = (torch.arange(512, device=dev)[None] < lens[:, None]).float()
        rows = ops.row_index(mask); live = int(mask.sum())
    outs, line = {}, []
    for name, env in (("t64", {"MADE_LINEAR_TILE": "64"}), ("t128", {"MADE_LINEAR_TILE": "128"}), ("wst", {"MADE_LINEAR_TILE": "0"})):
        os.environ.update(env)
        out = torch.zeros(M, N, device=dev, dtype=dt)
        for _ in range(5):
            ops.linear(A, W, b, out=out, R=R, rows=rows, act=ops.ACT_RELU)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(REP):
            ops.linear(A, W, b, out=out, R=R, rows=rows, act=ops.ACT_RELU)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / REP
        outs[name] = out.float()
        line.append(f"{name} {us:7.1f} us {2 * live * N * K / us * 1e-6:6.0f} TFLOP/s")
    err = max(float((outs[k] - outs["t64"]).abs().max()) for k in outs)
    print(f"M={M} (live {live}) N={N} K={K} res={int(res)} gather={int(gather)}: " + " | ".join(line) + f" | max |diff| vs t64 {err:.3g}", flush=True)
os.environ.pop("MADE_LINEAR_TILE", None)

"""made_linear on the encoder-sized shapes, one process, every kernel choice (MADE_LINEAR_TILE is read per call):
1000 = round 1's single-stage kernels, 2128 / 2256 = the LDS-DMA ring at 128 x 128 / 256 x 256.  Per-launch time from HIP events
around 20 back-to-back launches (rotating over 3 buffer sets), so launch gaps are included, as in a step."""
import math, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mgsv_amd import ops
dev = torch.device("cuda"); dt = torch.bfloat16
shapes = [(18432, 512, 512, False, False, 0), (18432, 512, 512, True, False, 0), (32768, 512, 512, True, True, 0), (18432, 1536, 512, False, False, 0),
          (32768, 1536, 512, False, True, 0), (18432, 1024, 512, False, False, ops.ACT_RELU), (32768, 1024, 512, False, True, ops.ACT_GELU),
          (18432, 512, 1024, True, False, 0), (32768, 512, 1024, True, True, 0), (32768, 512, 768, True, True, 0)]
tiles = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [1000, 2128]
print(f"{'M x N x K (R=residual g=gather a=act)':44s}" + "".join(f"{t:>12d}" for t in tiles))
for M, N, K, res, gather, act in shapes:
    sets = []
    for i in range(3):
        A = torch.randn(M, K, device=dev).to(dt); W = (torch.randn(N, K, device=dev) / math.sqrt(K)).to(dt)
        out = torch.zeros(M, N, device=dev, dtype=dt)
        R = torch.randn(M, N, device=dev).to(dt) if res else None
        sets.append((A, W, out, R))
    b = torch.randn(N, device=dev)
    rows = None
    if gather:
        lens = torch.randint(12, 513, (M // 512,), device=dev)
        mask = (torch.arange(512, device=dev)[None] < lens[:, None]).float()
        rows = ops.row_index(mask)
        nvalid = int(rows[1].item())
    else:
        nvalid = M
    line = f"{M:6d}x{N:5d}x{K:5d} {'R' if res else ' '}{'g' if gather else ' '}{'a' if act else ' '} valid={nvalid:6d}      "
    for t in tiles:
        os.environ["MADE_LINEAR_TILE"] = str(t)
        def run20():
            for i in range(20):
                A, W, out, R = sets[i % 3]
                ops.linear(A, W, b, out=out, R=R, rows=rows, act=act)
        run20()
        torch.cuda.synchronize()
        st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):               # a graph replay takes the host's launch rate (ctypes: ~20 us per call) out of the number
                run20()
            g.replay(); torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(3):
                g.replay()
            e.record(); torch.cuda.synchronize()
        us = s.elapsed_time(e) * 1e3 / 60
        line += f"{us:7.1f}us{2.0 * nvalid * N * K / us / 1e6:5.0f}"
    print(line, flush=True)

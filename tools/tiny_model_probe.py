"""Per-stage time of a chain of dependent tiny Linears (hipGraph replay) against the bytes one workgroup pulls in: rows M in {16, 64},
K = N in {256, 512, 1024}, weights rotating over 24 copies.  Tests the model  t = t0 + bytes_per_workgroup / (per-CU fill rate)."""
import math, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mgsv_amd import ops
dev = torch.device("cuda"); dt = torch.bfloat16
def bench(fn, n=48, reps=5):
    fn(0); torch.cuda.synchronize()
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for i in range(n): fn(i)
        g.replay(); torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(reps): g.replay()
        e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / (n * reps)
for K in (256, 512, 1024):
    N = K
    Ws = [(torch.randn(N, K, device=dev) / math.sqrt(K)).to(dt) for _ in range(24)]
    for M in (16, 32, 64):
        xs = [torch.randn(M, K, device=dev).to(dt), torch.empty(M, K, device=dev, dtype=dt)]
        def fn(i):
            ops.linear(xs[i & 1], Ws[i % 24], None, out=xs[(i + 1) & 1])
        t = bench(fn)
        kb = (min(M, 64) * K * 2 + 32 * K * 2) / 1024
        print(f"M={M:3d} N=K={K:5d}: {t:6.2f} us per dependent stage; {N // 32} workgroups x {kb:.0f} KB")
# the same chain issued as ordinary stream launches (the library's tape: one C loop of hipLaunchKernel) instead of a hipGraph
from mgsv_amd.tape import LaunchTape
K = N = 512; M = 64
Ws = [(torch.randn(N, K, device=dev) / math.sqrt(K)).to(dt) for _ in range(24)]
xs = [torch.randn(M, K, device=dev).to(dt), torch.empty(M, K, device=dev, dtype=dt)]
with LaunchTape.record() as tp:
    for i in range(48):
        ops.linear(xs[i & 1], Ws[i % 24], None, out=xs[(i + 1) & 1])
torch.cuda.synchronize()
for _ in range(3): tp.replay()
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(10): tp.replay()
e.record(); torch.cuda.synchronize()
print(f"M=64 N=K=512 as stream launches (tape replay): {s.elapsed_time(e) * 1e3 / 480:.2f} us per dependent stage")
st = torch.cuda.Stream()
with torch.cuda.stream(st):
    with LaunchTape.record() as tp2:
        for i in range(48):
            ops.linear(xs[i & 1], Ws[i % 24], None, out=xs[(i + 1) & 1])
    torch.cuda.synchronize()
    for _ in range(3): tp2.replay()
    torch.cuda.synchronize()
    s.record()
    for _ in range(10): tp2.replay()
    e.record(); torch.cuda.synchronize()
print(f"   ... on a non-default stream: {s.elapsed_time(e) * 1e3 / 480:.2f} us per dependent stage")

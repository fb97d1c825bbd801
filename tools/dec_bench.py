"""made_dec_stage / tiny made_linear in isolation: warm (same buffers every launch) and cold (rotating over weight copies that together
exceed the caches), under graph replay; shows what a decoder-sized launch costs apart from everything around it."""
import math, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mgsv_amd import ops
dev = torch.device("cuda"); dt = torch.bfloat16
M, K = 64, 512
def bench(fn, n=40, reps=5):
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
for N in (512, 1024, 4096):
    ncopy = 64 if N <= 1024 else 24
    Ws = [(torch.randn(N, K, device=dev) / math.sqrt(K)).to(dt) for _ in range(ncopy)]
    b = torch.randn(N, device=dev)
    z = torch.randn(M, K, device=dev); g1 = torch.ones(K, device=dev); b1 = torch.zeros(K, device=dev)
    xo = torch.empty(M, K, device=dev, dtype=dt)
    out32 = torch.empty(M, N, device=dev); outb = torch.empty(M, N, device=dev, dtype=dt)
    A = torch.randn(M, K, device=dev).to(dt); R = torch.randn(M, N, device=dev).to(dt)
    junk = torch.empty(512 << 20, device=dev, dtype=torch.uint8)
    res = {}
    res["dec_stage warm"] = bench(lambda i: ops.dec_stage(z, Ws[0], b, out32, ln=(g1, b1), x_out=xo))
    res["dec_stage cold W"] = bench(lambda i: ops.dec_stage(z, Ws[i % ncopy], b, out32, ln=(g1, b1), x_out=xo))
    res["dec_stage noLN warm"] = bench(lambda i: ops.dec_stage(z, Ws[0], b, out32))
    res["tiny warm"] = bench(lambda i: ops.linear(A, Ws[0], b, R=R, out=out32))
    res["tiny cold W"] = bench(lambda i: ops.linear(A, Ws[i % ncopy], b, R=R, out=out32))
    res["l2norm (trivial kernel)"] = bench(lambda i: ops.l2norm_rows(z, out_f32=out32[:, :K]))
    def flushy(i):
        junk[: 256 << 20].zero_()
        ops.dec_stage(z, Ws[i % ncopy], b, out32, ln=(g1, b1), x_out=xo)
    def flushonly(i):
        junk[: 256 << 20].zero_()
    res["dec_stage after 256MB memset (minus memset)"] = bench(flushy, n=10) - bench(flushonly, n=10)
    print(f"N={N}: " + "  ".join(f"{k}: {v:.1f}us" for k, v in res.items()), flush=True)

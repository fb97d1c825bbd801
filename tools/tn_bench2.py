"""made_gemm_tn (direct-to-LDS kernel) under graph replay: with / without the fused bias gradient, with / without the row gather,
single launches against the grouped launch of one layer's five products.  Shows where the weight-gradient time goes."""
import sys, os
os.environ.setdefault("MADE_DEBUG_VARIANTS", "1")          # (measurement knobs are honoured only under this switch)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mgsv_amd import ops, ops_train as tr
dev = "cuda"
def bench(fn, n=10, reps=3):
    fn(); torch.cuda.synchronize()
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(n): fn()
        g.replay(); torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(reps): g.replay()
        e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / (n * reps)
M = 34688
lens = torch.randint(12, 543, (64,), device=dev)
mask = (torch.arange(542, device=dev)[None] < lens[:, None]).float()
rows = ops.row_index(mask)
nv = int(rows[1].item())
mk = lambda n: torch.randn(M, n, device=dev).bfloat16()
for N, K in ((512, 512), (1024, 512), (512, 1024), (1536, 512)):
    A, B = mk(N), mk(K)
    C = torch.zeros(N, K, device=dev); cs = torch.zeros(N, device=dev)
    r = {}
    r["gather+colsum"] = bench(lambda: tr.gemm_tn(A, B, C, accumulate=True, colsum=cs, rows=rows))
    r["gather"] = bench(lambda: tr.gemm_tn(A, B, C, accumulate=True, rows=rows))
    Ad, Bd = A[:nv].contiguous(), B[:nv].contiguous()
    r["dense+colsum"] = bench(lambda: tr.gemm_tn(Ad, Bd, C, accumulate=True, colsum=cs))
    r["dense"] = bench(lambda: tr.gemm_tn(Ad, Bd, C, accumulate=True))
    for sp in (8, 32):
        r[f"gather split{sp}"] = bench(lambda: tr.gemm_tn(A, B, C, accumulate=True, rows=rows, split_m=sp))
    print(f"N={N} K={K} valid rows {nv}: " + "  ".join(f"{k}: {v:.1f}us ({2.0 * nv * N * K / v / 1e6:.0f} TF)" for k, v in r.items()), flush=True)
# one layer's five products, grouped (MADE_TN_TILE=128 selects the 128 x 128-tile kernel)
probs = []
for N, K in ((512, 1024), (1024, 512), (512, 512), (1024, 512), (512, 512)):
    probs.append((mk(N), mk(K), torch.zeros(N, K, device=dev), torch.zeros(N, device=dev)))
fl = sum(2.0 * nv * p[0].shape[1] * p[1].shape[1] for p in probs)
for sp in (None, 8, 16, 24, 32):
    t = bench(lambda: tr.gemm_tn_grouped(probs, rows=rows, split_m=sp), n=4)
    print(f"grouped x5 split={sp}: {t:.1f}us ({fl / t / 1e6:.0f} TF)")
t = bench(lambda: tr.gemm_tn_grouped([(a, b, c, None) for a, b, c, _ in probs], rows=rows), n=4)
print(f"grouped x5 no colsum: {t:.1f}us ({fl / t / 1e6:.0f} TF)")
t = bench(lambda: [tr.gemm_tn(a, b, c, accumulate=True, colsum=s_, rows=rows) for a, b, c, s_ in probs], n=4)
print(f"five single launches: {t:.1f}us ({fl / t / 1e6:.0f} TF)")

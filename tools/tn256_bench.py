"""One DETR-encoder layer's five weight-gradient products as the training step launches them (made_gemm_tn_grouped, 256 x 256 tiles, row gather,
bias gradients), alone under graph replay.  With MADE_LIB_PATH=tools/_ab/tn256_skip<mask>.so (tools/variant_build.sh tn256_skip<m> gemm_tn_glds
"-DTN256_SKIP=<m>"): what the flush / the MFMAs / the operand loads cost."""
import sys, os
os.environ.setdefault("MADE_DEBUG_VARIANTS", "1")          # (measurement knobs are honoured only under this switch)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mgsv_amd import ops, ops_train as tr
dev = "cuda"
def bench(fn, n=6, reps=5):
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
B, L = 64, 542
M = B * L
g = torch.Generator().manual_seed(0)
lv, la = torch.randint(5, 31, (B,), generator=g), torch.randint(12, 513, (B,), generator=g)
pos = torch.arange(L)[None]
mask = ((pos < lv[:, None]) | ((pos >= 30) & (pos < 30 + la[:, None]))).float().to(dev)
rows = ops.row_index(mask)
nv = int(rows[1].item())
mk = lambda n: torch.randn(M, n, device=dev).bfloat16()
shapes = ((512, 1024), (1024, 512), (512, 512), (1024, 512), (512, 512))
if len(sys.argv) > 1 and sys.argv[1] == "x3":            # the three layers' products in one launch (what deferring them to one launch would run)
    shapes = shapes * 3
probs = [(mk(N), mk(K), torch.zeros(N, K, device=dev), torch.zeros(N, device=dev)) for N, K in shapes]
fl = sum(2.0 * nv * p[0].shape[1] * p[1].shape[1] for p in probs)
groups = [probs[i:i + 8] for i in range(0, len(probs), 8)]
holder = {}
def workspace(n):
    if "ws" not in holder or holder["ws"].numel() < n:
        holder["ws"] = tr.gemm_tn_grouped_workspace(torch.device(dev), n)
    return holder["ws"]
t = bench(lambda: [tr.gemm_tn_grouped(gr, rows=rows, workspace=workspace) for gr in groups])
print(f"flush={'atomics' if os.environ.get('MADE_TN256_ATOMIC_FLUSH') else 'workspace'} lib={os.environ.get('MADE_LIB_PATH', 'product')}  {len(shapes)} products, valid rows {nv}: {t:7.1f} us  ({fl / t / 1e6:.0f} TFLOP/s)")


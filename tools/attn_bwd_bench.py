"""Microbenchmark: attention forward / backward at the DETR-encoder shape, with and without dropout (GPU box)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mgsv_amd import ops, ops_train as tr

B, H, hd, L = 64, 8, 64, 542
D = H * hd
qkv = torch.randn(B, L, 3 * D, device="cuda").bfloat16()
q, k, v = qkv[:, :, :D], qkv[:, :, D:2 * D], qkv[:, :, 2 * D:]
lens = torch.randint(200, L + 1, (B,), device="cuda")
mask = (torch.arange(L, device="cuda")[None] < lens[:, None]).float()
O = torch.empty(B, L, D, device="cuda", dtype=torch.bfloat16); dO = torch.randn_like(O)
lse = torch.empty(B, H, L, device="cuda"); delta = torch.empty_like(lse)
dqkv = torch.empty_like(qkv)

def timeit(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3

for p in (0.0, 0.1):
    drop = (1, 2, p)
    f = timeit(lambda: ops.attention(q, k, v, O, H, key_mask=mask, q_skip_mask=mask, lse=lse, drop=drop))
    b = timeit(lambda: tr.attention_bwd(q, k, v, O, dO, dqkv[:, :, :D], dqkv[:, :, D:2 * D], dqkv[:, :, 2 * D:], lse, delta, H,
                                        key_mask=mask, q_skip_mask=mask, drop=drop))
    print(f"p={p}: fwd {f:.1f} us, bwd {b:.1f} us")

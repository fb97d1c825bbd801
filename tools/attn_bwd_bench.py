"""Microbenchmark: attention forward / backward at the DETR-encoder shape (B=64, H=8, hd=64, L=576): dense batch, ragged batch
(valid lengths as bench.py draws them), ragged with the longest-first issue order, with and without dropout (GPU box)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mgsv_amd import ops, ops_train as tr

B, H, hd, L = 64, 8, 64, 576
D = H * hd
qkv = torch.randn(B, L, 3 * D, device="cuda").bfloat16()
q, k, v = qkv[:, :, :D], qkv[:, :, D:2 * D], qkv[:, :, 2 * D:]
O = torch.empty(B, L, D, device="cuda", dtype=torch.bfloat16); dO = torch.randn_like(O)
lse = torch.empty(B, H, L, device="cuda"); delta = torch.empty_like(lse)
dqkv = torch.empty_like(qkv)
g = torch.Generator().manual_seed(0)
lv, la = torch.randint(5, 65, (B,), generator=g), torch.randint(12, 513, (B,), generator=g)
pos = torch.arange(L)[None]
ragged = ((pos < lv[:, None]) | ((pos >= 64) & (pos < 64 + la[:, None]))).float().cuda()
dense = torch.ones(B, L, device="cuda")


def timeit(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


for name, mask, use_order in (("dense", dense, False), ("ragged", ragged, False), ("ragged+order", ragged, True)):
    order = ops.batch_order(mask) if use_order else None
    n = mask.sum(1)
    work = float((n * n).sum() / (B * L * L))
    for p in (0.0, 0.1):
        drop = (1, 2, p)
        f = timeit(lambda: ops.attention(q, k, v, O, H, key_mask=mask, q_skip_mask=mask, lse=lse, drop=drop, order=order))
        b = timeit(lambda: tr.attention_bwd(q, k, v, O, dO, dqkv[:, :, :D], dqkv[:, :, D:2 * D], dqkv[:, :, 2 * D:], lse, delta, H,
                                            key_mask=mask, q_skip_mask=mask, drop=drop, order=order))
        gf = 4 * B * H * L * L * hd * work / 1e9
        print(f"{name:13s} work={work:.2f} p={p}: fwd {f:6.1f} us ({gf / f * 1e3:6.1f} TF), bwd {b:6.1f} us ({2.5 * gf / b * 1e3:6.1f} TF)")

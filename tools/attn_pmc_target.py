"""Attention forward + backward at the DETR-encoder shape of the training step (B=64, H=8, hd=64, L=542, ragged valid lengths as bench.py
draws them, longest-sample-first issue order, dropout p = 0.1): the target of the rocprofv3 --pmc passes of tools/pmc_sq_round4.sh."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mgsv_amd import ops, ops_train as tr
B, H, hd, L = 64, 8, 64, 542
D = H * hd
p = float(os.environ.get("P", "0.1"))
qkv = torch.randn(B, L, 3 * D, device="cuda").bfloat16()
q, k, v = qkv[:, :, :D], qkv[:, :, D:2 * D], qkv[:, :, 2 * D:]
O = torch.empty(B, L, D, device="cuda", dtype=torch.bfloat16); dO = torch.randn_like(O)
lse = torch.empty(B, H, L, device="cuda"); delta = torch.empty_like(lse)
dqkv = torch.empty_like(qkv)
g = torch.Generator().manual_seed(0)
lv, la = torch.randint(5, 31, (B,), generator=g), torch.randint(12, 513, (B,), generator=g)
pos = torch.arange(L)[None]
mask = ((pos < lv[:, None]) | ((pos >= 30) & (pos < 30 + la[:, None]))).float().cuda()
order = ops.batch_order(mask)
drop = (1, 2, p)
bits = torch.empty(*ops.attention_bits_shape(B, H, L, L), device="cuda", dtype=torch.int32) if os.environ.get("BITS", "1") != "0" else None
def run():
    ops.attention(q, k, v, O, H, key_mask=mask, q_skip_mask=mask, lse=lse, drop=drop, order=order, keep_bits=bits)
    tr.attention_bwd(q, k, v, O, dO, dqkv[:, :, :D], dqkv[:, :, D:2 * D], dqkv[:, :, 2 * D:], lse, delta, H, key_mask=mask, q_skip_mask=mask, drop=drop, order=order,
                     keep_bits=bits)
for _ in range(3): run()
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(10): run()
e.record(); torch.cuda.synchronize()
n = mask.sum(1)
work = float((n * n).sum() / (B * L * L))
gf = 4 * B * H * L * L * hd * work / 1e9
us = s.elapsed_time(e) / 10 * 1e3
print(f"[bits {0 if bits is None else 1}] attention fwd + bwd, ragged (executed fraction {work:.2f}), p = {p}: {us:.1f} us per pair of calls = {3.5 * gf / us * 1e3:.1f} TFLOP/s executed (fwd 1x + bwd 2.5x of {gf:.1f} GFLOP)")

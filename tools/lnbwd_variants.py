"""made_layernorm_bwd alone at the DETR-encoder shape (34 688 token rows, ~54 % valid, D = 512, bf16): the forms the training step launches
(plain; + add; + add + dropped copy), each with and without the parameter-gradient flush, against a plain device copy of the same bytes.
    python tools/lnbwd_variants.py            (MADE_LNBWD_NB / MADE_LNBWD_RF: grid cap / rows in flight per wave)
"""
import os, sys, torch
os.environ.setdefault("MADE_DEBUG_VARIANTS", "1")          # (measurement knobs are honoured only under this switch)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mgsv_amd import ops_train as tr, _lib

dev = "cuda"
B, L, D = 64, 542, 512
g = torch.Generator().manual_seed(0)
lv, la = torch.randint(5, 31, (B,), generator=g), torch.randint(12, 513, (B,), generator=g)
pos = torch.arange(L)[None]
mask = ((pos < lv[:, None]) | ((pos >= 30) & (pos < 30 + la[:, None]))).float().to(dev).reshape(-1)
live = int(mask.sum().item())
x = torch.randn(B * L, D, device=dev).bfloat16(); dy = torch.randn(B * L, D, device=dev).bfloat16()
add = torch.randn(B * L, D, device=dev).bfloat16(); dx = torch.empty_like(x); dxd = torch.empty_like(x)
gamma = torch.ones(D, device=dev); dg = torch.zeros(D, device=dev); db = torch.zeros(D, device=dev)
drop = (11, 3, 0.1)


def timed(fn, n=60):
    for _ in range(6):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


print(f"rows {B * L}, live {live}, NB={os.environ.get('MADE_LNBWD_NB', '-')} RF={os.environ.get('MADE_LNBWD_RF', '-')}")
forms = {
    "plain": dict(),
    "add": dict(add=add),
    "add+drop": dict(add=add, dx_drop=dxd, drop=drop),
}
for name, kw in forms.items():
    for flush in (True, False):
        t = timed(lambda: tr.layernorm_bwd(x, gamma, dy, dx, dgamma=dg if flush else None, dbeta=db if flush else None, row_skip=mask, **kw))
        tensors = 3 + (1 if "add" in kw else 0) + (1 if "dx_drop" in kw else 0)
        mb = live * D * 2 * tensors / 1e6
        print(f"  {name:9s} flush={int(flush)}  {t:7.2f} us   {mb:6.1f} MB of live rows  {mb / t / 1e3 * 1e3:6.2f} GB/ms = {mb / t:5.2f} TB/s")
# yardstick: a device copy moving the same bytes (half read, half written)
for tensors in (3, 5):
    n = live * D * tensors // 2
    src = torch.empty(n, device=dev, dtype=torch.bfloat16); dst = torch.empty_like(src)
    t = timed(lambda: dst.copy_(src))
    print(f"  copy of {n * 4 / 1e6:6.1f} MB (read + write): {t:7.2f} us = {n * 4 / 1e6 / t:5.2f} TB/s")

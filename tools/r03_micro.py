"""Round-3 micro-benchmarks (hipGraph replay of 20 calls each, so launch overhead stays out):
  * made_layernorm_bwd at the DETR-encoder shape with / without the parameter-gradient flush, grid caps via MADE_LNBWD_NB
  * made_attention_wide at the decoder's cross-attention shape: keys split over workgroups (merged by a second launch; the in-launch merge measured in profiles/r03_micro_merge_in_launch_vs_second_launch.txt was removed)
  * made_attention_wide_bwd at the same shape for several key splits
usage: python tools/r03_micro.py [ln|wide|all]"""
import math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mgsv_amd import ops, ops_train as tr

dev, dt = torch.device("cuda"), torch.bfloat16
what = sys.argv[1] if len(sys.argv) > 1 else "all"


def timeit(fn, iters=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(iters): fn()
    g.replay(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); g.replay(); g.replay(); e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / (2 * iters) * 1e3


B, L, D = 64, 542, 512
g = torch.Generator().manual_seed(0)
lv, la = torch.randint(5, 31, (B,), generator=g), torch.randint(12, 513, (B,), generator=g)
pos = torch.arange(L)[None]
mask2 = ((pos < lv[:, None]) | ((pos >= 30) & (pos < 30 + la[:, None]))).float().to(dev)
mask = mask2.reshape(-1)
nvalid = int(mask.sum())

if what in ("ln", "all"):
    x = torch.randn(B * L, D, device=dev).to(dt); dy = torch.randn(B * L, D, device=dev).to(dt); dx = torch.empty_like(x); dxd = torch.empty_like(x)
    gamma = torch.ones(D, device=dev); dg = torch.zeros(D, device=dev); db = torch.zeros(D, device=dev)
    byts = nvalid * D * 2 * 4
    for flush in (True, False):
        for drop in (None, (1, 2, 0.1)):
            t = timeit(lambda: tr.layernorm_bwd(x, gamma, dy, dx, dgamma=dg if flush else None, dbeta=db if flush else None, dx_drop=dxd, drop=drop, row_skip=mask))
            print(f"layernorm_bwd rows={nvalid} of {B * L} D={D} flush={flush} dropout={drop is not None} NB={os.environ.get('MADE_LNBWD_NB', 'default')}: "
                  f"{t:7.1f} us  {byts / t / 1e3:7.1f} GB/s", flush=True)

if what in ("wide", "all"):
    NQ, hd = 8, 64
    scale = 1 / math.sqrt(hd)
    q = (torch.randn(B, NQ, D, device=dev) * 0.5).to(dt); dO = (torch.randn(B, NQ, D, device=dev) * 0.3).to(dt)
    k, v = torch.randn(B, L, D, device=dev).to(dt), torch.randn(B, L, D, device=dev).to(dt)
    O = torch.empty(B, NQ, 1, D, device=dev, dtype=dt)
    ssum, lse = torch.empty(B * NQ, device=dev), torch.empty(B * NQ, device=dev)
    byts = 2.0 * nvalid * D * 2
    for ns in (1, 2, 4, 8):
        po, pml = torch.empty(B * ns * NQ * D, device=dev), torch.empty(B * ns * NQ * 4, device=dev)
        t = timeit(lambda: ops.attention_wide(q.view(B, NQ, 1, D), k, v, O, scale=scale, key_mask=mask2, n_split=ns, part_o=po, part_ml=pml,
                                              drop=(1, 2, 0.1), sum_out=ssum, lse_out=lse))
        print(f"attention_wide fwd B={B} NQ={NQ} L={L} D={D} n_split={ns} {'+ merge launch' if ns > 1 else ''}: "
              f"{t:7.1f} us  {byts / t / 1e3:7.1f} GB/s", flush=True)
    Lp = (L + 7) // 8 * 8
    Pd = torch.empty(B, 2, NQ, Lp, device=dev, dtype=dt); dQ = torch.empty(B, NQ, D, device=dev, dtype=dt)
    dattc = torch.randn(B, D, device=dev).to(dt); bv = torch.randn(D, device=dev)
    for ns in (1, 2, 4, 8):
        part = torch.empty(B * ns * NQ * D, device=dev)
        t = timeit(lambda: tr.attention_wide_bwd(q, dO, O.view(B, NQ, D), k, v, lse.view(B, NQ), Pd[:, 0], Pd[:, 1], dQ, scale=scale, key_mask=mask2,
                                                 ssum=ssum.view(B, NQ), dattc=dattc, vbias=bv, hd=hd, drop=(1, 2, 0.1), n_split=ns, part_dq=part))
        print(f"attention_wide_bwd B={B} NQ={NQ} L={L} D={D} n_split={ns}: {t:7.1f} us  {byts / t / 1e3:7.1f} GB/s", flush=True)

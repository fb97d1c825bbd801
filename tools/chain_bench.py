"""One decoder-layer-like program of 14 dependent 64-row stages: separate launches vs made_chain with N workgroups (in isolation)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mgsv_amd import ops, ops_train as tr
dev = "cuda"
Bq, D, H, Fd = 64, 512, 8, 1024
hd = D // H
g = torch.Generator(device="cpu").manual_seed(0)
rnd = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(dev)
NL = 6
Ws = [{k: rnd(n, kk, sc=kk ** -0.5).bfloat16() for k, (n, kk) in dict(v=(D, D), o=(D, D), q=(D, D), f1=(Fd, D), f2=(D, Fd), kt=(D, D), vh=(D, D), o2=(D, D)).items()} for _ in range(NL)]
bias = {k: rnd(n, sc=0.1) for k, n in dict(v=D, o=D, q=D, f1=Fd, f2=D, vh=D).items()}
ln = [(1 + rnd(D, sc=0.1), rnd(D, sc=0.1)) for _ in range(3)]
x0, qp, s_h = rnd(Bq, D).bfloat16(), rnd(1, D, sc=0.3).bfloat16(), torch.rand(Bq, H, device=dev) + 0.5
seed = torch.full((1,), 1234, device=dev, dtype=torch.int64)
drop = lambda site: (seed, site, 0.1)
E = lambda *s: torch.empty(s, device=dev, dtype=torch.bfloat16)
bufs = [[E(Bq, D) for _ in range(12)] + [E(Bq, H, D), E(Bq, Fd)] for _ in range(NL)]
def layer(l, x):
    W = Ws[l]
    v, att, ta, t1, t1q, qc, attc, tb, t2, tcx, t3, hs, qpr, hh = bufs[l]
    ops.linear(x, W["v"], bias["v"], out=v)
    tr.gate_rows(v, att, drop=drop(1), drop_ld=H, drop_col_div=hd)
    ops.linear(att, W["o"], bias["o"], R=x, out=ta, drop=drop(2))
    ops.layernorm_add(ta, ln[0][0], ln[0][1], qp.expand(Bq, D), t1, t1q)
    ops.linear(t1q, W["q"], bias["q"], out=qc)
    ops.linear(qc[:, :hd], W["kt"][:, :hd], None, M=Bq, N=D, K=hd, batch=H, a_z_stride=hd, w_z_stride=hd,
               segs=[ops.Seg(out=qpr, ldo=D, rows_per_batch=1, out_batch_stride=qpr.stride(0), out_z_stride=D)])
    pooled = qpr.view(Bq, H * D)
    ops.linear(pooled[:, :D], W["vh"][:hd], None, M=Bq, N=hd, K=D, batch=H, a_z_stride=D, w_z_stride=hd * D,
               segs=[ops.Seg(out=attc, ldo=D, out_z_stride=hd)])
    tr.head_bias(attc, s_h, bias["vh"], H)
    ops.linear(attc, W["o2"], bias["o"], R=t1, out=tb, drop=drop(3))
    ops.layernorm(tb, ln[1][0], ln[1][1], out=t2)
    ops.linear(t2, W["f1"], bias["f1"], act=ops.ACT_RELU, out=hh, drop=drop(4))
    ops.linear(hh, W["f2"], bias["f2"], R=t2, out=tcx, drop=drop(5))
    ops.layernorm_add(tcx, ln[2][0], ln[2][1], qp.expand(Bq, D), t3, t1q)
    ops.layernorm(t3, ln[0][0], ln[0][1], out=hs)
    return t3
def program():
    x = x0
    for l in range(NL):
        x = layer(l, x)
def timeit(f, n=30):
    for _ in range(5): f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6
print(f"separate launches: {timeit(program):.1f} us for {NL} layers x 14 stages")
for nwg in (16, 32, 64, 128):
    ops.ChainRecorder.N_WG = nwg
    state = {}
    def chained():
        with ops.ChainRecorder(state, dev):
            program()
    print(f"made_chain, {nwg} workgroups: {timeit(chained):.1f} us")

# ---- where a stage's time goes (workgroup 0's cycle stamps; 16 workgroups)
from mgsv_amd import _lib
ops.ChainRecorder.N_WG = 16
stamps = torch.zeros(4 * 14 * NL + 64, dtype=torch.int64, device=dev)
state = {}
def chained_one():
    with ops.ChainRecorder(state, dev):
        program()
for _ in range(3): chained_one()
torch.cuda.synchronize()
_lib.lib().made_chain_debug_stamps(stamps.data_ptr())
chained_one()
torch.cuda.synchronize()
_lib.lib().made_chain_debug_stamps(None)
st = stamps.cpu().numpy().reshape(-1, 4)[:14 * NL]
names = ["v", "gate", "sa_out", "ln1+add", "q", "q'fold(z=8)", "vproj(z=8)", "head_bias", "ca_out", "ln2", "ff1", "ff2", "ln3+add", "norm"]
import numpy as np
work = (st[:, 1] - st[:, 0]).reshape(NL, 14)[1:].mean(0)
rel = (st[:, 2] - st[:, 1]).reshape(NL, 14)[1:].mean(0)
bar = (st[:, 3] - st[:, 2]).reshape(NL, 14)[1:].mean(0)
print("cycles per stage (workgroup 0): work / release fence / barrier + acquire")
for n, w, r, b in zip(names, work, rel, bar):
    print(f"  {n:12s} {w:8.0f} {r:8.0f} {b:8.0f}")

"""The micro-benchmark BASELINE.json's north_star names: the X-Pool cross-attention core (QK^T over the segments, softmax, P.V)
at B = 64 videos x 64 tracks, T_a = 512 segments, d = 512, bf16 (reference modules/transformer.py:110-119).  HBM-bound by
construction (4.3 GFLOP against >= 67 MB of K / U): reports time, GB/s against the 8 TB/s HBM peak and TFLOP/s
against the 2.5 PFLOP/s bf16 MFMA peak, for several key splits; then the same contraction at retrieval scale, where MFMA binds."""
import math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mgsv_amd import ops

dev, dt = torch.device("cuda"), torch.bfloat16


def timeit(fn, iters=20, warm=3):
    """hipGraph replay of `iters` calls: the Python / ctypes launch cost (10-20 us per call) stays out of the measurement."""
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(iters): fn()
    g.replay(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); g.replay(); g.replay(); e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / (2 * iters) * 1e3


def run(Nv, Nm, S, D, splits):
    q = torch.randn(Nv, D, device=dev).to(dt)
    k, u = torch.randn(Nm, S, D, device=dev).to(dt), torch.randn(Nm, S, D, device=dev).to(dt)
    lens = torch.randint(12, S + 1, (Nm,), device=dev)
    o = torch.empty(Nm * Nv, D, device=dev, dtype=dt)
    flops = 4.0 * Nv * Nm * S * D
    for name, mask in (("full-length tracks", None), ("ragged tracks (12..S segments)", (torch.arange(S, device=dev)[None] < lens[:, None]).float())):
        frac = 1.0 if mask is None else float(mask.mean())
        byts = 2.0 * Nm * S * D * 2 * frac + Nv * D * 2 + Nm * Nv * D * 2
        for ns in splits:
            part_o = torch.empty(Nm * ns * Nv * D, device=dev) if ns > 1 else None
            part_ml = torch.empty(Nm * ns * Nv * 4, device=dev) if ns > 1 else None
            t = timeit(lambda: ops.attention_wide(q.view(1, Nv, 1, D), k, u, o.view(Nm, Nv, 1, D), scale=1 / math.sqrt(D), key_mask=mask,
                                                  shared_q=True, n_split=ns, part_o=part_o, part_ml=part_ml))
            print(f"  Nv={Nv} Nm={Nm} S={S} D={D} {name:32s} n_split={ns} {'+ merge launch' if ns > 1 else '              '}: "
                  f"{t:8.1f} us  {byts / t / 1e3:7.1f} GB/s ({byts / t / 1e3 / 8000 * 100:4.1f}% of HBM peak)  "
                  f"{flops * frac / t / 1e6:7.1f} TFLOP/s ({flops * frac / t / 1e6 / 2500 * 100:4.1f}% of bf16 MFMA peak)", flush=True)


def run_inbatch(Nv, Nm, S, D):
    """made_xpool_inbatch (round 4): scores per (track, 128 segments), P.V per (track, 128 value columns) -- two launches, no f32 partials."""
    q = torch.randn(Nv, D, device=dev).to(dt)
    k, u = torch.randn(Nm, S, D, device=dev).to(dt), torch.randn(Nm, S, D, device=dev).to(dt)
    lens = torch.randint(12, S + 1, (Nm,), device=dev)
    o = torch.empty(Nm, Nv, D, device=dev, dtype=dt)
    ws = torch.zeros(ops.xpool_inbatch_ws_bytes(Nm, S), device=dev, dtype=torch.uint8)
    flops = 4.0 * Nv * Nm * S * D
    for name, mask in (("full-length tracks", None), ("ragged tracks (12..S segments)", (torch.arange(S, device=dev)[None] < lens[:, None]).float())):
        frac = 1.0 if mask is None else float(mask.mean())
        byts = 2.0 * Nm * S * D * 2 * frac + Nv * D * 2 + Nm * Nv * D * 2
        t = timeit(lambda: ops.xpool_inbatch(q, k, u, mask, o, scale=1 / math.sqrt(D), ws=ws))
        print(    f"  Nv={Nv} Nm={Nm} S={S} D={D} {name:32s} made_xpool_inbatch (2 launches): "
                  f"{t:8.1f} us  {byts / t / 1e3:7.1f} GB/s ({byts / t / 1e3 / 8000 * 100:4.1f}% of HBM peak)  "
              f"{flops * frac / t / 1e6:7.1f} TFLOP/s ({flops * frac / t / 1e6 / 2500 * 100:4.1f}% of bf16 MFMA peak)", flush=True)


print("in-batch X-Pool attention core, the shape north_star names:")
run_inbatch(64, 64, 512, 512)
run(64, 64, 512, 512, (1, 2, 4, 8))
print("the scripts' native shape:")
run_inbatch(64, 64, 96, 256)
run(64, 64, 96, 256, (1, 2, 4))
print("retrieval scale (unfused attention core only; the product path is made_xpool_fused, tools/xpool_only.py):")
run(4096, 128, 96, 256, (1,))

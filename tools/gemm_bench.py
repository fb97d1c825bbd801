"""Micro-benchmark of made_linear / made_attention on the hot path's shapes (HIP events, one process).
    python tools/gemm_bench.py [--dtype bf16]"""
import argparse, math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mgsv_amd import ops
from mgsv_amd.ops import Seg

def timeit(fn, iters=20, warm=3):
    for _ in range(warm): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3   # us

def main():
    ap = argparse.ArgumentParser(); ap.add_argument("--dtype", default="bf16"); a = ap.parse_args()
    dt = torch.bfloat16 if a.dtype == "bf16" else torch.float32
    dev = torch.device("cuda")
    shapes = [(34688, 512, 512), (34688, 1536, 512), (34688, 1024, 512), (34688, 512, 1024), (34688, 6144, 512),
              (32768, 512, 768), (1920, 512, 512), (64, 512, 512), (64, 1024, 512), (8192, 8192, 8192) if a.dtype == "bf16" else (4096, 4096, 4096)]
    for M, N, K in shapes:
        A = torch.randn(M, K, device=dev).to(dt); W = (torch.randn(N, K, device=dev) / math.sqrt(K)).to(dt)
        b = torch.randn(N, device=dev); out = torch.empty(M, N, device=dev, dtype=dt); R = torch.randn(M, N, device=dev).to(dt)
        t0 = timeit(lambda: ops.linear(A, W, b, out=out))
        t1 = timeit(lambda: ops.linear(A, W, b, out=out, act=ops.ACT_RELU, R=R))
        fl = 2.0 * M * N * K
        print(f"linear {a.dtype} M={M:6d} N={N:5d} K={K:5d}: plain {t0:9.1f} us {fl/t0/1e6:8.1f} TF | relu+res {t1:9.1f} us {fl/t1/1e6:8.1f} TF", flush=True)
    # transposed segment
    M, N, K, B, T = 34688, 1536, 512, 64, 542
    A = torch.randn(M, K, device=dev).to(dt); W = (torch.randn(N, K, device=dev) / math.sqrt(K)).to(dt); b = torch.randn(N, device=dev)
    P = torch.randn(M, K, device=dev).to(dt)
    qk = torch.empty(M, 1024, device=dev, dtype=dt); vt = torch.zeros(B, 512, 576, device=dev, dtype=dt)
    t = timeit(lambda: ops.linear(A, W, b, A2=P, segs=[Seg(out=qk, use_a2=True), Seg(out=vt, col_begin=1024, transposed=True, ldo=576, rows_per_batch=T, out_batch_stride=512 * 576)]))
    print(f"in_proj (+pos, V^T) M={M} N={N} K={K}: {t:9.1f} us {2.0*M*N*K/t/1e6:8.1f} TF")
    for (Bq, H, hd, Lq, Lk) in [(64, 8, 64, 542, 542), (64, 8, 64, 512, 512), (64, 8, 64, 30, 30), (64, 8, 64, 1, 542)]:
        D = H * hd
        q = torch.randn(Bq, Lq, D, device=dev).to(dt); k = torch.randn(Bq, Lk, D, device=dev).to(dt)
        vt = torch.randn(Bq, D, ops.round_up(Lk, 64), device=dev).to(dt); o = torch.empty(Bq, Lq, D, device=dev, dtype=dt)
        km = torch.ones(Bq, Lk, device=dev)
        t = timeit(lambda: ops.attention(q, k, vt, o, H, key_mask=km, Lk=Lk))
        print(f"attention {a.dtype} B={Bq} H={H} hd={hd} Lq={Lq} Lk={Lk}: {t:9.1f} us {4.0*Bq*H*Lq*Lk*hd/t/1e6:8.1f} TF")

if __name__ == "__main__":
    main()

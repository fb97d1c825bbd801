"""Vendor-GEMM yardstick (tools only, never on the product path): torch.matmul (hipBLASLt / rocBLAS) in bf16 on the training
step's heaviest GEMM shapes, alone on the chip, beside made_linear / made_gemm_tn alone on the same operands.

    python tools/gemm_yardstick.py            # table on stdout
    rocprofv3 --kernel-trace --stats ... -- python3 tools/gemm_yardstick.py   # adds the vendor kernels' symbols (tile configs)

Every arm: 5 warm-up calls, then ROUNDS rounds of ITERS back-to-back calls between two events on the current stream, arms
interleaved per round (cdna_hip_programming.md rule 24); median and min of the per-call time.  Operands are uniform random
(rule 25).  Forward / dX form: C[M,N] = A[M,K] W[N,K]^T (NT).  dW form: C[N,K] = dY[M,N]^T X[M,K] (TN, f32 output for ours,
bf16 -> f32 for the vendor's, which is what autograd's mm does before the optimizer's cast)."""
import os, statistics, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from mgsv_amd import ops, ops_train

ROUNDS = int(os.environ.get("ROUNDS", 7))
ITERS = int(os.environ.get("ITERS", 20))
dev = torch.device("cuda", 0)


def arms_time(arms):
    """arms: {name: fn}; returns {name: (median_us, min_us)} of the per-call time over interleaved rounds."""
    for fn in arms.values():
        for _ in range(5):
            fn()
    torch.cuda.synchronize()
    res = {k: [] for k in arms}
    for _ in range(ROUNDS):
        for k, fn in arms.items():
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(ITERS):
                fn()
            e.record()
            e.synchronize()
            res[k].append(s.elapsed_time(e) / ITERS * 1e3)
    return {k: (statistics.median(v), min(v)) for k, v in res.items()}


def rnd(*shape):
    return (torch.rand(*shape, device=dev) * 2 - 1).to(torch.bfloat16)


def main():
    print(f"# torch {torch.__version__}  device {torch.cuda.get_device_name(0)}  TORCH_BLAS_PREFER_HIPBLASLT={os.environ.get('TORCH_BLAS_PREFER_HIPBLASLT')}"
          f"  rounds={ROUNDS} iters={ITERS}")
    Ms = [int(x) for x in os.environ.get("MS", "17920,34688").split(",")]
    nt = [(512, 512), (1024, 512), (1536, 512), (512, 1024), (512, 768)]           # (N, K)
    print("## forward / dX form  C[M,N] = A[M,K] W[N,K]^T  (bf16 in, bf16 out); us = median (min)")
    print(f"{'M':>6} {'N':>5} {'K':>5} | {'made_linear':>22} {'TF':>6} | {'torch.matmul':>22} {'TF':>6} | {'F.linear+bias':>22} | ours/vendor")
    for M in Ms:
        for N, K in nt:
            A, W, b = rnd(M, K), rnd(N, K) * (K ** -0.5), torch.randn(N, device=dev)
            bb = b.to(torch.bfloat16)
            out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
            out2 = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
            Wt = W.t()
            r = arms_time({
                "ours": lambda: ops.linear(A, W, b, out=out),
                "mm": lambda: torch.matmul(A, Wt, out=out2),
                "lin": lambda: torch.nn.functional.linear(A, W, bb),
            })
            fl = 2.0 * M * N * K
            print(f"{M:6d} {N:5d} {K:5d} | {r['ours'][0]:9.1f} ({r['ours'][1]:9.1f}) {fl / r['ours'][0] / 1e6:6.0f} | "
                  f"{r['mm'][0]:9.1f} ({r['mm'][1]:9.1f}) {fl / r['mm'][0] / 1e6:6.0f} | {r['lin'][0]:9.1f} ({r['lin'][1]:9.1f}) | "
                  f"{r['ours'][0] / r['mm'][0]:5.2f}", flush=True)
            # parity of the two arms on this operand set (bf16 outputs of f32 accumulations)
            err = (out.float() - (out2.float() + b)).abs().max().item()
            assert err < 0.1, err
    tn = [(512, 512), (1024, 512), (1536, 512), (512, 1024), (512, 768)]           # (N, K): dW [N, K]
    print("## dW form  C[N,K] = dY[M,N]^T X[M,K]  (bf16 in; ours accumulates into f32, vendor writes bf16 and f32)")
    print(f"{'M':>6} {'N':>5} {'K':>5} | {'made_gemm_tn':>22} {'TF':>6} | {'torch.matmul bf16':>22} {'TF':>6} | {'torch.matmul ->f32':>22} | ours/vendor")
    for M in Ms:
        for N, K in tn:
            dY, X = rnd(M, N), rnd(M, K)
            Cw = torch.zeros(N, K, device=dev, dtype=torch.float32)
            cs = torch.zeros(N, device=dev, dtype=torch.float32)
            Cv = torch.empty(N, K, device=dev, dtype=torch.bfloat16)
            dYt = dY.t()
            r = arms_time({
                "ours": lambda: ops_train.gemm_tn(dY, X, Cw, accumulate=True, colsum=cs),
                "mm": lambda: torch.matmul(dYt, X, out=Cv),
                "mm32": lambda: torch.matmul(dYt, X).float(),
            })
            fl = 2.0 * M * N * K
            print(f"{M:6d} {N:5d} {K:5d} | {r['ours'][0]:9.1f} ({r['ours'][1]:9.1f}) {fl / r['ours'][0] / 1e6:6.0f} | "
                  f"{r['mm'][0]:9.1f} ({r['mm'][1]:9.1f}) {fl / r['mm'][0] / 1e6:6.0f} | {r['mm32'][0]:9.1f} ({r['mm32'][1]:9.1f}) | "
                  f"{r['ours'][0] / r['mm'][0]:5.2f}", flush=True)
    # the grouped weight-gradient launch of one DETR-encoder layer (q|k, v, out, ffn1, ffn2 over the same rows) against five vendor calls
    M = Ms[0]
    probs = [(1024, 512), (512, 512), (512, 512), (1024, 512), (512, 1024)]
    ours = [(rnd(M, n), None, torch.zeros(n, k, device=dev), torch.zeros(n, device=dev)) for n, k in probs]
    Xs = {512: rnd(M, 512), 1024: rnd(M, 1024)}
    ours = [(a, Xs[k], c, s) for (a, _, c, s), (n, k) in zip(ours, probs)]
    outs = [torch.empty(n, k, device=dev, dtype=torch.bfloat16) for n, k in probs]
    def vendor():
        for (a, x, _, _), o in zip(ours, outs):
            torch.matmul(a.t(), x, out=o)
    r = arms_time({"ours": lambda: ops_train.gemm_tn_grouped(ours), "mm": vendor})
    fl = sum(2.0 * M * n * k for n, k in probs)
    print(f"## one encoder layer's five dW products, M={M}: made_gemm_tn_grouped {r['ours'][0]:.1f} us ({fl / r['ours'][0] / 1e6:.0f} TF) | "
          f"five torch.matmul {r['mm'][0]:.1f} us ({fl / r['mm'][0] / 1e6:.0f} TF)")


if __name__ == "__main__":
    main()

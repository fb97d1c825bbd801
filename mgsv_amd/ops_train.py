"""Tensor-level wrappers over the training (backward) entry points of the C ABI (include/made_hip.h).

Same rules as ops.py: PyTorch provides device memory and the current stream; every op enqueues hand-written HIP
kernels from libmade_hip.so and nothing falls back to ATen.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import torch

from . import _lib
from ._lib import BF16, F32, MadeAttnBwdArgs, MadeDropout, MadeGemmTNArgs, check, lib
from .ops import _f32, _p, _stream, _timed, dt_of

Tensor = torch.Tensor
Pair = Tuple[int, int]


def gemm_tn(A: Tensor, B: Tensor, Cout: Tensor, *, alpha: float = 1.0, accumulate: bool = False, split_m: Optional[int] = None,
            row_mask: Optional[Tensor] = None, colsum: Optional[Tensor] = None, batch: Pair = (1, 1),
            a_zs: Pair = (0, 0), b_zs: Pair = (0, 0), c_zs: Pair = (0, 0), mask_zs: Pair = (0, 0),
            colsum_zs: Pair = (0, 0)) -> Tensor:
    """Cout[N,K] (+)= alpha * A[M,N]^T B[M,K]  (made_gemm_tn).  A, B, Cout are the 2-D views of batch element 0; the
    *_zs pairs are the element strides of the two batch levels.  colsum[N] += alpha * column sums of A."""
    M, N = A.shape
    K = B.shape[1]
    assert B.shape[0] == M and tuple(Cout.shape) == (N, K), (A.shape, B.shape, Cout.shape)
    assert A.stride(1) == 1 and B.stride(1) == 1 and Cout.stride(1) == 1
    assert A.dtype == B.dtype
    nz = batch[0] * batch[1]
    if split_m is None:
        split_m = 1
        if accumulate:
            slab = 64 if A.dtype == torch.bfloat16 else 32
            tiles = ((N + 127) // 128) * ((K + 127) // 128) * nz
            split_m = max(1, min(512 // max(tiles, 1), (M + slab - 1) // slab, 256))
    a = MadeGemmTNArgs()
    a.A, a.B, a.C = _p(A), _p(B), _p(Cout)
    a.ab_dtype, a.c_dtype = dt_of(A), dt_of(Cout)
    a.M, a.N, a.K = M, N, K
    a.lda, a.ldb, a.ldc = A.stride(0), B.stride(0), Cout.stride(0)
    a.batch1, a.batch2 = batch
    a.a_zs1, a.a_zs2 = a_zs
    a.b_zs1, a.b_zs2 = b_zs
    a.c_zs1, a.c_zs2 = c_zs
    a.row_mask = _p(_f32(row_mask, "row_mask"))
    a.mask_zs1, a.mask_zs2 = mask_zs
    a.alpha, a.accumulate, a.split_m = float(alpha), int(bool(accumulate)), int(split_m)
    a.colsum = _p(_f32(colsum, "colsum"))
    a.colsum_zs1, a.colsum_zs2 = colsum_zs
    flops = 2.0 * M * N * K * nz
    nbytes = float((M * N + M * K) * A.element_size() * nz + N * K * Cout.element_size())
    _timed("made_gemm_tn", flops, nbytes, lambda: check(lib().made_gemm_tn(C.byref(a), _stream()), "made_gemm_tn"),
           f"M={M} N={N} K={K} z={nz}")
    return Cout


def dropout_desc(seed: int, site: int, p: float) -> MadeDropout:
    d = MadeDropout()
    d.seed, d.site, d.p = int(seed) & 0xFFFFFFFFFFFFFFFF, int(site) & 0xFFFFFFFF, float(p)
    return d


def attention_bwd(Q: Tensor, K: Tensor, V: Tensor, O: Tensor, dO: Tensor, dQ: Tensor, dK: Tensor, dV: Tensor, lse: Tensor,
                  delta: Tensor, H: int, *, key_mask: Optional[Tensor] = None, q_skip_mask: Optional[Tensor] = None,
                  scale: Optional[float] = None, drop=None) -> None:
    """Gradients of ops.attention (made_attention_bwd).  All [B, L, H*hd] views with unit inner stride."""
    import math
    for t in (Q, K, V, O, dO, dQ, dK, dV):
        assert t.dim() == 3 and t.stride(2) == 1 and t.dtype == Q.dtype
    B, Lq, D = Q.shape
    hd = D // H
    a = MadeAttnBwdArgs()
    a.Q, a.K, a.V, a.O, a.dO, a.dQ, a.dK, a.dV = _p(Q), _p(K), _p(V), _p(O), _p(dO), _p(dQ), _p(dK), _p(dV)
    a.lse, a.delta = _p(_f32(lse, "lse")), _p(_f32(delta, "delta"))
    assert delta.numel() >= B * H * Lq
    a.dtype, a.hd = dt_of(Q), hd
    a.B, a.H, a.Lq, a.Lk = B, H, Lq, K.shape[1]
    a.q_bs, a.ldq, a.k_bs, a.ldk, a.v_bs, a.ldv = Q.stride(0), Q.stride(1), K.stride(0), K.stride(1), V.stride(0), V.stride(1)
    a.o_bs, a.ldo, a.do_bs, a.lddo = O.stride(0), O.stride(1), dO.stride(0), dO.stride(1)
    a.dq_bs, a.lddq, a.dk_bs, a.lddk, a.dv_bs, a.lddv = dQ.stride(0), dQ.stride(1), dK.stride(0), dK.stride(1), dV.stride(0), dV.stride(1)
    a.key_mask = _p(_f32(key_mask, "key_mask"))
    a.q_skip_mask = _p(_f32(q_skip_mask, "q_skip_mask"))
    a.scale = (1.0 / math.sqrt(hd)) if scale is None else scale
    if drop is not None and drop[2] > 0.0:
        a.drop.seed, a.drop.site, a.drop.p = int(drop[0]), int(drop[1]), float(drop[2])
    flops = 10.0 * B * H * Lq * a.Lk * hd * 1.4          # 7 products of 2*Lq*Lk*hd
    _timed("made_attention_bwd", flops, float(Q.element_size() * B * D * (4 * Lq + 4 * a.Lk)),
           lambda: check(lib().made_attention_bwd(C.byref(a), _stream()), "made_attention_bwd"), f"B={B} H={H} hd={hd} Lq={Lq} Lk={a.Lk}")

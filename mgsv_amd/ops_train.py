"""Tensor-level wrappers over the training (backward) entry points of the C ABI (include/made_hip.h).

Same rules as ops.py: PyTorch provides device memory and the current stream; every op enqueues hand-written HIP
kernels from libmade_hip.so and nothing falls back to ATen.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import torch

from . import _lib
from ._lib import BF16, F32, MadeAttnBwdArgs, MadeDropout, MadeGemmTNArgs, MadeGemmTNGroup, check, lib
from .ops import _f32, _p, _stream, _timed, dt_of, set_drop

Tensor = torch.Tensor
Pair = Tuple[int, int]


_TN_SLAB_US = 1.3       # (0.7 / 2.5 / 4.0 alternated in the step: 4.70 / 4.72 / 4.73 against 4.69 ms; profiles/r06_ab_tn_splits.txt)


def _tn_splits(tiles: int, nslab: int, gathered: bool, cap: int) -> int:
    """Reduction splits of the 128 x 128-tile weight-gradient kernels (two workgroups per CU): the split count that minimises
        rounds of 512 workgroups x slabs per workgroup x 1.3 us  +  tiles x splits x 64 KB of f32 atomic adds at 1.2 TB/s
    (docs/EXPERIMENTS.md 3f-4: the atomic units' rate; a gathered batch keeps about 9 / 16 of its slabs).  Long reductions come out at the 16 splits
    measured best in rounds 2-3; short ones (the video tower's 1 920 rows, the 4 096 pair rows) at 2-8 instead of 14-32, whose atomic adds cost several
    times the products.  cap: most splits the caller allows; a workgroup holds at most 64 slabs' row indices."""
    if _lib.variant_env("MADE_TN_SPLIT_MODEL", "1") == "0":   # (measurement knob: rounds 2-5's rule -- fill two workgroups per CU, at least 8 splits)
        sp = max(8, (512 // max(tiles, 1)) // 8 * 8) if cap == 64 and tiles < 64 else max(1, 2048 // max(tiles, 1))
        while (nslab + sp - 1) // sp > 64:
            sp += 8
        return min(sp, max(1, nslab))
    live = max(1, (nslab * 9) // 16 if gathered else nslab)
    best, best_t = 1, float("inf")
    for sp in (1, 2, 3, 4, 6, 8, 16, 24, 32, 40, 48, 56, 64):
        if sp > max(1, cap) or sp > nslab:
            break
        if (nslab + sp - 1) // sp > 64:
            continue
        t = ((tiles * sp + 511) // 512) * ((live + sp - 1) // sp) * _TN_SLAB_US + tiles * sp * 0.0546
        if t < best_t:
            best, best_t = sp, t
    while (nslab + best - 1) // best > 64:                    # (nothing admissible above: take what the row-index budget needs)
        best += 1 if best < 8 else 8
    return best


def gemm_tn(A: Tensor, B: Tensor, Cout: Tensor, *, alpha: float = 1.0, accumulate: bool = False, split_m: Optional[int] = None,
            row_mask: Optional[Tensor] = None, row_groups: Optional[Tensor] = None, rows=None, colsum: Optional[Tensor] = None,
            batch: Pair = (1, 1),
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
            fast = A.dtype == torch.bfloat16 and N % 128 == 0 and K % 128 == 0 and nz == 1 and (row_mask is None or rows is not None)
            if fast:                                          # direct-to-LDS kernel, two workgroups per CU (8 splits or more: the tiles of a split on one XCD)
                split_m = _tn_splits(tiles, (M + slab - 1) // slab, rows is not None, 64)
            else:
                # (at least eight slabs per split: fewer do not pay for a tile of atomic adds -- one slab per split until round 6: 4.70 against 4.72 ms)
                split_m = max(1, min(512 // max(tiles, 1), ((M + slab - 1) // slab) // 8, 256))
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
    a.row_group_valid = _p(_f32(row_groups, "row_groups")) if (row_mask is not None and nz == 1) else None
    if rows is not None and nz == 1:                          # (row_index int32 [M], n_rows int32 [1]) from ops.row_index
        a.row_index, a.n_rows = _p(rows[0]), _p(rows[1])
    a.mask_zs1, a.mask_zs2 = mask_zs
    a.alpha, a.accumulate, a.split_m = float(alpha), int(bool(accumulate)), int(split_m)
    a.colsum = _p(_f32(colsum, "colsum"))
    a.colsum_zs1, a.colsum_zs2 = colsum_zs
    flops = 2.0 * M * N * K * nz
    nbytes = float((M * N + M * K) * A.element_size() * nz + N * K * Cout.element_size())
    _timed("made_gemm_tn", flops, nbytes, lambda: check(lib().made_gemm_tn(C.byref(a), _stream()), "made_gemm_tn"),
           ("rows", rows[1], M) if (rows is not None and nz == 1) else f"M={M} N={N} K={K} z={nz}")
    return Cout


def gemm_tn_grouped_workspace(device, nbytes: int) -> Tensor:
    """A workspace for gemm_tn_grouped (contents arbitrary).  One per stream: launches that share one must be ordered."""
    return torch.empty(int(nbytes), dtype=torch.uint8, device=device)


def gemm_tn_grouped(problems, rows=None, alpha: float = 1.0, split_m: Optional[int] = None, workspace=None) -> None:
    """Several weight gradients over the same rows in one launch (made_gemm_tn_grouped).  problems: list of (dY [M, N], X [M, K],
    dW [N, K] f32, db [N] f32 or None); bf16 operands, N and K multiples of 128.  rows = (row_index, n_rows) or None.
    workspace: a callable nbytes -> uint8 tensor (gemm_tn_grouped_workspace; the caller keeps one per stream) -- with it the 256 x 256-tile
    form exchanges the tile partials through it instead of adding them to dW with atomics (include/made_hip.h: MadeGemmTNGroup.workspace)."""
    assert 1 <= len(problems) <= 8
    M = problems[0][0].shape[0]
    g = MadeGemmTNGroup()
    g.n_problems, g.alpha, g.M = len(problems), float(alpha), M
    tiles, flops, nbytes = 0, 0.0, 0.0
    for i, (A, B, Cw, cs) in enumerate(problems):
        assert A.dim() == 2 and B.dim() == 2 and A.shape[0] == M and B.shape[0] == M and A.stride(1) == 1 and B.stride(1) == 1
        assert A.dtype == torch.bfloat16 and B.dtype == torch.bfloat16 and Cw.dtype == torch.float32 and Cw.stride(1) == 1
        N, K = A.shape[1], B.shape[1]
        assert tuple(Cw.shape) == (N, K) and N % 128 == 0 and K % 128 == 0, (A.shape, B.shape, Cw.shape)
        p = g.p[i]
        p.A, p.B, p.C, p.colsum = _p(A), _p(B), _p(Cw), _p(_f32(cs, "colsum"))
        p.N, p.K, p.lda, p.ldb, p.ldc = N, K, A.stride(0), B.stride(0), Cw.stride(0)
        tiles += (N // 128) * (K // 128)
        flops += 2.0 * M * N * K
        nbytes += float(M * (N + K) * 2 + N * K * 4)
    if rows is not None:
        g.row_index, g.n_rows = _p(rows[0]), _p(rows[1])
    import os
    big = (all(pr[0].shape[1] % 256 == 0 and pr[1].shape[1] % 256 == 0 for pr in problems) and 4096 <= M <= 36864
           and _lib.variant_env("MADE_TN_TILE", "256") == "256")
    if big:
        # 256 x 256 tiles, one eight-wave workgroup per CU, the (tile, slab) units dealt out evenly by the kernel itself
        g.tile_size = 256
        if split_m is None:
            split_m = 1
        # the tile partials meet in a workspace instead of being added to the gradients with atomics (MADE_TN256_ATOMIC_FLUSH=1, measurement knob: the atomics)
        if workspace is not None and _lib.variant_env("MADE_TN256_ATOMIC_FLUSH", "") in ("", "0"):
            need = int(lib().made_gemm_tn_grouped_workspace(C.byref(g)))
            if need > 0:
                ws = workspace(need)
                assert ws.dtype == torch.uint8 and ws.numel() >= need and ws.is_contiguous()
                from . import tape as _tape
                _tape.keep(ws)
                g.workspace, g.workspace_bytes = ws.data_ptr(), ws.numel()
    if split_m is None:
        split_m = _tn_splits(tiles, (M + 63) // 64, rows is not None, 64)
    g.split_m = int(split_m)
    _timed("made_gemm_tn", flops, nbytes, lambda: check(lib().made_gemm_tn_grouped(C.byref(g), _stream()), "made_gemm_tn_grouped"),
           ("rows", rows[1], M) if rows is not None else f"grouped x{len(problems)} M={M}")


def row_groups(mask: Tensor, out: Optional[Tensor] = None) -> Tensor:
    """flags per 32 rows of a flat [M] mask (made_row_groups): lets made_gemm_tn skip slabs of padded tokens."""
    m = mask.reshape(-1)
    M = m.numel()
    if out is None:
        out = torch.empty((M + 31) // 32, device=m.device, dtype=torch.float32)
    check(lib().made_row_groups(_p(_f32(m, "mask")), M, _p(out), _stream()), "made_row_groups")
    return out


def dropout_desc(seed: int, site: int, p: float) -> MadeDropout:
    d = MadeDropout()
    set_drop(d, (seed, site, p))
    return d


def attention_bwd(Q: Tensor, K: Tensor, V: Tensor, O: Tensor, dO: Tensor, dQ: Tensor, dK: Tensor, dV: Tensor, lse: Tensor,
                  delta: Tensor, H: int, *, key_mask: Optional[Tensor] = None, q_skip_mask: Optional[Tensor] = None,
                  scale: Optional[float] = None, drop=None, order: Optional[Tensor] = None, keep_bits: Optional[Tensor] = None) -> None:
    """Gradients of ops.attention (made_attention_bwd).  All [B, L, H*hd] views with unit inner stride.  keep_bits: the decisions the
    forward call stored (same mask as re-drawing it; bf16 path)."""
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
    if order is not None:
        assert order.dtype == torch.int32 and order.numel() == a.B and order.is_contiguous()
        a.batch_order = _p(order)
    a.scale = (1.0 / math.sqrt(hd)) if scale is None else scale
    if drop is not None and drop[2] > 0.0:
        set_drop(a.drop, drop)
        if keep_bits is not None:
            assert keep_bits.dtype == torch.int32 and keep_bits.dim() == 2 and keep_bits.is_contiguous()
            assert keep_bits.shape[0] >= B * H * ((a.Lk + 31) // 32) and keep_bits.shape[1] >= Lq
            a.keep_bits, a.ld_bits = _p(keep_bits), keep_bits.shape[1]
    flops = 10.0 * B * H * Lq * a.Lk * hd                # five products of 2*Lq*Lk*hd (S, dP, dV, dK, dQ): cdna_hip_programming.md, attention backward
    _timed("made_attention_bwd", flops, float(Q.element_size() * B * D * (4 * Lq + 4 * a.Lk)),
           lambda: check(lib().made_attention_bwd(C.byref(a), _stream()), "made_attention_bwd"),
           ("attn", key_mask, q_skip_mask) if (key_mask is not None and key_mask.dim() == 2) else f"B={B} H={H} hd={hd} Lq={Lq} Lk={a.Lk}")


def _drop_ptr(drop):
    if drop is None or drop[2] <= 0.0:
        return None
    return C.byref(dropout_desc(*drop))


def _dt(t: Optional[Tensor]) -> int:
    return dt_of(t) if t is not None else F32


def layernorm_bwd(x: Tensor, gamma: Tensor, dy: Tensor, dx: Tensor, *, dgamma: Optional[Tensor], dbeta: Optional[Tensor],
                  add: Optional[Tensor] = None, dx_drop: Optional[Tensor] = None, drop=None, drop_ld: int = 0,
                  row_skip: Optional[Tensor] = None, eps: float = 1e-5) -> Tensor:
    """dx = LN'(dy) (+ add); dx_drop = dropout(dx); dgamma / dbeta accumulated.  x may be a [B,T,D] view (batch stride)."""
    if x.dim() == 3:
        rpb, xbs, ldx = x.shape[1], x.stride(0), x.stride(1)
        rows, D = x.shape[0] * x.shape[1], x.shape[2]
    else:
        rpb, xbs, ldx = 0, 0, x.stride(0)
        rows, D = x.shape
    assert dy.dim() == 2 and dx.dim() == 2 and dy.shape[0] == rows and dx.shape[0] == rows
    check(lib().made_layernorm_bwd(_p(x), dt_of(x), ldx, rpb, xbs, _p(_f32(gamma, "gamma")), _p(dy), dt_of(dy), dy.stride(0),
                                   _p(add), _dt(add), add.stride(0) if add is not None else 0,
                                   _p(dx), dt_of(dx), dx.stride(0),
                                   _p(dx_drop), dx_drop.stride(0) if dx_drop is not None else 0, _drop_ptr(drop), drop_ld,
                                   _p(_f32(dgamma, "dgamma")), _p(_f32(dbeta, "dbeta")), rows, D, eps,
                                   _p(_f32(row_skip, "row_skip")), _stream()), "made_layernorm_bwd")
    return dx


def dec_stage_bwd(xa: Tensor, gamma_a: Tensor, dy: Tensor, W: Tensor, out: Tensor, *, dgamma_a: Optional[Tensor], dbeta_a: Optional[Tensor],
                  xb: Optional[Tensor] = None, gamma_b: Optional[Tensor] = None, dgamma_b: Optional[Tensor] = None, dbeta_b: Optional[Tensor] = None,
                  add: Optional[Tensor] = None, dx_out: Optional[Tensor] = None, a_out: Optional[Tensor] = None, drop_a=None,
                  G: Optional[Tensor] = None, gate_scale: float = 1.0, drop_o=None, drop_o_ld: int = 0, drop_o_col_div: int = 1,
                  R: Optional[Tensor] = None, eps: float = 1e-5) -> Tensor:
    """A LayerNorm backward (or two stacked ones: xb / gamma_b / add) in the prologue of the dX product that consumes it
    (made_dec_stage_bwd): dx = LN'(dy; xa, gamma_a) -> dx_out, A = dropout(dx; drop_a) -> a_out, out = dropout(A W^T * [G != 0] * gate_scale;
    drop_o) + R.  All row tensors bf16; W [N, K]; the norm's width K = 256 or 512."""
    from . import _lib as L
    M, K = xa.shape
    N = W.shape[0]
    for t in (xa, dy, xb, add, dx_out, a_out, W, G, R, out):
        assert t is None or (t.dim() == 2 and t.dtype == torch.bfloat16 and t.stride(1) == 1)
    assert W.shape[1] == K and tuple(out.shape) == (M, N)
    a = L.MadeDecStageBwdArgs()
    a.xa, a.gamma_a, a.ldxa = _p(xa), _p(_f32(gamma_a, "gamma_a")), xa.stride(0)
    if xb is not None:
        a.xb, a.gamma_b, a.ldxb = _p(xb), _p(_f32(gamma_b, "gamma_b")), xb.stride(0)
    a.dy, a.lddy = _p(dy), dy.stride(0)
    if add is not None:
        a.add, a.ldadd = _p(add), add.stride(0)
    if dx_out is not None:
        a.dx_out, a.lddx = _p(dx_out), dx_out.stride(0)
    if a_out is not None:
        a.a_out, a.lda_out = _p(a_out), a_out.stride(0)
    a.dgamma_a, a.dbeta_a = _p(_f32(dgamma_a, "dgamma_a")), _p(_f32(dbeta_a, "dbeta_a"))
    a.dgamma_b, a.dbeta_b = _p(_f32(dgamma_b, "dgamma_b")), _p(_f32(dbeta_b, "dbeta_b"))
    if drop_a is not None and drop_a[2] > 0.0:
        set_drop(a.drop_a, drop_a)
    a.drop_a_ld = K
    a.W, a.ldw = _p(W), W.stride(0)
    if G is not None:
        a.G, a.ldg = _p(G), G.stride(0)
    a.gate_scale, a.eps = float(gate_scale), float(eps)
    if drop_o is not None and drop_o[2] > 0.0:
        set_drop(a.drop_o, drop_o)
    a.drop_o_ld, a.drop_o_col_div = (drop_o_ld or N), int(drop_o_col_div)
    if R is not None:
        a.R, a.ldr = _p(R), R.stride(0)
    a.out, a.ldo = _p(out), out.stride(0)
    a.M, a.N, a.K = M, N, K
    _timed("dec_stage_bwd_kernel", 2.0 * M * N * K, float(M * K * 2 * 2 * ((N + 31) // 32) + N * K * 2 + M * N * 2),
           lambda: check(lib().made_dec_stage_bwd(C.byref(a), _stream()), "made_dec_stage_bwd"), f"M={M} N={N} K={K}")
    return out


def layernorm_bwd2(xa: Tensor, gamma_a: Tensor, xb: Tensor, gamma_b: Tensor, dy: Tensor, dx: Tensor, *, dgamma_a: Tensor, dbeta_a: Tensor,
                   dgamma_b: Tensor, dbeta_b: Tensor, add: Optional[Tensor] = None, dx_drop: Optional[Tensor] = None, drop=None,
                   drop_ld: int = 0, eps: float = 1e-5) -> Tensor:
    """Two chained LayerNorms backward (made_layernorm_bwd2): g = LN_b'(dy; xb) + add, dx = LN_a'(g; xa), dx_drop = dropout(dx)."""
    rows, D = xa.shape
    for t in (xa, xb, dy, dx, add, dx_drop):
        assert t is None or (t.dim() == 2 and t.dtype == xa.dtype and t.stride(1) == 1 and t.shape[0] >= rows)
    check(lib().made_layernorm_bwd2(_p(xa), _p(_f32(gamma_a, "gamma_a")), xa.stride(0), _p(xb), _p(_f32(gamma_b, "gamma_b")), xb.stride(0),
                                    _p(dy), dy.stride(0), _p(add), add.stride(0) if add is not None else 0, _p(dx), dx.stride(0),
                                    _p(dx_drop), dx_drop.stride(0) if dx_drop is not None else 0, _drop_ptr(drop), drop_ld, dt_of(xa),
                                    _p(_f32(dgamma_a, "dgamma_a")), _p(_f32(dbeta_a, "dbeta_a")), _p(_f32(dgamma_b, "dgamma_b")),
                                    _p(_f32(dbeta_b, "dbeta_b")), rows, D, eps, _stream()), "made_layernorm_bwd2")
    return dx


def pool_bwd(mean: Tensor, dvec: Tensor, mask: Tensor, out: Tensor, in1: Optional[Tensor] = None, in2: Optional[Tensor] = None,
             eps: float = 1e-12) -> Tensor:
    """out[b,t,:] = mask * (in1 + in2 + d masked-mean / normalise); in1 / in2 / out are [B,T,D] views."""
    B, T, D = out.shape
    def bs(t): return (t.stride(0), t.stride(1)) if t is not None else (0, 0)
    check(lib().made_pool_bwd(_p(_f32(mean, "mean")), _p(_f32(dvec, "dvec")), _p(_f32(mask, "mask")),
                              _p(in1), _dt(in1), *bs(in1), _p(in2), _dt(in2), *bs(in2),
                              _p(out), dt_of(out), *bs(out), B, T, D, eps, _stream()), "made_pool_bwd")
    return out


def l2norm_bwd(x: Tensor, dy: Tensor, dx: Optional[Tensor] = None, *, accumulate: bool = False, dx_alt: Optional[Tensor] = None,
               eps: float = 1e-12, dy_rows_per: int = 1) -> None:
    rows, D = x.shape
    check(lib().made_l2norm_bwd(_p(x), dt_of(x), x.stride(0), _p(_f32(dy, "dy")), dy.stride(0), dy_rows_per,
                                _p(dx), dx.stride(0) if dx is not None else 0, int(accumulate),
                                _p(dx_alt), _dt(dx_alt), dx_alt.stride(0) if dx_alt is not None else 0,
                                rows, D, eps, _stream()), "made_l2norm_bwd")


def clip_loss_bwd(sims: Tensor, logit_scale: Tensor, weight: float, upstream: Optional[Tensor], lse_ws: Tensor, dsims: Tensor,
                  dsims_t: Optional[Tensor], d_logit_scale: Optional[Tensor], accumulate: bool = False,
                  row_exclude: Optional[Tensor] = None) -> None:
    n = sims.shape[0]
    assert lse_ws.numel() >= 2 * n and dsims.is_contiguous()
    assert row_exclude is None or (row_exclude.shape == sims.shape and row_exclude.is_contiguous())
    check(lib().made_clip_loss_bwd(_p(_f32(sims, "sims")), sims.stride(0), n, _p(logit_scale), float(weight), _p(upstream),
                                   _p(lse_ws), _p(dsims), _p(dsims_t), int(accumulate), _p(d_logit_scale),
                                   _p(_f32(row_exclude, "row_exclude")), _stream()),
          "made_clip_loss_bwd")


def xpool_tail_bwd(y: Tensor, gamma: Tensor, beta: Tensor, video: Tensor, dsims: Tensor, dy: Tensor, Nm: int, Nv: int, *,
                   dy_drop: Optional[Tensor] = None, drop=None, dgamma: Optional[Tensor] = None, dbeta: Optional[Tensor] = None,
                   dvideo: Optional[Tensor] = None, dpool: Optional[Tensor] = None, dpool_scale: float = 1.0, eps: float = 1e-5) -> None:
    """dpool [Nm, D] f32: an extra gradient of the pooled rows, dpool[m] * dpool_scale for every video n (include/made_hip.h)."""
    D = y.shape[1]
    check(lib().made_xpool_tail_bwd(_p(y), dt_of(y), y.stride(0), _p(gamma), _p(beta), _p(_f32(video, "video")), video.stride(0),
                                    _p(_f32(dsims, "dsims")), dsims.stride(0), _p(dy), dt_of(dy), dy.stride(0), _p(dy_drop),
                                    _drop_ptr(drop), _p(dgamma), _p(dbeta), _p(dvideo), dvideo.stride(0) if dvideo is not None else 0,
                                    _p(_f32(dpool, "dpool")), dpool.stride(0) if dpool is not None else 0, float(dpool_scale),
                                    Nm, Nv, D, eps, _stream()), "made_xpool_tail_bwd")


def softmax_bwd(S: Tensor, dP: Tensor, mask: Optional[Tensor], rows_per_mask: int, scale: float, Pd: Tensor, dS: Tensor,
                dSt: Optional[Tensor], rows_per_batch: int, L: int, *, extra: Optional[Tensor] = None, drop=None,
                ldo: Optional[int] = None, ldt: int = 0, out_batch_stride: int = 0, t_batch_stride: int = 0) -> None:
    """S, dP [rows, >= L] f32; Pd / dS: element (z, i, k) at z*out_batch_stride + i*ldo + k; dSt: z*t_batch_stride + k*ldt + i."""
    rows = S.shape[0]
    check(lib().made_softmax_bwd(_p(_f32(S, "S")), S.stride(0), _p(dP), dP.stride(0), _p(mask), rows_per_mask,
                                 _p(extra), float(scale), _drop_ptr(drop), _p(Pd), _p(dS), _p(dSt), dt_of(Pd),
                                 Pd.stride(-2) if ldo is None else ldo, ldt, out_batch_stride, t_batch_stride,
                                 rows, rows_per_batch, L, _stream()), "made_softmax_bwd")


def attention_wide_bwd(Q: Tensor, dO: Tensor, O: Tensor, K: Tensor, V: Tensor, lse: Tensor, Pd: Tensor, dS: Tensor, dQ: Tensor, *, scale: float,
                       key_mask: Optional[Tensor] = None, ssum: Optional[Tensor] = None, extra: Optional[Tensor] = None,
                       dattc: Optional[Tensor] = None, vbias: Optional[Tensor] = None, hd: int = 0, drop=None, n_split: int = 1,
                       part_dq: Optional[Tensor] = None) -> Tensor:
    """Backward of the decoder's memory-space cross-attention in ONE launch (made_attention_wide_bwd).  Q / dO / O / dQ [B, NQ, D] bf16
    (NQ <= 8), K / V [B, L, D] bf16, lse / ssum / extra [B, NQ] f32, Pd / dS [B, NQ, ld_p >= L] bf16 views."""
    assert Q.dim() == 3 and dO.shape == Q.shape and O.shape == Q.shape and dQ.shape == Q.shape and K.dim() == 3 and V.shape == K.shape
    B, NQ, D = Q.shape
    L = K.shape[1]
    for t in (Q, dO, O, K, V, Pd, dS, dQ):
        assert t.dtype == torch.bfloat16 and t.stride(-1) == 1
    assert Pd.dim() == 3 and dS.dim() == 3 and Pd.stride() == dS.stride() and Pd.shape[0] == B and Pd.shape[1] == NQ
    a = _lib.MadeWideAttnBwdArgs()
    a.Q, a.dO, a.O, a.K, a.V = _p(Q), _p(dO), _p(O), _p(K), _p(V)
    a.key_mask, a.lse, a.ssum, a.extra = _p(_f32(key_mask, "key_mask")), _p(_f32(lse, "lse")), _p(_f32(ssum, "ssum")), _p(_f32(extra, "extra"))
    if dattc is not None:
        assert dattc.dim() == 2 and dattc.dtype == torch.bfloat16 and dattc.stride(1) == 1 and vbias is not None
        a.dattc, a.ld_dattc, a.vbias, a.hd = _p(dattc), dattc.stride(0), _p(_f32(vbias, "vbias")), hd
    a.Pd, a.dS, a.p_bs, a.ld_p = _p(Pd), _p(dS), Pd.stride(0), Pd.stride(1)
    a.dQ, a.dq_bs, a.ld_dq = _p(dQ), dQ.stride(0), dQ.stride(1)
    a.B, a.NQ, a.L, a.D = B, NQ, L, D
    a.q_bs, a.ld_q, a.do_bs, a.ld_do, a.o_bs, a.ld_o = Q.stride(0), Q.stride(1), dO.stride(0), dO.stride(1), O.stride(0), O.stride(1)
    a.k_bs, a.ldk, a.v_bs, a.ldv = K.stride(0), K.stride(1), V.stride(0), V.stride(1)
    a.scale = float(scale)
    if n_split > 1:
        assert part_dq is not None and part_dq.dtype == torch.float32 and part_dq.numel() >= B * n_split * NQ * D
        a.n_split, a.part_dq = n_split, _p(part_dq)
    if drop is not None and drop[2] > 0.0:
        set_drop(a.drop, drop)
    _timed("made_attention_wide_bwd", 6.0 * B * NQ * L * D, 2.0 * B * (2 * L * D + 4 * NQ * D + 2 * NQ * L),
                lambda: check(lib().made_attention_wide_bwd(C.byref(a), _stream()), "made_attention_wide_bwd"), f"B={B} NQ={NQ} L={L} D={D}")
    return dQ


def head_bias(x: Tensor, s: Tensor, bias: Tensor, H: int) -> None:
    rows, D = x.shape
    check(lib().made_head_bias(_p(x), dt_of(x), x.stride(0), _p(_f32(s, "s")), _p(bias), rows, H, D // H, _stream()), "made_head_bias")


def head_bias_bwd(dy: Tensor, s: Tensor, bias: Tensor, dbias: Tensor, ds: Tensor, H: int) -> None:
    rows, D = dy.shape
    check(lib().made_head_bias_bwd(_p(dy), dt_of(dy), dy.stride(0), _p(_f32(s, "s")), _p(bias), _p(dbias), _p(ds), rows, H, D // H,
                                   _stream()), "made_head_bias_bwd")


def add3(out: Tensor, a: Tensor, b: Optional[Tensor] = None, c: Optional[Tensor] = None, b_mod: int = 0) -> Tensor:
    for t in (out, a, b, c):
        assert t is None or t.is_contiguous()
    check(lib().made_add3(_p(out), dt_of(out), _p(a), dt_of(a), _p(b), _dt(b), _p(c), _dt(c), out.numel(), b_mod, _stream()), "made_add3")
    return out


def gate_rows(x: Tensor, out: Tensor, *, G: Optional[Tensor] = None, gate: int = 0, scale: float = 1.0, drop=None, drop_ld: int = 0,
              drop_col_div: int = 1, row_skip: Optional[Tensor] = None) -> Tensor:
    """out = dropout(x * act'(G) * scale), row-wise (made_gate_rows); x / G / out [rows, cols] with unit inner stride; dropout
    element index row*drop_ld + col // drop_col_div."""
    rows, cols = x.shape
    assert x.stride(1) == 1 and out.stride(1) == 1 and out.shape == x.shape and (G is None or (G.shape == x.shape and G.stride(1) == 1))
    check(lib().made_gate_rows(_p(x), dt_of(x), x.stride(0), _p(G), _dt(G), G.stride(0) if G is not None else 0, gate, scale,
                               _drop_ptr(drop), drop_ld, drop_col_div, _p(out), dt_of(out), out.stride(0), _p(_f32(row_skip, "row_skip")), rows, cols,
                               _stream()), "made_gate_rows")
    return out


def colsum(x: Tensor, out: Tensor) -> None:
    """out[c] += sum_rows x[row, c]."""
    rows, cols = x.shape
    check(lib().made_colsum(_p(x), dt_of(x), x.stride(0), rows, cols, _p(_f32(out, "out")), _stream()), "made_colsum")


def posbn_relu(x: Tensor, weight: Tensor, bias: Tensor, running_mean: Optional[Tensor], running_var: Optional[Tensor], momentum: float,
               batch_stats: bool, save_mean: Tensor, save_rstd: Tensor, out: Tensor, eps: float = 1e-5) -> Tensor:
    """out = relu(BatchNorm1d over positions(x)); x / out [B, T, F] (contiguous rows).  made_posbn_relu_fwd."""
    B, T, F = x.shape
    assert x.is_contiguous() and out.is_contiguous() and out.shape == x.shape
    check(lib().made_posbn_relu_fwd(_p(x), dt_of(x), F, _p(_f32(weight, "weight")), _p(_f32(bias, "bias")), _p(_f32(running_mean, "running_mean")),
                                    _p(_f32(running_var, "running_var")), float(momentum), float(eps), int(bool(batch_stats)),
                                    _p(_f32(save_mean, "save_mean")), _p(_f32(save_rstd, "save_rstd")), _p(out), dt_of(out), F, B, T, F, _stream()),
          "made_posbn_relu_fwd")
    return out


def posbn_relu_bwd(x: Tensor, y: Tensor, dy: Tensor, weight: Tensor, save_mean: Tensor, save_rstd: Tensor, batch_stats: bool, dx: Tensor,
                   dweight: Optional[Tensor], dbias: Optional[Tensor]) -> Tensor:
    """dx of posbn_relu (stored), dweight / dbias accumulated.  made_posbn_relu_bwd."""
    B, T, F = x.shape
    for t in (x, y, dy, dx):
        assert t.is_contiguous() and tuple(t.shape) == (B, T, F)
    check(lib().made_posbn_relu_bwd(_p(x), dt_of(x), F, _p(y), dt_of(y), F, _p(dy), dt_of(dy), F, _p(_f32(weight, "weight")),
                                    _p(_f32(save_mean, "save_mean")), _p(_f32(save_rstd, "save_rstd")), int(bool(batch_stats)),
                                    _p(dx), dt_of(dx), F, _p(_f32(dweight, "dweight")), _p(_f32(dbias, "dbias")), B, T, F, _stream()),
          "made_posbn_relu_bwd")
    return dx


def set_criterion_bwd(logits: Tensor, spans: Tensor, targets: Tensor, pi: Tensor, ti: Tensor, cnt: Tensor, pq: Optional[Tensor],
                      vid_sum: Optional[Tensor], empty_weight: Tensor, fg: int, weights: Tensor, upstream: Optional[Tensor],
                      d_logits: Tensor, d_spans: Tensor, d_pq: Optional[Tensor], d_vid_sum: Optional[Tensor],
                      temperature: float = 0.07, ld_out: int = 2, through_sigmoid: bool = False) -> None:
    nd, B, Q, _ = logits.shape
    G = targets.shape[1]
    Dc = pq.shape[-1] if pq is not None else 0
    check(lib().made_set_criterion_bwd(_p(logits), _p(spans), _p(targets), _p(pi), _p(ti), _p(cnt), _p(pq), _p(vid_sum),
                                       _p(empty_weight), nd, B, Q, G, Dc, fg, temperature, _p(weights), _p(upstream),
                                       _p(d_logits), _p(d_spans), ld_out, int(through_sigmoid), _p(d_pq), _p(d_vid_sum), _stream()),
          "made_set_criterion_bwd")

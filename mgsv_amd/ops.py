"""Tensor-level wrappers over the C ABI (include/made_hip.h).

PyTorch is used for device memory and the current stream only; every op here enqueues one
hand-written HIP kernel from libmade_hip.so and nothing falls back to ATen.  All tensors must live
on the GPU; shapes/strides are validated by the library (negative status -> MadeError).
"""
from __future__ import annotations

import ctypes as C
import math
from dataclasses import dataclass
from typing import Optional, Sequence

import torch

from . import _lib, tape as _tape
from ._lib import (ACT_GELU, ACT_NONE, ACT_QUICKGELU, ACT_RELU, ACT_SIGMOID, BF16, F32,  # noqa: F401
                   MadeAttnArgs, MadeDecStageArgs, MadeFinishArgs, MadeLinearArgs, MadeLinearSeg, MadeWideAttnArgs, check, lib)

Tensor = torch.Tensor


def dt_of(t: Tensor) -> int:
    if t.dtype == torch.float32:
        return F32
    if t.dtype == torch.bfloat16:
        return BF16
    raise TypeError(f"unsupported dtype {t.dtype}")


def torch_dtype(code: int):
    return torch.float32 if code == F32 else torch.bfloat16


def _p(t: Optional[Tensor]):
    if t is None:
        return None
    if not t.is_cuda:
        raise _lib.MadeError("libmade_hip ops need GPU tensors (no CPU fallback)")
    if _tape._recording is not None:                          # a launch tape holds raw pointers: keep what it points at alive
        _tape._recording._keep.append(t)
    return t.data_ptr()


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _f32(t: Optional[Tensor], name: str) -> Optional[Tensor]:
    if t is not None and (t.dtype != torch.float32 or not t.is_contiguous()):
        raise TypeError(f"{name} must be a contiguous float32 tensor")
    return t


def set_drop(d, drop) -> None:
    """fill a MadeDropout from (seed, site, p); `seed` is an int, or a 1-element int64 device tensor the kernels read at run
    time (MadeDropout.seed_device: a captured graph of the training step then draws new masks on every replay)."""
    seed, site, p = drop
    if isinstance(seed, Tensor):
        assert seed.dtype == torch.int64 and seed.numel() == 1 and seed.is_cuda
        d.seed, d.seed_device = 0, seed.data_ptr()
    else:
        d.seed, d.seed_device = int(seed) & 0xFFFFFFFFFFFFFFFF, None
    d.site, d.p = int(site) & 0xFFFFFFFF, float(p)


def round_up(x: int, m: int) -> int:
    return (x + m - 1) // m * m


class KernelTimer:
    """Optional per-launch timing with HIP events on the launch stream (bench.py's roofline leg).
    `with KernelTimer() as kt: engine.forward(...)` brackets every made_linear / made_attention launch
    with a pair of events; `kt.summary()` synchronises and returns per-kind totals."""

    def __init__(self):
        self.records = []

    def __enter__(self):
        global _timer
        self._prev, _timer = _timer, self
        return self

    def __exit__(self, *exc):
        global _timer
        _timer = self._prev

    def summary(self):
        torch.cuda.synchronize()
        out = {}
        frac_cache = {}
        for kind, s, e, flops, nbytes, desc in self.records:
            d = out.setdefault(kind, dict(launches=0, ms=0.0, flops=0.0, bytes=0.0, flops_nominal=0.0))
            d["launches"] += 1
            d["ms"] += s.elapsed_time(e)
            frac = 1.0
            if isinstance(desc, tuple) and desc[0] == "rows":   # row gather: (tag, n_rows tensor, M): only the valid rows are computed
                frac = min(1.0, (math.ceil(int(desc[1].item()) / 128) * 128) / max(desc[2], 1))
            elif isinstance(desc, tuple) and desc[0] == "attn":  # (tag, key mask [B,Lk] | None, query skip mask [B,Lq] | None): ragged batch
                key = ("attn", desc[1].data_ptr() if desc[1] is not None else 0, desc[2].data_ptr() if desc[2] is not None else 0)
                if key not in frac_cache:
                    fk = (desc[1] != 0).float().mean(dim=1) if desc[1] is not None else 1.0
                    fq = (desc[2] != 0).float().mean(dim=1) if desc[2] is not None else 1.0
                    f = fk * fq
                    frac_cache[key] = float(f.mean().item()) if isinstance(f, torch.Tensor) else 1.0
                frac = frac_cache[key]
            elif isinstance(desc, tuple) and desc[0] == "frac":  # (tag, executed fraction) computed by the caller
                frac = float(desc[1])
            elif isinstance(desc, tuple):               # (text, skip mask, rows per tile): count only tiles that were executed
                _, mask, tile = desc
                key = (mask.data_ptr(), mask.numel(), tile)
                if key not in frac_cache:
                    n = mask.numel()
                    pad = (-n) % tile
                    m = torch.nn.functional.pad(mask.reshape(-1) != 0, (0, pad)).view(-1, tile)
                    frac_cache[key] = float(m.any(dim=1).float().mean().item())
                frac = frac_cache[key]
            d["flops"] += flops * frac
            d["bytes"] += nbytes * frac
            d["flops_nominal"] += flops
        return out


_timer: Optional[KernelTimer] = None


def _timed(kind: str, flops: float, nbytes: float, launch, desc: str = ""):
    if _timer is None:
        return launch()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    r = launch()
    e.record()
    _timer.records.append((kind, s, e, flops, nbytes, desc))
    return r


@dataclass
class Seg:
    """One column segment of made_linear's output (include/made_hip.h: MadeLinearSeg)."""
    out: Tensor
    col_begin: int = 0
    ldo: Optional[int] = None          # default: out.stride(-2)
    transposed: bool = False
    rows_per_batch: int = 0
    out_batch_stride: int = 0
    out_z_stride: int = 0
    use_a2: bool = False


# made_linear_variant codes (include/made_hip.h: MadeLinearVariant) -> KernelTimer kinds = rocprofv3 kernel symbols
LINEAR_VARIANTS = {0: "linear_f32", 1: "linear_f32in_bf16", 2: "linear_kernel<bf16,bf16>", 3: "linear_tiny_kernel", 4: "linear_skinny_kernel",
                   5: "linear_glds_kernel<3,.,128>", 6: "linear_glds_kernel<1,.,64>", 7: "linear_glds_kernel<1,.,128>",
                   8: "linear_big_kernel<256>", 9: "linear_t16_kernel", 10: "linear_big_kernel<128>",
                   11: "linear_glds_f32<64>", 12: "linear_glds_f32<128>"}


LINEAR_LOG = None                                             # a list: every linear() call appends its form (tools/linear_calls.py)


def linear(A: Tensor, W: Tensor, bias: Optional[Tensor] = None, *, out: Optional[Tensor] = None,
           segs: Optional[Sequence[Seg]] = None, A2: Optional[Tensor] = None, a2_row_mod: int = 0,
           a2_replace: bool = False, a_row_mask: Optional[Tensor] = None, act: int = ACT_NONE, R: Optional[Tensor] = None,
           r_row_mod: int = 0, out_row_mask: Optional[Tensor] = None, out_dtype: Optional[torch.dtype] = None,
           tile_skip_mask: Optional[Tensor] = None, batch: int = 1, a_z_stride: int = 0, w_z_stride: int = 0, M: Optional[int] = None,
           N: Optional[int] = None, K: Optional[int] = None, gate: int = 0, G: Optional[Tensor] = None, gate_scale: float = 1.0,
           Zout: Optional[Tensor] = None, drop=None, drop_ld: Optional[int] = None, rows=None, drop_col_div: int = 1,
           bias_row_scale: Optional[Tensor] = None, bias_z_stride: int = 0) -> Tensor:
    """out = act(A' W^T + bias) (+R).  A [M,K] (row stride free, unit inner stride), W [N,K].
    Training extras: Zout receives the pre-activation; gate/G multiply by act'(G) * gate_scale; drop = (seed, site, p)
    applies the stateless dropout of include/made_hip.h after act/gate (element index row * drop_ld + col)."""
    assert A.dim() == 2 and W.dim() == 2 and A.stride(1) == 1 and W.stride(1) == 1
    M = A.shape[0] if M is None else M
    K = A.shape[1] if K is None else K
    N = W.shape[0] if N is None else N
    a = MadeLinearArgs()
    a.A, a.a_dtype, a.w_dtype, a.lda = _p(A), dt_of(A), dt_of(W), A.stride(0)
    if A2 is not None:
        assert A2.dim() == 2 and A2.stride(1) == 1 and A2.dtype == A.dtype
        a.A2, a.lda2, a.a2_row_mod = _p(A2), A2.stride(0), a2_row_mod
        a.a2_replace = 1 if a2_replace else 0
    a.a_row_mask = _p(_f32(a_row_mask, "a_row_mask"))
    a.W, a.ldw = _p(W), W.stride(0)
    a.bias = _p(_f32(bias, "bias"))
    a.M, a.N, a.K = M, N, K
    a.batch, a.a_z_stride, a.w_z_stride = batch, a_z_stride, w_z_stride
    a.act = act
    if R is not None:
        assert R.dim() == 2 and R.stride(1) == 1
        a.R, a.r_dtype, a.ldr, a.r_row_mod = _p(R), dt_of(R), R.stride(0), r_row_mod
    a.out_row_mask = _p(_f32(out_row_mask, "out_row_mask"))
    a.tile_skip_mask = _p(_f32(tile_skip_mask, "tile_skip_mask"))
    if gate:
        assert G is not None and G.dim() == 2 and G.stride(1) == 1
        a.G, a.g_dtype, a.gate, a.ldg, a.gate_scale = _p(G), dt_of(G), gate, G.stride(0), float(gate_scale)
    if Zout is not None:
        assert Zout.dim() == 2 and Zout.stride(1) == 1
        a.Zout, a.z_dtype, a.ldz = _p(Zout), dt_of(Zout), Zout.stride(0)
    if drop is not None and drop[2] > 0.0:
        set_drop(a.drop, drop)
        a.drop_ld = N if drop_ld is None else drop_ld
        a.drop_col_div = int(drop_col_div)
    if rows is not None:                                      # row gather: (row_index int32 [M], n_rows int32 [1]) from row_index()
        a.row_index, a.n_rows = _p(rows[0]), _p(rows[1])
    if bias_row_scale is not None:                            # bias[z * bias_z_stride + col] * bias_row_scale[row, z] (tiny-M kernel)
        assert tuple(bias_row_scale.shape) == (M, batch)
        a.bias_row_scale, a.bias_z_stride = _p(_f32(bias_row_scale, "bias_row_scale")), bias_z_stride
    if segs is None:
        if out is None:
            out = torch.empty((M, N) if batch == 1 else (batch, M, N), device=A.device,
                              dtype=out_dtype or W.dtype)
        segs = [Seg(out=out, out_z_stride=(out.stride(0) if (batch > 1 and out.dim() == 3) else 0))]
    a.nseg = len(segs)
    for i, s in enumerate(segs):
        sg = a.seg[i]
        sg.col_begin, sg.out, sg.out_dtype = s.col_begin, _p(s.out), dt_of(s.out)
        sg.transposed = 1 if s.transposed else 0
        sg.ldo = s.ldo if s.ldo is not None else s.out.stride(-2)
        sg.rows_per_batch, sg.out_batch_stride, sg.out_z_stride = s.rows_per_batch, s.out_batch_stride, s.out_z_stride
        sg.use_a2 = 1 if s.use_a2 else 0
    esz = 4 if a.w_dtype == F32 else 2
    kind = "linear_" + ("f32" if a.w_dtype == F32 else ("bf16" if a.a_dtype == BF16 else "f32in_bf16"))
    if _timer is not None and kind == "linear_bf16":          # timed runs label the launch with the kernel it dispatches to
        kind = LINEAR_VARIANTS[lib().made_linear_variant(C.byref(a))]
    elif _timer is not None and kind == "linear_f32" and lib().made_linear_variant(C.byref(a)) >= 11:
        kind = LINEAR_VARIANTS[lib().made_linear_variant(C.byref(a))]
    flops = 2.0 * M * N * K * batch
    nbytes = batch * (M * K * (4 if a.a_dtype == F32 else 2) + N * K * esz + M * N * esz)
    desc = f"M={M} N={N} K={K} z={batch} nseg={len(segs)} a2={int(A2 is not None)} R={int(R is not None)} act={act} tr={int(any(s_.transposed for s_ in segs))}"
    if LINEAR_LOG is not None:                                # (tools/linear_calls.py: which forms of made_linear a step uses)
        LINEAR_LOG.append((LINEAR_VARIANTS.get(lib().made_linear_variant(C.byref(a)), "?") if kind in ("linear_bf16", "linear_f32") else kind, M, N, K, batch,
                           f"nseg={len(segs)} a2={int(A2 is not None)}/{int(bool(a2_replace))} R={None if R is None else str(R.dtype)[6:]}/{r_row_mod} act={act} gate={gate} "
                           f"G={None if G is None else str(G.dtype)[6:]} Z={None if Zout is None else str(Zout.dtype)[6:]} "
                           f"drop={0 if drop is None else drop[2]}/{drop_col_div} rows={int(rows is not None)} orm={int(out_row_mask is not None)} "
                           f"arm={int(a_row_mask is not None)} out={str(segs[0].out.dtype)[6:]} rpb={segs[0].rows_per_batch} tr={int(any(s_.transposed for s_ in segs))}"))
    _timed(kind, flops, nbytes, lambda: check(lib().made_linear(C.byref(a), _stream()), "made_linear"),
           ("rows", rows[1], M) if rows is not None else ((desc, tile_skip_mask, 128) if tile_skip_mask is not None else desc))
    return segs[0].out


def row_index(mask: Tensor, out=None):
    """(row_index int32 [M], n_rows int32 [1]) of a token mask (made_row_index): the `rows=` argument of linear / gemm_tn."""
    m = mask.reshape(-1)
    M = m.numel()
    if out is not None:
        idx, n = out
        assert idx.dtype == torch.int32 and idx.numel() >= M and n.dtype == torch.int32
    else:
        idx = torch.empty(M, device=m.device, dtype=torch.int32)
        n = torch.empty(1, device=m.device, dtype=torch.int32)
    check(lib().made_row_index(_p(_f32(m, "mask")), M, _p(idx), _p(n), _stream()), "made_row_index")
    return idx, n


def batch_order(mask: Tensor, out: Optional[Tensor] = None) -> Tensor:
    """int32 [B]: samples of a [B, T] token mask by descending valid length (made_batch_order): `order=` of attention / _bwd."""
    assert mask.dim() == 2
    B, T = mask.shape
    if out is None:
        out = torch.empty(B, device=mask.device, dtype=torch.int32)
    assert out.dtype == torch.int32 and out.numel() == B
    check(lib().made_batch_order(_p(_f32(mask.contiguous(), "mask")), B, T, _p(out), _stream()), "made_batch_order")
    return out


def linear_splitk(A: Tensor, W: Tensor, bias: Optional[Tensor], ws: Tensor, split_k: int, *, A2: Optional[Tensor] = None,
                  a2_row_mod: int = 0, act: int = ACT_NONE, R: Optional[Tensor] = None, r_row_mod: int = 0,
                  out: Optional[Tensor] = None, ln1=None, ln1_out: Optional[Tensor] = None, ln2=None,
                  ln2_out: Optional[Tensor] = None, eps: float = 1e-5) -> None:
    """Skinny Linear (few rows, e.g. the decoder's B*Q): K is split over grid.z so the launch fills the chip, the raw f32
    partials land in `ws` ([>= split_k*M*N] f32) and made_splitk_finish sums them and applies bias / act / residual, then
    optionally LayerNorm (ln1 = (gamma, beta)) and a second LayerNorm on top (ln2)."""
    assert A.dim() == 2 and W.dim() == 2 and A.stride(1) == 1 and W.stride(1) == 1
    M, K = A.shape
    N = W.shape[0]
    assert ws.dtype == torch.float32 and ws.is_contiguous() and ws.numel() >= split_k * M * N
    a = MadeLinearArgs()
    a.A, a.a_dtype, a.w_dtype, a.lda = _p(A), dt_of(A), dt_of(W), A.stride(0)
    use_a2 = A2 is not None
    if use_a2:
        a.A2, a.lda2, a.a2_row_mod = _p(A2), A2.stride(0), a2_row_mod
    a.W, a.ldw = _p(W), W.stride(0)
    a.M, a.N, a.K = M, N, K
    a.batch = 1
    a.nseg, a.split_k, a.split_ws = 1, max(split_k, 2), _p(ws)
    sg = a.seg[0]
    sg.col_begin, sg.out, sg.out_dtype, sg.ldo, sg.use_a2 = 0, _p(ws), F32, N, 1 if use_a2 else 0
    esz = 4 if a.w_dtype == F32 else 2
    _timed("linear_splitk_f32" if a.w_dtype == F32 else "linear_kernel<bf16,bf16>", 2.0 * M * N * K, M * K * esz + N * K * esz + M * N * 4 * a.split_k,
           lambda: check(lib().made_linear(C.byref(a), _stream()), "made_linear(split_k)"), f"M={M} N={N} K={K} split={a.split_k}")
    splitk_finish(ws, a.split_k, M, N, bias, act=act, R=R, r_row_mod=r_row_mod, out=out, ln1=ln1, ln1_out=ln1_out, ln2=ln2, ln2_out=ln2_out,
                  eps=eps)


def splitk_finish(ws: Tensor, split_k: int, M: int, N: int, bias: Optional[Tensor], *, act: int = ACT_NONE, R: Optional[Tensor] = None,
                  r_row_mod: int = 0, out: Optional[Tensor] = None, ln1=None, ln1_out: Optional[Tensor] = None, ln2=None,
                  ln2_out: Optional[Tensor] = None, eps: float = 1e-5) -> None:
    """made_splitk_finish: sum the `split_k` f32 partials in ws ([split_k, M, N]), + bias, act, + residual -> out; then optionally
    LayerNorm (ln1 = (gamma, beta)) -> ln1_out and a second LayerNorm on top (ln2) -> ln2_out."""
    assert ws.dtype == torch.float32 and ws.is_contiguous() and ws.numel() >= split_k * M * N
    f = MadeFinishArgs()
    f.ws, f.split_k, f.M, f.N = _p(ws), split_k, M, N
    f.bias, f.act = _p(_f32(bias, "bias")), act
    if R is not None:
        assert R.dim() == 2 and R.stride(1) == 1
        f.R, f.r_dtype, f.ldr, f.r_row_mod = _p(R), dt_of(R), R.stride(0), r_row_mod
    if out is not None:
        f.out, f.out_dtype, f.ldo = _p(out), dt_of(out), out.stride(0)
    if ln1 is not None:
        f.ln1_g, f.ln1_b = _p(_f32(ln1[0], "ln1.g")), _p(_f32(ln1[1], "ln1.b"))
        if ln1_out is not None:
            f.ln1_out, f.ln1_dtype, f.ln1_ld = _p(ln1_out), dt_of(ln1_out), ln1_out.stride(0)
    if ln2 is not None:
        f.ln2_g, f.ln2_b = _p(_f32(ln2[0], "ln2.g")), _p(_f32(ln2[1], "ln2.b"))
        f.ln2_out, f.ln2_dtype, f.ln2_ld = _p(ln2_out), dt_of(ln2_out), ln2_out.stride(0)
    f.eps = eps
    check(lib().made_splitk_finish(C.byref(f), _stream()), "made_splitk_finish")


def dec_stage(Zin: Tensor, W: Tensor, bias: Optional[Tensor], out: Tensor, *, ln=None, ln2=None, x2_out: Optional[Tensor] = None,
              add: Optional[Tensor] = None, x_out: Optional[Tensor] = None, R: Optional[Tensor] = None, res_from_x: bool = False,
              act: int = ACT_NONE, eps: float = 1e-5, a_out: Optional[Tensor] = None, drop=None, drop_ld: Optional[int] = None,
              drop_col_div: int = 1) -> Tensor:
    """One decoder stage with the previous stage's LayerNorm in its prologue (made_dec_stage): x = LayerNorm(Zin; ln) (ln = (gamma,
    beta) or None: x = Zin), x2_out = LayerNorm(x; ln2), x_out = bf16(x), out = dropout(act((x + add) W^T + bias)) + R (or + x).
    Zin [M, K] f32 (eval chain) or bf16 (training chain: then a_out = bf16(x) + add and drop = (seed, site, p) are available);
    W [N, K] bf16; add [rows, K] bf16 (row modulo); R [M, N] bf16; out [M, N] f32 or bf16."""
    assert Zin.dim() == 2 and W.dim() == 2 and out.dim() == 2 and W.dtype == torch.bfloat16
    assert Zin.stride(1) == 1 and W.stride(1) == 1 and out.stride(1) == 1
    M, K = Zin.shape
    N = W.shape[0]
    assert W.shape[1] == K and tuple(out.shape) == (M, N)
    a = MadeDecStageArgs()
    a.Zin, a.ldz, a.zin_dtype = _p(Zin), Zin.stride(0), dt_of(Zin)
    if ln is not None:
        a.ln_g, a.ln_b = _p(_f32(ln[0], "ln.g")), _p(_f32(ln[1], "ln.b"))
    if ln2 is not None:
        assert x2_out is not None and x2_out.dtype == torch.bfloat16 and x2_out.stride(1) == 1
        a.ln2_g, a.ln2_b, a.x2_out, a.ldx2 = _p(_f32(ln2[0], "ln2.g")), _p(_f32(ln2[1], "ln2.b")), _p(x2_out), x2_out.stride(0)
    if add is not None:
        assert add.dim() == 2 and add.dtype == torch.bfloat16 and add.is_contiguous() and add.shape[1] == K
        a.add, a.add_row_mod = _p(add), add.shape[0]
    if x_out is not None:
        assert x_out.dtype == torch.bfloat16 and x_out.stride(1) == 1 and tuple(x_out.shape) == (M, K)
        a.x_out, a.ldx = _p(x_out), x_out.stride(0)
    if a_out is not None:
        assert a_out.dtype == torch.bfloat16 and a_out.stride(1) == 1 and tuple(a_out.shape) == (M, K) and Zin.dtype == torch.bfloat16
        a.a_out, a.lda_out = _p(a_out), a_out.stride(0)
    if drop is not None and drop[2] > 0.0:
        assert Zin.dtype == torch.bfloat16
        set_drop(a.drop, drop)
        a.drop_ld = N if drop_ld is None else drop_ld
        a.drop_col_div = int(drop_col_div)
    a.W, a.ldw, a.bias = _p(W), W.stride(0), _p(_f32(bias, "bias"))
    if R is not None:
        assert R.dtype == torch.bfloat16 and R.stride(1) == 1 and tuple(R.shape) == (M, N)
        a.R, a.ldr = _p(R), R.stride(0)
    a.out, a.ldo, a.out_dtype = _p(out), out.stride(0), dt_of(out)
    a.act, a.res_from_x, a.eps = act, 1 if res_from_x else 0, eps
    a.M, a.N, a.K = M, N, K
    _timed("dec_stage_kernel", 2.0 * M * N * K, float(M * K * Zin.element_size() * ((N + 31) // 32) + N * K * 2 + M * N * out.element_size()),
           lambda: check(lib().made_dec_stage(C.byref(a), _stream()), "made_dec_stage"), f"M={M} N={N} K={K}")
    return out


def attention(Q: Tensor, K: Tensor, V: Tensor, O: Tensor, H: int, *, key_mask: Optional[Tensor] = None,
              q_mask: Optional[Tensor] = None, scale: Optional[float] = None, Lk: Optional[int] = None,
              q_skip_mask: Optional[Tensor] = None, lse: Optional[Tensor] = None, drop=None,
              order: Optional[Tensor] = None, keep_bits: Optional[Tensor] = None) -> Tensor:
    """softmax(Q K^T scale + mask) V.  Q [B,Lq,H*hd], K / V [B,Lk,H*hd], O [B,Lq,H*hd]
    (any batch/row strides, unit inner stride).  Lk defaults to K.shape[1].
    Training: lse [B,H,Lq] f32 receives the log-sum-exp, drop = (seed, site, p) drops attention weights; keep_bits
    (int32, shape attention_bits_shape(B, H, Lq, Lk)) receives the decisions, one bit per score, for attention_bwd."""
    assert Q.dim() == 3 and K.dim() == 3 and V.dim() == 3 and O.dim() == 3
    assert Q.stride(2) == 1 and K.stride(2) == 1 and V.stride(2) == 1 and O.stride(2) == 1
    assert Q.dtype == K.dtype == V.dtype == O.dtype
    B, Lq, D = Q.shape
    hd = D // H
    a = MadeAttnArgs()
    a.Q, a.K, a.V, a.O = _p(Q), _p(K), _p(V), _p(O)
    a.dtype, a.hd = dt_of(Q), hd
    a.B, a.H, a.Lq, a.Lk = B, H, Lq, (K.shape[1] if Lk is None else Lk)
    a.q_bs, a.ldq = Q.stride(0), Q.stride(1)
    a.k_bs, a.ldk = K.stride(0), K.stride(1)
    a.v_bs, a.ldv = V.stride(0), V.stride(1)
    a.o_bs, a.ldo = O.stride(0), O.stride(1)
    a.key_mask = _p(_f32(key_mask, "key_mask"))
    a.q_mask = _p(_f32(q_mask, "q_mask"))
    a.q_skip_mask = _p(_f32(q_skip_mask, "q_skip_mask"))
    a.scale = (1.0 / math.sqrt(hd)) if scale is None else scale
    if order is not None:                       # issue order of the batch (batch_order): longest sample first
        assert order.dtype == torch.int32 and order.numel() == B and order.is_contiguous()
        a.batch_order = _p(order)
    if lse is not None:
        assert lse.dtype == torch.float32 and lse.is_contiguous() and lse.numel() == B * H * Lq
        a.lse = _p(lse)
    if drop is not None and drop[2] > 0.0:
        set_drop(a.drop, drop)
        if keep_bits is not None:                   # [B*H*ceil(Lk/32), ld] int32 words: the dropout decisions, for attention_bwd
            assert keep_bits.dtype == torch.int32 and keep_bits.dim() == 2 and keep_bits.is_contiguous()
            assert keep_bits.shape[0] >= B * H * ((a.Lk + 31) // 32) and keep_bits.shape[1] >= Lq
            a.keep_bits, a.ld_bits = _p(keep_bits), keep_bits.shape[1]
    esz = 4 if a.dtype == F32 else 2
    flops = 4.0 * B * H * Lq * a.Lk * hd
    nbytes = esz * B * D * (2 * Lq + 2 * a.Lk)
    _timed("attention_" + ("f32" if a.dtype == F32 else "bf16"), flops, nbytes,
           lambda: check(lib().made_attention(C.byref(a), _stream()), "made_attention"),
           ("attn", key_mask, q_skip_mask) if (key_mask is not None and key_mask.dim() == 2) else f"B={B} H={H} hd={hd} Lq={Lq} Lk={a.Lk}")
    return O


def attention_bits_shape(B: int, H: int, Lq: int, Lk: int):
    """shape of the dropout-decision cache of made_attention / made_attention_bwd (MadeAttnArgs.keep_bits): one row of
    32 * ceil(Lq / 32) words per (batch, head, 32-key tile)"""
    return (B * H * ((Lk + 31) // 32), 32 * ((Lq + 31) // 32))


def attention_bits_decode(bits: Tensor, B: int, H: int, Lq: int, Lk: int) -> Tensor:
    """keep_bits -> bool [B, H, Lq, Lk] (tests): word (b, h, key tile kt, slot(q)) bit j = element (q, 32 kt + j) is kept; the 32
    slots of a query group are permuted (slot(q) = 2 ((q & 3) + 4 (q >> 3)) + ((q >> 2) & 1), csrc/common.h made_keep_slot)"""
    nkt, ldq = (Lk + 31) // 32, bits.shape[1]
    w = bits[:B * H * nkt].view(B, H, nkt, ldq)
    q = torch.arange(Lq, device=bits.device)
    ql = q & 31
    slot = (q & ~31) + 2 * ((ql & 3) + 4 * (ql >> 3)) + ((ql >> 2) & 1)
    w = w[:, :, :, slot]                                                   # [B, H, nkt, Lq]
    b = (w[..., None] >> torch.arange(32, device=bits.device, dtype=torch.int32)) & 1     # [B, H, nkt, Lq, 32]
    return b.permute(0, 1, 3, 2, 4).reshape(B, H, Lq, nkt * 32)[..., :Lk].bool()


def attention_wide(Q: Tensor, K: Tensor, V: Tensor, O: Tensor, *, scale: float, Kadd: Optional[Tensor] = None,
                   key_mask: Optional[Tensor] = None, shared_q: bool = False, n_split: int = 1,
                   part_o: Optional[Tensor] = None, part_ml: Optional[Tensor] = None, drop=None,
                   sum_out: Optional[Tensor] = None, lse_out: Optional[Tensor] = None) -> Tensor:
    """Single-head attention with head dim = D.  Q [B|1, NQ1, NQ2, D], K/Kadd/V [B, L, D], O [B, NQ1, NQ2, D]
    (strided views fine, unit inner stride).  shared_q: the same queries for every batch entry (Q.shape[0] == 1)."""
    assert Q.dim() == 4 and O.dim() == 4 and K.dim() == 3 and V.dim() == 3
    assert Q.stride(3) == 1 and O.stride(3) == 1 and K.stride(2) == 1 and V.stride(2) == 1
    assert Q.dtype == K.dtype == V.dtype
    B, L, D = K.shape
    a = MadeWideAttnArgs()
    a.Q, a.K, a.V, a.O = _p(Q), _p(K), _p(V), _p(O)
    a.key_mask = _p(_f32(key_mask, "key_mask"))
    a.dtype, a.o_dtype = dt_of(Q), dt_of(O)
    a.B, a.NQ1, a.NQ2, a.L, a.D = B, O.shape[1], O.shape[2], L, D
    a.q_bs, a.q_s1, a.q_s2 = (0 if shared_q else Q.stride(0)), Q.stride(1), Q.stride(2)
    a.k_bs, a.ldk = K.stride(0), K.stride(1)
    a.v_bs, a.ldv = V.stride(0), V.stride(1)
    if Kadd is not None:
        assert Kadd.dim() == 3 and Kadd.stride(2) == 1 and Kadd.dtype == K.dtype
        a.Kadd, a.kadd_bs, a.ldkadd = _p(Kadd), Kadd.stride(0), Kadd.stride(1)
    a.o_bs, a.o_s1, a.o_s2 = O.stride(0), O.stride(1), O.stride(2)
    a.scale = scale
    nq = a.NQ1 * a.NQ2
    if n_split > 1:
        if part_o is None:
            part_o = torch.empty(B * n_split * nq * D, device=Q.device, dtype=torch.float32)
            part_ml = torch.empty(B * n_split * nq * 4, device=Q.device, dtype=torch.float32)
        assert part_o.numel() >= B * n_split * nq * D and part_ml.numel() >= B * n_split * nq * 4
        a.n_split, a.part_o, a.part_ml = n_split, _p(_f32(part_o, "part_o")), _p(_f32(part_ml, "part_ml"))
    if drop is not None and drop[2] > 0.0:
        set_drop(a.drop, drop)
    if lse_out is not None:
        assert lse_out.dtype == torch.float32 and lse_out.is_contiguous() and lse_out.numel() >= B * nq
        a.lse_out = _p(lse_out)
    if sum_out is not None:
        assert sum_out.dtype == torch.float32 and sum_out.is_contiguous() and sum_out.numel() >= B * nq
        a.sum_out = _p(sum_out)
    esz = 4 if a.dtype == F32 else 2
    _timed("attention_wide_" + ("f32" if a.dtype == F32 else "bf16"), 4.0 * B * nq * L * D,
           esz * B * (2 * L * D + 2 * nq * D),
           lambda: check(lib().made_attention_wide(C.byref(a), _stream()), "made_attention_wide"),
           f"B={B} NQ={nq} L={L} D={D} kadd={int(Kadd is not None)}")
    return O


def layernorm(x: Tensor, gamma: Tensor, beta: Tensor, out: Optional[Tensor] = None, eps: float = 1e-5,
              out_dtype: Optional[torch.dtype] = None, row_skip: Optional[Tensor] = None) -> Tensor:
    """LayerNorm over the last axis.  x: [rows, D] (row stride free) or a [B, T, D] view of a larger buffer
    (batch and row strides free); out: [rows, D] / [B*T, D] with free row stride."""
    assert x.dim() in (2, 3) and x.stride(-1) == 1
    if x.dim() == 3:
        rows, D = x.shape[0] * x.shape[1], x.shape[2]
        ldx, rpb, xbs = x.stride(1), x.shape[1], x.stride(0)
    else:
        rows, D = x.shape
        ldx, rpb, xbs = x.stride(0), 0, 0
    if out is None:
        out = torch.empty((rows, D), device=x.device, dtype=out_dtype or x.dtype)
    assert out.dim() == 2 and out.stride(1) == 1 and out.shape[0] >= rows
    check(lib().made_layernorm(_p(x), dt_of(x), ldx, rpb, xbs, _p(_f32(gamma, "gamma")), _p(_f32(beta, "beta")),
                               _p(out), dt_of(out), out.stride(0), rows, D, eps, _p(_f32(row_skip, "row_skip")), _stream()),
          "made_layernorm")
    return out


def layernorm_add(x: Tensor, gamma: Optional[Tensor], beta: Optional[Tensor], add: Tensor, out: Optional[Tensor], out2: Tensor,
                  eps: float = 1e-5, row_skip: Optional[Tensor] = None) -> Tensor:
    """out = LayerNorm(x) (or x when gamma is None; `out` may be None then), out2 = out + add.  All [rows, D]."""
    assert x.dim() == 2 and add.dim() == 2 and out2.dim() == 2 and x.stride(1) == 1 and add.stride(1) == 1 and out2.stride(1) == 1
    rows, D = x.shape
    ydt = dt_of(out2)
    if out is not None:
        assert out.dtype == out2.dtype and out.stride(1) == 1
    check(lib().made_layernorm_add(_p(x), dt_of(x), x.stride(0), _p(_f32(gamma, "gamma")), _p(_f32(beta, "beta")),
                                   _p(out), ydt, out.stride(0) if out is not None else 0, _p(add), dt_of(add), add.stride(0),
                                   _p(out2), out2.stride(0), rows, D, eps, _p(_f32(row_skip, "row_skip")), _stream()), "made_layernorm_add")
    return out2


def cast_mask_rows(x: Tensor, mask: Optional[Tensor], out: Tensor) -> Tensor:
    """out[r] = x[r] * (mask[r] != 0), f32 -> out.dtype; x, out [rows, D]."""
    assert x.dim() == 2 and out.dim() == 2 and x.dtype == torch.float32 and x.stride(1) == 1 and out.stride(1) == 1
    check(lib().made_cast_mask_rows(_p(x), x.stride(0), _p(_f32(mask, "mask")), _p(out), dt_of(out), out.stride(0),
                                    x.shape[0], x.shape[1], _stream()), "made_cast_mask_rows")
    return out


def pack_music_records(seg: Tensor, mask: Tensor, music: Tensor, out: Tensor, pack_dtype: torch.dtype) -> Tensor:
    """made_pack_music_records: out [n_pad, rec] uint8 <- per track [S * D embeddings in pack_dtype | S mask floats | D pooled floats | pad];
    rows past seg.shape[0] are zero-filled (the sharded retrieval's one all-gather buffer, mgsv_amd/retrieval.py)."""
    n, S, D = seg.shape
    assert seg.stride(2) == 1 and seg.stride(1) == D and out.dtype == torch.uint8 and out.dim() == 2 and out.is_contiguous() and out.shape[0] >= n
    mask, music = _f32(mask.contiguous(), "mask"), _f32(music.contiguous(), "music")
    check(lib().made_pack_music_records(_p(seg), dt_of(seg), seg.stride(0) if n > 0 else S * D, _p(mask), S, _p(music), D, _p(out),
                                        F32 if pack_dtype == torch.float32 else BF16, out.shape[1], n, out.shape[0], S, D, _stream()),
          "made_pack_music_records")
    return out


def row_affine(x: Tensor, scale: Tensor, shift: Tensor, act: int = ACT_NONE, out: Optional[Tensor] = None) -> Tensor:
    """out[r] = act(x[r] * scale[r % P] + shift[r % P]) with P = scale.numel() (made_row_affine); x [rows, cols], in place by default."""
    assert x.dim() == 2 and x.stride(1) == 1 and scale.numel() == shift.numel()
    out = x if out is None else out
    check(lib().made_row_affine(_p(x), dt_of(x), x.stride(0), _p(_f32(scale, "scale")), _p(_f32(shift, "shift")), scale.numel(), act,
                                _p(out), dt_of(out), out.stride(0), x.shape[0], x.shape[1], _stream()), "made_row_affine")
    return out


def concat_cols(a: Optional[Tensor], c: Optional[Tensor], out: Tensor) -> Tensor:
    """out[b] = [a[b] ; c[b]] for contiguous f32 [B, *] tensors (made_concat_cols): the DETR token mask."""
    B = out.shape[0]
    ca = a.shape[1] if a is not None else 0
    cc = c.shape[1] if c is not None else 0
    assert out.is_contiguous() and out.dtype == torch.float32 and out.shape[1] == ca + cc
    check(lib().made_concat_cols(_p(_f32(a, "a")) if a is not None else None, ca, _p(_f32(c, "c")) if c is not None else None, cc, _p(out), B, _stream()),
          "made_concat_cols")
    return out


def masked_mean(x: Tensor, mask: Optional[Tensor], out: Optional[Tensor] = None) -> Tensor:
    """x [B,T,D] (unit inner stride), mask [B,T] or None (plain sum) -> [B,D] f32."""
    assert x.dim() == 3 and x.stride(2) == 1
    B, T, D = x.shape
    if out is None:
        out = torch.empty((B, D), device=x.device, dtype=torch.float32)
    check(lib().made_masked_mean(_p(x), dt_of(x), x.stride(0), x.stride(1), _p(_f32(mask, "mask")), _p(_f32(out, "out")),
                                 B, T, D, _stream()), "made_masked_mean")
    return out


def l2norm_rows(x: Tensor, out_f32: Optional[Tensor] = None, out_alt: Optional[Tensor] = None, eps: float = 1e-12):
    assert x.dim() == 2 and x.stride(1) == 1
    rows, D = x.shape
    if out_f32 is None and out_alt is None:
        out_f32 = torch.empty((rows, D), device=x.device, dtype=torch.float32)
    ldy = (out_f32 if out_f32 is not None else out_alt).stride(0)
    if out_f32 is not None and out_alt is not None:
        assert out_f32.stride(0) == out_alt.stride(0)
    check(lib().made_l2norm_rows(_p(x), dt_of(x), x.stride(0), _p(out_f32), _p(out_alt),
                                 dt_of(out_alt) if out_alt is not None else F32, ldy, rows, D, eps, _stream()),
          "made_l2norm_rows")
    return out_f32 if out_f32 is not None else out_alt


def sine_pe(mask: Tensor, dim_t: Tensor, out: Optional[Tensor] = None, out_dtype=torch.float32) -> Tensor:
    B, L = mask.shape
    D = dim_t.shape[0]
    if out is None:
        out = torch.empty((B, L, D), device=mask.device, dtype=out_dtype)
    assert out.is_contiguous()
    check(lib().made_sine_pe(_p(_f32(mask, "mask")), _p(_f32(dim_t, "dim_t")), _p(out), dt_of(out), B, L, D, _stream()),
          "made_sine_pe")
    return out


def masked_softmax(logits: Tensor, mask: Optional[Tensor], probs: Tensor, S: int, scale: float) -> Tensor:
    """logits [M_outer,R,>=S] f32 -> probs [M_outer,R,S_pad] (pad columns zeroed); mask [M_outer,S]."""
    assert logits.dim() == 3 and probs.dim() == 3 and logits.dtype == torch.float32
    assert logits.stride(2) == 1 and probs.stride(2) == 1
    Mo, R = logits.shape[0], logits.shape[1]
    assert logits.stride(0) == R * logits.stride(1) and probs.stride(0) == R * probs.stride(1)
    S_pad = probs.shape[2]
    check(lib().made_masked_softmax(_p(logits), logits.stride(1), _p(_f32(mask, "mask")), mask.stride(0) if mask is not None else 0,
                                    _p(probs), dt_of(probs), probs.stride(1), Mo, R, S, S_pad, scale, _stream()),
          "made_masked_softmax")
    return probs


def xpool_tail(y: Tensor, gamma: Tensor, beta: Tensor, video: Tensor, sims: Tensor, Nm: int, Nv: int,
               pooled_out: Optional[Tensor] = None, eps: float = 1e-5) -> Tensor:
    """y [Nm*Nv,D] -> LayerNorm3 -> cosine with video[n] -> sims[n, m] (sims: [Nv, >=Nm] f32)."""
    assert y.dim() == 2 and y.stride(1) == 1 and video.dim() == 2 and video.stride(1) == 1
    D = y.shape[1]
    if pooled_out is not None:
        assert pooled_out.is_contiguous() and pooled_out.dtype == torch.float32
    check(lib().made_xpool_tail(_p(y), dt_of(y), y.stride(0), _p(_f32(gamma, "gamma")), _p(_f32(beta, "beta")),
                                _p(video), video.stride(0), _p(pooled_out), _p(sims), sims.stride(0), Nm, Nv, D, eps, _stream()),
          "made_xpool_tail")
    return sims


def xpool_fused_ws_floats(Nv: int, Nm: int, D: int = 256) -> int:
    """Workspace of made_xpool_fused in floats (include/made_hip.h): per-video LayerNorm3 / cosine terms, the folded Linear
    (W'' in bf16, two vectors) and four ints per track."""
    return Nv * (D + 2) + 4 + 2 * D + D * D // 2 + 4 * Nm


def xpool_fused(Q: Tensor, K: Tensor, U: Tensor, key_mask: Optional[Tensor], ln2, Wl: Tensor, bl: Tensor, ln3, vn: Tensor,
                sims: Tensor, scale: float, eps: float = 1e-5, ws: Optional[Tensor] = None, prepare_ws: bool = True) -> Tensor:
    """All-pairs X-Pool scoring in one launch (made_xpool_fused; bf16, D = 256): Q [Nv,D], K / U [Nm,S,D] (unit inner stride),
    key_mask [Nm,S] or None, ln2 / ln3 = (gamma, beta) f32, Wl [D,D] bf16, bl f32, vn [Nv,D] f32 (L2-normalised videos)
    -> sims[n, m] written into `sims` ([Nv, >= Nm] f32 view)."""
    from ._lib import MadeXpoolFusedArgs
    Nv, D = Q.shape
    Nm, S, _ = K.shape
    assert Q.dtype == K.dtype == U.dtype == Wl.dtype == torch.bfloat16 and Q.stride(1) == 1 and K.stride(2) == 1 and U.stride(2) == 1
    assert U.shape == K.shape and Wl.shape == (D, D) and Wl.stride(1) == 1 and vn.shape == (Nv, D) and vn.dtype == torch.float32
    assert sims.dtype == torch.float32 and sims.stride(1) == 1 and sims.shape[0] == Nv and sims.shape[1] >= Nm
    a = MadeXpoolFusedArgs()
    a.Q, a.ldq = _p(Q), Q.stride(0)
    a.K, a.U, a.k_bs, a.ldk, a.u_bs, a.ldu = _p(K), _p(U), K.stride(0), K.stride(1), U.stride(0), U.stride(1)
    a.key_mask = _p(_f32(key_mask.contiguous(), "key_mask")) if key_mask is not None else None
    a.ln2_g, a.ln2_b, a.ln3_g, a.ln3_b = _p(_f32(ln2[0], "ln2")), _p(_f32(ln2[1], "ln2")), _p(_f32(ln3[0], "ln3")), _p(_f32(ln3[1], "ln3"))
    a.Wl, a.ldw, a.bl = _p(Wl), Wl.stride(0), _p(_f32(bl, "bl"))
    a.vn, a.ldvn = _p(vn), vn.stride(0)
    a.sims, a.ld_sims = _p(sims), sims.stride(0)
    a.Nv, a.Nm, a.S, a.D, a.scale, a.eps = Nv, Nm, S, D, scale, eps
    if ws is None:
        ws = torch.empty(xpool_fused_ws_floats(Nv, Nm, D), device=Q.device, dtype=torch.float32)
        prepare_ws = True
    assert ws.dtype == torch.float32 and ws.is_contiguous() and ws.numel() >= xpool_fused_ws_floats(Nv, Nm, D)
    a.ws, a.prepare_ws = _p(ws), 1 if prepare_ws else 0
    flops = 2.0 * Nv * Nm * (2 * S * D + D * D)
    desc = ""
    if _timer is not None and key_mask is not None:             # executed work: the valid segments of each track only
        valid = float((key_mask != 0).sum().item())
        desc = ("frac", (2.0 * Nv * (2 * valid * D + Nm * D * D)) / flops)
    _timed("xpool_fused", flops, float(2 * (K.numel() + U.numel()) + 4 * Nv * Nm + 2 * Q.numel()),
           lambda: check(lib().made_xpool_fused(C.byref(a), _stream()), "made_xpool_fused"), desc)
    return sims


def xpool_sims_ws_bytes(Nv: int, Nm: int, D: int = 256) -> int:
    """Workspace of made_xpool_sims in bytes: per-video LayerNorm3 / cosine terms and 32 ints per track."""
    return int(lib().made_xpool_sims_ws_bytes(Nv, Nm, D))


def xpool_sims(Q: Tensor, K: Tensor, UU: Tensor, key_mask: Optional[Tensor], av: Tensor, bv: Tensor, ln3, vn: Tensor, sims: Tensor, scale: float,
               eps: float = 1e-5, ws: Optional[Tensor] = None, prepare_ws: bool = True) -> Tensor:
    """All-pairs X-Pool scoring with the per-pair Linear moved onto the values (made_xpool_sims; bf16, D = 256, S <= 96): Q [Nv, D],
    K [Nm, S, D], UU [Nm, S, 2 D] = value rows u_s | W'' u_s (unit inner stride), key_mask [Nm, S] or None, av = b'' and bv = W'' 1 [D] f32,
    ln3 = (gamma, beta) f32, vn [Nv, D] f32 (L2-normalised videos) -> sims[n, m] written into `sims` ([Nv, >= Nm] f32 view)."""
    from ._lib import MadeXpoolSimsArgs
    Nv, D = Q.shape
    Nm, S, _ = K.shape
    assert Q.dtype == K.dtype == UU.dtype == torch.bfloat16 and Q.stride(1) == 1 and K.stride(2) == 1 and UU.stride(2) == 1
    assert UU.shape == (Nm, S, 2 * D) and vn.shape == (Nv, D) and vn.dtype == torch.float32 and av.shape == (D,) and bv.shape == (D,)
    assert sims.dtype == torch.float32 and sims.stride(1) == 1 and sims.shape[0] == Nv and sims.shape[1] >= Nm
    a = MadeXpoolSimsArgs()
    a.Q, a.ldq = _p(Q), Q.stride(0)
    a.K, a.UU, a.k_bs, a.ldk, a.u_bs, a.ldu = _p(K), _p(UU), K.stride(0), K.stride(1), UU.stride(0), UU.stride(1)
    a.key_mask = _p(_f32(key_mask.contiguous(), "key_mask")) if key_mask is not None else None
    a.av, a.bv = _p(_f32(av, "av")), _p(_f32(bv, "bv"))
    a.ln3_g, a.ln3_b = _p(_f32(ln3[0], "ln3")), _p(_f32(ln3[1], "ln3"))
    a.vn, a.ldvn = _p(vn), vn.stride(0)
    a.sims, a.ld_sims = _p(sims), sims.stride(0)
    a.Nv, a.Nm, a.S, a.D, a.scale, a.eps = Nv, Nm, S, D, scale, eps
    need = int(lib().made_xpool_sims_ws_bytes(Nv, Nm, D))
    if ws is None:
        ws = torch.empty(need, device=Q.device, dtype=torch.uint8)
        prepare_ws = True
    assert ws.is_contiguous() and ws.numel() * ws.element_size() >= need
    a.ws, a.prepare_ws = _p(ws), 1 if prepare_ws else 0
    flops = 2.0 * Nv * Nm * (2 * S * D + D * D)                    # the reference's work per pair (the kernel executes 3 S D per pair)
    desc = ""
    if _timer is not None and key_mask is not None:
        valid = float((key_mask != 0).sum().item())
        desc = ("frac", (2.0 * Nv * (2 * valid * D + Nm * D * D)) / flops)
    _timed("xpool_sims", flops, float(2 * (K.numel() + UU.numel()) + 4 * Nv * Nm + 2 * Q.numel()),
           lambda: check(lib().made_xpool_sims(C.byref(a), _stream()), "made_xpool_sims"), desc)
    return sims


def xpool_attention(Q: Tensor, K: Tensor, U: Tensor, key_mask: Optional[Tensor], out: Tensor, scale: float, normalize: bool = True,
                    eps: float = 1e-5, ws: Optional[Tensor] = None) -> Tensor:
    """The X-Pool attention at retrieval scale, head dim = D = 256 or 512, S <= 512 (made_xpool_attention): Q [Nv, D], K / U [Nm, S, D]
    bf16 (unit inner stride), key_mask [Nm, S] or None -> out [Nm, Nv, D] bf16: softmax_s(Q K^T * scale + mask) U, then (normalize)
    (x - mean) * rstd over D -- LayerNorm2 without its affine part, which the caller folds into the Linear behind it."""
    from ._lib import MadeXpoolAttnArgs
    Nv, D = Q.shape
    Nm, S, _ = K.shape
    assert Q.dtype == K.dtype == U.dtype == out.dtype == torch.bfloat16 and Q.stride(1) == 1 and K.stride(2) == 1 and U.stride(2) == 1
    assert U.shape == K.shape and out.dim() == 3 and tuple(out.shape) == (Nm, Nv, D) and out.stride(2) == 1 and out.stride(0) == Nv * out.stride(1)
    a = MadeXpoolAttnArgs()
    a.Q, a.ldq = _p(Q), Q.stride(0)
    a.K, a.U, a.k_bs, a.ldk, a.u_bs, a.ldu = _p(K), _p(U), K.stride(0), K.stride(1), U.stride(0), U.stride(1)
    a.key_mask = _p(_f32(key_mask.contiguous(), "key_mask")) if key_mask is not None else None
    a.out, a.ldo = _p(out), out.stride(1)
    a.Nv, a.Nm, a.S, a.D, a.scale, a.eps, a.normalize = Nv, Nm, S, D, scale, eps, 1 if normalize else 0
    if ws is None:
        ws = torch.empty(Nm * 32, device=Q.device, dtype=torch.int32)
    assert ws.dtype == torch.int32 and ws.is_contiguous() and ws.numel() >= Nm * 32
    a.ws = _p(ws)
    flops = 4.0 * Nv * Nm * S * D
    desc = ""
    if _timer is not None and key_mask is not None:             # executed work: the valid segments of each track only
        desc = ("frac", float((key_mask != 0).sum().item()) / float(Nm * S))
    _timed("xpool_attention", flops, float(2 * (K.numel() + U.numel()) + 2 * Nv * Nm * D + 2 * Q.numel()),
           lambda: check(lib().made_xpool_attention(C.byref(a), _stream()), "made_xpool_attention"), desc)
    return out


def xpool_inbatch_ws_bytes(Nm: int, S: int) -> int:
    return int(lib().made_xpool_inbatch_ws_bytes(Nm, S))


def xpool_inbatch(Q: Tensor, K: Tensor, U: Tensor, key_mask: Optional[Tensor], out: Tensor, scale: float, ws: Optional[Tensor] = None) -> Tensor:
    """The in-batch X-Pool contraction (made_xpool_inbatch; reference modules/transformer.py:110-119): Q [Nv <= 64, D], K / U [Nm, S <= 512, D]
    bf16 (unit inner stride), key_mask [Nm, S] or None -> out [Nm, Nv, D] bf16 or f32 = softmax_s(Q K^T * scale + mask) U.  Two launches:
    scores per (track, 128 segments), then P.V per (track, 128 value columns); ws: xpool_inbatch_ws_bytes(Nm, S) bytes."""
    from ._lib import MadeXpoolInbatchArgs
    Nv, D = Q.shape
    Nm, S, _ = K.shape
    assert Q.dtype == K.dtype == U.dtype == torch.bfloat16 and Q.stride(1) == 1 and K.stride(2) == 1 and U.stride(2) == 1
    assert U.shape == K.shape and out.dim() == 3 and tuple(out.shape) == (Nm, Nv, D) and out.stride(2) == 1
    a = MadeXpoolInbatchArgs()
    a.Q, a.ldq = _p(Q), Q.stride(0)
    a.K, a.U, a.k_bs, a.ldk, a.u_bs, a.ldu = _p(K), _p(U), K.stride(0), K.stride(1), U.stride(0), U.stride(1)
    a.key_mask = _p(_f32(key_mask.contiguous(), "key_mask")) if key_mask is not None else None
    a.out, a.out_dtype, a.o_bs, a.ldo = _p(out), dt_of(out), out.stride(0), out.stride(1)
    a.Nv, a.Nm, a.S, a.D, a.scale = Nv, Nm, S, D, scale
    need = xpool_inbatch_ws_bytes(Nm, S)
    if ws is None:
        ws = torch.empty(need, device=Q.device, dtype=torch.uint8)           # (contents arbitrary: the exchange tiles of the two launches)
    assert ws.is_contiguous() and ws.numel() * ws.element_size() >= need
    a.ws = _p(ws)
    desc = ""
    if _timer is not None and key_mask is not None:             # executed work: the valid segments of each track only
        desc = ("frac", float((key_mask != 0).sum().item()) / float(Nm * S))
    _timed("xpool_inbatch", 4.0 * Nv * Nm * S * D, float(2 * (K.numel() + U.numel()) + out.element_size() * Nv * Nm * D + 2 * Q.numel()),
           lambda: check(lib().made_xpool_inbatch(C.byref(a), _stream()), "made_xpool_inbatch"), desc)
    return out


def clip_loss(sims: Tensor, logit_scale: Tensor, loss_out: Tensor, weight: float = 1.0, accumulate: bool = False,
              row_exclude: Optional[Tensor] = None) -> Tensor:
    """Symmetric cross entropy of a square similarity matrix; row_exclude [n, n] f32 (1 = same-track negative left out of the
    video -> music softmax, reference modules/loss.py:90-114)."""
    assert sims.dim() == 2 and sims.shape[0] == sims.shape[1] and sims.stride(1) == 1 and sims.dtype == torch.float32
    assert row_exclude is None or (row_exclude.shape == sims.shape and row_exclude.is_contiguous())
    check(lib().made_clip_loss(_p(sims), sims.stride(0), sims.shape[0], _p(logit_scale), weight, 1 if accumulate else 0,
                               _p(loss_out), _p(_f32(row_exclude, "row_exclude")), _stream()), "made_clip_loss")
    return loss_out


def hungarian_match(pred_logits: Tensor, pred_spans: Tensor, targets: Tensor, fg_label: int,
                    w_span: float = 10.0, w_giou: float = 1.0, w_class: float = 4.0, cost_in: Optional[Tensor] = None):
    """pred_* [NS,Q,2] (NS = layers*B), targets [B,G,2] -> (pred_idx [NS,min(Q,G)] i64, tgt_idx, count [NS] i32,
    status [1] i32, cost [NS,Q,G] f32).  No host sync: inspect `status` later (1 = SciPy would raise)."""
    NS, Q, _ = pred_logits.shape
    B, G, _ = targets.shape
    dev = pred_logits.device
    width = min(Q, G)
    if cost_in is not None:
        assert cost_in.shape == (NS, Q, G) and cost_in.dtype == torch.float32 and cost_in.is_contiguous()
        cost = cost_in
    else:
        cost = torch.empty((NS, Q, G), device=dev, dtype=torch.float32)
    pi = torch.empty((NS, width), device=dev, dtype=torch.int64)
    ti = torch.empty((NS, width), device=dev, dtype=torch.int64)
    cnt = torch.empty((NS,), device=dev, dtype=torch.int32)
    status = _tape.zero_(torch.empty((1,), device=dev, dtype=torch.int32))
    check(lib().made_hungarian_match(_p(_f32(pred_logits, "pred_logits")), _p(_f32(pred_spans, "pred_spans")),
                                     _p(_f32(targets, "targets")), NS, B, Q, G, fg_label, w_span, w_giou, w_class,
                                     _p(cost), 1 if cost_in is not None else 0, _p(pi), _p(ti), _p(cnt), _p(status), _stream()), "made_hungarian_match")
    return pi, ti, cnt, status, cost


def set_criterion(pred_logits: Tensor, pred_spans: Tensor, targets: Tensor, pred_idx: Tensor, tgt_idx: Tensor,
                  count: Tensor, proj_queries: Optional[Tensor], vid_sum: Optional[Tensor], empty_weight: Tensor,
                  fg_label: int, weights: Tensor, temperature: float = 0.07):
    """-> (losses [n_layers,5] f32, total [1] f32)."""
    nl, B, Q, _ = pred_logits.shape
    G = targets.shape[1]
    Dc = proj_queries.shape[-1] if proj_queries is not None else 0
    dev = pred_logits.device
    losses = torch.empty((nl, 5), device=dev, dtype=torch.float32)
    total = torch.empty((1,), device=dev, dtype=torch.float32)
    check(lib().made_set_criterion(_p(_f32(pred_logits, "pred_logits")), _p(_f32(pred_spans, "pred_spans")),
                                   _p(_f32(targets, "targets")), _p(pred_idx), _p(tgt_idx), _p(count),
                                   _p(_f32(proj_queries, "proj_queries")), _p(_f32(vid_sum, "vid_sum")),
                                   _p(_f32(empty_weight, "empty_weight")), nl, B, Q, G, Dc, fg_label, temperature,
                                   _p(_f32(weights, "weights")), _p(losses), _p(total), _stream()), "made_set_criterion")
    return losses, total

"""GPU: the backward kernels of the training path against plain PyTorch fp32 references of the same op
(tolerances are stated per test; bf16 inputs are compared after the same rounding of the inputs)."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def T():
    from mgsv_amd import ops, ops_train
    return ops, ops_train


def _rand(*shape, dtype, seed):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return torch.randn(*shape, generator=g).to("cuda").to(dtype)


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-5), (torch.bfloat16, 2e-3)])
@pytest.mark.parametrize("M,N,K", [(300, 256, 128), (1000, 384, 520), (77, 128, 256), (4096, 512, 512), (64, 40, 8)])
def test_gemm_tn_matches_torch(T, dtype, tol, M, N, K):
    ops, tr = T
    A, B = _rand(M, N, dtype=dtype, seed=1), _rand(M, K, dtype=dtype, seed=2)
    mask = (torch.rand(M, device="cuda") > 0.3).float()
    ref = (A.float() * mask[:, None]).t() @ B.float()
    # plain store
    Cst = torch.full((N, K), 7.0, device="cuda")
    tr.gemm_tn(A, B, Cst, row_mask=mask)
    scale = float(ref.abs().max())
    assert float((Cst - ref).abs().max()) <= tol * scale
    # accumulate with the reduction split over workgroups, plus the bias gradient
    Cacc = torch.ones(N, K, device="cuda")
    cs = torch.full((N,), 2.0, device="cuda")
    tr.gemm_tn(A, B, Cacc, row_mask=mask, accumulate=True, alpha=0.5, colsum=cs)
    assert float((Cacc - (1.0 + 0.5 * ref)).abs().max()) <= tol * scale
    cref = 2.0 + 0.5 * (A.float() * mask[:, None]).sum(0)
    assert float((cs - cref).abs().max()) <= tol * max(float(cref.abs().max()), 1.0) * 4
    # compute-dtype output
    Cc = torch.empty(N, K, device="cuda", dtype=dtype)
    tr.gemm_tn(A, B, Cc)
    ref2 = A.float().t() @ B.float()
    assert float((Cc.float() - ref2).abs().max()) <= max(tol, 1e-2 if dtype == torch.bfloat16 else 0) * float(ref2.abs().max())


@pytest.fixture
def split_products():
    """made_set_f32_products(1) for the duration of a test: every f32 product as three bf16 products on split operands (csrc/common.h)"""
    from mgsv_amd import _lib
    _lib.check(_lib.lib().made_set_f32_products(1), "made_set_f32_products")
    try:
        yield
    finally:
        _lib.check(_lib.lib().made_set_f32_products(0), "made_set_f32_products")


def test_split_bf16_products_of_the_training_kernels(T, split_products):
    """The split-product mode on the kernels of the training path (the engine's "f32x3" is an inference mode -- tests/test_trainer_gpu.py --, but the
    library-wide switch reaches every f32 product): made_gemm_tn, made_attention and made_attention_bwd in f32 against f32 torch math at
    2e-4 of the result's scale (exact mode: 2e-5 / 3e-5) -- the operands' dropped bits (2^-17 each), not an indexing mistake."""
    ops, tr = T
    from mgsv_amd import _lib
    assert _lib.lib().made_get_f32_products() == 1
    for M, N, K in ((300, 256, 128), (1000, 384, 520), (4096, 512, 512)):
        A, B = _rand(M, N, dtype=torch.float32, seed=1), _rand(M, K, dtype=torch.float32, seed=2)
        mask = (torch.rand(M, device="cuda") > 0.3).float()
        ref = (A * mask[:, None]).t() @ B
        C = torch.full((N, K), 7.0, device="cuda")
        tr.gemm_tn(A, B, C, row_mask=mask)
        err = float((C - ref).abs().max()) / float(ref.abs().max())
        assert 1e-7 < err <= 2e-4, (M, N, K, err)               # (not bit-equal to the exact mode either: the switch did reach the kernel)
    for B_, H, hd, Lq, Lk in ((2, 8, 64, 150, 150), (3, 4, 32, 70, 200), (1, 2, 128, 96, 50)):
        D = H * hd
        qkv = _rand(B_, max(Lq, Lk), 3 * D, dtype=torch.float32, seed=11)
        q, k, v = qkv[:, :Lq, :D], qkv[:, :Lk, D:2 * D], qkv[:, :Lk, 2 * D:]
        lens = torch.tensor([Lk - (i * 37) % max(Lk // 2, 1) for i in range(B_)], device="cuda")
        key_mask = (torch.arange(Lk, device="cuda")[None, :] < lens[:, None]).float()
        q_skip = key_mask if Lq == Lk else None
        qf, kf, vf = [t.clone().requires_grad_(True) for t in (q, k, v)]
        o_ref, lse_ref = _attn_ref(qf, kf, vf, H, key_mask, None, 0.0)
        dO = _rand(B_, Lq, D, dtype=torch.float32, seed=12)
        valid_q = (q_skip if q_skip is not None else torch.ones(B_, Lq, device="cuda"))[:, :, None]
        (o_ref * dO * valid_q).sum().backward()
        O = torch.zeros(B_, Lq, D, device="cuda")
        lse = torch.empty(B_, H, Lq, device="cuda")
        ops.attention(q, k, v, O, H, key_mask=key_mask, q_skip_mask=q_skip, lse=lse)
        sel = valid_q.bool().expand_as(O)
        assert float((O - o_ref)[sel].abs().max()) <= 2e-4 * float(o_ref.detach().abs().max())
        dqkv = torch.full((B_, max(Lq, Lk), 3 * D), float("nan"), device="cuda")
        delta = torch.empty(B_, H, Lq, device="cuda")
        tr.attention_bwd(q, k, v, O, dO, dqkv[:, :Lq, :D], dqkv[:, :Lk, D:2 * D], dqkv[:, :Lk, 2 * D:], lse, delta, H, key_mask=key_mask, q_skip_mask=q_skip)
        gscale = max(float(g.abs().max()) for g in (qf.grad, kf.grad, vf.grad))
        for name, got, ref in (("dq", dqkv[:, :Lq, :D], qf.grad), ("dk", dqkv[:, :Lk, D:2 * D], kf.grad), ("dv", dqkv[:, :Lk, 2 * D:], vf.grad)):
            assert torch.isfinite(got).all(), name
            assert float((got - ref).abs().max()) <= 2e-4 * max(float(ref.abs().max()), 0.05 * gscale) * 2, (name, hd)


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-5), (torch.bfloat16, 2e-3)])
def test_gemm_tn_skips_padded_slabs(T, dtype, tol):
    """prefix-valid sequences (long runs of padding): with the 32-row validity flags whole slabs are skipped, result unchanged
    even when the padded rows hold NaN."""
    ops, tr = T
    Bn, L, N, K = 12, 300, 256, 384
    lens = torch.tensor([300, 10, 150, 0, 299, 33, 64, 65, 1, 200, 128, 31], device="cuda")
    mask = (torch.arange(L, device="cuda")[None] < lens[:, None]).float().reshape(-1)
    M = Bn * L
    A, B = _rand(M, N, dtype=dtype, seed=1), _rand(M, K, dtype=dtype, seed=2)
    ref = (A.float() * mask[:, None]).t() @ B.float()
    A[mask == 0] = float("nan"); B[mask == 0] = float("nan")
    groups = tr.row_groups(mask)
    assert groups.numel() == (M + 31) // 32
    gref = torch.nn.functional.pad(mask, (0, (-M) % 32)).view(-1, 32).amax(1)
    assert torch.equal(groups, gref)
    for split in (1, 7, None):
        C = torch.zeros(N, K, device="cuda")
        cs = torch.zeros(N, device="cuda")
        tr.gemm_tn(A, B, C, accumulate=True, row_mask=mask, row_groups=groups, split_m=split, colsum=cs)
        assert torch.isfinite(C).all()
        assert float((C - ref).abs().max()) <= tol * float(ref.abs().max())


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-5), (torch.bfloat16, 2e-3)])
def test_gemm_tn_batched_and_summed(T, dtype, tol):
    ops, tr = T
    Z1, Z2, M, N, K = 3, 4, 70, 128, 256
    A, B = _rand(Z1, Z2, M, N, dtype=dtype, seed=3), _rand(Z1, Z2, M, K, dtype=dtype, seed=4)
    ref = torch.einsum("abmn,abmk->abnk", A.float(), B.float())
    Cb = torch.empty(Z1, Z2, N, K, device="cuda")
    tr.gemm_tn(A[0, 0], B[0, 0], Cb[0, 0], batch=(Z1, Z2), a_zs=(A.stride(0), A.stride(1)), b_zs=(B.stride(0), B.stride(1)),
               c_zs=(Cb.stride(0), Cb.stride(1)))
    assert float((Cb - ref).abs().max()) <= tol * float(ref.abs().max())
    # C stride 0 on the inner level: sum over it
    Cs = torch.zeros(Z1, N, K, device="cuda")
    tr.gemm_tn(A[0, 0], B[0, 0], Cs[0], batch=(Z1, Z2), a_zs=(A.stride(0), A.stride(1)), b_zs=(B.stride(0), B.stride(1)),
               c_zs=(Cs.stride(0), 0), accumulate=True)
    assert float((Cs - ref.sum(1)).abs().max()) <= tol * float(ref.sum(1).abs().max())
    # strided operands (a column block of a wider buffer), unaligned tiny head
    wide = _rand(200, 3 * 128, dtype=dtype, seed=5)
    X = _rand(200, 64, dtype=dtype, seed=6)
    Cw = torch.empty(128, 64, device="cuda")
    tr.gemm_tn(wide[:, 128:256], X, Cw)
    r = wide[:, 128:256].float().t() @ X.float()
    assert float((Cw - r).abs().max()) <= tol * float(r.abs().max())
    d2 = _rand(384, 2, dtype=torch.float32, seed=7)
    X2 = _rand(384, 256, dtype=torch.float32, seed=8)
    C2 = torch.zeros(2, 256, device="cuda")
    cs = torch.zeros(2, device="cuda")
    tr.gemm_tn(d2, X2, C2, accumulate=True, colsum=cs)
    assert float((C2 - d2.t() @ X2).abs().max()) <= 1e-4 * float((d2.t() @ X2).abs().max())
    assert float((cs - d2.sum(0)).abs().max()) <= 1e-4


def _keep(seed, site, p, shape):
    from mgsv_amd import dropout as dr
    n = int(np.prod(shape))
    return torch.from_numpy(dr.keep_mask(seed, site, p, n).reshape(shape)).cuda()


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-5), (torch.bfloat16, 2e-2)])
def test_linear_training_epilogue(T, dtype, tol):
    """Zout (pre-activation), act, dropout, residual; and the backward gates act'(G) * dropout."""
    ops, tr = T
    from mgsv_amd import _lib
    M, N, K = 300, 256, 128
    A, W = _rand(M, K, dtype=dtype, seed=1), _rand(N, K, dtype=dtype, seed=2) * 0.1
    bias = _rand(N, dtype=torch.float32, seed=3)
    R = _rand(M, N, dtype=dtype, seed=4)
    seed, site, p = 99, 1234567, 0.3
    keep = _keep(seed, site, p, (M, N)).float()
    z_ref = A.float() @ W.float().t() + bias
    for act, fn in ((ops.ACT_GELU, torch.nn.functional.gelu), (ops.ACT_RELU, torch.relu)):
        Z = torch.empty(M, N, device="cuda", dtype=dtype)
        out = ops.linear(A, W, bias, act=act, R=R, Zout=Z, drop=(seed, site, p))
        ref = fn(z_ref) * keep / (1 - p) + R.float()
        assert float((Z.float() - z_ref).abs().max()) <= tol * float(z_ref.abs().max())
        assert float((out.float() - ref).abs().max()) <= tol * float(ref.abs().max()) + (1e-6 if dtype == torch.float32 else 0.05)
    # backward gates: out = (A W^T) * act'(G) * scale, then dropout
    G = _rand(M, N, dtype=dtype, seed=5)
    base = A.float() @ W.float().t()
    g = G.float().clone().requires_grad_(True)
    torch.nn.functional.gelu(g).sum().backward()
    out = ops.linear(A, W, None, gate=_lib.GATE_GELU_Z, G=G, drop=(seed, site, p))
    ref = base * g.grad * keep / (1 - p)
    assert float((out.float() - ref).abs().max()) <= tol * float(ref.abs().max()) + 1e-6
    Gr = torch.relu(G)
    out = ops.linear(A, W, None, gate=_lib.GATE_RELU_OUT, G=Gr, gate_scale=1.25)
    ref = base * (Gr.float() != 0).float() * 1.25
    assert float((out.float() - ref).abs().max()) <= tol * float(ref.abs().max()) + 1e-6
    Gs = torch.sigmoid(G.float()).to(dtype)
    out = ops.linear(A, W, None, gate=_lib.GATE_SIGMOID_OUT, G=Gs)
    ref = base * Gs.float() * (1 - Gs.float())
    assert float((out.float() - ref).abs().max()) <= tol * float(ref.abs().max()) + 1e-6
    g2 = G.float().clone().requires_grad_(True)
    (g2 * torch.sigmoid(1.702 * g2)).sum().backward()
    out = ops.linear(A, W, None, gate=_lib.GATE_QUICKGELU_Z, G=G)
    ref = base * g2.grad
    assert float((out.float() - ref).abs().max()) <= tol * float(ref.abs().max()) + 1e-6


@pytest.mark.parametrize("K,N", [(512, 1024), (512, 512), (1024, 512), (256, 640)])
def test_linear_big_tile_training_epilogue(T, K, N, monkeypatch):
    """The persistent big-tile kernel (csrc/linear.hip linear_big_kernel, 128 x 256 tiles) with the training epilogue: Zout, ReLU,
    stateless dropout, residual, the backward's ReLU gate, row gather -- against f32 math on the bf16 operands, and against the
    single-stage kernel (MADE_LINEAR_TILE=64) on the same call: the same dropout draws, sums within bf16 rounding of each other."""
    ops, tr = T
    from mgsv_amd import _lib
    M, Tn = 12288, 512
    dtype = torch.bfloat16
    A, W = _rand(M, K, dtype=dtype, seed=1), _rand(N, K, dtype=dtype, seed=2) * (1.0 / math.sqrt(K))
    bias = _rand(N, dtype=torch.float32, seed=3)
    R = _rand(M, N, dtype=dtype, seed=4)
    lens = torch.tensor([(53 * i) % Tn + 1 for i in range(M // Tn)])
    mask = (torch.arange(Tn)[None] < lens[:, None]).float().cuda()
    rows = ops.row_index(mask)
    valid = mask.reshape(-1) != 0
    seed, site, p = 77, 424242, 0.1
    keep = _keep(seed, site, p, (M, N)).float()
    z_ref = A.float() @ W.float().t() + bias
    ref = torch.relu(z_ref) * keep / (1 - p) + R.float()
    got = {}
    for tile in ("256", "64"):
        monkeypatch.setenv("MADE_LINEAR_TILE", tile)
        Z = torch.full((M, N), float("nan"), device="cuda", dtype=dtype)
        out = torch.full((M, N), float("nan"), device="cuda", dtype=dtype)
        ops.linear(A, W, bias, act=ops.ACT_RELU, R=R, Zout=Z, drop=(seed, site, p), out=out, rows=rows)
        torch.cuda.synchronize()
        assert torch.isnan(out[~valid].float()).all() and torch.isnan(Z[~valid].float()).all()      # rows outside the list stay untouched
        assert float((Z[valid].float() - z_ref[valid]).abs().max()) <= 2e-2 * float(z_ref.abs().max())
        assert float((out[valid].float() - ref[valid]).abs().max()) <= 2e-2 * float(ref.abs().max()) + 0.05
        got[tile] = (out[valid].float(), Z[valid].float())
    assert ((got["256"][0] == 0) == (got["64"][0] == 0)).float().mean() > 0.999                   # the same draws
    assert float((got["256"][1] - got["64"][1]).abs().max()) <= 2e-2 * float(z_ref.abs().max())
    # the backward's gate on the same kernel: out = (A W^T) * [G != 0] * scale, no gather
    monkeypatch.setenv("MADE_LINEAR_TILE", "512")
    G = torch.relu(_rand(M, N, dtype=dtype, seed=5))
    out = ops.linear(A, W, None, gate=_lib.GATE_RELU_OUT, G=G, gate_scale=1.0 / 0.9)
    base = A.float() @ W.float().t()
    refg = base * (G.float() != 0).float() / 0.9
    assert float((out.float() - refg).abs().max()) <= 2e-2 * float(refg.abs().max()) + 1e-6
    # gate + dropout + a row-periodic f32 residual + the output row mask, f32 output (the straight-line epilogue's remaining branches)
    monkeypatch.setenv("MADE_LINEAR_TILE", "256")
    Rp = _rand(Tn, N, dtype=torch.float32, seed=6)
    orm = (torch.arange(M, device="cuda") % 7 != 3).float()
    out32 = torch.full((M, N), float("nan"), device="cuda", dtype=torch.float32)
    ops.linear(A, W, bias, gate=_lib.GATE_RELU_OUT, G=G, gate_scale=0.5, drop=(seed, site, p), R=Rp, r_row_mod=Tn, out_row_mask=orm, out=out32)
    ref32 = ((base + bias) * (G.float() != 0).float() * 0.5 * keep / (1 - p) + Rp.repeat(M // Tn, 1)) * orm[:, None]
    assert float((out32 - ref32).abs().max()) <= 2e-2 * float(ref32.abs().max()) + 1e-6


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_repack_ragged_shapes_and_store_words(T, dtype):
    """made_repack (64 x 64 tiles, vector and scalar paths): W and W^T copies of matrices with ragged shapes -- 2-row heads with a padded
    reduction dimension, rows / columns that are not multiples of 4 or 64, masters at unaligned offsets of one flat buffer -- in ONE launch;
    made_store_words: a few 32-bit words as kernel arguments."""
    import ctypes as C
    from mgsv_amd import _lib
    ops, tr = T
    shapes = [(2, 512), (130, 70), (512, 1024), (64, 64), (7, 5), (256, 258), (1, 64)]
    total = sum(r * c for r, c in shapes) + 3 * len(shapes)
    flat = _rand(total, dtype=torch.float32, seed=11)
    descs, keep, off, tiles = [], [], 1, 0                  # (offset 1: the first master is 4-byte aligned only)
    for r, c in shapes:
        m = flat[off:off + r * c].view(r, c)
        off += r * c + 3
        w = torch.full((r, c), float("nan"), device="cuda", dtype=dtype)
        ld = max(r, 64)
        wt = torch.zeros(c, ld, device="cuda", dtype=dtype)
        d = _lib.MadeRepackDesc()
        d.src, d.w, d.wt = m.data_ptr(), w.data_ptr(), wt.data_ptr()
        d.rows, d.cols, d.wt_ld, d.tile_begin = r, c, ld, tiles
        d.dtype = ops.dt_of(w)
        tiles += ((r + 63) // 64) * ((c + 63) // 64)
        descs.append(d); keep.append((m, w, wt))
    arr = (_lib.MadeRepackDesc * len(descs))(*descs)
    dev_descs = torch.from_numpy(np.frombuffer(bytes(arr), dtype=np.uint8).copy()).cuda()
    _lib.check(_lib.lib().made_repack(dev_descs.data_ptr(), len(descs), tiles, torch.cuda.current_stream().cuda_stream), "made_repack")
    torch.cuda.synchronize()
    for (m, w, wt), (r, c) in zip(keep, shapes):
        assert torch.equal(w, m.to(dtype)), (r, c)
        assert torch.equal(wt[:, :r], m.to(dtype).t()), (r, c)
        assert bool((wt[:, r:] == 0).all())                 # the padded reduction columns stay zero
    # made_store_words
    buf = torch.zeros(6, device="cuda", dtype=torch.int32)
    words = (C.c_uint32 * 3)(0xDEADBEEF, 7, 0x80000001)
    _lib.check(_lib.lib().made_store_words(buf[1:].data_ptr(), words, 3, torch.cuda.current_stream().cuda_stream), "made_store_words")
    torch.cuda.synchronize()
    assert buf.cpu().numpy().astype(np.uint32).tolist() == [0, 0xDEADBEEF, 7, 0x80000001, 0, 0]


def _attn_ref(q, k, v, H, key_mask, keep, p, scale=None):
    B, Lq, D = q.shape
    Lk, hd = k.shape[1], D // H
    sc = hd ** -0.5 if scale is None else scale
    qh = q.view(B, Lq, H, hd).transpose(1, 2); kh = k.view(B, Lk, H, hd).transpose(1, 2); vh = v.view(B, Lk, H, hd).transpose(1, 2)
    s = qh @ kh.transpose(-1, -2) * sc
    if key_mask is not None:
        s = s.masked_fill((key_mask == 0)[:, None, None, :], float("-inf"))
    lse = torch.logsumexp(s, dim=-1)
    a = torch.softmax(s, dim=-1)
    if keep is not None:
        a = a * keep / (1 - p)
    return (a @ vh).transpose(1, 2).reshape(B, Lq, D), lse


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 3e-5), (torch.bfloat16, 2.5e-2)])
@pytest.mark.parametrize("B,H,hd,Lq,Lk,p", [(2, 8, 64, 150, 150, 0.0), (3, 4, 32, 70, 200, 0.1), (2, 2, 64, 1, 1, 0.5),
                                            (2, 8, 64, 542, 542, 0.8), (1, 2, 128, 96, 50, 0.0)])
def test_attention_forward_dropout_lse_and_backward(T, dtype, tol, B, H, hd, Lq, Lk, p):
    ops, tr = T
    D = H * hd
    qkv = _rand(B, max(Lq, Lk), 3 * D, dtype=dtype, seed=11)
    q, k, v = qkv[:, :Lq, :D], qkv[:, :Lk, D:2 * D], qkv[:, :Lk, 2 * D:]
    lens = torch.tensor([Lk - (i * 37) % max(Lk // 2, 1) for i in range(B)], device="cuda")
    key_mask = (torch.arange(Lk, device="cuda")[None, :] < lens[:, None]).float()
    q_skip = key_mask if Lq == Lk else None
    seed, site = 4242, 777
    keep = _keep(seed, site, p, (B, H, Lq, Lk)).float() if p > 0 else None
    qf, kf, vf = [t.float().clone().requires_grad_(True) for t in (q, k, v)]
    o_ref, lse_ref = _attn_ref(qf, kf, vf, H, key_mask, keep, p)
    dO = _rand(B, Lq, D, dtype=dtype, seed=12)
    valid_q = (q_skip if q_skip is not None else torch.ones(B, Lq, device="cuda"))[:, :, None]
    (o_ref * dO.float() * valid_q).sum().backward()

    O = torch.zeros(B, Lq, D, device="cuda", dtype=dtype)
    lse = torch.empty(B, H, Lq, device="cuda")
    ops.attention(q, k, v, O, H, key_mask=key_mask, q_skip_mask=q_skip, lse=lse, drop=(seed, site, p))
    sel = valid_q.bool().expand_as(O)
    scale_o = float(o_ref.detach().abs().max())
    assert float((O.float() - o_ref)[sel].abs().max()) <= tol * scale_o
    lsel = valid_q[:, None, :, 0].bool().expand_as(lse)
    assert float((lse - lse_ref)[lsel].abs().max()) <= (1e-4 if dtype == torch.float32 else 3e-2)

    dqkv = torch.full((B, max(Lq, Lk), 3 * D), float("nan"), device="cuda", dtype=dtype)
    dq, dk, dv = dqkv[:, :Lq, :D], dqkv[:, :Lk, D:2 * D], dqkv[:, :Lk, 2 * D:]
    delta = torch.empty(B, H, Lq, device="cuda")
    tr.attention_bwd(q, k, v, O, dO, dq, dk, dv, lse, delta, H, key_mask=key_mask, q_skip_mask=q_skip, drop=(seed, site, p))
    gscale = max(float(g.abs().max()) for g in (qf.grad, kf.grad, vf.grad))      # (dq = dk = 0 exactly when Lk == 1)
    for name, got, ref in (("dq", dq, qf.grad), ("dk", dk, kf.grad), ("dv", dv, vf.grad)):
        got = got.float()
        assert torch.isfinite(got).all(), name
        err = float((got - ref).abs().max())
        assert err <= tol * max(float(ref.abs().max()), 0.05 * gscale) * 2, (name, err, float(ref.abs().max()))
    if p > 0:
        # the forward's decision cache (one bit per score, made_attention's keep_bits): equal to the stateless mask where it is defined
        # (valid queries, keys up to the sample's last valid one), and a backward that reads it gives the gradients of the one that
        # re-draws the mask, bit for bit (bf16; the f32 parity path always re-draws)
        bits = torch.full(ops.attention_bits_shape(B, H, Lq, Lk), 0x5A5A5A5A, device="cuda", dtype=torch.int32)
        O2 = torch.zeros_like(O)
        ops.attention(q, k, v, O2, H, key_mask=key_mask, q_skip_mask=q_skip, lse=lse, drop=(seed, site, p), keep_bits=bits)
        torch.cuda.synchronize()
        assert torch.equal(O2, O)
        got_bits = ops.attention_bits_decode(bits, B, H, Lq, Lk).float()
        defined = (key_mask[:, None, None, :] * valid_q[:, None, :, :]).bool().expand_as(keep)
        assert torch.equal(got_bits[defined], keep[defined])
        if dtype == torch.bfloat16:
            # ... and a backward that reads the cache gives the gradients of the one that re-draws the mask: bit for bit where both run the split
            # kernels (head dims 32 / 128), within bf16 rounding where the cache selects the single-pass kernel (head dim 64: one recomputation
            # of the probabilities for dQ, dK and dV; other summation order)
            dqkv2 = torch.full_like(dqkv, float("nan"))
            tr.attention_bwd(q, k, v, O, dO, dqkv2[:, :Lq, :D], dqkv2[:, :Lk, D:2 * D], dqkv2[:, :Lk, 2 * D:], lse, delta, H, key_mask=key_mask,
                             q_skip_mask=q_skip, drop=(seed, site, p), keep_bits=bits)
            torch.cuda.synchronize()
            for a_, b_, ref in ((dqkv2[:, :Lq, :D], dq, qf.grad), (dqkv2[:, :Lk, D:2 * D], dk, kf.grad), (dqkv2[:, :Lk, 2 * D:], dv, vf.grad)):
                if hd == 64:
                    assert torch.isfinite(a_.float()).all()
                    err = float((a_.float() - ref).abs().max())
                    assert err <= tol * max(float(ref.abs().max()), 0.05 * gscale) * 2, err
                else:
                    assert torch.equal(a_.contiguous().view(torch.int16), b_.contiguous().view(torch.int16))


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-5), (torch.bfloat16, 2e-2)])
@pytest.mark.parametrize("rows,D", [(1000, 512), (333, 256), (64, 1024), (20011, 512), (1500, 768)])
def test_layernorm_bwd(T, dtype, tol, rows, D):
    ops, tr = T
    x, dy, add = _rand(rows, D, dtype=dtype, seed=1), _rand(rows, D, dtype=dtype, seed=2), _rand(rows, D, dtype=dtype, seed=3)
    gamma = _rand(D, dtype=torch.float32, seed=4) * 0.3 + 1.0
    skip = (torch.rand(rows, device="cuda") > 0.2).float()
    xr = x.float().clone().requires_grad_(True)
    gr = gamma.clone().requires_grad_(True)
    br = torch.zeros(D, device="cuda", requires_grad=True)
    y = torch.nn.functional.layer_norm(xr, (D,), gr, br)
    (y * dy.float() * skip[:, None]).sum().backward()
    seed, site, p = 5, 6, 0.25
    dx = torch.empty(rows, D, device="cuda", dtype=dtype)
    dxd = torch.empty(rows, D, device="cuda", dtype=dtype)
    dg, db = torch.ones(D, device="cuda"), torch.ones(D, device="cuda")
    dx.fill_(3.0); dxd.fill_(3.0)
    tr.layernorm_bwd(x, gamma, dy, dx, dgamma=dg, dbeta=db, add=add, dx_drop=dxd, drop=(seed, site, p), row_skip=skip)
    ref = xr.grad + add.float()
    v = skip.bool()
    sc = float(ref[v].abs().max())
    assert float((dx.float() - ref)[v].abs().max()) <= tol * sc
    keep = _keep(seed, site, p, (rows, D)).float()
    assert float((dxd.float() - ref * keep / (1 - p))[v].abs().max()) <= tol * sc / (1 - p)
    assert bool((dx[~v] == 3.0).all()) and bool((dxd[~v] == 3.0).all())       # skipped rows are left untouched
    assert float((dg - 1 - gr.grad).abs().max()) <= tol * float(gr.grad.abs().max()) * 2
    assert float((db - 1 - br.grad).abs().max()) <= tol * float(br.grad.abs().max()) * 2


def test_pool_and_l2norm_bwd(T):
    ops, tr = T
    B, Tn, D = 5, 37, 256
    local = _rand(B, Tn, D, dtype=torch.float32, seed=1).requires_grad_(True)
    lens = torch.tensor([37, 20, 5, 1, 30], device="cuda")
    mask = (torch.arange(Tn, device="cuda")[None] < lens[:, None]).float()
    mean = (local * mask[:, :, None]).sum(1) / mask.sum(1, keepdim=True)
    vec = torch.nn.functional.normalize(mean, dim=-1)
    dvec = _rand(B, D, dtype=torch.float32, seed=2)
    in1 = _rand(B, Tn, D, dtype=torch.float32, seed=3)
    in2 = _rand(B, Tn + 3, D, dtype=torch.bfloat16, seed=4)[:, 3:]
    (vec * dvec).sum().backward()
    out = torch.empty(B, Tn, D, device="cuda")
    tr.pool_bwd(mean.detach().contiguous(), dvec, mask, out, in1=in1, in2=in2)
    ref = (local.grad + in1 + in2.float()) * mask[:, :, None]
    assert float((out - ref).abs().max()) <= 1e-5 * float(ref.abs().max())
    # row L2 normalisation
    x = _rand(50, D, dtype=torch.float32, seed=5).requires_grad_(True)
    dy = _rand(50, D, dtype=torch.float32, seed=6)
    (torch.nn.functional.normalize(x, dim=-1) * dy).sum().backward()
    dx = torch.ones(50, D, device="cuda")
    alt = torch.empty(50, D, device="cuda", dtype=torch.bfloat16)
    tr.l2norm_bwd(x.detach(), dy, dx, accumulate=True, dx_alt=alt)
    assert float((dx - 1 - x.grad).abs().max()) <= 1e-5 * float(x.grad.abs().max())
    assert float((alt.float() - x.grad).abs().max()) <= 1e-2 * float(x.grad.abs().max())


def test_clip_loss_bwd(T):
    ops, tr = T
    n = 70
    sims = (_rand(n, n, dtype=torch.float32, seed=1) * 0.3).requires_grad_(True)
    ls = torch.tensor([3.5], device="cuda", requires_grad=True)
    z = sims * ls.exp()
    loss = 0.5 * (-torch.diagonal(torch.log_softmax(z, 1)).mean() - torch.diagonal(torch.log_softmax(z, 0)).mean())
    up = torch.tensor([0.7], device="cuda")
    (loss * 1.3 * up).backward()
    ws = torch.empty(2 * n, device="cuda")
    d, dt_ = torch.empty(n, n, device="cuda"), torch.empty(n, n, device="cuda")
    dls = torch.zeros(1, device="cuda")
    tr.clip_loss_bwd(sims.detach(), ls.detach(), 1.3, up, ws, d, dt_, dls)
    assert float((d - sims.grad).abs().max()) <= 2e-5 * float(sims.grad.abs().max())
    assert float((dt_ - sims.grad.t()).abs().max()) <= 2e-5 * float(sims.grad.abs().max())
    assert abs(float(dls) - float(ls.grad)) <= 1e-4 * abs(float(ls.grad))


@pytest.mark.parametrize("Nm,Nv,D", [(6, 9, 256), (20, 24, 768)])      # (the second: more than 256 rows at D > 512 -- the eight-wave workgroups)
@pytest.mark.parametrize("dtype,tol", [(torch.float32, 3e-5), (torch.bfloat16, 2e-2)])
def test_xpool_tail_bwd(T, dtype, tol, Nm, Nv, D):
    ops, tr = T
    y = _rand(Nm * Nv, D, dtype=dtype, seed=1)
    gamma = _rand(D, dtype=torch.float32, seed=2) * 0.2 + 1
    beta = _rand(D, dtype=torch.float32, seed=3) * 0.1
    video = _rand(Nv, D, dtype=torch.float32, seed=4)
    dsims = _rand(Nv, Nm, dtype=torch.float32, seed=5)
    yr, gr, br, vr = [t.float().clone().requires_grad_(True) for t in (y, gamma, beta, video)]
    p_ = torch.nn.functional.layer_norm(yr, (D,), gr, br).view(Nm, Nv, D)
    sims = torch.einsum("nd,mnd->nm", vr / vr.norm(dim=-1, keepdim=True), p_ / p_.norm(dim=-1, keepdim=True))
    (sims * dsims).sum().backward()
    dy = torch.empty(Nm * Nv, D, device="cuda", dtype=dtype)
    dyd = torch.empty_like(dy)
    dg, db, dv = torch.zeros(D, device="cuda"), torch.zeros(D, device="cuda"), torch.zeros(Nv, D, device="cuda")
    tr.xpool_tail_bwd(y, gamma, beta, video, dsims, dy, Nm, Nv, dy_drop=dyd, drop=(3, 4, 0.3), dgamma=dg, dbeta=db, dvideo=dv)
    sc = float(yr.grad.abs().max())
    assert float((dy.float() - yr.grad).abs().max()) <= tol * sc
    keep = _keep(3, 4, 0.3, (Nm * Nv, D)).float()
    assert float((dyd.float() - yr.grad * keep / 0.7).abs().max()) <= tol * sc / 0.7
    assert float((dg - gr.grad).abs().max()) <= 1e-4 * float(gr.grad.abs().max())
    assert float((db - br.grad).abs().max()) <= 1e-4 * float(br.grad.abs().max())
    assert float((dv - vr.grad).abs().max()) <= 1e-4 * float(vr.grad.abs().max())


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-5), (torch.bfloat16, 1e-2)])
def test_softmax_bwd_rows(T, dtype, tol):
    ops, tr = T
    Z, rpb, L = 5, 8, 77
    ldo = 80
    S = _rand(Z * rpb, L, dtype=torch.float32, seed=1)
    dPd = _rand(Z * rpb, L, dtype=torch.float32, seed=2)
    extra = _rand(Z * rpb, dtype=torch.float32, seed=3)
    lens = torch.tensor([77, 40, 9, 60, 1], device="cuda")
    mask = (torch.arange(L, device="cuda")[None] < lens[:, None]).float()
    scale, p, seed, site = 0.37, 0.2, 8, 9
    keep = _keep(seed, site, p, (Z * rpb, L)).float()
    Sr = S.clone().requires_grad_(True)
    logits = (Sr * scale).masked_fill((mask == 0).repeat_interleave(rpb, 0), float("-inf"))
    Pd_ref = torch.softmax(logits, -1) * keep / (1 - p)
    (Pd_ref * (dPd + extra[:, None])).sum().backward()
    Pd = torch.empty(Z * rpb, ldo, device="cuda", dtype=dtype)
    dS = torch.empty_like(Pd)
    dSt = torch.zeros(Z, L, rpb, device="cuda", dtype=dtype)
    tr.softmax_bwd(S, dPd, mask, rpb, scale, Pd, dS, dSt, rpb, L, extra=extra, drop=(seed, site, p), ldt=rpb)
    assert float((Pd[:, :L].float() - Pd_ref).abs().max()) <= tol
    assert float(Pd[:, L:].abs().max()) == 0 and float(dS[:, L:].abs().max()) == 0
    sc = float(Sr.grad.abs().max())
    assert float((dS[:, :L].float() - Sr.grad).abs().max()) <= tol * sc
    assert float((dSt.float() - Sr.grad.view(Z, rpb, L).transpose(1, 2)).abs().max()) <= tol * sc


def test_head_bias_and_add3(T):
    ops, tr = T
    rows, H, hd = 10, 8, 32
    x = _rand(rows, H * hd, dtype=torch.float32, seed=1)
    s = _rand(rows, H, dtype=torch.float32, seed=2)
    bias = _rand(H * hd, dtype=torch.float32, seed=3)
    ref = x + (s[:, :, None] * bias.view(H, hd)[None]).reshape(rows, -1)
    y = x.clone()
    tr.head_bias(y, s, bias, H)
    assert float((y - ref).abs().max()) <= 1e-6
    dy = _rand(rows, H * hd, dtype=torch.float32, seed=4)
    dbias, ds = torch.zeros(H * hd, device="cuda"), torch.empty(rows, H, device="cuda")
    tr.head_bias_bwd(dy, s, bias, dbias, ds, H)
    assert float((dbias - (dy.view(rows, H, hd) * s[:, :, None]).sum(0).reshape(-1)).abs().max()) <= 1e-5
    assert float((ds - (dy.view(rows, H, hd) * bias.view(1, H, hd)).sum(-1)).abs().max()) <= 1e-5
    a, b, c = _rand(1000, dtype=torch.float32, seed=5), _rand(1000, dtype=torch.bfloat16, seed=6), _rand(1000, dtype=torch.float32, seed=7)
    out = torch.empty(1000, device="cuda", dtype=torch.bfloat16)
    tr.add3(out, a, b, c)
    assert float((out.float() - (a + b.float() + c)).abs().max()) <= 3e-2


@pytest.mark.parametrize("Q,G", [(1, 1), (3, 2), (5, 4)])
def test_set_criterion_bwd(T, Q, G):
    """against autograd through the oracle's criterion (same matcher indices)."""
    ops, tr = T
    from oracle import made_oracle as O
    from mgsv_amd.config import cfg_native
    cfg = cfg_native(); cfg.num_moment_queries = Q
    nd, B, Dc = 3, 6, 64
    g = torch.Generator().manual_seed(5)
    logits = torch.randn(nd, B, Q, 2, generator=g)
    spans = torch.rand(nd, B, Q, 2, generator=g) * 0.5 + 0.2
    tg = torch.rand(B, G, 2, generator=g) * 0.4 + 0.2
    if G > 1:
        tg[1, 1, 1] = 0.0                                              # a zero-width target is dropped
    pq = torch.nn.functional.normalize(torch.randn(nd, B, Q, Dc, generator=g), dim=-1)
    pv = torch.nn.functional.normalize(torch.randn(B, 7, Dc, generator=g), dim=-1)
    P = {"criterion.empty_weight": torch.tensor([1.0, 0.1]) if cfg.foreground_label == 0 else torch.tensor([0.1, 1.0])}
    lr, sr, pqr, pvr = [t.clone().requires_grad_(True) for t in (logits, spans, pq, pv)]
    wd = O.criterion_weight_dict(cfg)
    total = 0
    for l in range(nd):
        ld = O.set_criterion({"pred_logits": lr[l], "pred_spans": sr[l], "proj_queries": pqr[l], "proj_vid_mem": pvr}, tg, P, cfg)
        total = total + sum(v * wd[k] for k, v in ld.items() if k in wd)
    up = 0.9
    (total * up).backward()
    dev = "cuda"
    lg, sp, tgd, pqd = logits.to(dev), spans.to(dev), tg.to(dev), pq.to(dev).contiguous()
    vid_sum = pv.sum(1).to(dev).contiguous()
    pi, ti, cnt, status, cost = ops.hungarian_match(lg.view(nd * B, Q, 2), sp.view(nd * B, Q, 2), tgd, cfg.foreground_label)
    w = torch.tensor([4.0, 1.0, 0.8, 0.0, 0.2], device=dev)
    dl, ds = torch.empty_like(lg), torch.empty_like(sp)
    dpq, dvs = torch.empty_like(pqd), torch.zeros_like(vid_sum)
    tr.set_criterion_bwd(lg, sp, tgd, pi, ti, cnt, pqd, vid_sum, P["criterion.empty_weight"].to(dev), cfg.foreground_label, w,
                         torch.tensor([up], device=dev), dl, ds, dpq, dvs)
    for name, got, ref in (("dlogits", dl, lr.grad), ("dspans", ds, sr.grad), ("dpq", dpq, pqr.grad), ("dvid_sum", dvs, pvr.grad[:, 0])):
        assert float((got.cpu() - ref).abs().max()) <= 2e-5 * max(float(ref.abs().max()), 1e-3), name


@pytest.mark.parametrize("Bn,N", [(9, 384), (60, 640), (200, 1280)])     # skinny / 64-row-tile / 128-row-tile kernels in bf16
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_row_gather_linear_and_gemm_tn_bit_identical_on_valid_rows(T, dtype, Bn, N):
    """rows=(row_index, n_rows): the GEMMs touch the valid tokens only; their results equal the ungathered call bit for bit,
    padded rows are neither read (they hold NaN here) nor written (sentinel survives)."""
    ops, tr = T
    L, K = 150, 256
    lens = torch.tensor([150, 3, 77, 0, 149, 128, 1, 64, 100], device="cuda")
    if Bn > 9:
        g = torch.Generator().manual_seed(Bn)
        lens = torch.cat([lens, torch.randint(0, L + 1, (Bn - 9,), generator=g).cuda()])
    mask = (torch.arange(L, device="cuda")[None] < lens[:, None]).float().reshape(-1)
    M = Bn * L
    rows = ops.row_index(mask)
    idx_ref = torch.nonzero(mask).flatten().int()
    assert int(rows[1]) == idx_ref.numel() and torch.equal(rows[0][:idx_ref.numel()], idx_ref)
    assert bool((rows[0][idx_ref.numel():] == idx_ref[-1]).all())
    A, W = _rand(M, K, dtype=dtype, seed=1), _rand(N, K, dtype=dtype, seed=2) * 0.1
    bias = _rand(N, dtype=torch.float32, seed=3)
    R = _rand(M, N, dtype=dtype, seed=4)
    full = ops.linear(A, W, bias, act=ops.ACT_GELU, R=R, drop=(5, 6, 0.3))
    A2 = A.clone(); A2[mask == 0] = float("nan")
    R2 = R.clone(); R2[mask == 0] = float("nan")
    out = torch.full((M, N), 7.0, device="cuda", dtype=dtype)
    ops.linear(A2, W, bias, act=ops.ACT_GELU, R=R2, drop=(5, 6, 0.3), out=out, rows=rows)
    v = mask.bool()
    assert torch.equal(out[v], full[v])
    assert bool((out[~v] == 7.0).all())
    # weight gradient over the valid rows only
    dY, X = _rand(M, N, dtype=dtype, seed=7), _rand(M, K, dtype=dtype, seed=8)
    ref = (dY.float() * mask[:, None]).t() @ X.float()
    dY[mask == 0] = float("nan"); X[mask == 0] = float("nan")
    for split in (1, 5, None):
        Cg = torch.zeros(N, K, device="cuda"); cs = torch.zeros(N, device="cuda")
        tr.gemm_tn(dY, X, Cg, accumulate=True, colsum=cs, rows=rows, split_m=split)
        assert torch.isfinite(Cg).all() and torch.isfinite(cs).all()
        tol = 2e-5 if dtype == torch.float32 else 2e-3
        assert float((Cg - ref).abs().max()) <= tol * float(ref.abs().max())


@pytest.mark.parametrize("M,N,K", [(1000, 256, 128), (4133, 512, 512), (333, 128, 384), (20000, 1024, 512)])
def test_gemm_tn_direct_to_lds_path(T, M, N, K):
    """bf16, N and K multiples of 128, no row mask (or a row list): the 3-stage direct-to-LDS kernel, incl. a ragged last slab."""
    ops, tr = T
    A, B = _rand(M, N, dtype=torch.bfloat16, seed=1), _rand(M, K, dtype=torch.bfloat16, seed=2)
    ref = A.float().t() @ B.float()
    sc = float(ref.abs().max())
    for split in (1, 3, None):
        C = torch.zeros(N, K, device="cuda"); cs = torch.zeros(N, device="cuda")
        tr.gemm_tn(A, B, C, accumulate=True, colsum=cs, split_m=split)
        assert float((C - ref).abs().max()) <= 2e-3 * sc, split
        assert float((cs - A.float().sum(0)).abs().max()) <= 2e-3 * float(A.float().sum(0).abs().max()) + 0.05
    Cs = torch.empty(N, K, device="cuda", dtype=torch.bfloat16)
    tr.gemm_tn(A, B, Cs)
    assert float((Cs.float() - ref).abs().max()) <= 1e-2 * sc
    mask = (torch.rand(M, device="cuda") > 0.4).float()
    rows = ops.row_index(mask)
    refm = (A.float() * mask[:, None]).t() @ B.float()
    A2 = A.clone(); A2[mask == 0] = float("nan")
    C = torch.zeros(N, K, device="cuda")
    tr.gemm_tn(A2, B, C, accumulate=True, rows=rows)
    assert torch.isfinite(C).all() and float((C - refm).abs().max()) <= 2e-3 * float(refm.abs().max())


@pytest.mark.parametrize("M", [4096, 9000, 34688])
@pytest.mark.parametrize("tile", ["256", "128"])
def test_gemm_tn_grouped_tiles(T, M, tile, monkeypatch):
    """several weight gradients over the same (gathered) rows in one launch: the 256 x 256-tile kernel (eight waves) and the 128 x 128
    one give the same gradients and bias gradients, ragged last slab and a device-side row count included."""
    ops, tr = T
    monkeypatch.setenv("MADE_TN_TILE", tile)
    shapes = [(512, 1024), (1024, 512), (256, 512), (512, 256)]
    mask = (torch.rand(M, device="cuda") > 0.25).float()
    rows = ops.row_index(mask)
    for use_rows in (False, True):
        probs, refs = [], []
        for i, (N, K) in enumerate(shapes):
            A, B = _rand(M, N, dtype=torch.bfloat16, seed=10 + i), _rand(M, K, dtype=torch.bfloat16, seed=20 + i)
            Am = A.float() * mask[:, None] if use_rows else A.float()
            refs.append((Am.t() @ B.float(), Am.sum(0)))
            if use_rows:
                A = A.clone(); A[mask == 0] = float("nan")
            probs.append((A, B, torch.ones(N, K, device="cuda"), torch.ones(N, device="cuda") if i != 2 else None))
        for split in (None, 16):
            for (A, B, Cw, cs) in probs:
                Cw.fill_(1.0)
                if cs is not None: cs.fill_(1.0)
            tr.gemm_tn_grouped(probs, rows=rows if use_rows else None, alpha=0.5, split_m=split)
            for (A, B, Cw, cs), (rC, rs) in zip(probs, refs):
                assert torch.isfinite(Cw).all()
                assert float((Cw - (1.0 + 0.5 * rC)).abs().max()) <= 2e-3 * float(rC.abs().max()), (use_rows, split)
                if cs is not None:
                    assert float((cs - (1.0 + 0.5 * rs)).abs().max()) <= 2e-3 * float(rs.abs().max()) + 0.05


@pytest.mark.parametrize("M", [4096, 9000, 34688])
@pytest.mark.parametrize("shapes", [[(512, 1024), (1024, 512), (256, 512), (512, 256)],
                                    [(512, 1024), (1024, 512), (512, 512), (1024, 512), (512, 512)], [(256, 256)]])
def test_gemm_tn_grouped_workspace_flush(T, M, shapes):
    """the 256 x 256-tile weight-gradient launch with a workspace (tile partials stored, summed by a second launch instead of added with
    atomics): same gradients as the f32 product, BITWISE the same from launch to launch (fixed summing order), the workspace reusable as it
    is, tile counts that do and do not divide the workgroups of an XCD, row gather on and off."""
    ops, tr = T
    mask = (torch.rand(M, device="cuda") > 0.45).float()
    rows = ops.row_index(mask)
    holder = {}

    def workspace(n):
        if "ws" not in holder or holder["ws"].numel() < n:
            holder["ws"] = tr.gemm_tn_grouped_workspace(torch.device("cuda"), n)
        return holder["ws"]
    for use_rows in (False, True):
        probs, refs = [], []
        for i, (N, K) in enumerate(shapes):
            A, B = _rand(M, N, dtype=torch.bfloat16, seed=10 + i), _rand(M, K, dtype=torch.bfloat16, seed=20 + i)
            Am = A.float() * mask[:, None] if use_rows else A.float()
            refs.append((Am.t() @ B.float(), Am.sum(0)))
            if use_rows:
                A = A.clone(); A[mask == 0] = float("nan")
            probs.append((A, B, torch.ones(N, K, device="cuda"), torch.ones(N, device="cuda") if i != 2 else None))
        first = None
        for rep in range(3):
            for (A, B, Cw, cs) in probs:
                Cw.fill_(1.0)
                if cs is not None: cs.fill_(1.0)
            tr.gemm_tn_grouped(probs, rows=rows if use_rows else None, alpha=0.5, workspace=workspace)
            for (A, B, Cw, cs), (rC, rs) in zip(probs, refs):
                assert torch.isfinite(Cw).all()
                assert float((Cw - (1.0 + 0.5 * rC)).abs().max()) <= 2e-3 * float(rC.abs().max()), (use_rows, rep)
                if cs is not None:
                    assert float((cs - (1.0 + 0.5 * rs)).abs().max()) <= 2e-3 * float(rs.abs().max()) + 0.05
            got = [p[2].clone() for p in probs]
            if first is None:
                first = got
            else:
                assert all(torch.equal(a, b) for a, b in zip(first, got)), "the workspace flush sums in a fixed order"
        assert "ws" in holder


def test_clip_loss_same_music_exclusion_forward_and_backward(T):
    """row_exclude of made_clip_loss / made_clip_loss_bwd = the same-music-aware InfoNCE of reference modules/loss.py:90-114
    (oracle restatement info_nce_same_music): loss and d(loss)/d(sims), d/d(logit_scale) against its autograd."""
    ops, tr = T
    from oracle import made_oracle as O
    n = 12
    ids = [str(i % 5) for i in range(n)]                      # tracks shared by two or three videos
    sims = (_rand(n, n, dtype=torch.float32, seed=3) * 0.3).contiguous()
    ls = torch.tensor([math.log(1 / 0.07)], device="cuda")
    idx = torch.tensor([int(i) for i in ids])
    ex = (idx[:, None] == idx[None, :]).float()
    ex.fill_diagonal_(0.0)
    ex = ex.cuda()
    loss = torch.zeros(1, device="cuda")
    ops.clip_loss(sims, ls, loss, weight=1.0, row_exclude=ex)
    s_ref = sims.cpu().double().requires_grad_(True)
    l_ref = ls.cpu().double()[0].requires_grad_(True)
    ref = O.info_nce_same_music(s_ref, l_ref, ids)
    ref.backward()
    assert abs(float(loss) - float(ref)) <= 1e-5 * max(1.0, abs(float(ref)))
    ds, dst, gls = torch.zeros(n, n, device="cuda"), torch.zeros(n, n, device="cuda"), torch.zeros(1, device="cuda")
    tr.clip_loss_bwd(sims, ls, 1.0, None, torch.empty(2 * n, device="cuda"), ds, dst, gls, row_exclude=ex)
    np.testing.assert_allclose(ds.cpu().numpy(), s_ref.grad.float().numpy(), atol=2e-6, rtol=1e-4)
    np.testing.assert_allclose(dst.cpu().numpy(), s_ref.grad.float().numpy().T, atol=2e-6, rtol=1e-4)
    np.testing.assert_allclose(float(gls), float(l_ref.grad), rtol=1e-4, atol=1e-5)
    assert float(ds[0, 5].abs()) > 0 and bool((ex[0, 5] == 1))       # an excluded pair still gets the column-direction gradient


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-5), (torch.bfloat16, 2e-2)])
@pytest.mark.parametrize("B,Tn,F", [(5, 20, 1024), (3, 7, 256), (128, 50, 256), (1, 3, 8)])
@pytest.mark.parametrize("batch_stats", [True, False])
def test_posbn_relu_forward_backward_match_torch_batchnorm(T, dtype, tol, B, Tn, F, batch_stats):
    """nn.BatchNorm1d(num_features = positions) on [B, positions, F] followed by ReLU (EmbeddingNet, reference model_Base.py:216-249):
    batch statistics + running-buffer update in train mode, the running statistics in eval mode; dx / dweight / dbias of both."""
    ops, tr = T
    x, dy = _rand(B, Tn, F, dtype=dtype, seed=1) * 1.5 + 0.3, _rand(B, Tn, F, dtype=dtype, seed=2)
    momentum = 0.99
    bn = torch.nn.BatchNorm1d(Tn, momentum=momentum).cuda()
    with torch.no_grad():
        bn.weight.copy_(_rand(Tn, dtype=torch.float32, seed=3) * 0.3 + 1.0); bn.bias.copy_(_rand(Tn, dtype=torch.float32, seed=4) * 0.2)
        bn.running_mean.copy_(_rand(Tn, dtype=torch.float32, seed=5) * 0.1 + 0.3); bn.running_var.copy_(torch.rand(Tn, device="cuda") + 1.5)
    rm, rv = bn.running_mean.clone(), bn.running_var.clone()
    bn.train(batch_stats)
    xr = x.float().clone().requires_grad_(True)
    yr = torch.relu(bn(xr))
    (yr * dy.float()).sum().backward()
    y = torch.empty_like(x)
    sm, sr = torch.empty(Tn, device="cuda"), torch.empty(Tn, device="cuda")
    w, b = bn.weight.detach().clone(), bn.bias.detach().clone()
    tr.posbn_relu(x, w, b, rm, rv, momentum, batch_stats, sm, sr, out=y)
    sc = float(yr.abs().max())
    assert float((y.float() - yr).abs().max()) <= tol * sc
    if B * F > 1 or not batch_stats:
        assert float((rm - bn.running_mean).abs().max()) <= 1e-5 and float((rv - bn.running_var).abs().max()) <= 1e-4 * float(bn.running_var.max())
    dx = torch.empty_like(x)
    dw, db = torch.ones(Tn, device="cuda"), torch.ones(Tn, device="cuda")
    tr.posbn_relu_bwd(x, y, dy, w, sm, sr, batch_stats, dx, dw, db)
    if dtype == torch.float32:      # (bf16: a value the rounding of y moves across the ReLU's zero flips a whole gradient entry -- compared in f32 only)
        assert float((dx.float() - xr.grad).abs().max()) <= 5 * tol * float(xr.grad.abs().max())
        assert float((dw - 1 - bn.weight.grad).abs().max()) <= 5 * tol * max(float(bn.weight.grad.abs().max()), 1.0)
        assert float((db - 1 - bn.bias.grad).abs().max()) <= 5 * tol * max(float(bn.bias.grad.abs().max()), 1.0)
    else:
        cos = torch.nn.functional.cosine_similarity(dx.float().reshape(-1), xr.grad.reshape(-1), dim=0)
        assert float(cos) >= 0.995


# -------------------------------------------------------------------------------- round 3: the fused training decoder chain
@pytest.mark.parametrize("D,N,p", [(512, 512, 0.1), (512, 1024, 0.1), (256, 256, 0.0), (512, 512, 0.0)])
def test_dec_stage_training_options_match_the_separate_launches(T, D, N, p):
    """made_dec_stage with bf16 raw rows: LayerNorm prologue (+ add), x_out / a_out, ReLU, dropout per element and per group of
    columns -- against made_layernorm_add + made_linear with the same stateless masks (reference music_detr/transformer.py:273-307
    forward_post in train mode).  The dropout pattern is identical; values agree to bf16 rounding (one rounding point differs)."""
    ops, tr = T
    M = 64
    z = _rand(M, D, dtype=torch.bfloat16, seed=1)
    W = (_rand(N, D, dtype=torch.float32, seed=2) / math.sqrt(D)).bfloat16()
    b = _rand(N, dtype=torch.float32, seed=3) * 0.1
    g, be = 1 + 0.1 * _rand(D, dtype=torch.float32, seed=4), 0.1 * _rand(D, dtype=torch.float32, seed=5)
    add = _rand(1, D, dtype=torch.bfloat16, seed=6)
    R = _rand(M, N, dtype=torch.bfloat16, seed=7)
    seed = torch.full((1,), 4242, device="cuda", dtype=torch.int64)
    for col_div, act in ((1, ops.ACT_RELU), (64, ops.ACT_NONE)):
        drop = (seed, 77, p) if p > 0 else None
        kw = dict(drop=drop, drop_ld=(N // col_div if col_div > 1 else N), drop_col_div=col_div)
        x_ref, a_ref = torch.empty(M, D, device="cuda", dtype=torch.bfloat16), torch.empty(M, D, device="cuda", dtype=torch.bfloat16)
        ops.layernorm_add(z, g, be, add.expand(M, D), x_ref, a_ref)
        y_ref = ops.linear(a_ref, W, b, act=act, R=R, out=torch.empty(M, N, device="cuda", dtype=torch.bfloat16), **kw)
        x_out, a_out = torch.full_like(x_ref, float("nan")), torch.full_like(a_ref, float("nan"))
        y = torch.full((M, N), float("nan"), device="cuda", dtype=torch.bfloat16)
        ops.dec_stage(z, W, b, y, ln=(g, be), add=add, x_out=x_out, a_out=a_out, R=R, act=act, **kw)
        torch.cuda.synchronize()
        assert float((x_out.float() - x_ref.float()).abs().max()) <= 2 ** -6        # one bf16 ulp at |x| < 4
        assert float((a_out.float() - a_ref.float()).abs().max()) <= 2 ** -5
        assert float((y.float() - y_ref.float()).abs().max()) <= 0.08, float((y.float() - y_ref.float()).abs().max())
        if p > 0:
            # where the reference output is exactly R (the branch was dropped), so is the fused one: same mask
            dropped_ref, dropped = (y_ref == R), (y == R)
            assert float((dropped_ref != dropped).float().mean()) <= 2e-3 and 0.05 < float(dropped.float().mean()) < 0.6
    # no LayerNorm, no add (layer 0's first stage reads the content query itself)
    y0 = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    ops.dec_stage(z, W, b, y0)
    y0_ref = ops.linear(z, W, b)
    torch.cuda.synchronize()
    assert float((y0.float() - y0_ref.float()).abs().max()) <= 2 ** -6 * max(1.0, float(y0_ref.float().abs().max()))


@pytest.mark.parametrize("B,NQ,L,D,p", [(64, 8, 542, 512, 0.1), (5, 8, 97, 256, 0.0), (3, 24, 300, 512, 0.1)])
def test_wide_attention_key_slices_and_lse(T, B, NQ, L, D, p):
    """made_attention_wide with keys split over workgroups (merged by the second launch, in slice order): the same rows launch after
    launch, close to the unsplit launch; lse_out = the log-sum-exp of the scaled scores."""
    ops, tr = T
    q = _rand(B, NQ, 1, D, dtype=torch.bfloat16, seed=1)
    k, v = _rand(B, L, D, dtype=torch.bfloat16, seed=2), _rand(B, L, D, dtype=torch.bfloat16, seed=3)
    lens = torch.randint(L // 3, L + 1, (B,), device="cuda")
    mask = (torch.arange(L, device="cuda")[None] < lens[:, None]).float()
    scale, ns = 1 / math.sqrt(64), 4
    drop = (123, 5, p) if p > 0 else None
    o0, s0 = torch.empty(B, NQ, 1, D, device="cuda", dtype=torch.bfloat16), torch.empty(B * NQ, device="cuda")
    ops.attention_wide(q, k, v, o0, scale=scale, key_mask=mask, n_split=1, drop=drop, sum_out=s0)
    o1, s1 = torch.empty(B, NQ, 1, D, device="cuda", dtype=torch.bfloat16), torch.empty(B * NQ, device="cuda")
    ops.attention_wide(q, k, v, o1, scale=scale, key_mask=mask, n_split=ns, drop=drop, sum_out=s1)
    for rep in range(3):
        o2, s2, lse = torch.full_like(o1, float("nan")), torch.full_like(s1, float("nan")), torch.empty(B * NQ, device="cuda")
        ops.attention_wide(q, k, v, o2, scale=scale, key_mask=mask, n_split=ns, drop=drop, sum_out=s2, lse_out=lse)
        torch.cuda.synchronize()
        assert torch.equal(o1, o2) and torch.equal(s1, s2), rep
    assert float((o1.float() - o0.float()).abs().max()) <= 2e-2 * max(1.0, float(o0.float().abs().max()))
    assert float((s1 - s0).abs().max()) <= 1e-3
    S = torch.einsum("bqd,bld->bql", q[:, :, 0].float(), k.float()) * scale + torch.where(mask == 0, float("-inf"), 0.0)[:, None]
    ref = torch.logsumexp(S, dim=-1).reshape(-1)
    assert float((lse - ref).abs().max()) <= 2e-3 * max(1.0, float(ref.abs().max()))


@pytest.mark.parametrize("N,K,batch", [(512, 512, 1), (512, 1024, 1), (64, 512, 8), (512, 128, 1)])
def test_linear_16x16_tile_kernel_matches_the_default_one(T, N, K, batch, monkeypatch):
    """at most 64 rows: 16 x 16 tiles (the default) against the 64 x 32-tile kernel (MADE_LINEAR_TILE=32): the same result up to the
    order of the K sums, training epilogue (dropout, residual, pre-activation output) included -- the draws do not depend on the tiling."""
    ops, tr = T
    from mgsv_amd import _lib
    M = 64
    A = _rand(M, K * batch, dtype=torch.bfloat16, seed=1)
    W = _rand(N * batch, K, dtype=torch.bfloat16, seed=2) * 0.1
    bias = _rand(N * batch, dtype=torch.float32, seed=3)
    R = _rand(M, N * batch, dtype=torch.bfloat16, seed=4)
    outs = []
    for tile in ("32", "0"):
        monkeypatch.setenv("MADE_LINEAR_TILE", tile)
        if batch == 1:
            o = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
            z = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
            ops.linear(A, W, bias, out=o, R=R, Zout=z, act=ops.ACT_RELU, drop=(11, 7, 0.2))
            outs.append((o, z))
        else:
            o = torch.empty(M, N * batch, device="cuda", dtype=torch.bfloat16)
            ops.linear(A[:, :K], W[:N], bias, M=M, N=N, K=K, batch=batch, a_z_stride=K, w_z_stride=N * K,
                       segs=[ops.Seg(out=o, ldo=N * batch, out_z_stride=N)])
            outs.append((o,))
    for a_, b_ in zip(outs[0], outs[1]):
        assert float((a_.float() - b_.float()).abs().max()) <= 2e-2 * max(1.0, float(a_.float().abs().max()))
    keep0, keep1 = outs[0][0] == R, outs[1][0] == R                    # (dropped elements leave the residual alone)
    if batch == 1:
        assert float((keep0 != keep1).float().mean()) < 1e-3


def test_linear_per_head_bias_scaled_by_a_row_factor(T):
    """made_linear's bias_row_scale (the value bias of the memory-space cross-attention under dropout: bias[h*hd + j] * s[row, h])
    against made_linear + made_head_bias."""
    ops, tr = T
    M, D, H = 64, 512, 8
    hd = D // H
    pooled = _rand(M, H * D, dtype=torch.bfloat16, seed=1)
    Wv = (_rand(D, D, dtype=torch.float32, seed=2) / math.sqrt(D)).bfloat16()
    bv = _rand(D, dtype=torch.float32, seed=3)
    s = torch.rand(M, H, device="cuda") + 0.5
    ref = torch.empty(M, D, device="cuda", dtype=torch.bfloat16)
    ops.linear(pooled[:, :D], Wv[:hd], None, M=M, N=hd, K=D, batch=H, a_z_stride=D, w_z_stride=hd * D, segs=[ops.Seg(out=ref, ldo=D, out_z_stride=hd)])
    tr.head_bias(ref, s, bv, H)
    got = torch.full_like(ref, float("nan"))
    ops.linear(pooled[:, :D], Wv[:hd], bv, M=M, N=hd, K=D, batch=H, a_z_stride=D, w_z_stride=hd * D, segs=[ops.Seg(out=got, ldo=D, out_z_stride=hd)],
               bias_row_scale=s, bias_z_stride=hd)
    torch.cuda.synchronize()
    exact = torch.einsum("mhk,hjk->mhj", pooled.float().view(M, H, D), Wv.float().view(H, hd, D)) + s[:, :, None] * bv.view(1, H, hd)
    assert float((got.float().view(M, H, hd) - exact).abs().max()) <= 2 ** -7 * float(exact.abs().max())
    assert float((got.float() - ref.float()).abs().max()) <= 2 ** -6 * float(exact.abs().max())


@pytest.mark.parametrize("B,L,D,p,ns", [(64, 542, 512, 0.1, 4), (5, 146, 256, 0.0, 2), (3, 60, 512, 0.1, 1), (7, 333, 256, 0.2, 4)])
def test_wide_attention_backward_in_one_launch(T, B, L, D, p, ns):
    """made_attention_wide_bwd against torch autograd of the same memory-space attention (reference music_detr/transformer.py:293-296
    in train mode: softmax over the valid keys, dropout on the weights, value bias weighted by the dropped weights' sum): Pd, dS, dQ',
    with lse / ssum / O taken from made_attention_wide's forward, the value-bias term reduced from d attc, keys split over workgroups
    (summed by the second launch); repeated launches reproduce the result bit for bit."""
    ops, tr = T
    H = NQ = 8
    hd = D // H
    Lp = (L + 7) // 8 * 8
    scale = 1 / math.sqrt(hd)
    q = (_rand(B, NQ, D, dtype=torch.float32, seed=1) * 0.5).bfloat16()
    k, v = _rand(B, L, D, dtype=torch.bfloat16, seed=2), _rand(B, L, D, dtype=torch.bfloat16, seed=3)
    dO = (_rand(B, NQ, D, dtype=torch.float32, seed=4) * 0.3).bfloat16()
    dattc = (_rand(B, D, dtype=torch.float32, seed=5) * 0.3).bfloat16()
    bv = _rand(D, dtype=torch.float32, seed=6) * 0.2
    lens = torch.randint(max(L // 3, 1), L + 1, (B,), device="cuda")
    lens[0] = L
    mask = (torch.arange(L, device="cuda")[None] < lens[:, None]).float()
    seed, site = 77, 31
    drop = (seed, site, p) if p > 0 else None
    # forward on the device: O, lse, ssum
    O = torch.empty(B, NQ, 1, D, device="cuda", dtype=torch.bfloat16)
    ssum, lse = torch.empty(B * NQ, device="cuda"), torch.empty(B * NQ, device="cuda")
    ops.attention_wide(q.view(B, NQ, 1, D), k, v, O, scale=scale, key_mask=mask, n_split=4, drop=drop, sum_out=ssum, lse_out=lse)
    # reference: autograd through the same forward in f32 on the bf16 operands
    qr = q.float().requires_grad_(True)
    S = torch.einsum("bqd,bld->bql", qr, k.float()) * scale
    P = torch.softmax(S.masked_fill((mask == 0)[:, None, :], float("-inf")), -1)
    keep = _keep(seed, site, p, (B * NQ, L)).float().view(B, NQ, L) if p > 0 else torch.ones(B, NQ, L, device="cuda")
    Pd_ref = P * keep / (1 - p)
    Pd_ref.retain_grad()
    pooled = torch.einsum("bql,bld->bqd", Pd_ref, v.float())
    extra = (dattc.float().view(B, H, hd) * bv.view(1, H, hd)).sum(-1)                   # [B, NQ]: d(sum of dropped weights)
    loss = (pooled * dO.float()).sum() + (Pd_ref.sum(-1) * extra).sum()
    gS, = torch.autograd.grad(loss, S, retain_graph=True)
    gq, = torch.autograd.grad(loss, qr)
    np.testing.assert_allclose(ssum.view(B, NQ).cpu().numpy(), Pd_ref.sum(-1).detach().cpu().numpy(), atol=2e-2)
    Pd = torch.full((B, 2, NQ, Lp), float("nan"), device="cuda", dtype=torch.bfloat16)
    dQ = torch.full((B, NQ, D), float("nan"), device="cuda", dtype=torch.bfloat16)
    part = torch.empty(B * max(ns, 1) * NQ * D, device="cuda")
    outs = []
    for rep in range(3):
        Pd.fill_(float("nan")); dQ.fill_(float("nan"))
        tr.attention_wide_bwd(q, dO, O.view(B, NQ, D), k, v, lse.view(B, NQ), Pd[:, 0], Pd[:, 1], dQ, scale=scale, key_mask=mask,
                              ssum=ssum.view(B, NQ), dattc=dattc, vbias=bv, hd=hd, drop=drop, n_split=ns, part_dq=part)
        torch.cuda.synchronize()
        outs.append((Pd.clone(), dQ.clone()))
    assert all(torch.equal(outs[0][0], o_[0]) and torch.equal(outs[0][1], o_[1]) for o_ in outs[1:])
    got_pd, got_ds = Pd[:, 0, :, :L].float(), Pd[:, 1, :, :L].float()
    assert bool(torch.isfinite(Pd).all()) and float(Pd[:, :, :, L:].abs().max() if Lp > L else 0.0) == 0.0
    assert float((got_pd - Pd_ref.detach()).abs().max()) <= 1e-2
    gS = gS * scale                                               # (the kernel's dS is the gradient of the RAW dot product)
    assert float((got_ds - gS).abs().max()) <= 2e-2 * max(float(gS.abs().max()), 1e-3) + 5e-4
    assert float((dQ.float() - gq).abs().max()) <= 3e-2 * float(gq.abs().max()) + 1e-3
    masked = (mask == 0)[:, None, :].expand(B, NQ, L)
    assert float(got_pd[masked].abs().max() if masked.any() else 0.0) == 0.0 and float(got_ds[masked].abs().max() if masked.any() else 0.0) == 0.0
    # the same with the value-bias term handed in
    dQ2 = torch.empty_like(dQ)
    tr.attention_wide_bwd(q, dO, O.view(B, NQ, D), k, v, lse.view(B, NQ), Pd[:, 0], Pd[:, 1], dQ2, scale=scale, key_mask=mask,
                          ssum=ssum.view(B, NQ), extra=extra.contiguous(), drop=drop, n_split=ns, part_dq=part)
    torch.cuda.synchronize()
    assert float((dQ2.float() - dQ.float()).abs().max()) <= 2e-2 * float(gq.abs().max()) + 1e-3


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-5), (torch.bfloat16, 2e-2)])
@pytest.mark.parametrize("rows,D", [(64, 512), (37, 256)])
def test_two_chained_layernorms_backward(T, dtype, tol, rows, D):
    """made_layernorm_bwd2 (norm 3 of a decoder layer followed by the shared output norm) against torch autograd; the dropped copy
    carries the stateless mask."""
    ops, tr = T
    xa = _rand(rows, D, dtype=dtype, seed=1)
    ga, ba = 1 + 0.1 * _rand(D, dtype=torch.float32, seed=2), 0.1 * _rand(D, dtype=torch.float32, seed=3)
    gb = 1 + 0.1 * _rand(D, dtype=torch.float32, seed=4)
    dy, add = _rand(rows, D, dtype=dtype, seed=5), _rand(rows, D, dtype=dtype, seed=6)
    xr = xa.float().requires_grad_(True)
    t3 = torch.nn.functional.layer_norm(xr, (D,), ga, ba, 1e-5)
    xb = t3.detach().to(dtype)                                    # what the forward saved
    t3b = xb.float().requires_grad_(True)
    hs = torch.nn.functional.layer_norm(t3b, (D,), gb, torch.zeros(D, device="cuda"), 1e-5)
    gbr = gb.clone().requires_grad_(True)
    hs2 = torch.nn.functional.layer_norm(t3b, (D,), gbr, torch.zeros(D, device="cuda", requires_grad=True), 1e-5)
    (hs2 * dy.float()).sum().backward()
    g_mid = t3b.grad + add.float()
    gar = ga.clone().requires_grad_(True); bar = ba.clone().requires_grad_(True)
    t3r = torch.nn.functional.layer_norm(xr, (D,), gar, bar, 1e-5)
    (t3r * g_mid).sum().backward()
    dx, dxd = torch.empty(rows, D, device="cuda", dtype=dtype), torch.empty(rows, D, device="cuda", dtype=dtype)
    dga, dba, dgb, dbb = (torch.ones(D, device="cuda") for _ in range(4))
    p, seed, site = 0.1, 3, 4
    tr.layernorm_bwd2(xa, ga, xb, gb, dy, dx, dgamma_a=dga, dbeta_a=dba, dgamma_b=dgb, dbeta_b=dbb, add=add, dx_drop=dxd, drop=(seed, site, p))
    torch.cuda.synchronize()
    sc = float(xr.grad.abs().max())
    assert float((dx.float() - xr.grad).abs().max()) <= tol * sc
    keep = _keep(seed, site, p, (rows, D)).float()
    assert float((dxd.float() - dx.float() * keep / (1 - p)).abs().max()) <= tol * sc
    for got, ref in ((dga, gar.grad), (dba, bar.grad), (dgb, gbr.grad), (dbb, dy.float().sum(0))):
        assert float((got - 1 - ref).abs().max()) <= 5 * tol * max(float(ref.abs().max()), 1.0)


@pytest.mark.parametrize("D,N,mode,gate,col_div", [(512, 512, 0, False, 1), (512, 512, 0, False, 64), (512, 1024, 2, True, 1), (512, 1024, 1, True, 1),
                                                    (256, 1024, 2, True, 1), (256, 256, 0, False, 32), (256, 1024, 1, True, 1)])
def test_dec_stage_bwd_matches_the_separate_launches(T, D, N, mode, gate, col_div):
    """made_dec_stage_bwd (a LayerNorm backward -- or norm 3 + the output norm -- in the prologue of the dX product that consumes it)
    against made_layernorm_bwd / made_layernorm_bwd2 + made_linear with the same stateless masks: same dropout pattern, values to
    bf16 rounding, parameter gradients to f32 summation order."""
    ops, tr = T
    from mgsv_amd import _lib
    M, p = 64, 0.1
    two = mode == 2                                               # mode 0: one norm; 1: one norm, dy + add; 2: two stacked norms + add
    bf = torch.bfloat16
    xa, dy = _rand(M, D, dtype=bf, seed=1), (_rand(M, D, dtype=torch.float32, seed=2) * 0.3).to(bf)
    xb, add = _rand(M, D, dtype=bf, seed=3), (_rand(M, D, dtype=torch.float32, seed=4) * 0.3).to(bf)
    ga, gb = 1 + 0.1 * _rand(D, dtype=torch.float32, seed=5), 1 + 0.1 * _rand(D, dtype=torch.float32, seed=6)
    W = (_rand(N, D, dtype=torch.float32, seed=7) / math.sqrt(D)).to(bf)
    Gt = torch.relu(_rand(M, N, dtype=torch.float32, seed=8)).to(bf)
    R = _rand(M, N, dtype=bf, seed=9)
    seed = torch.full((1,), 99, device="cuda", dtype=torch.int64)
    drop_a, drop_o = (seed, 11, p), ((seed, 12, p) if col_div > 1 else None)
    E = lambda *s_: torch.full(s_, float("nan"), device="cuda", dtype=bf)
    Z = lambda: torch.zeros(D, device="cuda")
    # separate launches
    dx0, ad0, out0 = E(M, D), E(M, D), E(M, N)
    pg0 = [Z(), Z(), Z(), Z()]
    if two:
        tr.layernorm_bwd2(xa, ga, xb, gb, dy, dx0, dgamma_a=pg0[0], dbeta_a=pg0[1], dgamma_b=pg0[2], dbeta_b=pg0[3], add=add, dx_drop=ad0, drop=drop_a)
    else:
        dy_in = dy if mode == 0 else (dy.float() + add.float()).to(bf)      # (the fused form adds in f32: one bf16 rounding apart)
        tr.layernorm_bwd(xa, ga, dy_in, dx0, dgamma=pg0[0], dbeta=pg0[1], dx_drop=ad0, drop=drop_a)
    ops.linear(ad0, W, None, out=out0, R=R, **(dict(gate=_lib.GATE_RELU_OUT, G=Gt, gate_scale=1.25) if gate else {}),
               **(dict(drop=drop_o, drop_ld=N // col_div, drop_col_div=col_div) if drop_o else {}))
    # fused
    dx1, ad1, out1 = E(M, D), E(M, D), E(M, N)
    pg1 = [Z(), Z(), Z(), Z()]
    tr.dec_stage_bwd(xa, ga, dy, W, out1, dgamma_a=pg1[0], dbeta_a=pg1[1], dx_out=dx1, a_out=ad1, drop_a=drop_a, R=R,
                     **(dict(xb=xb, gamma_b=gb, dgamma_b=pg1[2], dbeta_b=pg1[3], add=add) if two else (dict(add=add) if mode == 1 else {})),
                     **(dict(G=Gt, gate_scale=1.25) if gate else {}),
                     **(dict(drop_o=drop_o, drop_o_ld=N // col_div, drop_o_col_div=col_div) if drop_o else {}))
    torch.cuda.synchronize()
    sc = float(dx0.float().abs().max())
    rt = 2 ** -7 if mode != 1 else 2 ** -5                        # (mode 1: the reference rounded dy + add to bf16 first)
    assert float((dx1.float() - dx0.float()).abs().max()) <= rt * sc
    assert float((ad1.float() - ad0.float()).abs().max()) <= rt * sc / (1 - p)
    assert float(((ad1 == 0) != (ad0 == 0)).float().mean()) <= 1e-3                       # the same mask
    so = float(out0.float().abs().max())
    assert float((out1.float() - out0.float()).abs().max()) <= 0.03 * so, float((out1.float() - out0.float()).abs().max()) / so
    for a_, b_ in zip(pg0, pg1):
        assert float((a_ - b_).abs().max()) <= (1e-3 if mode != 1 else 2e-2) * max(float(a_.abs().max()), 1e-3) + 1e-5


def test_launch_tape_drops_duplicate_cross_stream_dependencies(T):
    """made_tape_end keeps ONE record + wait where a program asks for the same cross-stream dependency several times in a row (every pair costs the
    waiting stream >= 6 us on this stack), a stream's later record is a dependency of its own, and the replayed result is the eager one."""
    ops, tr = T
    from mgsv_amd import tape as _tape
    x = torch.arange(4096, device="cuda", dtype=torch.float32)
    y, z = torch.zeros_like(x), torch.zeros_like(x)
    cur, side = torch.cuda.current_stream(), torch.cuda.Stream()

    def program():
        side.wait_stream(cur); side.wait_stream(cur); side.wait_stream(cur)        # three times the same dependency
        with torch.cuda.stream(side):
            tr.add3(y, x, x)                                                       # y = 2 x
        tr.add3(z, x)                                                              # z = x (main stream, beside it)
        cur.wait_stream(side); cur.wait_stream(side)                               # twice the same join
        tr.add3(z, z, y)                                                           # z = 3 x
        side.wait_stream(cur)                                                      # a NEW point of the main stream: kept
        with torch.cuda.stream(side):
            tr.add3(y, z, x)                                                       # y = 4 x
        cur.wait_stream(side)
    program(); torch.cuda.synchronize()
    with _tape.LaunchTape.record() as tp:
        program()
    torch.cuda.synchronize()
    k, w, o = tp.counts()
    assert k == 4 and w == 0 and o == 8, (k, w, o)               # 4 record + wait pairs of the 7 asked for
    y.zero_(); z.zero_(); torch.cuda.synchronize()
    tp.replay(); torch.cuda.synchronize()
    assert torch.equal(z, 3 * x) and torch.equal(y, 4 * x)
    tp.close()

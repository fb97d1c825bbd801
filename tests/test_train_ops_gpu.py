"""GPU: the backward kernels of the training path against plain PyTorch fp32 references of the same op
(tolerances are stated per test; bf16 inputs are compared after the same rounding of the inputs)."""
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

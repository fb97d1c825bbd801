"""Nothing that runs BESIDE a kernel may change its results.

Round 3 left one deviation open: one element of one 16 x 16 product tile of a `made_dec_stage_bwd` launch of the decoder's backward chain
came out one bf16 ulp off in 15-20 % of the first captured steps whenever two launches of the retrieval branch ran on the second stream
beside it (reference work: music_detr/transformer.py:273-307, backward).  Round 4 found the cause with the probe these tests are made of
(tools/dec_corun_probe.py, profiles/r04_*_dec_corun_probe*.txt): packed-FP32 instructions (v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32), which
hipcc's SLP vectoriser puts into the kernel's LayerNorm backward, give results that depend on which other kernel shares the CU -- 55 % of the
launches beside a register-staged Linear, 0 of 2000 in the same source built without them, whose solo results are the packed build's solo
results bit for bit.  The library is built with -fno-slp-vectorize; these tests are the guard."""
import math
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mgsv_amd import ops, ops_train as tr, synth  # noqa: E402
from mgsv_amd.config import cfg_headline  # noqa: E402

pytestmark = pytest.mark.gpu
bf = torch.bfloat16


def _rand(*shape, seed, dtype=torch.float32, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(dtype).cuda()


def test_decoder_backward_stage_beside_streaming_launches_is_bit_identical_to_its_solo_run():
    """One stage of the backward chain (norm 3 -> FFN-2 dX with the ReLU gate: 64 x 1024 x 512, g = dy + add, dropout) launched 600 times per
    co-runner while a second stream runs it without pause: every output of every launch equals the solo run's, bit for bit.  The
    co-runners: the register-staged Linear that moved 55 % of the launches of the packed-FP32 build, the single-stage LDS-DMA Linear,
    the retrieval branch's batched score product (the launches round 3's bisection named)."""
    M, D, N = 64, 512, 1024
    xa, dy, add = _rand(M, D, seed=1, dtype=bf), _rand(M, D, seed=2, scale=0.3, dtype=bf), _rand(M, D, seed=4, scale=0.3, dtype=bf)
    ga = 1 + 0.1 * _rand(D, seed=5)
    W = _rand(N, D, seed=7, scale=1 / math.sqrt(D), dtype=bf)
    Gt, R = torch.relu(_rand(M, N, seed=8)).to(bf), _rand(M, N, seed=9, dtype=bf)
    seed = torch.full((1,), 99, device="cuda", dtype=torch.int64)
    Z = lambda: torch.zeros(D, device="cuda")

    def stage(out, dx, ad):
        tr.dec_stage_bwd(xa, ga, dy, W, out, dgamma_a=Z(), dbeta_a=Z(), dx_out=dx, a_out=ad, drop_a=(seed, 11, 0.1), R=R, add=add, G=Gt, gate_scale=1.25)

    ref = [torch.empty(M, N, device="cuda", dtype=bf), torch.empty(M, D, device="cuda", dtype=bf), torch.empty(M, D, device="cuda", dtype=bf)]
    stage(*ref)
    torch.cuda.synchronize()
    As, Ws, Os = _rand(32768, 512, seed=23, dtype=bf), _rand(512, 512, seed=24, scale=0.05, dtype=bf), torch.empty(32768, 512, device="cuda", dtype=bf)
    ones = torch.ones(32768, device="cuda")
    Ab, Wb, Ob = _rand(64, 64, 512, seed=21, dtype=bf), _rand(64, 512, 512, seed=22, scale=0.05, dtype=bf), torch.empty(64, 64, 512, device="cuda", dtype=bf)
    corun = {
        "register-staged Linear": lambda: ops.linear(As, Ws, None, out=Os, a_row_mask=ones),
        "LDS-DMA Linear": lambda: ops.linear(As, Ws, None, out=Os),
        "batched score product": lambda: ops.linear(Ab.view(-1, 512), Wb.view(-1, 512), None, out=Ob, batch=64, a_z_stride=64 * 512, w_z_stride=512 * 512,
                                                    M=64, N=512, K=512),
    }
    ring = [[torch.empty_like(r) for r in ref] for _ in range(8)]
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    for name, co in corun.items():
        bad = torch.zeros(3, device="cuda", dtype=torch.int64)
        torch.cuda.synchronize()
        for i in range(600):
            with torch.cuda.stream(s2):
                co()
                if i % 4 == 0:
                    co()
            with torch.cuda.stream(s1):
                o = ring[i % 8]
                stage(*o)
                for j in range(3):
                    bad[j] += (o[j].view(torch.int16) != ref[j].view(torch.int16)).any().long()
        torch.cuda.synchronize()
        assert bad.tolist() == [0, 0, 0], (name, bad.tolist())


def test_backward_chain_of_the_recorded_step_is_bit_reproducible_with_the_retrieval_branch_beside_it(monkeypatch):
    """The recorded training step (launch tape) with round 3's mitigation switched OFF (MADE_RET_SPLIT=0: the retrieval branch's two batched
    products are issued beside the decoder's backward chain again), replayed 500 times with one seed and zero learning rates: the chain's
    hand-off rows and every per-layer gradient stack equal the first replay's, bit for bit."""
    from mgsv_amd.trainer import MadeTrainer
    monkeypatch.setenv("MADE_RET_SPLIT", "0")
    cfg = cfg_headline()
    B, Tv, Ta = 64, cfg.max_v_frames, cfg.max_snippet_num
    dev = torch.device("cuda", 0)
    sd = synth.make_state_dict(cfg, seed=0)
    inp = synth.make_inputs(cfg, B, Tv, Ta, seed=1)
    trn = MadeTrainer(cfg, sd, device=dev, dtype="bf16")
    t = {k: torch.from_numpy(v).to(dev) for k, v in inp.items() if isinstance(v, np.ndarray)}
    g = trn.capture_train_step(t["frame_feats"], t["segment_feats"], t["frame_masks"], t["segment_masks"], t["spans_target"], max_grad_norm=1.0, mode="tape")
    tb = tuple(g.inputs[k] for k in ("frame_feats", "segment_feats", "frame_masks", "segment_masks", "spans_target"))
    tw = trn._train_buffers(B, Tv, Ta)
    watch = {"dchain": tw["dchain"]}
    watch.update({k: v for k, v in tw["dstack"].items() if k.startswith("g_") or k == "dt1q"})
    g.step(*tb, seed=7, lrs=(0.0, 0.0, 0.0))
    torch.cuda.synchronize()
    ref = {k: v.clone() for k, v in watch.items()}
    differing = 0
    for _ in range(500):
        g.step(*tb, seed=7, lrs=(0.0, 0.0, 0.0))
        torch.cuda.synchronize()
        differing += int(any(not torch.equal(v.view(torch.int16) if v.dtype == bf else v, ref[k].view(torch.int16) if v.dtype == bf else ref[k])
                             for k, v in watch.items()))
    g.close()
    assert differing == 0, f"{differing} of 500 replays differ from the first"


@pytest.mark.gpu
@pytest.mark.parametrize("which", ["xpool_inbatch", "xpool_attention", "xpool_fused", "xpool_sims", "xpool_sims_pq64"])
def test_retrieval_kernels_bit_identical_beside_small_workgroups(which, monkeypatch):
    """The retrieval kernels hand LDS reads to inline assembly (transposing reads, counted waits).  A register that such a read has been given is an
    ordinary value to the compiler: if it copies it before the data has arrived the kernel is right alone on the chip and wrong beside another
    kernel's workgroups on the same CU (round 4: the first made_xpool_inbatch, 10-30 % of the launches).  Each kernel 60 times beside a stream of
    64-row Linears, bit-identical to its solo run."""
    import math
    from mgsv_amd import ops
    dev, dt = torch.device("cuda"), torch.bfloat16
    g = torch.Generator(device=dev).manual_seed(0)
    if which.startswith("xpool_inbatch"):
        Nv, Nm, S, D = 64, 64, 512, 512
    elif which == "xpool_attention":
        Nv, Nm, S, D = 1024, 16, 512, 512
    else:
        Nv, Nm, S, D = 2048, 64, 96, 256
    q = torch.randn(Nv, D, device=dev, generator=g).to(dt)
    k, u = torch.randn(Nm, S, D, device=dev, generator=g).to(dt), torch.randn(Nm, S, D, device=dev, generator=g).to(dt)
    lens = torch.randint(12, S + 1, (Nm,), device=dev, generator=g)
    mask = (torch.arange(S, device=dev)[None] < lens[:, None]).float()
    if which == "xpool_fused":
        W = (torch.randn(D, D, device=dev, generator=g) / math.sqrt(D)).to(dt)
        vec = lambda: torch.randn(D, device=dev, generator=g) * 0.1
        ln2, ln3, bl = (1 + vec(), vec()), (1 + vec(), vec()), vec()
        vn = torch.nn.functional.normalize(torch.randn(Nv, D, device=dev, generator=g), dim=-1)
        shape, odt = (Nv, Nm), torch.float32
        run = lambda o: ops.xpool_fused(q, k, u, mask, ln2, W, bl, ln3, vn, o, scale=1 / math.sqrt(D))
    elif which.startswith("xpool_sims"):
        monkeypatch.setenv("MADE_XPOOL_SIMS_PQ", "64" if which.endswith("pq64") else "32")       # round 5's 64-video kernel / the default
        vec = lambda: torch.randn(D, device=dev, generator=g) * 0.1
        ln3, av, bv = (1 + vec(), vec()), vec(), vec()
        vn = torch.nn.functional.normalize(torch.randn(Nv, D, device=dev, generator=g), dim=-1)
        uu = torch.cat([u, torch.randn(Nm, S, D, device=dev, generator=g).to(dt)], -1)
        shape, odt = (Nv, Nm), torch.float32
        run = lambda o: ops.xpool_sims(q, k, uu, mask, av, bv, ln3, vn, o, scale=1 / math.sqrt(D))
    elif which == "xpool_attention":
        shape, odt = (Nm, Nv, D), dt
        run = lambda o: ops.xpool_attention(q, k, u, mask, o, scale=1 / math.sqrt(D))
    else:
        shape, odt = (Nm, Nv, D), dt
        ws = torch.zeros(ops.xpool_inbatch_ws_bytes(Nm, S), device=dev, dtype=torch.uint8)
        run = lambda o: ops.xpool_inbatch(q, k, u, mask, o, scale=1 / math.sqrt(D), ws=ws)
    solo = torch.empty(shape, device=dev, dtype=odt)
    run(solo); torch.cuda.synchronize()
    x1, W1 = torch.randn(64, 512, device=dev, generator=g).to(dt), torch.randn(512, 512, device=dev, generator=g).to(dt)
    side = torch.cuda.Stream()
    outs = [torch.empty(shape, device=dev, dtype=odt) for _ in range(60)]
    for o in outs:
        with torch.cuda.stream(side):
            for _ in range(4):
                ops.linear(x1, W1)
        run(o)
    torch.cuda.synchronize()
    differing = sum(0 if torch.equal(o.view(torch.int32 if odt == torch.float32 else torch.int16), solo.view(torch.int32 if odt == torch.float32 else torch.int16)) else 1 for o in outs)
    assert differing == 0, f"{which}: {differing} of 60 launches differ from the solo run"

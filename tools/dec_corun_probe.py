"""Does anything running BESIDE the decoder's backward chain change its results?  (one element of one 16 x 16 product tile of a
`dec_stage_bwd_kernel` launch came out one bf16 ulp off in 15-20 % of the first captured steps when two batched products of the retrieval
branch ran on the second stream beside the chain.)

  micro : one `made_dec_stage_bwd` launch (the norm-3 stage of a decoder layer: 64 x 1024 x 512, g = dy + add, ReLU gate, dropout) repeated
          N times on one stream while a second stream runs a co-runner without pause; every output is compared with the solo run's, bit
          for bit.  Co-runners: the retrieval branch's batched score product (64 problems of 64 x 512 x 512), a streaming Linear over
          33 MB of rows, a plain streaming add over 66 MB.
  step  : the recorded training step (launch tape) replayed N times with the same seed and zero learning rates, MADE_RET_SPLIT as given in
          the environment; the backward chain's hand-off rows (`dchain`) and per-layer gradient stacks are compared with the first
          replay's, bit for bit.

usage: python tools/dec_corun_probe.py micro|step [N]"""
import math, os, sys
os.environ.setdefault("MADE_DEBUG_VARIANTS", "1")          # (measurement knobs are honoured only under this switch)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mgsv_amd import ops, ops_train as tr, _lib

dev = torch.device("cuda", 0)
bf = torch.bfloat16


def _rand(*shape, seed, dtype=torch.float32, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(dtype).to(dev)


def stage_inputs(M=64, D=512, N=1024):
    xa, dy = _rand(M, D, seed=1, dtype=bf), _rand(M, D, seed=2, scale=0.3, dtype=bf)
    add = _rand(M, D, seed=4, scale=0.3, dtype=bf)
    ga = 1 + 0.1 * _rand(D, seed=5)
    W = _rand(N, D, seed=7, scale=1 / math.sqrt(D), dtype=bf)
    Gt = torch.relu(_rand(M, N, seed=8)).to(bf)
    R = _rand(M, N, seed=9, dtype=bf)
    seed = torch.full((1,), 99, device=dev, dtype=torch.int64)
    return dict(xa=xa, dy=dy, add=add, ga=ga, W=W, Gt=Gt, R=R, drop_a=(seed, 11, 0.1))


def run_stage(I, out, dx, ad, pg):
    tr.dec_stage_bwd(I["xa"], I["ga"], I["dy"], I["W"], out, dgamma_a=pg[0], dbeta_a=pg[1], dx_out=dx, a_out=ad, drop_a=I["drop_a"], R=I["R"],
                     add=I["add"], G=I["Gt"], gate_scale=1.25)


def corunners():
    # (a) the retrieval branch's batched score product: 64 problems of 64 x 512 x 512 (A [64, 64, 512], W [64, 512, 512])
    Ab, Wb = _rand(64, 64, 512, seed=21, dtype=bf), _rand(64, 512, 512, seed=22, scale=0.05, dtype=bf)
    Ob = torch.empty(64, 64, 512, device=dev, dtype=bf)
    # (b) a streaming Linear: 32768 x 512 x 512, through each of made_linear's large-problem kernels
    As, Ws, Os = _rand(32768, 512, seed=23, dtype=bf), _rand(512, 512, seed=24, scale=0.05, dtype=bf), torch.empty(32768, 512, device=dev, dtype=bf)
    ones = torch.ones(32768, device=dev)
    # (c) a streaming add over 2 x 33 MB
    x1, x2, xo = _rand(1 << 24, seed=25, dtype=bf), _rand(1 << 24, seed=26, dtype=bf), torch.empty(1 << 24, device=dev, dtype=bf)
    # (d) flash attention forward, a LayerNorm over 32768 rows
    qkv = _rand(16, 512, 1536, seed=27, dtype=bf); ao = torch.empty(16, 512, 512, device=dev, dtype=bf)
    lg, lb, lo = torch.ones(512, device=dev), torch.zeros(512, device=dev), torch.empty(32768, 512, device=dev, dtype=bf)

    def lin(tile):
        def f():
            os.environ["MADE_LINEAR_TILE"] = str(tile)
            ops.linear(As, Ws, None, out=Os)
            os.environ.pop("MADE_LINEAR_TILE")
        return f
    return {
        "batched 64 x (64 x 512 x 512)": lambda: ops.linear(Ab.view(-1, 512), Wb.view(-1, 512), None, out=Ob, batch=64, a_z_stride=64 * 512,
                                                            w_z_stride=512 * 512, M=64, N=512, K=512),
        "Linear 32768x512x512, LDS-DMA 64-row": lin(64),
        "Linear 32768x512x512, LDS-DMA 128-row": lin(128),
        "Linear 32768x512x512, register-staged": lambda: ops.linear(As, Ws, None, out=Os, a_row_mask=ones),
        "torch.matmul 32768x512x512": lambda: torch.matmul(As, Ws.t(), out=Os),
        "streaming add 2 x 33 MB": lambda: torch.add(x1, x2, out=xo),
        "flash attention 16 x 8 x 512 x 64": lambda: ops.attention(qkv[:, :, :512], qkv[:, :, 512:1024], qkv[:, :, 1024:], ao, 8),
        "LayerNorm 32768 x 512": lambda: ops.layernorm(As, lg, lb, out=lo),
    }


def micro(N):
    I = stage_inputs()
    M, D, N_ = 64, 512, 1024
    Z = lambda: torch.zeros(D, device=dev)
    ref = [torch.empty(M, N_, device=dev, dtype=bf), torch.empty(M, D, device=dev, dtype=bf), torch.empty(M, D, device=dev, dtype=bf)]
    run_stage(I, *ref, [Z(), Z()])
    torch.cuda.synchronize()
    import zlib
    crc = [zlib.crc32(r.view(torch.int16).cpu().numpy().tobytes()) for r in ref]
    print(f"micro: solo run checksums (out, dx_out, a_out): {crc[0]:08x} {crc[1]:08x} {crc[2]:08x}")
    ring = [[torch.empty_like(r) for r in ref] for _ in range(8)]
    pg = [Z(), Z()]
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    total = 0
    only = os.environ.get("CORUN")                             # substring of the co-runner's name
    print(f"micro: lib={os.path.basename(_lib.LIB_PATH)}")
    for name, co in [("nothing", None)] + [(k, v) for k, v in corunners().items() if only is None or only in k]:
        bad = torch.zeros(3, device=dev, dtype=torch.int64)
        torch.cuda.synchronize()
        for i in range(N):
            if co is not None:
                with torch.cuda.stream(s2):
                    co()
                    if i % 4 == 0: co()
            with torch.cuda.stream(s1):
                o = ring[i % 8]
                run_stage(I, *o, pg)
                for j in range(3):
                    bad[j] += (o[j].view(torch.int16) != ref[j].view(torch.int16)).any().long()
        torch.cuda.synchronize()
        b = bad.tolist()
        total += sum(b)
        print(f"micro, beside {name:38s}: {N} launches, differing from the solo run: out {b[0]}, dx_out {b[1]}, a_out {b[2]}")
    return total


def step(N):
    from mgsv_amd import synth
    from mgsv_amd.config import cfg_headline
    from mgsv_amd.trainer import MadeTrainer
    cfg = cfg_headline()
    B, Tv, Ta = 64, cfg.max_v_frames, cfg.max_snippet_num
    sd = synth.make_state_dict(cfg, seed=0)
    inp = synth.make_inputs(cfg, B, Tv, Ta, seed=1)
    trn = MadeTrainer(cfg, sd, device=dev, dtype="bf16")
    t = {k: torch.from_numpy(v).to(dev) for k, v in inp.items() if isinstance(v, np.ndarray)}
    batch = (t["frame_feats"], t["segment_feats"], t["frame_masks"], t["segment_masks"], t["spans_target"])
    g = trn.capture_train_step(*batch, max_grad_norm=1.0, mode=os.environ.get("MODE", "tape"))
    tb = tuple(g.inputs[k] for k in ("frame_feats", "segment_feats", "frame_masks", "segment_masks", "spans_target"))
    tw = trn._train_buffers(B, Tv, Ta)
    watch = {"dchain": tw["dchain"]}
    for k, v in tw["dstack"].items():
        if k.startswith("g_") or k == "dt1q":
            watch[k] = v
    for k in ("dgN", "dhs"):
        if k in tw: watch[k] = tw[k]
    g.step(*tb, seed=7, lrs=(0.0, 0.0, 0.0))
    torch.cuda.synchronize()
    ref = {k: v.clone() for k, v in watch.items()}
    bad = {k: 0 for k in watch}
    steps_bad = 0
    first = None
    for i in range(N):
        g.step(*tb, seed=7, lrs=(0.0, 0.0, 0.0))
        torch.cuda.synchronize()
        any_bad = False
        for k, v in watch.items():
            if not torch.equal(v.view(torch.int16) if v.dtype == bf else v, ref[k].view(torch.int16) if v.dtype == bf else ref[k]):
                bad[k] += 1; any_bad = True
                if first is None and k == "dchain":
                    d = (v.float() - ref[k].float()).abs()
                    idx = torch.nonzero(d > 0)
                    first = (i, idx[:6].tolist(), float(d.max()))
        steps_bad += any_bad
    print(f"step ({g.mode}), MADE_RET_SPLIT={os.environ.get('MADE_RET_SPLIT', '1')}, lib={os.path.basename(_lib.LIB_PATH)}: {steps_bad} of {N} replays differ from the first; "
          f"per buffer: { {k: v for k, v in bad.items() if v} }" + (f"; first: replay {first[0]}, dchain [layer, slot, row, col] {first[1]}, max |diff| {first[2]:.3g}" if first else ""))
    return steps_bad


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "micro"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 500
    bad = micro(n) if what == "micro" else step(n)
    sys.exit(1 if bad and os.environ.get("STRICT", "0") == "1" else 0)

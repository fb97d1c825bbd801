#!/usr/bin/env python3
"""bench.py -- throughput of the MaDe hot path on MI355X (contract: see the task statement / DESIGN.md).

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is one eval-mode `Uni_model.forward` (reference model/model_Uni.py:177-322: both temporal
encoders, X-Pool similarities, DETR encoder/decoder, heads, retrieval loss, Hungarian matcher and set
criterion) over one batch of B=64 synthetic video-music pairs at BASELINE.json configs[1]
(T_v=30, T_a=512, D=512), inputs resident in HBM.  N>1: every rank runs its own batch (pairs are
independent; no data-path collective), value = all pairs / max-over-ranks time ("weak").
Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from mgsv_amd import ops, synth  # noqa: E402
from mgsv_amd.config import cfg_headline  # noqa: E402
from mgsv_amd.engine import MadeEngine  # noqa: E402

PEAK_TFLOPS = {"bf16": 2500.0, "f32": 157.3}        # dense MFMA peaks, /opt/skills/guides/MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=20)
    p.add_argument("--warmup", type=int, default=5)
    p.add_argument("--dtype", choices=["bf16", "f32"], default="bf16")
    p.add_argument("--batch", type=int, default=64)
    p.add_argument("--launch", choices=["graph", "eager"], default="graph")
    p.add_argument("--workload", choices=["forward", "retrieval", "train"], default="forward")
    p.add_argument("--nv", type=int, default=53000)
    p.add_argument("--nm", type=int, default=4000)
    p.add_argument("--seg", type=int, default=96)
    p.add_argument("--ta", type=int, default=0, help="train workload: number of music segments (default: configs[1]'s 512)")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--cpu-steps", type=int, default=2)
    p.add_argument("--in-flight", type=int, default=2,
                   help="forward workload: independent batches in flight (one engine, workspace, stream and graph each); steps "
                        "are issued round-robin over them")
    return p.parse_args()


def cpu_baseline(cfg, sd, inp, steps: int):
    """The oracle (CPU restatement validated against the reference) on this box's host cores."""
    from oracle import made_oracle as O
    P = O.to_torch_params(sd)
    n = torch.get_num_threads()

    def one():
        with torch.no_grad():
            O.forward(P, cfg, inp["frame_feats"], inp["segment_feats"], inp["frame_masks"], inp["segment_masks"],
                      inp["spans_target"], v_duration=inp["v_duration"])

    one()                                            # warm-up (allocator, thread pool)
    t0 = time.perf_counter()
    for _ in range(steps):
        one()
    dt = (time.perf_counter() - t0) / steps
    B = inp["frame_feats"].shape[0]
    return dict(value=B / dt, unit="pairs/s", cores=n, kind="port",
                sample=f"{steps} eval forwards of the same B={B} batch (oracle/made_oracle.py, torch CPU f32, {n} threads), {dt:.2f} s each")


def cpu_baseline_retrieval(cfg, sd, S: int, n_v: int = 2048, n_m: int = 256):
    """The oracle's all-pairs retrieval scoring (reference test-MaDe.py:386-403, video chunks of 512) on a bounded sample."""
    from oracle import made_oracle as O
    P = O.to_torch_params(sd)
    ri = synth.make_retrieval_inputs(n_v, n_m, S, cfg.D, seed=3)
    n = torch.get_num_threads()
    with torch.no_grad():
        O.retrieval_sim_matrix(P, cfg, ri["video_embeds"][:512], ri["segment_embeds"], ri["segment_masks"], ri["music_embeds"])   # warm-up
        t0 = time.perf_counter()
        O.retrieval_sim_matrix(P, cfg, ri["video_embeds"], ri["segment_embeds"], ri["segment_masks"], ri["music_embeds"])
        dt = time.perf_counter() - t0
    byts = 4.0 * (n_m * S * cfg.D + n_m * S + n_v * cfg.D + n_m * cfg.D + n_v * n_m)
    return dict(value=round(byts / dt / 1e9, 4), unit="GB/s", cores=n, kind="port", pairs_per_s=round(n_v * n_m / dt, 1),
                sample=f"one pass over {n_v} videos x {n_m} tracks, S={S}, D={cfg.D} (oracle/made_oracle.py, torch CPU f32, {n} threads), {dt:.2f} s")


def cpu_baseline_train(cfg, sd, inp, B: int = 16):
    """The oracle's train-mode forward (dropout on) + autograd backward on a bounded sample: the first B samples of the batch."""
    from oracle import made_oracle as O
    P = O.to_torch_params(sd)
    names = [k for k, v in P.items() if v.is_floating_point() and not k.endswith(".pe") and k != "criterion.empty_weight"]
    for k in names:
        P[k].requires_grad_(True)
    sub = {k: (v[:B] if isinstance(v, np.ndarray) else v) for k, v in inp.items()}
    n = torch.get_num_threads()

    def one(seed):
        r = O.forward(P, cfg, sub["frame_feats"], sub["segment_feats"], sub["frame_masks"], sub["segment_masks"], sub["spans_target"],
                      v_duration=sub["v_duration"], drop=O.Drop(seed, p_detr=cfg.detr_dropout))
        (r["retrieval_loss"] + r["localization_loss"]).backward()
        for k in names:
            P[k].grad = None

    one(1)
    t0 = time.perf_counter()
    one(2)
    dt = time.perf_counter() - t0
    return dict(value=round(B / dt, 3), unit="pairs/s", cores=n, kind="port",
                sample=f"one train-mode forward + backward of the first {B} samples (oracle/made_oracle.py autograd, torch CPU f32, {n} threads; "
                       f"no optimizer step), {dt:.2f} s")


def retrieval_main(args, rank, world, local, dist):
    """BASELINE.json configs[3]: all-pairs video x music similarity, videos row-sharded over the ranks, music side
    all-gathered once per pass (RCCL).  A step = one full pass (exchange + scoring).  Strong scaling: the problem is fixed."""
    from mgsv_amd.config import cfg_native
    from mgsv_amd.retrieval import ShardedRetrieval, shard_rows
    cfg = cfg_native()
    dev = torch.device("cuda", local)
    eng = MadeEngine(cfg, synth.make_state_dict(cfg, seed=0), device=dev, dtype=args.dtype)
    N_v, N_m, S, D = args.nv, args.nm, args.seg, cfg.D
    vlo, vhi = shard_rows(N_v, world, rank)
    mlo, mhi = shard_rows(N_m, world, rank)
    g = torch.Generator(device=dev).manual_seed(2 + rank)
    v = torch.nn.functional.normalize(torch.randn(vhi - vlo, D, device=dev, generator=g), dim=-1)
    seg = torch.randn(mhi - mlo, S, D, device=dev, generator=g)
    lens = torch.randint(12, S + 1, (mhi - mlo,), device=dev, generator=g)
    mask = (torch.arange(S, device=dev)[None] < lens[:, None]).float()
    seg = (seg * mask[:, :, None]).to(eng.tc)
    mu = torch.nn.functional.normalize(torch.randn(mhi - mlo, D, device=dev, generator=g), dim=-1)
    sr = ShardedRetrieval(lambda a, b, c, d: eng.retrieval_sim_matrix(a, b, c, d))

    def step():
        return sr.sim_rows(v, seg, mask, mu)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(max(args.warmup, 1)):
        rows = step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        rows = step()
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        te = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(te, op=dist.ReduceOp.MAX)
        elapsed = float(te.item())
    assert rows.shape == (vhi - vlo, N_m) and bool(torch.isfinite(rows).all())
    alg_bytes = 4.0 * (N_m * S * D + N_m * S + N_v * D + N_m * D + N_v * N_m)      # inputs once + sim matrix once
    pairs = float(N_v) * N_m
    roof = None
    if rank == 0:
        # dominant kernel, timed live with HIP events on its launch stream: the per-pair chain (made_xpool_fused when the engine
        # takes that path: bf16, D = 256); algorithmic flops per pair = 4 S D (scores + pooling) + 2 D^2 (Linear), SURVEY 8(d)
        with ops.KernelTimer() as kt:
            step()
        summ = kt.summary()
        k = "xpool_fused" if "xpool_fused" in summ else max(summ, key=lambda n: summ[n]["ms"])
        ms_k = summ[k]["ms"]
        fl = pairs / world * (4.0 * S * D + 2.0 * D * D) if k == "xpool_fused" else summ[k]["flops"]
        peak = PEAK_TFLOPS[args.dtype]
        roof = dict(bound="mfma", kernel=k, achieved=round(fl / (ms_k * 1e-3) / 1e12, 2), peak=peak, unit="TFLOP/s",
                    frac=round(fl / (ms_k * 1e-3) / 1e12 / peak, 4), traffic=None, launches_per_step=summ[k]["launches"],
                    avg_launch_us=round(ms_k * 1e3 / max(summ[k]["launches"], 1), 1),
                    algorithmic_gflop_per_launch=round(fl / max(summ[k]["launches"], 1) / 1e9, 1))
    if rank == 0:
        sec = elapsed / args.steps
        print(json.dumps({
            "metric": "retrieval sim-matrix GB/s (all-pairs video x music, X-Pool + dual tower)", "value": round(alg_bytes / sec / 1e9, 3),
            "unit": "GB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(sec * 1e3, 3),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"BASELINE.json configs[3]: N_v={N_v}, N_m={N_m}, S={S}, D={D}; videos row-sharded, music all-gathered",
                       "pairs_per_s": round(pairs / sec, 1), "algorithmic_gb": round(alg_bytes / 1e9, 3)},
            "roofline": roof,
            "cpu_baseline": (cpu_baseline_retrieval(cfg, synth.make_state_dict(cfg, seed=0), S) if (world == 1 and not args.no_cpu_baseline) else None)}))


def train_main(args, rank, world, local, dist):
    """BASELINE.json configs[2] (N = 1) / configs[4] (N > 1, per-rank B = 64): one full training iteration per step --
    train-mode forward (dropout on), matcher + criterion, hand-written backward, data-parallel gradient all-reduce (RCCL, one
    flat f32 buffer), three-group clipping + Adam, re-derivation of the bf16 weights (reference train-MaDe.py:337-381)."""
    from mgsv_amd.trainer import MadeTrainer
    cfg = cfg_headline()
    if args.ta:
        cfg.max_snippet_num = args.ta
        cfg.audio_attention_seqlen = max(cfg.audio_attention_seqlen, args.ta)
    B, Tv, Ta = args.batch, cfg.max_v_frames, cfg.max_snippet_num
    dev = torch.device("cuda", local)
    sd = synth.make_state_dict(cfg, seed=0)
    inp = synth.make_inputs(cfg, B, Tv, Ta, seed=1 + rank)
    trn = MadeTrainer(cfg, sd, device=dev, dtype=args.dtype)
    t = {k: torch.from_numpy(v).to(dev) for k, v in inp.items() if isinstance(v, np.ndarray)}
    it = [0]

    def step():
        it[0] += 1
        return trn.train_step(t["frame_feats"], t["segment_feats"], t["frame_masks"], t["segment_masks"], t["spans_target"],
                              seed=it[0], lrs=(1e-4, 1e-4, 1e-4), max_grad_norm=1.0, dist=dist)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(max(args.warmup, 1)):
        out = step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        te = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(te, op=dist.ReduceOp.MAX)
        elapsed = float(te.item())
    assert bool(torch.isfinite(out["localization_loss"]).all()) and bool(torch.isfinite(out["retrieval_loss"]).all())
    assert bool(torch.isfinite(trn.flat_param).all())
    per_kernel, roof = {}, None
    if rank == 0:
        with ops.KernelTimer() as kt:
            for _ in range(2):
                step()
        summ = kt.summary()
        for k, v in summ.items():
            per_kernel[k] = dict(launches_per_step=v["launches"] // 2, ms_per_step=round(v["ms"] / 2, 4),
                                 tflops=round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 2))
        mm = [k for k in summ if k.startswith("linear_") or k == "made_gemm_tn"]
        fl = sum(summ[k]["flops"] for k in mm)
        ms_ = sum(summ[k]["ms"] for k in mm)
        peak = PEAK_TFLOPS[args.dtype]
        roof = dict(bound="mfma", kernel="made_linear + made_gemm_tn (all GEMMs of the step)", achieved=round(fl / (ms_ * 1e-3) / 1e12, 2),
                    peak=peak, unit="TFLOP/s", frac=round(fl / (ms_ * 1e-3) / 1e12 / peak, 4), traffic=None)
    if rank == 0:
        sec = elapsed / args.steps
        print(json.dumps({
            "metric": "video-music pairs/s, full training step (fwd + bwd + matcher + clip/Adam), B=64 per GPU",
            "value": round(world * B / sec, 1), "unit": "pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(sec * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype,
            "data": "synthetic",
            "config": {"workload": f"BASELINE.json configs[{2 if world == 1 else 4}]: B={B}/GPU, T_v={Tv}, T_a={Ta}, D={cfg.D}, train mode (dropout on), "
                                   "f32 master weights + Adam, f32 gradient accumulation",
                       "global_batch": world * B, "parallelism": f"dp{world}: the flat f32 gradient buffer all-reduced in two buckets, the large one under the encoders' backward",
                       "launch": "eager"},
            "roofline": roof,
            "cpu_baseline": (cpu_baseline_train(cfg, sd, inp) if (world == 1 and not args.no_cpu_baseline) else None),
            "kernels": per_kernel}))


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run"
    assert torch.cuda.is_available(), "bench.py needs a GPU (the hot path has no CPU fallback)"
    torch.cuda.set_device(local)
    dist = None
    if world > 1 or "RANK" in os.environ:           # launched by torch.distributed.run: one process per GPU, RCCL
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))      # backend "nccl" is RCCL on ROCm

    if args.workload in ("retrieval", "train"):
        (retrieval_main if args.workload == "retrieval" else train_main)(args, rank, world, local, dist)
        if dist is not None:
            dist.destroy_process_group()
        return

    cfg = cfg_headline()
    B, Tv, Ta = args.batch, cfg.max_v_frames, cfg.max_snippet_num
    sd = synth.make_state_dict(cfg, seed=0)
    inp = synth.make_inputs(cfg, B, Tv, Ta, seed=1 + rank)
    dev = torch.device("cuda", local)
    # Steps are independent batches.  --in-flight N keeps N of them in flight: N engines (own workspace), each on its own HIP
    # stream with its own captured graph, steps issued round-robin.  One batch's decoder -- a chain of ~70 dependent launches
    # that leaves most of the chip idle -- then runs beside another batch's encoders.  Lane 0 is what the roofline leg times.
    n_lanes = max(1, args.in_flight)
    engines = [MadeEngine(cfg, sd, device=dev, dtype=args.dtype) for _ in range(n_lanes)]
    eng = engines[0]
    lane_inp = [inp] + [synth.make_inputs(cfg, B, Tv, Ta, seed=1 + rank + 1000 * l) for l in range(1, n_lanes)]
    lane_t = [{k: torch.from_numpy(v).to(dev) for k, v in li.items() if isinstance(v, np.ndarray)} for li in lane_inp]
    t = lane_t[0]
    lane_stream = [torch.cuda.Stream() for _ in range(n_lanes)] if n_lanes > 1 else [torch.cuda.current_stream()]

    def step(l=0):
        tl = lane_t[l]
        return engines[l].forward(tl["frame_feats"], tl["segment_feats"], tl["frame_masks"], tl["segment_masks"], tl["spans_target"])

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    outs = [step(l) for l in range(n_lanes)]         # allocates the workspaces
    torch.cuda.synchronize()
    launch = args.launch
    graphs = None
    if launch == "graph":
        try:
            graphs = []
            for l in range(n_lanes):
                side = torch.cuda.Stream()
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    step(l)
                torch.cuda.current_stream().wait_stream(side)
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    outs[l] = step(l)
                graphs.append(g)
        except Exception as ex:                      # report, do not hide: fall back to eager launches
            print(f"[bench] hipGraph capture failed ({type(ex).__name__}: {ex}); using eager launches", file=sys.stderr)
            launch, graphs = "eager", None
    out = outs[0]
    issued = [0]

    def run():
        l = issued[0] % n_lanes
        issued[0] += 1
        if n_lanes == 1:
            graphs[0].replay() if graphs else step(0)
            return
        with torch.cuda.stream(lane_stream[l]):
            graphs[l].replay() if graphs else step(l)

    for _ in range(args.warmup):
        run()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        run()
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        te = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(te, op=dist.ReduceOp.MAX)
        elapsed = float(te.item())
    ms = elapsed / args.steps * 1e3
    value = world * B * args.steps / elapsed

    # one batch alone (nothing else in flight): what a single forward takes end to end
    lat_ms = ms
    if n_lanes > 1:
        barrier()
        t1 = time.perf_counter()
        for _ in range(10):
            graphs[0].replay() if graphs else step(0)
        torch.cuda.synchronize()
        lat_ms = (time.perf_counter() - t1) / 10 * 1e3

    # sanity: the step really produced finite losses and a valid matching
    torch.cuda.synchronize()
    for o in outs:
        assert int(o["matcher_status"].cpu()) == 0
        assert bool(torch.isfinite(o["localization_loss"]).all()) and bool(torch.isfinite(o["retrieval_loss"]).all())

    # ---- roofline leg: HIP events around every launch of the dominant kernel, same stream, same step
    roof = None
    per_kernel = {}
    if rank == 0:
        with ops.KernelTimer() as kt:
            for _ in range(3):
                step()
        summ = kt.summary()
        # dominant kernel = the made_linear kernel symbol that does most of the step's arithmetic (timed launches are labelled
        # with the kernel they dispatch to, made_linear_variant, so this average sits beside rocprofv3's per-symbol average;
        # ranking by summed event time would favour the many tiny launches, whose event pairs cost as much as the kernels)
        lin = [k for k in summ if k.startswith("linear_")]
        dom = max(lin, key=lambda k: summ[k]["flops"]) if lin else max(summ, key=lambda k: summ[k]["ms"])
        d = summ[dom]
        achieved = d["flops"] / (d["ms"] * 1e-3) / 1e12
        peak = PEAK_TFLOPS[args.dtype]
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "r01_pmc_summary.json")
        if os.path.isfile(pmc):
            try:
                traffic = json.load(open(pmc)).get(dom, {}).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        all_lin = dict(launches_per_step=sum(summ[k]["launches"] for k in lin) // 3,
                       tflops=round(sum(summ[k]["flops"] for k in lin) / (sum(summ[k]["ms"] for k in lin) * 1e-3) / 1e12, 2)) if lin else None
        roof = dict(bound="mfma", kernel=f"{dom} (made_linear)", achieved=round(achieved, 2), peak=peak,
                    unit="TFLOP/s", frac=round(achieved / peak, 4), traffic=traffic,
                    launches_per_step=d["launches"] // 3, avg_launch_us=round(d["ms"] / d["launches"] * 1e3, 2),
                    algorithmic_gflop_per_launch=round(d["flops"] / d["launches"] / 1e9, 3),
                    algorithmic_mb_per_launch=round(d["bytes"] / d["launches"] / 1e6, 3),
                    all_made_linear_kernels=all_lin,
                    note="flops/bytes count the gathered (valid-token) rows only: padded tokens are not computed "
                         f"({100 * (1 - d['flops'] / max(d['flops_nominal'], 1)):.0f}% of this kernel's nominal work)")
        for k, v in summ.items():
            per_kernel[k] = dict(launches_per_step=v["launches"] // 3, ms_per_step=round(v["ms"] / 3, 4),
                                 tflops=round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 2),
                                 gbs=round(v["bytes"] / (v["ms"] * 1e-3) / 1e9, 1))

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(cfg, sd, inp, args.cpu_steps)

    if rank == 0:
        line = {
            "metric": "video-music pairs/s, eval forward (cross-modal transformer + DETR + matcher), B=64",
            "value": round(value, 1), "unit": "pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"BASELINE.json configs[1]: B={B}, T_v={Tv}, T_a={Ta}, D={cfg.D}, enc={cfg.detr_enc_layers}, "
                                   f"dec={cfg.detr_dec_layers}, Q={cfg.num_moment_queries}, concat fusion, fwd-only",
                       "global_batch": world * B, "launch": launch, "batches_in_flight": n_lanes,
                       "single_batch_ms": round(lat_ms, 4), "peak_hbm_gb": round(torch.cuda.max_memory_allocated() / 2**30, 2),
                       "parallelism": f"dp{world} (independent batches, no collective)",
                       "accumulate": "f32", "activations": args.dtype},
            "roofline": roof, "cpu_baseline": cpu, "kernels": per_kernel,
        }
        print(json.dumps(line))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

"""What the bf16 mode's error against the f32 oracle actually is, per output and per shape (VERDICT r4 'weak' 3): the forward outputs the tests bound
(tests/test_engine_gpu.py::test_bf16_forward_within_stated_tolerance, test_bench_shapes_gpu.py::test_eval_forward_at_ta1024_matches_oracle) and the
parameter gradients at the bench shapes (1 - cosine and relative L2 per tensor, dropout on).  The tests' bounds are 2x the maxima printed here.
    python tools/bf16_error_probe.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np, torch
from mgsv_amd import synth
from mgsv_amd.config import cfg_headline
from mgsv_amd.engine import MadeEngine
from mgsv_amd.trainer import MadeTrainer
from oracle import made_oracle as O
import test_engine_gpu as TE

KEYS = ("video_feats", "music_feats", "sims_dual", "sims_single", "pred_logits", "pred_spans")
worst = {k: 0.0 for k in KEYS}
worst_loss = 0.0
cases = dict(TE._cases())
c = cfg_headline(); c.max_snippet_num = 1024; c.audio_attention_seqlen = 1024
cases["cfg4_Ta1024_B3"] = (c, 3, 30, 1024)
print("## forward, bf16 engine against the f32 oracle: max |difference| per output")
for name, (cfg, B, Tv, Ta) in cases.items():
    sd = synth.make_state_dict(cfg, seed=0)
    inp = synth.make_inputs(cfg, B, Tv, Ta, seed=1)
    out = MadeEngine(cfg, sd, dtype="bf16").forward_numpy(inp)
    with torch.no_grad():
        ref = O.forward(O.to_torch_params(sd), cfg, inp["frame_feats"], inp["segment_feats"], inp["frame_masks"], inp["segment_masks"],
                        inp["spans_target"], v_duration=inp["v_duration"])
    errs = {k: float(np.abs(out[k] - ref[k].numpy()).max()) for k in KEYS}
    ll = abs(float(out["localization_loss"]) - float(ref["localization_loss"])) / max(1.0, abs(float(ref["localization_loss"])))
    rl = abs(float(out["retrieval_loss"]) - float(ref["retrieval_loss"])) / max(1.0, abs(float(ref["retrieval_loss"])))
    for k in KEYS:
        worst[k] = max(worst[k], errs[k])
    worst_loss = max(worst_loss, ll, rl)
    print(f"{name:36s} " + " ".join(f"{k} {v:.2e}" for k, v in errs.items()) + f" | loc loss rel {ll:.2e} ret loss rel {rl:.2e}", flush=True)
print("maxima: " + " ".join(f"{k} {v:.2e}" for k, v in worst.items()) + f" | losses rel {worst_loss:.2e}")

print("## gradients at the bench shapes, bf16 trainer against the oracle's f32 autograd, dropout on (seed 4321)")
for ta, B in ((512, 4), (1024, 2)):
    cfg = cfg_headline()
    if ta != 512:
        cfg.max_snippet_num = ta; cfg.audio_attention_seqlen = ta
    sd = synth.make_state_dict(cfg, seed=0)
    inp = synth.make_inputs(cfg, B, 30, ta, seed=1)
    trn = MadeTrainer(cfg, sd, dtype="bf16")
    for dropout in (True, False):
        trn.training_dropout = dropout
        res = trn.loss_and_grads(inp, seed=4321)
        P = O.to_torch_params(sd)
        for n in trn.param_names:
            P[n].requires_grad_(True)
        drop = O.Drop(4321, p_detr=cfg.detr_dropout) if dropout else None
        r = O.forward(P, cfg, inp["frame_feats"], inp["segment_feats"], inp["frame_masks"], inp["segment_masks"], inp["spans_target"],
                      v_duration=inp["v_duration"], drop=drop)
        (r["retrieval_loss"] + r["localization_loss"]).backward()
        rows = []
        gmax = max(float(P[n].grad.abs().max()) for n in trn.param_names if P[n].grad is not None)
        for n in trn.param_names:
            if P[n].grad is None:
                continue
            ref = P[n].grad.double().numpy().reshape(-1); got = res["grads"][n].astype(np.float64).reshape(-1)
            nr = np.linalg.norm(ref)
            if nr < 1e-6 * gmax * np.sqrt(ref.size):
                continue
            rows.append((1 - float(got @ ref / (np.linalg.norm(got) * nr + 1e-30)), float(np.linalg.norm(got - ref) / nr), n))
        rows.sort(reverse=True)
        lr = abs(res["retrieval_loss"] - float(r["retrieval_loss"])) / max(1.0, abs(float(r["retrieval_loss"])))
        ll = abs(res["localization_loss"] - float(r["localization_loss"])) / max(1.0, abs(float(r["localization_loss"])))
        print(f"T_a={ta} B={B} dropout={dropout}: losses rel {lr:.2e} {ll:.2e}; worst 1-cos {rows[0][0]:.3e} ({rows[0][2]}), median {np.median([x[0] for x in rows]):.2e}; "
              f"worst rel L2 {max(x[1] for x in rows):.3e}, median {np.median([x[1] for x in rows]):.2e}", flush=True)
        for x in rows[:4]:
            print(f"      1-cos {x[0]:.3e} rel {x[1]:.3e} {x[2]}")

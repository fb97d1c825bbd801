"""One configuration of the training path against the oracle's autograd, outside pytest: losses and the worst parameter gradients.
    python tools/train_variant_probe.py '{"dim_input": 128, "SA_temporal_heads": 4, "detr_nheads": 4}' [f32|bf16] [dropout 0|1] [seeds, comma separated] [B,Tv,Ta]
Prints, per seed, both losses (HIP / oracle) and the five largest relative L2 errors (and 1 - cosine) over the parameter tensors."""
import json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from mgsv_amd import synth
from mgsv_amd.config import cfg_native
from mgsv_amd.trainer import MadeTrainer
from oracle import made_oracle as O

ov = json.loads(sys.argv[1]) if len(sys.argv) > 1 else {}
dtype = sys.argv[2] if len(sys.argv) > 2 else "f32"
dropout = bool(int(sys.argv[3])) if len(sys.argv) > 3 else True
seeds = [int(x) for x in (sys.argv[4] if len(sys.argv) > 4 else "1234").split(",")]
B, Tv, Ta = [int(x) for x in (sys.argv[5] if len(sys.argv) > 5 else "3,20,40").split(",")]
cfg = cfg_native()
for k, v in ov.items():
    setattr(cfg, k, v)
sd = synth.make_state_dict(cfg, seed=0)
inp = synth.make_inputs(cfg, B, Tv, Ta, seed=1)
trn = MadeTrainer(cfg, sd, dtype=dtype)
trn.training_dropout = dropout
for seed in seeds:
    res = trn.loss_and_grads(inp, seed=seed)
    P = {k: (v.double() if v.is_floating_point() else v) for k, v in O.to_torch_params(sd).items()}
    for n in trn.param_names:
        P[n].requires_grad_(True)
    drop = O.Drop(seed, p_detr=cfg.detr_dropout) if dropout else None
    r = O.forward(P, cfg, inp["frame_feats"], inp["segment_feats"], inp["frame_masks"], inp["segment_masks"], inp["spans_target"],
                  v_duration=inp["v_duration"], drop=drop)
    (r["retrieval_loss"] + r["localization_loss"]).backward()
    rows = []
    for n in trn.param_names:
        ref = (P[n].grad if P[n].grad is not None else torch.zeros_like(P[n])).numpy().reshape(-1)
        got = res["grads"][n].astype(np.float64).reshape(-1)
        nr = np.linalg.norm(ref)
        if nr == 0:
            continue
        rows.append((float(np.linalg.norm(got - ref) / nr), 1 - float(got @ ref / (np.linalg.norm(got) * nr + 1e-30)), n))
    rows.sort(reverse=True)
    print(f"seed {seed}: retrieval {res['retrieval_loss']:.6f} / {float(r['retrieval_loss']):.6f}  localization {res['localization_loss']:.6f} / {float(r['localization_loss']):.6f}")
    for rel, omc, n in rows[:int(os.environ.get("TOP", 5))]:
        print(f"    rel {rel:.3e}  1-cos {omc:.3e}  {n}")
    if os.environ.get("DUMP"):                               # DUMP=<parameter name>: where its gradient differs
        n = os.environ["DUMP"]
        ref = P[n].grad.numpy().reshape(-1); got = res["grads"][n].astype(np.float64).reshape(-1)
        d = np.abs(got - ref); idx = np.argsort(-d)[:8]
        print(f"    {n}: |ref| max {np.abs(ref).max():.3e}; largest differences at", [(int(i), float(got[i]), float(ref[i])) for i in idx])
if os.environ.get("REPEAT"):
    # run-to-run: the same seed again, gradients compared bit for bit
    a = trn.loss_and_grads(inp, seed=seeds[0])
    for i in range(int(os.environ["REPEAT"])):
        b = trn.loss_and_grads(inp, seed=seeds[0])
        diff = [(float(np.abs(a["grads"][n] - b["grads"][n]).max()), n) for n in trn.param_names if not np.array_equal(a["grads"][n], b["grads"][n])]
        print(f"repeat {i}: {len(diff)} tensors differ", sorted(diff, reverse=True)[:4])

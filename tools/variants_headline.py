"""One training step of every round-2 training variant at the headline size (B = 64, T_a = 512, D = 512, bf16): finite, timed."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mgsv_amd import synth
from mgsv_amd.config import cfg_headline
from mgsv_amd.trainer import MadeTrainer
variants = [dict(with_cls_token=1, audio_attention_seqlen=600), dict(agg_module="mlp", video_transformer_depth=0, audio_transformer_depth=0),
            dict(vmr_fusion="XA-video-music", vmr_loss="single"), dict(moment_query_type="xpool"), dict(vmr_loss="dual_single_feature_fuse"),
            dict(with_cls_token=1, mml_fusion="CA", audio_attention_seqlen=600), dict(num_moment_queries=3)]
for ov in variants:
    cfg = cfg_headline()
    for k, v in ov.items(): setattr(cfg, k, v)
    B, Tv, Ta = 64, cfg.max_v_frames, cfg.max_snippet_num
    try:
        trn = MadeTrainer(cfg, synth.make_state_dict(cfg, seed=0), dtype="bf16")
        inp = synth.make_inputs(cfg, B, Tv, Ta, seed=1)
        t = {k: torch.from_numpy(v).cuda() for k, v in inp.items() if isinstance(v, np.ndarray)}
        for i in range(3):
            o = trn.train_step(t["frame_feats"], t["segment_feats"], t["frame_masks"], t["segment_masks"], t["spans_target"], seed=i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(10):
            o = trn.train_step(t["frame_feats"], t["segment_feats"], t["frame_masks"], t["segment_masks"], t["spans_target"], seed=10 + i)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 10 * 1e3
        ok = bool(torch.isfinite(trn.flat_param).all()) and bool(torch.isfinite(o["localization_loss"]).all()) and bool(torch.isfinite(o["retrieval_loss"]).all())
        print(ov, "finite" if ok else "NOT FINITE", f"{ms:.2f} ms/step", float(o["retrieval_loss"]), float(o["localization_loss"]), flush=True)
    except Exception as e:
        print(ov, "FAILED:", type(e).__name__, str(e)[:300], flush=True)
    del trn
    torch.cuda.empty_cache()

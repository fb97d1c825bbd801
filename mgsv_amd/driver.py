"""Entry-point logic behind `train-MaDe.py` / `test-MaDe.py` (reference train-MaDe.py:27-760, test-MaDe.py:27-520).

Same command-line flags, defaults and derived fields as the reference's `parse_option` (table `OPTIONS`), same dataset
layout (`dataset/MGSV-EC/*.csv` + `features/Kuai_feature/{vit_feature1,ast_feature2p5}/{*_feature,*_mask}/<id>.pt`,
reference dataloaders/dataloader_MGSV_EC_feature.py:30-75), same loop bodies; the model is the drop-in `Uni_model`, the
end-of-epoch evaluation keeps the similarity matrix and the ranking on the GPU (mgsv_amd/utils/util_test.py).

What is different on purpose:
  * launched with `python -m torch.distributed.run` (or plain `python` for one GPU): one process per GPU, RCCL;
  * `--synthetic_features 1` replaces the `.pt` feature files (absent from this repository) by seeded random features of the
    same shapes, so the drivers run end to end from the CSV metadata alone; `--max_samples N` truncates a split;
  * `--fused_step 1` (default) uses MadeTrainer.train_step (fused clip + Adam); `0` runs the reference's torch optimizer path.
"""
from __future__ import annotations

import argparse
import contextlib
import datetime
import logging
import math
import os
import random
import sys
import time
from typing import Dict, List

import numpy as np
import torch

# (flag, type, default, choices) -- reference train-MaDe.py:30-141; test-MaDe.py:30-117 adds --test_best, drops the two learning rates
OPTIONS = [
    ("name", str, None, None), ("output_dir", str, "./logs", None), ("load_uni_model_path", str, "", None), ("resume_path", str, None, None),
    ("data", str, "kuai50k", None), ("train_data", str, "kuai50k", None), ("val_data", str, "kuai50k", None), ("test_data", str, "kuai50k", None),
    ("train_csv", str, "dataset/MGSV-EC/train_data.csv", None), ("val_csv", str, "dataset/MGSV-EC/val_data.csv", None),
    ("test_csv", str, "dataset/MGSV-EC/test_data.csv", None), ("image_resolution", int, 224, None), ("max_v_frames", int, 30, None),
    ("max_m_duration", int, 240, None), ("stride", float, 2.5, None), ("filter", float, 4, None), ("padding", int, 0, None),
    ("toph_moment", int, 1, None), ("gt_moment_num", int, 1, None),
    ("backbone_type", str, "transf+detr", ["baseline", "transf+detr"]), ("dim_input", int, 256, None),
    ("frozen_feature_path", str, "features/Kuai_feature", None), ("video_encoder_type", str, "ViT", ["ViT", "ViViT"]),
    ("audio_encoder_type", str, "AST", ["MERT", "AST", "DeepSim"]), ("temperature_init_value", float, 0.07, None),
    ("video_attention_seqlen", int, 250, None), ("video_transformer_depth", int, 1, None), ("audio_transformer_depth", int, 1, None),
    ("with_cls_token", int, 0, None), ("with_last_token", int, 0, None), ("with_act_after_proj", int, 0, None),
    ("transformer_is_share", int, 0, None), ("projection_is_share", int, 0, None), ("SA_temporal_heads", int, 8, None),
    ("agg_module", str, "transf", ["None", "transf", "mlp"]), ("downup_is_share", int, 0, None), ("downup_dim", int, 64, None),
    ("vmr_fusion", str, "XA-music", ["NO", "XA", "XA-video", "XA-music", "XA-video-music", "XA-music-video"]),
    ("vmr_loss", str, "dual_single_loss_fuse", ["dual", "single", "dual_single", "dual_single_oneloss", "dual_single_sim_fuse",
                                                 "dual_single_loss_fuse", "dual_single_feature_fuse"]),
    ("dual_single_loss_weight", float, 1.0, None), ("fusion_mask", int, 1, None), ("mml_fusion", str, "CA", ["CA", "concat", "add"]),
    ("mml_localization", str, "detr", ["detr", "regression"]), ("hidden_dim", int, 256, None),
    ("moment_query_type", str, "video", ["video", "xpool", "music", "random", "zero"]), ("span_loss_type", str, "l1", ["l1", "ce"]),
    ("fb_label", str, "01", ["01", "10"]), ("detr_hidden_dim", int, 256, None), ("detr_dropout", float, 0.1, None), ("detr_nheads", int, 8, None),
    ("detr_dim_feedforward", int, 1024, None), ("detr_enc_layers", int, 0, None), ("detr_dec_layers", int, 6, None),
    ("detr_pre_norm", bool, False, None), ("num_moment_queries", int, 1, None), ("decoder_SA", int, 0, None), ("predict_center", int, 0, None),
    ("reg_mlp_num_layers", int, 3, None), ("l1_loss", int, 1, None), ("aux_loss", int, 1, None), ("contrastive_align_loss", int, 1, None),
    ("moment_loss", int, 0, None), ("audio_short_cut", int, 1, None), ("contrastive_dim", int, 256, None),
    ("position_embedding", str, "sine", ["sine", "learned"]), ("input_dropout", float, 0.5, None), ("ret_loss_weight", float, 3.0, None),
    ("loc_loss_weight", float, 0.2, None),
    ("do_train", "flag", False, None), ("do_eval", "flag", False, None), ("start_epoch", int, 0, None), ("epochs", int, 5, None), ("seed", int, 42, None),
    ("batch_size_train", int, 512, None), ("batch_size_val", int, 128, None), ("num_workers", int, 1, None), ("ignore_same_music", int, 1, None),
    ("world_size", int, 0, None), ("rank", int, 0, None), ("local_rank", int, 0, None), ("gradient_accumulation_steps", int, 1, None),
    ("matching_lr", float, 1e-4, None), ("detection_lr", float, 1e-4, None), ("decay_rate", float, 0.9, None), ("max_grad_norm", float, 1.0, None),
    ("scheduler", str, "warmupcosine", ["warmupcosine", "warmuplinear", "warmupconstant", "constant", "exponential"]),
    ("lr_update_rate", int, 50, None), ("warmup_rate", float, 0.1, None), ("distance_type", str, "COS", None), ("num_display", int, 15, None),
    ("tb_writer", int, 1, None), ("save_model", int, 1, None), ("save_json", int, 0, None),
]
TEST_ONLY = [("test_best", int, 0, None)]
TEST_DROPS = {"matching_lr", "detection_lr"}                  # the test parser has no optimizer learning rates
# additions of this build (not in the reference)
EXTRA = [("synthetic_features", int, 0, None), ("max_samples", int, 0, None), ("compute_dtype", str, "bf16", ["bf16", "f32"]),
         ("fused_step", int, 1, None),
         ("eval_in_flight", int, 2, None)]                    # evaluation: independent batches in flight (engine workspace + stream each)


def build_parser(for_test: bool = False) -> argparse.ArgumentParser:
    p = argparse.ArgumentParser("test-Uni" if for_test else "train-Uni", add_help=False)
    for name, typ, default, choices in OPTIONS + (TEST_ONLY if for_test else []) + EXTRA:
        if for_test and name in TEST_DROPS:
            continue
        if typ == "flag":
            p.add_argument("--" + name, action="store_true")
        elif name == "name":
            p.add_argument("--name", required=True, type=str)
        else:
            p.add_argument("--" + name, type=typ, default=default, **({"choices": choices} if choices else {}))
    return p


def parse_option(argv=None, for_test: bool = False):
    """reference train-MaDe.py:27-173 incl. the derived fields and consistency checks."""
    args = build_parser(for_test).parse_args(argv)
    args.train_data += "_uni"; args.val_data += "_uni"
    if for_test:
        args.test_data += "_uni"
    args.max_snippet_num = int(args.max_m_duration / args.stride)
    if "transf" not in args.agg_module:
        args.video_transformer_depth = args.audio_transformer_depth = 0
    assert (args.moment_loss >= args.audio_short_cut) or (args.contrastive_align_loss >= args.audio_short_cut), \
        "moment loss must be 1 when audio_short_cut is 1"
    args.hidden_dim = args.detr_hidden_dim = args.dim_input
    if "XA" in args.vmr_fusion and "single" not in args.vmr_loss:
        raise ValueError("XA fusion must support single tower loss in VMR")
    if args.decoder_SA == 0 and args.num_moment_queries > 1:
        raise ValueError("decoder_SA must be 1 when num_moment_queries > 1")
    music_dir = {2.5: "ast_feature2p5", 5.0: "ast_feature5", 7.5: "ast_feature7p5", 10.0: "ast_feature10"}
    args.music_frozen_feature_path = os.path.join(args.frozen_feature_path, music_dir[args.stride])
    args.frame_frozen_feature_path = os.path.join(args.frozen_feature_path, "vit_feature1")
    args.local_rank = int(os.environ.get("LOCAL_RANK", args.local_rank))
    return args


# --------------------------------------------------------------------------------------------- data
class MGSV_EC_Dataset(torch.utils.data.Dataset):
    """One row of the split CSV -> (data_map, meta_map, spans_target), reference dataloader_MGSV_EC_feature.py:6-75."""

    def __init__(self, csv_path: str, args):
        import pandas as pd
        self.args = args
        self.csv = pd.read_csv(csv_path)
        if getattr(args, "max_samples", 0):
            self.csv = self.csv.iloc[:args.max_samples]
        # packed stores written by tools/pack_features.py take precedence over the per-id .pt files (mgsv_amd/feature_store.py)
        from .feature_store import PackedFeatures
        self.packed = {}
        for kind, root in (("vit", args.frame_frozen_feature_path), ("ast", args.music_frozen_feature_path)):
            path = os.path.join(root, f"{kind}.made")
            if os.path.isfile(path):
                self.packed[kind] = PackedFeatures(path)

    def __len__(self):
        return len(self.csv)

    def _features(self, root: str, kind: str, ident: str, T: int, dim: int, length_hint: float):
        if kind in self.packed:
            return self.packed[kind].get(ident)
        fp = os.path.join(root, f"{kind}_feature", f"{ident}.pt")
        mp = os.path.join(root, f"{kind}_mask", f"{ident}.pt")
        if os.path.isfile(fp) and os.path.isfile(mp):
            feats, mask = torch.load(fp, map_location="cpu"), torch.load(mp, map_location="cpu")
        elif getattr(self.args, "synthetic_features", 0):
            import zlib
            g = torch.Generator().manual_seed(zlib.crc32(f"{kind}:{ident}".encode()))
            n = max(1, min(T, int(round(length_hint))))
            mask = (torch.arange(T) < n).float()
            feats = torch.randn(T, dim, generator=g)
        else:
            raise FileNotFoundError(f"{fp} (pass --synthetic_features 1 to run without extracted features)")
        return feats.masked_fill(mask.unsqueeze(-1) == 0, 0).float(), mask.float()

    def __getitem__(self, idx):
        a, r = self.args, self.csv.iloc[idx]
        video_id, music_id = str(r["video_id"]), str(r["music_id"])
        m_duration = float(r["music_total_duration"])
        gt = torch.tensor([[float(r["music_start"]), float(r["music_end"])]])
        v_dur = float(r["video_end"]) - float(r["video_start"])
        meta = {"video_id": video_id, "music_id": music_id, "v_duration": torch.tensor(v_dur), "m_duration": torch.tensor(m_duration),
                "gt_moment": gt}
        # the reference clamps the end of the moment IN PLACE on the tensor it has already stored as meta_map["gt_moment"]
        # (dataloader_MGSV_EC_feature.py:18-27,46-75), so the IoU ground truth is clamped too
        g = gt
        g[:, 1] = torch.clamp(g[:, 1], max=a.max_m_duration)
        spans_target = torch.stack([(g[:, 0] + g[:, 1]) / 2.0 / a.max_m_duration, (g[:, 1] - g[:, 0]) / a.max_m_duration], dim=-1)
        ff, fm = self._features(a.frame_frozen_feature_path, "vit", video_id, a.max_v_frames, 512, v_dur)
        sf, sm = self._features(a.music_frozen_feature_path, "ast", music_id, a.max_snippet_num, 768, m_duration / a.stride)
        return {"frame_feats": ff, "frame_mask": fm, "segment_feats": sf, "segment_mask": sm}, meta, spans_target


def make_loader(csv_path: str, args, batch_size: int, train: bool, world: int, rank: int):
    ds = MGSV_EC_Dataset(csv_path, args)
    sampler = torch.utils.data.distributed.DistributedSampler(ds, num_replicas=world, rank=rank, shuffle=train) if world > 1 else None
    dl = torch.utils.data.DataLoader(ds, batch_size=max(1, batch_size // max(world, 1)), num_workers=args.num_workers,
                                     shuffle=(train and sampler is None), sampler=sampler, drop_last=train, pin_memory=True)
    return dl, len(ds), sampler


# --------------------------------------------------------------------------------------------- schedules
def lr_factor(args, step: int, warmup_steps: int, total: int) -> float:
    """reference utils/scheduler.py (LambdaLR factors of WarmupCosine / WarmupLinear / WarmupConstant / Constant)."""
    s = args.scheduler
    if s == "constant":
        return 1.0
    if s == "exponential":
        # the reference steps its ExponentialLR when total_step % lr_update_rate == 0, checked BEFORE the increment (train-MaDe.py:
        # 379-381): the first decay lands right after step 0, so step s has seen ceil(s / rate) decays
        r = max(args.lr_update_rate, 1)
        return args.decay_rate ** ((step + r - 1) // r)
    if step < warmup_steps:
        return float(step) / float(max(1.0, warmup_steps))
    if s == "warmupconstant":
        return 1.0
    if s == "warmuplinear":
        return max(0.0, float(total - step) / float(max(1.0, total - warmup_steps)))
    progress = float(step - warmup_steps) / float(max(1, total - warmup_steps))
    return max(0.0, 0.5 * (1.0 + math.cos(math.pi * 0.5 * 2.0 * progress)))


# --------------------------------------------------------------------------------------------- loops
def get_logger(path: str = None):
    lg = logging.getLogger("MaDe")
    lg.setLevel(logging.INFO)
    if not lg.handlers:
        h = logging.StreamHandler(sys.stdout)
        h.setFormatter(logging.Formatter("%(asctime)s %(message)s", "%H:%M:%S"))
        lg.addHandler(h)
        if path:
            os.makedirs(os.path.dirname(path), exist_ok=True)
            lg.addHandler(logging.FileHandler(path))
    return lg


def init_runtime(args):
    random.seed(args.seed); np.random.seed(args.seed); torch.manual_seed(args.seed)
    dist = None
    if "RANK" in os.environ and int(os.environ.get("WORLD_SIZE", "1")) >= 1:
        import torch.distributed as dist
        torch.cuda.set_device(args.local_rank)
        if not dist.is_initialized():
            dist.init_process_group("nccl" if int(os.environ.get("WORLD_SIZE", "1")) > 1 else "gloo")
        args.world_size, args.rank = dist.get_world_size(), dist.get_rank()
    else:
        args.world_size, args.rank = 1, 0
    args.gpu_num = max(args.world_size, 1)
    device = torch.device("cuda", args.local_rank)
    stamp = time.strftime("%m%d", time.localtime())
    args.path_log = os.path.join(args.output_dir, args.train_data, f"{stamp}+{args.name}")
    logger = get_logger(os.path.join(args.path_log, f"{args.name}.log") if args.rank == 0 else None)
    return device, dist, logger


def save_model(epoch, args, logger, model, optimizer=None, loss=None, best_model=False, best_name="best"):
    """reference utils/util_train.py:21-36: `pytorch_model.bin.<epoch | best_name>` under args.path_log with the keys epoch / loss /
    model_state_dict / optimizer_state_dict."""
    if args.save_model == 0:
        return None
    os.makedirs(args.path_log, exist_ok=True)
    path = os.path.join(args.path_log, f"pytorch_model.bin.{best_name}" if best_model else f"pytorch_model.bin.{epoch}")
    torch.save({"epoch": epoch, "loss": loss if loss is not None else "None",
                "model_state_dict": {k: v.detach().cpu() for k, v in model.state_dict().items()},
                "optimizer_state_dict": optimizer.state_dict() if optimizer is not None else "None"}, path)
    logger.info("Model saved to %s", path)
    return path


def load_model(args, logger, model, path=None):
    """reference utils/util_train.py:38-60 (stage 0): strict load of `model_state_dict` (or of a bare state dict); the frozen
    `vit_model.` / `ast_model.` tensors a reference checkpoint also carries are never used on the feature path and are dropped by
    Uni_model.load_state_dict.  Returns (model, resume_epoch, resume_loss)."""
    path = path or args.resume_path or args.load_uni_model_path
    ckpt = torch.load(path, map_location="cpu", weights_only=False)
    model.load_state_dict(ckpt["model_state_dict"] if "model_state_dict" in ckpt else ckpt, strict=True)
    if getattr(args, "rank", 0) == 0:
        logger.info("Model loaded from %s", path)
    return model, (ckpt["epoch"] if "epoch" in ckpt else 0), (ckpt["loss"] if "loss" in ckpt else 0)


def build_model(args, device, logger, load: bool = True):
    from .model import Uni_model
    path = args.resume_path or args.load_uni_model_path
    # a path that is given but is not a file fails loudly, as the reference's torch.load does (main_test passes load=False for a
    # directory of checkpoints and loads them one by one).  As in the reference (train-MaDe.py:684-691) resuming does not move the
    # epoch counter by itself: pass --start_epoch with --resume_path.
    if load and path and not os.path.isfile(path):
        raise FileNotFoundError(f"checkpoint {path!r} (--resume_path / --load_uni_model_path) is not a file")
    model = Uni_model(args, device, logger, compute_dtype=args.compute_dtype)
    if load and path:
        load_model(args, logger, model, path)
    return model


def _to_device(data_map, meta_map, spans_target, device):
    f32 = torch.float32
    return (data_map["frame_feats"].to(device, f32), data_map["segment_feats"].to(device, f32), data_map["frame_mask"].to(device, f32),
            data_map["segment_mask"].to(device, f32), spans_target.to(device, f32), meta_map["v_duration"].to(device, f32))


def _batch_iou(args, model, output_map, meta_map, device):
    """IoU of every sample's top prediction with its ground-truth moment, on the device.  DETR head: reference test-MaDe.py:304-313
    (made_span_iou); regression head: test-MaDe.py:331-340 + music_detr/span_utils.py:119-139,147-170 -- the one regressed span,
    centre / width -> start / end in seconds, start clamped at 0, end at min(max_m_duration, the track's duration)."""
    from .utils.util_test import detr_iou_device
    if "regression" in args.mml_localization:
        sp = output_map["pred_spans"][:, 0].float()
        gt = meta_map["gt_moment"].to(device, torch.float32).reshape(sp.shape[0], -1)[:, :2]
        dur = meta_map["m_duration"].to(device, torch.float32)
        st = ((sp[:, 0] - 0.5 * sp[:, 1]) * float(args.max_m_duration)).clamp(min=0)
        ed = torch.minimum(((sp[:, 0] + 0.5 * sp[:, 1]) * float(args.max_m_duration)).clamp(max=float(args.max_m_duration)), dur)
        inter = (torch.minimum(gt[:, 1], ed) - torch.maximum(gt[:, 0], st)).clamp(min=0)
        union = (ed - st) + (gt[:, 1] - gt[:, 0]) - inter
        return torch.where((gt[:, 0] < gt[:, 1]) & (union > 0), inter / union.clamp(min=1e-30), torch.zeros_like(inter))
    iou, _ = detr_iou_device(output_map["pred_logits"], output_map["pred_spans"], meta_map["gt_moment"].to(device), meta_map["m_duration"].to(device),
                             model.criterion.foreground_label, float(args.max_m_duration))
    return iou


def train_one_epoch(epoch, args, model, loader, optimizer, device, dist, logger, total_step, warmup_steps):
    """reference train-MaDe.py:300-425."""
    from .utils.util_test import IoU_metrics, detr_iou_device
    model.train()
    ious: List[float] = []
    t0 = time.time()
    meter = {"loss": 0.0, "ret": 0.0, "loc": 0.0, "n": 0}
    for step, (data_map, meta_map, spans_target) in enumerate(loader):
        ff, sf, fm, sm, tg, vdur = _to_device(data_map, meta_map, spans_target, device)
        fac = lr_factor(args, args.total_step, warmup_steps, total_step)
        if args.fused_step:
            trn = model._trainer_ready()
            wr = torch.tensor([args.ret_loss_weight], device=device)
            wl = torch.tensor([args.loc_loss_weight], device=device)
            model._train_seed += 1
            o = trn.train_step(ff, sf, fm, sm, tg, seed=model._train_seed, lrs=(args.matching_lr * fac, args.matching_lr * fac, args.detection_lr * fac),
                               max_grad_norm=args.max_grad_norm, w_ret=wr, w_loc=wl, dist=dist if args.world_size > 1 else None, v_duration=vdur)
            ret, loc = o["retrieval_loss"][0] * args.ret_loss_weight, o["localization_loss"][0] * args.loc_loss_weight
            om = o
        else:
            for g, base in zip(optimizer.param_groups, (args.matching_lr, args.matching_lr, args.detection_lr)):
                g["lr"] = base * fac
            om, lm, *_ = model(ff, sf, fm, sm, tg, v_duration=vdur, video_ids=meta_map["video_id"], music_ids=meta_map["music_id"], is_train=True)
            ret, loc = lm["retrieval_loss"] * args.ret_loss_weight, lm["localization_loss"] * args.loc_loss_weight
            loss = ret + loc
            loss.backward()
            if dist is not None and args.world_size > 1:
                flat = model._trainer.flat_grad
                dist.all_reduce(flat); flat.div_(args.world_size)
            torch.nn.utils.clip_grad_norm_(model.get_temporal_parameter(), args.max_grad_norm)
            torch.nn.utils.clip_grad_norm_(model.get_matching_parameter(), args.max_grad_norm)
            torch.nn.utils.clip_grad_norm_(model.get_detection_parameter(), args.max_grad_norm)
            optimizer.step(); optimizer.zero_grad()
        iou = _batch_iou(args, model, om, meta_map, device)
        ious.extend(iou.cpu().tolist())
        args.total_step += 1
        b = ff.shape[0]
        meter["loss"] += float(ret + loc) * b; meter["ret"] += float(ret) * b; meter["loc"] += float(loc) * b; meter["n"] += b
        if args.rank == 0 and (step + 1) % max(1, len(loader) // max(args.num_display, 1)) == 0:
            logger.info(f"Train [{epoch}/{args.epochs}, {step + 1}/{len(loader)}] loss {meter['loss'] / meter['n']:.4f} "
                        f"ret {meter['ret'] / meter['n']:.4f} loc {meter['loc'] / meter['n']:.4f} lr x{fac:.3f} "
                        f"{(time.time() - t0) / (step + 1) * 1e3:.1f} ms/step")
    return meter["loss"] / max(meter["n"], 1), IoU_metrics(ious) if ious else {"mIoU": 0.0}


@torch.no_grad()
def eval_epoch(epoch, args, model, loader, device, dist, logger):
    """reference train-MaDe.py:430-625 / test-MaDe.py:255-470: per-batch forward, then the all-pairs similarity matrix
    (X-Pool + dual tower), de-duplicated recall, span IoU and the composite metrics -- all but the final scalars on the GPU."""
    from .utils.util_test import Composite_metrics, IoU_metrics, Recall_metrics, detr_iou_device
    model.eval()
    t0 = time.time()
    vids, mids, V, M, S, SM, IOU = [], [], [], [], [], [], []
    # Batches are independent: keep `eval_in_flight` of them in flight, each on its own HIP stream with its own engine workspace
    # (one batch's decoder, a chain of dependent launches, then runs beside the next batch's encoders), and read nothing back
    # until the loop is over: the loss is accumulated on the device.
    n_lanes = max(1, int(getattr(args, "eval_in_flight", 2))) if device.type == "cuda" else 1
    streams = [torch.cuda.Stream(device=device) for _ in range(n_lanes)] if n_lanes > 1 else [None]
    loss_acc = [torch.zeros((), device=device) for _ in range(n_lanes)]
    n = 0
    for it, (data_map, meta_map, spans_target) in enumerate(loader):
        lane = it % n_lanes
        if streams[lane] is not None:
            streams[lane].wait_stream(torch.cuda.current_stream(device))
        with torch.cuda.stream(streams[lane]) if streams[lane] is not None else contextlib.nullcontext():
            ff, sf, fm, sm, tg, vdur = _to_device(data_map, meta_map, spans_target, device)
            om, lm, feat, mask, ids = model(ff, sf, fm, sm, tg, v_duration=vdur, video_ids=meta_map["video_id"], music_ids=meta_map["music_id"],
                                            is_train=False, lane=lane)
            loss_acc[lane] += (lm["retrieval_loss"] * args.ret_loss_weight + lm["localization_loss"] * args.loc_loss_weight) * ff.shape[0]
            n += ff.shape[0]
            V.append(feat["video_feats"].clone()); M.append(feat["music_feats"].clone()); S.append(feat["segment_feats"].clone()); SM.append(sm)
            vids.extend(meta_map["video_id"]); mids.extend(meta_map["music_id"])
            iou = _batch_iou(args, model, om, meta_map, device)
            IOU.append(iou)
    if device.type == "cuda":
        torch.cuda.synchronize(device)
    loss_sum = float(sum(loss_acc))
    video, music, seg, segm, iou = torch.cat(V), torch.cat(M), torch.cat(S), torch.cat(SM), torch.cat(IOU)
    if "XA" not in args.vmr_fusion or args.vmr_loss == "dual":
        sim = model._engine_ready().dual_sims(video, music)
    elif args.vmr_loss == "single":
        sim = model._engine_ready().xpool_sims(video, seg.to(model._engine_ready().tc), segm if args.fusion_mask == 1 else None)
    else:
        sim = model.retrieval_sim_matrix(video, seg, segm, music)
    ret_metrics, ranks, _ = Recall_metrics(sim, dedup=True, all_music_ids_list=mids)
    iou_list = iou.cpu().tolist()
    loc_metrics = IoU_metrics(iou_list)
    com_metrics = Composite_metrics(ranks, iou_list, None, vids, mids)
    if args.rank == 0:
        logger.info(f"Video-to-Music Retrieval  Eval >>> R@1: {ret_metrics['R1']:.2f} - R@5: {ret_metrics['R5']:.2f} - R@10: {ret_metrics['R10']:.1f}"
                    f" - R@25: {ret_metrics['R25']:.1f} - R@50: {ret_metrics['R50']:.1f} - R@100: {ret_metrics['R100']:.1f}"
                    f" - Median R: {ret_metrics['MedianR']:.1f} - Mean R: {ret_metrics['MeanR']:.1f} - MRR: {ret_metrics['MRR']:.4f}")
        logger.info(f"Music Moment Localization Eval >>> mIoU: {loc_metrics['mIoU']:.4f} - IoU0.5: {loc_metrics['IoU@0.5']:.2f} - IoU0.7: {loc_metrics['IoU@0.7']:.2f}")
        logger.info(f"Composite Eval >> IoU0.5 - R1: {com_metrics['R1_iou0.5']:.2f} - R10: {com_metrics['R10_iou0.5']:.2f} - R100: {com_metrics['R100_iou0.5']:.2f}"
                    f" >> IoU0.7 - R1: {com_metrics['R1_iou0.7']:.2f} - R10: {com_metrics['R10_iou0.7']:.2f} - R100: {com_metrics['R100_iou0.7']:.2f}")
        logger.info(f"Eval takes {datetime.timedelta(seconds=int(time.time() - t0))} ({n} pairs)")
    return loss_sum / max(n, 1), ret_metrics, loc_metrics, com_metrics


def main_train(argv=None):
    args = parse_option(argv, for_test=False)
    device, dist, logger = init_runtime(args)
    if args.rank == 0:
        for k in sorted(vars(args)):
            logger.info(f"--{k} {vars(args)[k]}")
    model = build_model(args, device, logger)
    val_loader, val_len, _ = make_loader(args.val_csv, args, args.batch_size_val, False, 1, 0)      # every rank scores the whole split
    results = {}
    if args.do_train:
        train_loader, train_len, sampler = make_loader(args.train_csv, args, args.batch_size_train, True, args.world_size, args.rank)
        total_step = len(train_loader) * args.epochs
        warmup_steps = int(total_step * args.warmup_rate)
        optimizer = None
        if not args.fused_step:
            optimizer = torch.optim.Adam([{"params": model.get_temporal_parameter(), "lr": args.matching_lr},
                                          {"params": model.get_matching_parameter(), "lr": args.matching_lr},
                                          {"params": model.get_detection_parameter(), "lr": args.detection_lr}])
        logger.info(f"train_length = {train_len}, val_length = {val_len}, total_step = {total_step}, warmup_steps = {warmup_steps}")
        args.total_step = 0
        # reference train-MaDe.py:686-727: four "best" checkpoints, each rewritten when its criterion is matched or beaten
        best = {"R1": dict(v=0.0, epoch=0, name="best_r1", strict=False), "mIoU": dict(v=0.0, epoch=0, name="best_iou", strict=False),
                "R1_iou0.5": dict(v=0.0, epoch=0, name="best_r1iou05", strict=True), "R1_iou0.7": dict(v=0.0, epoch=0, name="best_r1iou07", strict=False)}
        for epoch in range(args.start_epoch + 1, args.epochs + 1):
            if sampler is not None:
                sampler.set_epoch(epoch)
            tl, tm = train_one_epoch(epoch, args, model, train_loader, optimizer, device, dist, logger, total_step, warmup_steps)
            logger.info(f"Epoch {epoch}/{args.epochs} Finished, Train Loss: {tl:.4f}, train mIoU {tm['mIoU']:.4f}")
            vl, ret, loc, com = eval_epoch(epoch, args, model, val_loader, device, dist, logger)
            results[epoch] = dict(train_loss=tl, val_loss=vl, R1=ret["R1"], mIoU=loc["mIoU"], R5=ret["R5"], R1_iou05=com["R1_iou0.5"],
                                  R1_iou07=com["R1_iou0.7"])
            if args.rank == 0:
                now = {"R1": ret["R1"], "mIoU": loc["mIoU"], "R1_iou0.5": com["R1_iou0.5"], "R1_iou0.7": com["R1_iou0.7"]}
                for key, b in best.items():
                    if (now[key] > b["v"]) if b["strict"] else (now[key] >= b["v"]):
                        b["v"], b["epoch"] = now[key], epoch
                        save_model(epoch, args, logger, model, optimizer=None, loss=vl, best_model=True, best_name=b["name"])   # reference train-MaDe.py:686-727: optimizer=None
                logger.info("Best R1: %.4f in epoch %d, Best mIoU: %.4f in epoch %d, Best R1IoU0.5: %.4f in epoch %d, Best R1IoU0.7: %.4f in epoch %d",
                            best["R1"]["v"], best["R1"]["epoch"], best["mIoU"]["v"], best["mIoU"]["epoch"], best["R1_iou0.5"]["v"],
                            best["R1_iou0.5"]["epoch"], best["R1_iou0.7"]["v"], best["R1_iou0.7"]["epoch"])
    elif args.do_eval:
        vl, ret, loc, com = eval_epoch(0, args, model, val_loader, device, dist, logger)
        results[0] = dict(val_loss=vl, R1=ret["R1"], mIoU=loc["mIoU"])
    if dist is not None:
        dist.barrier(); dist.destroy_process_group()
    return results


def main_test(argv=None):
    """reference test-MaDe.py:472-520: --load_uni_model_path names one checkpoint file (`pytorch_model.*`), or a directory whose best
    checkpoints (--test_best 1) or per-epoch checkpoints (`pytorch_model.bin.<epoch>`, start_epoch + 1 .. epochs) are evaluated in
    turn.  Returns {checkpoint name: metrics}; a single file also under the keys loss / ret / loc / com."""
    args = parse_option(argv, for_test=True)
    device, dist, logger = init_runtime(args)
    model = build_model(args, device, logger, load=False)
    loader, n, _ = make_loader(args.test_csv, args, args.batch_size_val, False, 1, 0)
    logger.info(f"test_length = {n}")
    out = {}

    def run(path, tag):
        _, epoch, _ = load_model(args, logger, model, path)
        vl, ret, loc, com = eval_epoch(epoch, args, model, loader, device, dist, logger)
        out[tag] = dict(loss=vl, ret=ret, loc=loc, com=com, epoch=epoch)
        return out[tag]

    path = args.load_uni_model_path
    if path == "":
        # the reference does nothing without a checkpoint; scoring fresh weights is only useful as a plumbing check -- say so loudly
        logger.warning("test: no --load_uni_model_path given -- evaluating FRESHLY INITIALISED weights (plumbing check only)")
        vl, ret, loc, com = eval_epoch(0, args, model, loader, device, dist, logger)
        out.update(loss=vl, ret=ret, loc=loc, com=com)
    elif os.path.basename(path).split(".")[0] == "pytorch_model" and not os.path.isdir(path):
        out.update(run(path, os.path.basename(path)))
    elif args.test_best == 1:
        for name in ("pytorch_model.bin.best_r1iou07", "pytorch_model.bin.best_r1iou05", "pytorch_model.bin.best_r1", "pytorch_model.bin.best_iou"):
            f = os.path.join(path, name)
            if not os.path.exists(f):
                logger.info(f"Model {f} not exists")
                continue
            run(f, name)
    else:
        for epoch in range(args.start_epoch + 1, args.epochs + 1):
            f = os.path.join(path, f"pytorch_model.bin.{epoch}")
            if not os.path.exists(f):
                logger.info(f"Model {f} not exists")
                continue
            r = run(f, f"pytorch_model.bin.{epoch}")
            assert r["epoch"] == epoch, f"resume_epoch {r['epoch']} != epoch {epoch}"
    if dist is not None:
        dist.barrier(); dist.destroy_process_group()
    return out

"""Forward executor of the MaDe hot path on MI355X.

`MadeEngine` owns the packed device weights and a shape-keyed workspace and runs the
reference's `Uni_model.forward` (reference model/model_Uni.py:177-322, eval mode) plus the
all-pairs retrieval scoring of reference test-MaDe.py:386-403 as a fixed sequence of kernels
from libmade_hip.so, all on the current HIP stream with no host synchronisation (so the whole
step can be captured in a HIP graph).  PyTorch only provides device memory and the stream.

Data layout in HBM (B = batch, L = T_v + T_a under concat fusion, D = dim_input):
  * activations are row-major [rows, D] in the compute dtype (f32 or bf16), one row per token;
  * the DETR input `fus` [B, L, D] is written in place by the two temporal encoders' last GEMMs
    (frame rows 0..T_v-1, segment rows T_v..L-1) -- the reference's torch.cat never happens;
  * q | k | v of every attention live side by side in one [rows, 3D] buffer written by one GEMM; the
    attention kernels read V row-major and transpose it on the fly (ds_read_b64_tr_b16);
  * the decoder never projects the L memory rows: its cross-attention runs in memory space
    (made_attention_wide over memory + pos) with W_k folded onto the query and W_v / out_proj folded
    into one Linear on the pooled rows (removes 25 % of the forward flops).
"""
from __future__ import annotations

import math
import os
from typing import Dict, Optional

import numpy as np
import torch

from . import _lib, ops
from .config import MadeConfig
from .ops import Seg, round_up

Tensor = torch.Tensor


class MadeEngine:
    def __init__(self, cfg: MadeConfig, state_dict: Dict[str, object], device="cuda:0", dtype: str = "f32"):
        assert dtype in ("f32", "f32x3", "bf16")
        self.cfg = cfg
        self.device = torch.device(device)
        self.tc = torch.bfloat16 if dtype == "bf16" else torch.float32
        self.dtype_name = dtype
        # "f32x3": f32 storage and elementwise arithmetic like "f32"; the matrix products run as three bf16 products on split operands
        # (made_set_f32_products(1): include/made_hip.h) instead of the exact-f32 MFMA at 1/16 of the bf16 rate
        self._f32_products = 1 if dtype == "f32x3" else 0
        self._check_supported()
        self._ws: Dict[tuple, Dict[str, Tensor]] = {}
        self.load_state_dict(state_dict)

    # ------------------------------------------------------------------ config support
    def _check_supported(self):
        c = self.cfg
        unsupported = []
        if c.agg_module == "mlp":
            if c.video_transformer_depth != 0 or c.audio_transformer_depth != 0:
                unsupported.append("agg_module=mlp with a temporal transformer depth > 0 (the reference asserts 0, model_Base.py:308)")
        elif "transf" not in c.agg_module or c.video_transformer_depth < 1 or c.audio_transformer_depth < 1:
            unsupported.append(f"agg_module={c.agg_module} / temporal transformer depth 0")
        if "concat" not in c.mml_fusion and "CA" not in c.mml_fusion:
            unsupported.append(f"mml_fusion={c.mml_fusion}")
        if "XA" not in c.vmr_fusion or not ("music" in c.vmr_fusion or "video" in c.vmr_fusion):
            unsupported.append(f"vmr_fusion={c.vmr_fusion}")
        if "music" not in c.vmr_fusion and c.vmr_loss not in ("dual", "single"):
            unsupported.append(f"vmr_loss={c.vmr_loss} without the music-pooling tower (the reference fails there too)")
        if not ("detr" in c.mml_localization or "regression" in c.mml_localization):
            unsupported.append(f"mml_localization={c.mml_localization}")
        if c.audio_short_cut and not c.contrastive_align_loss:
            unsupported.append("audio_short_cut without contrastive_align_loss")
        if c.moment_query_type not in ("video", "music", "zero", "random", "xpool"):
            unsupported.append(f"moment_query_type={c.moment_query_type}")
        if c.moment_query_type == "xpool" and "music" not in c.vmr_fusion:
            unsupported.append("moment_query_type=xpool without the music-pooling tower (the reference fails there too)")
        if c.vmr_loss not in ("dual", "single", "dual_single_loss_fuse", "dual_single_sim_fuse", "dual_single_feature_fuse"):
            unsupported.append(f"vmr_loss={c.vmr_loss}")
        if c.detr_dec_layers < 1:
            unsupported.append("detr_dec_layers=0")
        if c.D not in (128, 256, 512):
            unsupported.append(f"dim_input={c.D} (the wide-head attention kernel is built for 128, 256 and 512)")
        if unsupported:
            raise NotImplementedError("MadeEngine (HIP path) does not cover yet: " + "; ".join(unsupported))

    # ------------------------------------------------------------------ weights
    def load_state_dict(self, sd: Dict[str, object]):
        """Pack reference-layout weights for the kernels: matrices in the compute dtype,
        biases / LayerNorm parameters / tables in f32."""
        dev, tc = self.device, self.tc
        c = self.cfg
        D = c.D

        def T(name) -> Tensor:
            v = sd[name]
            t = v.detach().clone() if isinstance(v, torch.Tensor) else torch.from_numpy(np.asarray(v).copy())
            return t.to(dev, torch.float32)

        P: Dict[str, Tensor] = {}

        def mat(key, t: Tensor):
            P[key] = t.to(tc).contiguous()

        def vec(key, t: Tensor):
            P[key] = t.to(torch.float32).contiguous()

        def lin(key, name):
            mat(key + ".w", T(name + ".weight"))
            vec(key + ".b", T(name + ".bias"))

        def ln(key, name):
            vec(key + ".g", T(name + ".weight"))
            vec(key + ".b", T(name + ".bias"))

        lin("vit_proj", "vit_proj")
        lin("ast_proj", "ast_proj")
        if c.agg_module == "mlp":                                        # reference model/model_Base.py:216-249,357-377
            for key, mod in (("video_mlp", "Video_encoder_projection"), ("audio_mlp", "Music_encoder_projection")):
                lin(key + ".0", mod + ".net.0"); lin(key + ".3", mod + ".net.3"); lin(key + ".6", mod + ".net.6")
                for bn in ("1", "4"):                                     # eval-mode BatchNorm1d over the token axis -> (scale, shift) per position
                    sc = T(f"{mod}.net.{bn}.weight") / torch.sqrt(T(f"{mod}.net.{bn}.running_var") + 1e-5)
                    vec(f"{key}.{bn}.scale", sc)
                    vec(f"{key}.{bn}.shift", T(f"{mod}.net.{bn}.bias") - T(f"{mod}.net.{bn}.running_mean") * sc)
        else:
            vec("pe_video", T("video_position_embedding.pe")[0])
            vec("pe_audio", T("audio_position_embedding.pe")[0])
            share = bool(c.transformer_is_share) and c.video_transformer_depth == c.audio_transformer_depth
            if c.with_cls_token:                                          # reference model/model_Base.py:314-321
                vec("cls_video", T("video_cls_token").view(-1))
                vec("cls_audio", T("audio_cls_token").view(-1))
            for mod, depth in (("video_transformer", c.video_transformer_depth), ("audio_transformer", c.audio_transformer_depth)):
                src = "share_transformer" if share else mod               # one block for both towers (model_Base.py:322-331)
                for l in range(depth):
                    p, q = f"{mod}.layers.{l}", f"{src}.layers.{l}"
                    ln(p + ".ln1", q + ".0")
                    mat(p + ".in.w", T(q + ".1.in_proj_weight")); vec(p + ".in.b", T(q + ".1.in_proj_bias"))
                    lin(p + ".out", q + ".1.out_proj")
                    ln(p + ".ln2", q + ".2")
                    lin(p + ".ff1", q + ".3.0")
                    lin(p + ".ff2", q + ".3.3")
                lin(mod + ".final", src + ".final_linear")
        towers = []
        if "music" in c.vmr_fusion:
            towers.append(("xa", "video_guided_to_music_pooling_cross_transformer"))
        if "video" in c.vmr_fusion:                                      # reference model/model_Uni.py:26-27
            towers.append(("xav", "music_guided_to_video_pooling_cross_transformer"))
        for key, xa in towers:
            ln(key + ".ln1", xa + ".layer_norm1"); ln(key + ".ln2", xa + ".layer_norm2"); ln(key + ".ln3", xa + ".layer_norm3")
            lin(key + ".q", xa + ".cross_attn.q_proj")
            mat(key + ".kv.w", torch.cat([T(xa + ".cross_attn.k_proj.weight"), T(xa + ".cross_attn.v_proj.weight")], 0))
            vec(key + ".kv.b", torch.cat([T(xa + ".cross_attn.k_proj.bias"), T(xa + ".cross_attn.v_proj.bias")], 0))
            lin(key + ".out", xa + ".cross_attn.out_proj")
            lin(key + ".lin", xa + ".linear_proj")
            # LayerNorm2's affine part and the residual of modules/transformer.py:176 folded into the Linear (float64 on the host), for
            # the retrieval path whose attention kernel emits xhat = (o - mean) * rstd:
            #   x = xhat g2 + b2,  x + (W x + b) = (W + I) diag(g2) xhat + ((W + I) b2 + b)
            W64 = T(xa + ".linear_proj.weight").double().cpu() + torch.eye(D, dtype=torch.float64)
            g2, b2 = T(xa + ".layer_norm2.weight").double().cpu(), T(xa + ".layer_norm2.bias").double().cpu()
            mat(key + ".linf.w", (W64 * g2[None, :]).float().to(dev))
            vec(key + ".linf.b", (W64 @ b2 + T(xa + ".linear_proj.bias").double().cpu()).float().to(dev))
            # W'' 1 of the stored (compute-dtype) W'': made_xpool_sims forms W'' xhat as k1 (W'' o) + k2 (W'' 1) from z = P.(U W''^T)
            vec(key + ".linf.rs", P[key + ".linf.w"].double().sum(1))
        vec("logit_scale", T("logit_scale").view(1))
        if "CA" in c.mml_fusion:                                          # reference model/model_Base.py:99-213
            ca = "video_music_fusion_cross_transformer"
            mat("ca.q.w", T(ca + ".layers.0.0.to_q.weight"))
            mat("ca.kv.w", T(ca + ".layers.0.0.to_kv.weight"))
            lin("ca.out", ca + ".layers.0.0.to_out.0")
            lin("ca.ff1", ca + ".layers.0.1.net.0")
            lin("ca.ff2", ca + ".layers.0.1.net.3")
            ln("ca.lnq", ca + ".attention_query_layer_norms.0")
            ln("ca.lnc", ca + ".attention_context_layer_norms.0")
            ln("ca.lnf", ca + ".ff_layer_norms.0")
            lin("ca.final", ca + ".final_linear")
        for l in range(c.detr_enc_layers):
            p = f"detr_transformer.encoder.layers.{l}"
            mat(p + ".in.w", T(p + ".self_attn.in_proj_weight")); vec(p + ".in.b", T(p + ".self_attn.in_proj_bias"))
            lin(p + ".out", p + ".self_attn.out_proj")
            lin(p + ".ff1", p + ".linear1"); lin(p + ".ff2", p + ".linear2")
            ln(p + ".ln1", p + ".norm1"); ln(p + ".ln2", p + ".norm2")
        if c.detr_pre_norm and c.detr_enc_layers > 0:
            ln("enc.norm", "detr_transformer.encoder.norm")              # (the reference has it with normalize_before only, transformer.py:34)
        H, hd = c.detr_nheads, D // c.detr_nheads
        for l in range(c.detr_dec_layers):
            p = f"detr_transformer.decoder.layers.{l}"
            mat(p + ".sa.in.w", T(p + ".self_attn.in_proj_weight")); vec(p + ".sa.in.b", T(p + ".self_attn.in_proj_bias"))
            lin(p + ".sa.out", p + ".self_attn.out_proj")
            # one query and one key: softmax == 1, so self-attention is out_proj(v_proj(x)); fold both
            # Linears (float64 on the host, once per load)
            sw, sb = T(p + ".self_attn.in_proj_weight").double().cpu(), T(p + ".self_attn.in_proj_bias").double().cpu()
            so, sob = T(p + ".self_attn.out_proj.weight").double().cpu(), T(p + ".self_attn.out_proj.bias").double().cpu()
            mat(p + ".sa.fold.w", (so @ sw[2 * D:]).float().to(dev)); vec(p + ".sa.fold.b", (so @ sb[2 * D:] + sob).float().to(dev))
            # cross-attention in memory space: move W_k onto the query and W_v (with out_proj) onto the pooled rows.
            #   q'_h = W_k,h^T (W_q,h x + b_q,h)            -> one Linear  D -> H*D   (b_k shifts all keys alike: no effect)
            #   out  = sum_h W_o[:, h] W_v,h pooled_h + W_o b_v + b_o -> one Linear H*D -> D
            w, b = T(p + ".multihead_attn.in_proj_weight").double().cpu(), T(p + ".multihead_attn.in_proj_bias").double().cpu()
            wo, bo = T(p + ".multihead_attn.out_proj.weight").double().cpu(), T(p + ".multihead_attn.out_proj.bias").double().cpu()
            wq_h, wk_h, wv_h = w[:D].view(H, hd, D), w[D:2 * D].view(H, hd, D), w[2 * D:].view(H, hd, D)
            mat(p + ".ca.qk.w", torch.einsum("hjn,hjk->hnk", wk_h, wq_h).reshape(H * D, D).float().to(dev))
            vec(p + ".ca.qk.b", torch.einsum("hjn,hj->hn", wk_h, b[:D].view(H, hd)).reshape(H * D).float().to(dev))
            mat(p + ".ca.vo.w", torch.einsum("mhj,hjn->mhn", wo.view(D, H, hd), wv_h).reshape(D, H * D).float().to(dev))
            mat(p + ".ca.v.w", w[2 * D:].float().to(dev)); mat(p + ".ca.o.w", wo.float().to(dev))    # the same two Linears, not folded (fused chain)
            vec(p + ".ca.vo.b", (wo @ b[2 * D:] + bo).float().to(dev))
            lin(p + ".ff1", p + ".linear1"); lin(p + ".ff2", p + ".linear2")
            ln(p + ".ln1", p + ".norm1"); ln(p + ".ln2", p + ".norm2"); ln(p + ".ln3", p + ".norm3")
        ln("dec.norm", "detr_transformer.decoder.norm")
        mat("query_embed", T("decoder_query_embed.weight"))
        if "regression" in c.mml_localization:                           # reference model/model_Uni.py:66-69
            for i in range(3):
                lin(f"reg_mlp.{i}", f"reg_mlp.layers.{i}")
        else:
            lin("class_embed", "class_embed")
            for i in range(3):
                lin(f"span_embed.{i}", f"span_embed.layers.{i}")
                if c.moment_loss:
                    lin(f"moment_embed.{i}", f"moment_embed.layers.{i}")
            if c.contrastive_align_loss:
                lin("proj_q", "contrastive_align_projection_query")
                lin("proj_v", "contrastive_align_projection_vid")
            vec("empty_weight", T("criterion.empty_weight"))
        # constants
        i = torch.arange(D, dtype=torch.float32)
        vec("dim_t", (10000.0 ** (2 * torch.div(i, 2, rounding_mode="floor") / D)).to(dev))
        w_contr = 0.2 if c.contrastive_align_loss else 0.0
        vec("crit_weights", torch.tensor([4.0 if c.l1_loss else 0.0, 1.0, 0.8, 0.0, w_contr], device=dev))
        self.P = P

    def _side_stream(self):
        if _lib.variant_env("MADE_ONE_STREAM", "0") == "1":     # debugging: every branch on the caller's stream, in issue order
            return torch.cuda.current_stream()
        if getattr(self, "_side", None) is None:
            self._side = torch.cuda.Stream(device=self.device)
        return self._side

    # ------------------------------------------------------------------ workspace
    def _buffers(self, B: int, Tv: int, Ta: int) -> Dict[str, Tensor]:
        key = (B, Tv, Ta)
        ws = self._ws.get(key)
        if ws is not None:
            return ws
        c, dev, tc = self.cfg, self.device, self.tc
        concat = "concat" in c.mml_fusion
        D, L, Q = c.D, (Tv + Ta if concat else Ta), c.num_moment_queries
        F_t, F_d, nd = c.temporal_ffn_dim, c.detr_dim_feedforward, c.detr_dec_layers
        Lmax = max(L, Ta + 1, Tv + 1) if c.with_cls_token else max(L, Ta, Tv)
        Lpad = round_up(Lmax, 64)
        rows = B * Lmax

        def E(*shape, dtype=None):
            return torch.empty(shape, device=dev, dtype=dtype or tc)

        def Z(*shape, dtype=None):
            return torch.zeros(shape, device=dev, dtype=dtype or tc)

        vr = B * (Tv + 1 if c.with_cls_token else Tv)
        ws = dict(
            # private scratch of the video branch (it runs on its own stream beside the audio branch)
            v_x0=E(vr, D), v_x1=E(vr, D), v_x2=E(vr, D), v_x3=E(vr, D), v_xin=E(vr * c.vit_dim), v_qkv=E(vr, 3 * D),
            v_att=E(vr, D), v_ffn=E(vr, F_t),
            fus=E(B, L, D), fus_mask=E(B, L, dtype=torch.float32), pos=E(B, L, D), srcpos=E(B * L, D),
            x0=E(rows, D), x1=E(rows, D), x2=E(rows, D), x3=E(rows, D), xin=E(B * max(Tv * c.vit_dim, Ta * c.ast_dim)),
            qkv=E(rows, 3 * D), att=E(rows, D), ffn=E(rows, max(F_t, F_d)),
            dq_all=E(B * Q, c.detr_nheads * D), dpool=E(B * Q, c.detr_nheads * D),
            dws=E(32 * B * Q * max(D, 256) * 4 * max(1, nd // 4), dtype=torch.float32),   # split-K partials of the skinny GEMMs
            part_o=E(B * 8 * c.detr_nheads * Q * D, dtype=torch.float32), part_ml=E(B * 8 * c.detr_nheads * Q * 4, dtype=torch.float32),
            vmean=E(B, D, dtype=torch.float32), mmean=E(B, D, dtype=torch.float32),
            video=E(B, D, dtype=torch.float32), music=E(B, D, dtype=torch.float32),
            tgt=E(B * Q, D), t1=E(B * Q, D), t2=E(B * Q, D), tx=E(B * Q, D),
            dqkv=E(B * Q, 3 * D), datt=E(B * Q, D),
            dz=E(3, B * Q, D, dtype=torch.float32), dv=E(B * Q, D), dzero=Z(B * Q, D, dtype=torch.float32),   # fused decoder chain: raw (pre-norm) rows
            dffn=E(B * Q, F_d), hs=E(nd, B * Q, D),
            logits=E(nd, B, Q, 2, dtype=torch.float32), spans=E(nd, B, Q, 2, dtype=torch.float32),
            h1=E(nd * B * Q, D), h2=E(nd * B * Q, D),
            sims_single=E(B, B, dtype=torch.float32), sims_dual=E(B, B, dtype=torch.float32), sims_vp_t=E(B, B, dtype=torch.float32),
            ret_loss=E(1, dtype=torch.float32),
            # row gather (valid-token lists) of the video / audio / fused sequences
            rows_v=(E(B * Tv, dtype=torch.int32), E(1, dtype=torch.int32)), rows_a=(E(B * Ta, dtype=torch.int32), E(1, dtype=torch.int32)),
            rows_f=(E(B * L, dtype=torch.int32), E(1, dtype=torch.int32)),
            order_v=E(B, dtype=torch.int32), order_a=E(B, dtype=torch.int32), order_f=E(B, dtype=torch.int32),
        )
        if c.contrastive_align_loss:
            Dc = c.contrastive_hdim
            ws.update(pq_raw=E(nd * B * Q, Dc, dtype=torch.float32), pq=E(nd, B, Q, Dc, dtype=torch.float32),
                      pv_raw=E(B * Tv, Dc, dtype=torch.float32), pv=E(B, Tv, Dc, dtype=torch.float32),
                      vid_sum=E(B, Dc, dtype=torch.float32))
        if not concat:                      # CA fusion: encoders write their own buffers, the fusion block writes `fus`
            inner = c.ca_heads * c.ca_dim_head
            ws.update(frame_buf=E(B, Tv, D), seg_buf=E(B, Ta, D), ca_nx=E(B * Ta, D), ca_nc=E(B * Tv, D),
                      ca_q=E(B * Ta, inner), ca_kv=E(B * Tv, 2 * inner), ca_att=E(B * Ta, inner),
                      ca_x=E(B * Ta, D), ca_h=E(B * Ta, c.ca_ffn_dim), ca_y=E(B * Ta, D))
        self._ws[key] = ws
        return ws

    # ------------------------------------------------------------------ building blocks
    def _mha_block(self, x: Tensor, B: int, T: int, w_in: Tensor, b_in: Tensor, key_mask: Optional[Tensor],
                   ws: Dict[str, Tensor], H: int, pos: Optional[Tensor] = None, skip: Optional[Tensor] = None, rows=None,
                   order: Optional[Tensor] = None) -> Tensor:
        """packed in-proj -> flash attention; x [B*T, D]; q,k from (x + pos), v from x; returns att [B*T, D]."""
        D = self.cfg.D
        qkv = ws["qkv"][:B * T]
        if pos is None:
            ops.linear(x, w_in, b_in, out=qkv, rows=rows)
        else:                                       # q and k are projected from x + pos (given precomputed), v from x
            ops.linear(x, w_in, b_in, A2=pos, a2_replace=True, rows=rows,
                       segs=[Seg(out=qkv, col_begin=0, use_a2=True), Seg(out=qkv[:, 2 * D:], col_begin=2 * D, ldo=qkv.stride(0))])
        q3 = qkv.view(B, T, 3 * D)
        att = ws["att"][:B * T]
        ops.attention(q3[:, :, :D], q3[:, :, D:2 * D], q3[:, :, 2 * D:], att.view(B, T, D), H, key_mask=key_mask,
                      q_skip_mask=key_mask if skip is not None else None, order=order)
        return att

    def _encode(self, feats: Tensor, mask: Tensor, which: str, wsall: Dict[str, Tensor], row_off: int, rows=None, order=None) -> None:
        """reference model/model_Base.py:544-617 -> writes fus[:, row_off:row_off+T] and mean/normalised vector."""
        c, P = self.cfg, self.P
        ws = wsall if which == "audio" else {**wsall, **{k[2:]: v for k, v in wsall.items() if k.startswith("v_")}}
        B, T, Kin = feats.shape
        D = c.D
        proj, mod, pe, depth = (("vit_proj", "video_transformer", "pe_video", c.video_transformer_depth) if which == "video"
                                else ("ast_proj", "audio_transformer", "pe_audio", c.audio_transformer_depth))
        nrow = B * T
        mflat = mask.reshape(-1)
        act = ops.ACT_QUICKGELU if c.with_act_after_proj else ops.ACT_NONE
        if "concat" in c.mml_fusion:
            local = ws["fus"][:, row_off:row_off + T]                   # [B, T, D] view
        else:
            local = ws["frame_buf"] if which == "video" else ws["seg_buf"]
        mean, vec = (ws["vmean"], ws["video"]) if which == "video" else (ws["mmean"], ws["music"])
        if c.agg_module == "mlp":
            return self._encode_mlp(feats, mask, which, ws, local, mean, vec)
        if c.with_cls_token:
            return self._encode_cls(feats, mask, which, ws, local, vec)
        if P[pe].shape[0] < T:
            raise ValueError(f"{which} position table holds {P[pe].shape[0]} positions < T={T} "
                             "(reference model/model_Base.py:533 raises here too)")
        # padded tokens (mask 0) are never read by a valid token: the GEMMs gather the valid rows only (`rows`), the row kernels
        # and the attention skip them; valid rows are bit-identical to the dense computation
        if self.tc == torch.bfloat16:       # mask + f32->bf16 once, then the direct-to-LDS GEMM
            xin = ops.cast_mask_rows(feats.view(nrow, Kin), mflat, ws["xin"][:nrow * Kin].view(nrow, Kin))
            x = ops.linear(xin, P[proj + ".w"], P[proj + ".b"], act=act, R=P[pe][:T], r_row_mod=T, out=ws["x0"][:nrow], rows=rows)
        else:
            x = ops.linear(feats.view(nrow, Kin), P[proj + ".w"], P[proj + ".b"], a_row_mask=mflat, act=act,
                           R=P[pe][:T], r_row_mod=T, out=ws["x0"][:nrow], rows=rows)
        x = self._temporal_layers(x, B, T, mask, mod, depth, ws, rows, order)
        ops.linear(x, P[mod + ".final.w"], P[mod + ".final.b"], out_row_mask=mflat, tile_skip_mask=mflat,
                   segs=[Seg(out=local, ldo=local.stride(1), rows_per_batch=T, out_batch_stride=local.stride(0))])
        ops.masked_mean(local, mask, out=mean)
        ops.l2norm_rows(mean, out_f32=vec)

    def _temporal_layers(self, x: Tensor, B: int, T: int, mask: Tensor, mod: str, depth: int, ws, rows, order) -> Tensor:
        """reference model/model_Base.py:82-91 without the final Linear; x [B*T, D] in ws["x0"]."""
        c, P = self.cfg, self.P
        nrow = B * T
        mflat = mask.reshape(-1)
        for l in range(depth):
            p = f"{mod}.layers.{l}"
            x1 = ops.layernorm(x, P[p + ".ln1.g"], P[p + ".ln1.b"], out=ws["x1"][:nrow], row_skip=mflat)
            att = self._mha_block(x1, B, T, P[p + ".in.w"], P[p + ".in.b"], mask, ws, c.SA_temporal_heads, skip=mflat, rows=rows, order=order)
            x2 = ops.linear(att, P[p + ".out.w"], P[p + ".out.b"], R=x1, out=ws["x2"][:nrow], rows=rows)
            x3 = ops.layernorm(x2, P[p + ".ln2.g"], P[p + ".ln2.b"], out=ws["x3"][:nrow], row_skip=mflat)
            h = ops.linear(x3, P[p + ".ff1.w"], P[p + ".ff1.b"], act=ops.ACT_GELU, out=ws["ffn"][:nrow, :c.temporal_ffn_dim], rows=rows)
            x = ops.linear(h, P[p + ".ff2.w"], P[p + ".ff2.b"], R=x3, out=ws["x0"][:nrow], rows=rows)
        return x

    def _encode_mlp(self, feats: Tensor, mask: Tensor, which: str, ws, local: Tensor, mean: Tensor, vec: Tensor) -> None:
        """agg_module = "mlp" (reference model/model_Base.py:567-570,606-609 with EmbeddingNet :216-249, eval mode): projection,
        Linear - BatchNorm1d - ReLU - Linear - BatchNorm1d - ReLU - Linear on every token, masked mean.  The BatchNorms' channel
        axis is the token position, so T must equal the length they were built for."""
        c, P = self.cfg, self.P
        B, T, Kin = feats.shape
        key, proj, Tbn = (("video_mlp", "vit_proj", c.max_v_frames) if which == "video" else ("audio_mlp", "ast_proj", c.max_snippet_num))
        if T != Tbn:
            raise ValueError(f"agg_module=mlp: {which} sequence length {T} != {Tbn} positions of its BatchNorm1d (the reference fails there too)")
        nrow = B * T
        mflat = mask.reshape(-1)
        act = ops.ACT_QUICKGELU if c.with_act_after_proj else ops.ACT_NONE
        if self.tc == torch.bfloat16:
            xin = ops.cast_mask_rows(feats.view(nrow, Kin), mflat, ws["xin"][:nrow * Kin].view(nrow, Kin))
            x = ops.linear(xin, P[proj + ".w"], P[proj + ".b"], act=act, out=ws["x0"][:nrow])
        else:
            x = ops.linear(feats.view(nrow, Kin), P[proj + ".w"], P[proj + ".b"], a_row_mask=mflat, act=act, out=ws["x0"][:nrow])
        h = ops.linear(x, P[key + ".0.w"], P[key + ".0.b"], out=ws["ffn"][:nrow, :1024])
        ops.row_affine(h, P[key + ".1.scale"], P[key + ".1.shift"], act=ops.ACT_RELU)
        y = ops.linear(h, P[key + ".3.w"], P[key + ".3.b"], out=ws["x1"][:nrow])
        ops.row_affine(y, P[key + ".4.scale"], P[key + ".4.shift"], act=ops.ACT_RELU)
        ops.linear(y, P[key + ".6.w"], P[key + ".6.b"], out_row_mask=mflat,
                   segs=[Seg(out=local, ldo=local.stride(1), rows_per_batch=T, out_batch_stride=local.stride(0))])
        ops.masked_mean(local, mask, out=mean)
        ops.l2norm_rows(mean, out_f32=vec)

    def _encode_cls(self, feats: Tensor, mask: Tensor, which: str, ws, local: Tensor, vec: Tensor) -> None:
        """with_cls_token (reference model/model_Base.py:527-530,572-574): a learned token is prepended (its mask entry is 1), the
        block runs on T + 1 positions, the clip vector is the token's output and the local features are the other T rows."""
        c, P = self.cfg, self.P
        B, T, Kin = feats.shape
        D = c.D
        proj, mod, pe, depth, cls = (("vit_proj", "video_transformer", "pe_video", c.video_transformer_depth, "cls_video") if which == "video"
                                     else ("ast_proj", "audio_transformer", "pe_audio", c.audio_transformer_depth, "cls_audio"))
        T1 = T + 1
        if P[pe].shape[0] < T1:
            raise ValueError(f"{which} position table holds {P[pe].shape[0]} positions < T+1={T1} (reference model/model_Base.py:533)")
        mask1 = torch.cat([torch.ones_like(mask[:, :1]), mask], dim=1).contiguous()
        rows1 = ops.row_index(mask1)
        order1 = ops.batch_order(mask1)
        act = ops.ACT_QUICKGELU if c.with_act_after_proj else ops.ACT_NONE
        x = ws["x0"][:B * T1]
        x3d = x.view(B, T1, D)
        x3d[:, 0] = (P[cls] + P[pe][0]).to(self.tc)                      # the token, with position 0's encoding
        seg = Seg(out=x3d[:, 1:], ldo=D, rows_per_batch=T, out_batch_stride=T1 * D)
        if self.tc == torch.bfloat16:
            xin = ops.cast_mask_rows(feats.view(B * T, Kin), mask.reshape(-1), ws["xin"][:B * T * Kin].view(B * T, Kin))
            ops.linear(xin, P[proj + ".w"], P[proj + ".b"], act=act, R=P[pe][1:T1], r_row_mod=T, segs=[seg])
        else:
            ops.linear(feats.view(B * T, Kin), P[proj + ".w"], P[proj + ".b"], a_row_mask=mask.reshape(-1), act=act,
                       R=P[pe][1:T1], r_row_mod=T, segs=[seg])
        x = self._temporal_layers(x, B, T1, mask1, mod, depth, ws, rows1, order1)
        y = ops.linear(x, P[mod + ".final.w"], P[mod + ".final.b"], out_row_mask=mask1.reshape(-1), out=ws["x1"][:B * T1])
        y3 = y.view(B, T1, D)
        ops.l2norm_rows(y3[:, 0], out_f32=vec)
        local.copy_(y3[:, 1:])

    # ------------------------------------------------------------------ X-Pool scoring
    def xpool_sims(self, video: Tensor, seg: Tensor, seg_mask: Tensor, sims_out: Optional[Tensor] = None,
                   pooled_out: Optional[Tensor] = None, chunk_m: Optional[int] = None, tower: str = "xa") -> Tensor:
        """sims[n, m] = <v_n/|v_n|, XA(v, seg, mask)[m, n]/|.|>: reference modules/transformer.py:156-180 +
        modules/metrics.py:10-24.  video [Nv, D] f32; seg [Nm, S, D] (compute dtype, strided ok); mask [Nm, S].
        tower = "xav": the music-guided video pooling block (reference model_Uni.py:203, metrics.py:26-41) -- same
        arithmetic with the roles swapped: pass (music, frames, frame mask) and read the result as sims[m, v].
        Music tracks are processed in chunks so the per-pair intermediates stay bounded."""
        P, tc, dev = self.P, self.tc, self.device
        Nv, D = video.shape
        Nm, S, _ = seg.shape
        if sims_out is None:
            sims_out = torch.empty(Nv, Nm, device=dev, dtype=torch.float32)
        chunk_m_given = chunk_m is not None
        if chunk_m is None:
            budget = 6 << 30                                             # bytes of per-pair intermediates per chunk
            per_m = Nv * 3 * D * tc.itemsize + 4 * S * D * tc.itemsize
            chunk_m = max(1, min(Nm, budget // max(per_m, 1)))
        v1 = ops.layernorm(video, P[tower + ".ln1.g"], P[tower + ".ln1.b"], out_dtype=tc)
        q = ops.linear(v1, P[tower + ".q.w"], P[tower + ".q.b"])
        hoist = Nv > S          # out_proj commutes with the softmax-weighted sum (rows sum to 1): apply it to U instead
        if tc == torch.bfloat16 and D == 256 and pooled_out is None and Nv >= 256:
            # retrieval scale: the per-pair chain (attention -> LayerNorm2 -> Linear + residual -> LayerNorm3 -> cosine) in one
            # kernel, nothing per pair ever written to HBM (made_xpool_fused); the per-track K / U projections stay GEMMs
            vn = ops.l2norm_rows(video)
            cm = min(Nm, max(1, (2 << 30) // (3 * S * D * tc.itemsize)), 65535)
            if S <= 96 and _lib.variant_env("MADE_XPOOL_SIMS", "1") != "0":
                # round 4: the per-pair Linear moved onto the values.  W'' o = sum_s p_s (W'' u_s), so u''_s = W'' u_s is made once per
                # segment (one more GEMM over the tracks) and the pair costs a second P.V product (2 S D flops) instead of the Linear (2 D^2):
                # 2.2x fewer flops per pair at S = 96.  The default for tracks of at most 96 segments (made_xpool_sims: 52.5 ms against
                # made_xpool_fused's 60.7-61.3 on the 53 k x 4 k set); MADE_XPOOL_SIMS=0: the fused kernel
                s1 = torch.empty(cm * S, D, device=dev, dtype=tc)
                kbuf = torch.empty(cm * S, D, device=dev, dtype=tc)
                ubuf = torch.empty(cm * S, D, device=dev, dtype=tc)
                uu = torch.empty(cm * S, 2 * D, device=dev, dtype=tc)
                xws = torch.empty(ops.xpool_sims_ws_bytes(Nv, cm, D), device=dev, dtype=torch.uint8)
                for m0 in range(0, Nm, cm):
                    n = min(cm, Nm - m0)
                    skip = seg_mask[m0:m0 + n].reshape(-1) if seg_mask is not None else None
                    ops.layernorm(seg[m0:m0 + n], P[tower + ".ln1.g"], P[tower + ".ln1.b"], out=s1[:n * S], row_skip=skip)
                    ops.linear(s1[:n * S], P[tower + ".kv.w"], P[tower + ".kv.b"], tile_skip_mask=skip,
                               segs=[Seg(out=kbuf, col_begin=0), Seg(out=ubuf, col_begin=D)])
                    ops.linear(ubuf[:n * S], P[tower + ".out.w"], P[tower + ".out.b"], out=uu[:n * S, :D], tile_skip_mask=skip)
                    ops.linear(uu[:n * S, :D], P[tower + ".linf.w"], None, out=uu[:n * S, D:], tile_skip_mask=skip)
                    ops.xpool_sims(q, kbuf[:n * S].view(n, S, D), uu[:n * S].view(n, S, 2 * D),
                                   seg_mask[m0:m0 + n] if seg_mask is not None else None, P[tower + ".linf.b"], P[tower + ".linf.rs"],
                                   (P[tower + ".ln3.g"], P[tower + ".ln3.b"]), vn, sims_out[:, m0:m0 + n], scale=1.0 / math.sqrt(D),
                                   ws=xws, prepare_ws=(m0 == 0))
                return sims_out
            s1 = torch.empty(cm * S, D, device=dev, dtype=tc)
            kbuf = torch.empty(cm * S, D, device=dev, dtype=tc)
            ubuf = torch.empty(cm * S, D, device=dev, dtype=tc)
            ubuf2 = torch.empty(cm * S, D, device=dev, dtype=tc)
            xws = torch.empty(ops.xpool_fused_ws_floats(Nv, cm, D), device=dev, dtype=torch.float32)
            for m0 in range(0, Nm, cm):
                n = min(cm, Nm - m0)
                skip = seg_mask[m0:m0 + n].reshape(-1) if seg_mask is not None else None
                ops.layernorm(seg[m0:m0 + n], P[tower + ".ln1.g"], P[tower + ".ln1.b"], out=s1[:n * S], row_skip=skip)
                ops.linear(s1[:n * S], P[tower + ".kv.w"], P[tower + ".kv.b"], tile_skip_mask=skip,
                           segs=[Seg(out=kbuf, col_begin=0), Seg(out=ubuf, col_begin=D)])
                ops.linear(ubuf[:n * S], P[tower + ".out.w"], P[tower + ".out.b"], out=ubuf2[:n * S], tile_skip_mask=skip)
                ops.xpool_fused(q, kbuf[:n * S].view(n, S, D), ubuf2[:n * S].view(n, S, D),
                                seg_mask[m0:m0 + n] if seg_mask is not None else None,
                                (P[tower + ".ln2.g"], P[tower + ".ln2.b"]), P[tower + ".lin.w"], P[tower + ".lin.b"],
                                (P[tower + ".ln3.g"], P[tower + ".ln3.b"]), vn, sims_out[:, m0:m0 + n], scale=1.0 / math.sqrt(D),
                                ws=xws, prepare_ws=(m0 == 0))
            return sims_out
        if tc == torch.bfloat16 and D in (256, 512) and S <= 512 and pooled_out is None and Nv >= 256 and _lib.variant_env("MADE_XPOOL_ATTN", "1") != "0":
            # retrieval scale at the widths / lengths made_xpool_fused does not serve (D = 512, or more than its segments): the attention
            # as ONE two-pass kernel per chunk of tracks (made_xpool_attention: scores of a whole track in LDS, the normalisation of
            # LayerNorm2 in its tail), then the folded Linear and LayerNorm3 + cosine.  Chunks of about 1 GB of per-pair rows per tensor:
            # measured (tools/retr512_breakdown.py, 8192 x 512 x S 512), chunks small enough to stay in the Infinity Cache lose more to
            # short launches (13.8 ms per pass at 12 tracks per chunk) than they gain (11.45 ms at 96).
            cm = chunk_m if chunk_m is not None and chunk_m_given else max(1, min(Nm, (1 << 30) // max(Nv * D * tc.itemsize, 1)))
            s1 = torch.empty(cm * S, D, device=dev, dtype=tc)
            kbuf = torch.empty(cm * S, D, device=dev, dtype=tc)
            ubuf = torch.empty(cm * S, D, device=dev, dtype=tc)
            ubuf2 = torch.empty(cm * S, D, device=dev, dtype=tc)
            xh = torch.empty(cm * Nv, D, device=dev, dtype=tc)
            y = torch.empty(cm * Nv, D, device=dev, dtype=tc)
            iws = torch.empty(cm * 32, device=dev, dtype=torch.int32)
            for m0 in range(0, Nm, cm):
                n = min(cm, Nm - m0)
                skip = seg_mask[m0:m0 + n].reshape(-1) if seg_mask is not None else None
                ops.layernorm(seg[m0:m0 + n], P[tower + ".ln1.g"], P[tower + ".ln1.b"], out=s1[:n * S], row_skip=skip)
                ops.linear(s1[:n * S], P[tower + ".kv.w"], P[tower + ".kv.b"], tile_skip_mask=skip,
                           segs=[Seg(out=kbuf, col_begin=0), Seg(out=ubuf, col_begin=D)])
                ops.linear(ubuf[:n * S], P[tower + ".out.w"], P[tower + ".out.b"], out=ubuf2[:n * S], tile_skip_mask=skip)
                ops.xpool_attention(q, kbuf[:n * S].view(n, S, D), ubuf2[:n * S].view(n, S, D),
                                    seg_mask[m0:m0 + n] if seg_mask is not None else None, xh[:n * Nv].view(n, Nv, D),
                                    scale=1.0 / math.sqrt(D), ws=iws)
                ops.linear(xh[:n * Nv], P[tower + ".linf.w"], P[tower + ".linf.b"], out=y[:n * Nv])
                ops.xpool_tail(y[:n * Nv], P[tower + ".ln3.g"], P[tower + ".ln3.b"], video, sims_out[:, m0:m0 + n], n, Nv)
            return sims_out
        cm = min(chunk_m, Nm)
        s1 = torch.empty(cm * S, D, device=dev, dtype=tc)
        kbuf = torch.empty(cm * S, D, device=dev, dtype=tc)
        ubuf = torch.empty(cm * S, D, device=dev, dtype=tc)
        ubuf2 = torch.empty(cm * S, D, device=dev, dtype=tc) if hoist else None
        o = torch.empty(cm * Nv, D, device=dev, dtype=tc)
        o2 = torch.empty(cm * Nv, D, device=dev, dtype=tc)
        o3 = torch.empty(cm * Nv, D, device=dev, dtype=tc)
        scale = 1.0 / math.sqrt(D)
        xib_ws = None
        for m0 in range(0, Nm, cm):
            n = min(cm, Nm - m0)
            skip = seg_mask[m0:m0 + n].reshape(-1) if seg_mask is not None else None    # masked segments are never attended to
            ops.layernorm(seg[m0:m0 + n], P[tower + ".ln1.g"], P[tower + ".ln1.b"], out=s1[:n * S], row_skip=skip)   # [n,S,D] view -> compact rows
            ops.linear(s1[:n * S], P[tower + ".kv.w"], P[tower + ".kv.b"], tile_skip_mask=skip,
                       segs=[Seg(out=kbuf, col_begin=0), Seg(out=ubuf, col_begin=D)])
            u = ubuf
            if hoist:
                u = ops.linear(ubuf[:n * S], P[tower + ".out.w"], P[tower + ".out.b"], out=ubuf2[:n * S], tile_skip_mask=skip)
            # all videos attend to each track's segments: softmax over segments, scores never leave the chip
            if tc == torch.bfloat16 and Nv <= 64 and S <= 512 and S * D >= 65536 and D in (256, 512) and _lib.variant_env("MADE_XPOOL_INBATCH", "1") != "0":
                # the in-batch shape (round 4): scores per (track, 128 segments), P.V per (track, 128 columns), bf16 probabilities between them
                if xib_ws is None:
                    xib_ws = torch.empty(ops.xpool_inbatch_ws_bytes(cm, S), device=dev, dtype=torch.uint8)      # (no counters in it since the one-launch form went: nothing to clear)
                ops.xpool_inbatch(q, kbuf[:n * S].view(n, S, D), u[:n * S].view(n, S, D), seg_mask[m0:m0 + n] if seg_mask is not None else None,
                                  o[:n * Nv].view(n, Nv, D), scale=scale, ws=xib_ws)
            else:
                ops.attention_wide(q.view(1, Nv, 1, D), kbuf[:n * S].view(n, S, D), u[:n * S].view(n, S, D),
                                   o[:n * Nv].view(n, Nv, 1, D), scale=scale,
                                   key_mask=seg_mask[m0:m0 + n] if seg_mask is not None else None, shared_q=True,
                                   n_split=1)   # (splitting the keys over workgroups + a merge launch is no faster here and costs the other stream CUs:
                                   # 1.2 % of the eval throughput with two batches in flight)
            rows = n * Nv
            if hoist:
                a2 = o[:rows]
            else:
                a2 = ops.linear(o[:rows], P[tower + ".out.w"], P[tower + ".out.b"], out=o2[:rows])
            a3 = ops.layernorm(a2, P[tower + ".ln2.g"], P[tower + ".ln2.b"], out=o3[:rows])
            y = ops.linear(a3, P[tower + ".lin.w"], P[tower + ".lin.b"], R=a3, out=o2[:rows] if hoist else o[:rows])
            ops.xpool_tail(y, P[tower + ".ln3.g"], P[tower + ".ln3.b"], video, sims_out[:, m0:m0 + n], n, Nv,
                           pooled_out=pooled_out[m0 * Nv:(m0 + n) * Nv] if pooled_out is not None else None)
        return sims_out

    def dual_sims(self, video: Tensor, music: Tensor, out: Optional[Tensor] = None, add: Optional[Tensor] = None) -> Tensor:
        """cos(v, m) (reference modules/loss.py:52-56), always with the exact-f32 MFMA; `add` is summed in."""
        vn = ops.l2norm_rows(video)
        mn = ops.l2norm_rows(music)
        Nv, Nm, D = vn.shape[0], mn.shape[0], vn.shape[1]
        if Nv <= 256 and Nm <= 256 and Nm % 4 == 0 and (add is None or add.stride(0) % 4 == 0):   # in-batch: one output tile -> split K over workgroups
            if out is None:
                out = torch.empty(Nv, Nm, device=vn.device, dtype=torch.float32)
            split = max(2, min(16, D // 32))
            wsp = torch.empty(split * Nv * Nm, device=vn.device, dtype=torch.float32)
            ops.linear_splitk(vn, mn, None, wsp, split, R=add, out=out)
            return out
        return ops.linear(vn, mn, None, R=add, out=out, out_dtype=torch.float32)

    def _set_products(self) -> None:
        """the library's process-wide f32 product mode <- this engine's (read by the f32 kernels' launchers)"""
        if self.tc == torch.float32:
            from . import _lib
            _lib.check(_lib.lib().made_set_f32_products(self._f32_products), "made_set_f32_products")

    def retrieval_sim_matrix(self, video_embeds: Tensor, segment_embeds: Tensor, segment_masks: Tensor,
                             music_embeds: Tensor, chunk_m: Optional[int] = None) -> Tensor:
        """reference test-MaDe.py:386-403: sim[Nv, Nm] = single (X-Pool) + dual (cosine)."""
        self._set_products()
        seg = segment_embeds.to(self.tc) if segment_embeds.dtype != self.tc else segment_embeds
        single = self.xpool_sims(video_embeds, seg, segment_masks if self.cfg.fusion_mask == 1 else None, chunk_m=chunk_m)
        return self.dual_sims(video_embeds, music_embeds, add=single)

    # ------------------------------------------------------------------ full forward
    @torch.no_grad()
    def forward(self, frame_feats: Tensor, segment_feats: Tensor, frame_masks: Tensor, segment_masks: Tensor,
                spans_target: Tensor, with_losses: bool = True, want_pooled: bool = False,
                v_duration: Optional[Tensor] = None) -> Dict[str, Tensor]:
        c, P = self.cfg, self.P
        self._set_products()
        regression = "regression" in c.mml_localization
        if c.predict_center == 1 and v_duration is None:
            raise ValueError("predict_center=1 needs v_duration (reference model/model_Uni.py:280-282)")
        B, Tv, _ = frame_feats.shape
        Ta = segment_feats.shape[1]
        concat = "concat" in c.mml_fusion
        D, L, Q, nd = c.D, (Tv + Ta if concat else Ta), c.num_moment_queries, c.detr_dec_layers
        H = c.detr_nheads
        ws = self._buffers(B, Tv, Ta)
        fm, sm = frame_masks.contiguous(), segment_masks.contiguous()

        # ---- temporal encoders (K1-K4) write straight into the fused DETR input.  The video branch (B*T_v rows: every
        # launch far smaller than the chip) runs on a second HIP stream beside the audio branch and joins before X-Pool.
        cur = torch.cuda.current_stream()
        side = self._side_stream()
        fus, fus_mask = ws["fus"], ws["fus_mask"]
        side.wait_stream(cur)
        # The audio branch (the launches that fill the chip) is enqueued first, the video branch (17 launches of at most a few
        # hundred workgroups) second on the side stream: eager launching is 1-3 % faster this way, graph replay unchanged.
        self._encode(segment_feats.contiguous(), sm, "audio", ws, Tv, rows=ops.row_index(sm, out=ws["rows_a"]),
                     order=ops.batch_order(sm, out=ws["order_a"]))
        with torch.cuda.stream(side):
            # the DETR mask and its sine position embedding depend on the masks only: off the critical path
            ops.concat_cols(fm if concat else None, sm, fus_mask)
            pos = ops.sine_pe(fus_mask, P["dim_t"], out=ws["pos"])
            rows_f = ops.row_index(fus_mask, out=ws["rows_f"])
            order_f = ops.batch_order(fus_mask, out=ws["order_f"])       # attention workgroups: longest sample first
            self._encode(frame_feats.contiguous(), fm, "video", ws, 0, rows=ops.row_index(fm, out=ws["rows_v"]),
                         order=ops.batch_order(fm, out=ws["order_v"]))
        cur.wait_stream(side)
        if concat:
            frame, seg = fus[:, :Tv], fus[:, Tv:]
        else:
            frame, seg = ws["frame_buf"], ws["seg_buf"]
            self._ca_fusion(ws, frame, seg, fm, sm, B, Tv, Ta)
        video, music = ws["video"], ws["music"]
        out: Dict[str, Tensor] = dict(video_feats=video, music_feats=music, frame_feats=frame, segment_feats=seg)

        # ---- X-Pool similarities + retrieval loss (K5-K7): independent of the DETR branch, so they run on the side stream
        # beside the (latency-bound) decoder and join at the end of the step
        side.wait_stream(cur)
        dec_early = None
        with torch.cuda.stream(side):
            if not regression and c.moment_query_type != "xpool" and not c.detr_pre_norm:
                # The query side of decoder layer 0 (initial queries -> self-attention block -> the folded cross-attention
                # query) reads nothing the DETR encoder produces: it runs here, beside the encoder, instead of at the head of
                # the decoder's chain of dependent launches.
                if self._fused_decoder():
                    self._dec_fused_query_side(ws, 0, self._dec_first_rows(ws, video, music))
                else:
                    self._decoder_queries(ws, video, music, None, B)
                    self._decoder_query_side(ws, 0, B)
                dec_early = torch.cuda.Event()
                dec_early.record(side)
            need_pooled = want_pooled or c.moment_query_type == "xpool" or c.vmr_loss == "dual_single_feature_fuse"
            pooled = torch.empty(B * B, D, device=self.device, dtype=torch.float32) if (need_pooled and "music" in c.vmr_fusion) else None
            if "music" in c.vmr_fusion:
                self.xpool_sims(video, seg, sm if c.fusion_mask == 1 else None, sims_out=ws["sims_single"], pooled_out=pooled)
            if "video" in c.vmr_fusion:
                # music-guided video pooling (reference model_Uni.py:203, metrics.py:26-41): the same block with the roles
                # swapped gives sims[m, v]; the reference adds its transpose to the music-pooling similarities (:247-251)
                vp = self.xpool_sims(music, frame, fm if c.fusion_mask == 1 else None, sims_out=ws["sims_vp_t"], tower="xav")
                out["sims_video_pooling"] = vp.t()
                if "music" in c.vmr_fusion:
                    ws["sims_single"].add_(vp.t())
                else:
                    ws["sims_single"].copy_(vp.t())
            self.dual_sims(video, music, out=ws["sims_dual"])
            if with_losses:
                self._retrieval_loss(ws, video, music, pooled=pooled)
        out.update(sims_single=ws["sims_single"], sims_dual=ws["sims_dual"])
        if pooled is not None:
            out["music_feats_pooled"] = pooled.view(B, B, D)
        if with_losses:
            out["retrieval_loss"] = ws["ret_loss"]

        # ---- DETR encoder (K8, K9)
        rows = B * L
        src = fus.view(rows, D)
        pos2 = pos.view(rows, D)
        srcpos = ws["srcpos"]
        fskip = fus_mask.view(-1)             # padded tokens of the fused sequence: skipped everywhere below
        enc_rows, order_e = rows_f, order_f
        if regression:
            # the regression head sums the memory over ALL positions, padded ones included (reference model_Uni.py:229), so
            # here the encoder computes them too
            fskip = enc_rows = None
        if c.detr_pre_norm:
            # reference music_detr/transformer.py:170-189 (forward_pre): the residual stream x is never normalised inside a layer; norm 1 feeds
            # the attention (q = k = LN1(x) + pos, v = LN1(x)), norm 2 the FFN, and one more norm follows the last layer (:33-35,107-108)
            x = src
            for l in range(c.detr_enc_layers):
                p = f"detr_transformer.encoder.layers.{l}"
                n1 = ws["x2"][:rows]
                ops.layernorm_add(x, P[p + ".ln1.g"], P[p + ".ln1.b"], pos2, n1, srcpos, row_skip=fskip)
                att = self._mha_block(n1, B, L, P[p + ".in.w"], P[p + ".in.b"], fus_mask, ws, H, pos=srcpos, skip=fskip, rows=enc_rows, order=order_e)
                xa = ops.linear(att, P[p + ".out.w"], P[p + ".out.b"], R=x, out=ws["x1"][:rows], rows=enc_rows)
                n2 = ops.layernorm(xa, P[p + ".ln2.g"], P[p + ".ln2.b"], out=ws["x2"][:rows], row_skip=fskip)
                h = ops.linear(n2, P[p + ".ff1.w"], P[p + ".ff1.b"], act=ops.ACT_RELU, out=ws["ffn"][:rows, :c.detr_dim_feedforward], rows=enc_rows)
                x = ops.linear(h, P[p + ".ff2.w"], P[p + ".ff2.b"], R=xa, out=ws["x0"][:rows], rows=enc_rows)
            if c.detr_enc_layers > 0:
                src = ws["x3"][:rows]
                ops.layernorm_add(x, P["enc.norm.g"], P["enc.norm.b"], pos2, src, srcpos, row_skip=fskip)
            else:
                ops.layernorm_add(src, None, None, pos2, None, srcpos, row_skip=fskip)
        else:
            ops.layernorm_add(src, None, None, pos2, None, srcpos, row_skip=fskip)      # layer 0: src + pos (no norm)
        for l in range(0 if c.detr_pre_norm else c.detr_enc_layers):
            p = f"detr_transformer.encoder.layers.{l}"
            att = self._mha_block(src, B, L, P[p + ".in.w"], P[p + ".in.b"], fus_mask, ws, H, pos=srcpos, skip=fskip, rows=enc_rows, order=order_e)
            x = ops.linear(att, P[p + ".out.w"], P[p + ".out.b"], R=src, out=ws["x1"][:rows], rows=enc_rows)
            s1 = ops.layernorm(x, P[p + ".ln1.g"], P[p + ".ln1.b"], out=ws["x2"][:rows], row_skip=fskip)
            h = ops.linear(s1, P[p + ".ff1.w"], P[p + ".ff1.b"], act=ops.ACT_RELU, out=ws["ffn"][:rows, :c.detr_dim_feedforward], rows=enc_rows)
            x = ops.linear(h, P[p + ".ff2.w"], P[p + ".ff2.b"], R=s1, out=ws["x1"][:rows], rows=enc_rows)
            # the norm that produces the next src also emits src + pos (next layer's q/k input, decoder's keys)
            src = ws["x3" if l % 2 == 0 else "x0"][:rows]
            ops.layernorm_add(x, P[p + ".ln2.g"], P[p + ".ln2.b"], pos2, src, srcpos, row_skip=fskip)
        memory = src
        out["memory"] = memory.view(B, L, D)
        if regression:
            return self._regression_head(out, ws, memory.view(B, L, D), fus_mask, spans_target, v_duration, with_losses, cur, side)

        # ---- DETR decoder (K10).  Cross-attention runs in memory space (made_attention_wide): no projection of
        # the L memory rows at all; self-attention collapses to one folded Linear when there is a single query.
        mem3, mempos3 = memory.view(B, L, D), srcpos.view(B, L, D)
        tgt = ws["tgt"]
        if dec_early is not None:
            cur.wait_event(dec_early)                                    # layer 0's query side was computed beside the encoder
        else:
            if c.moment_query_type == "xpool":                           # reference model_Uni.py:222-223: the track's pooled vectors,
                cur.wait_stream(side)                                    # averaged over the videos of the batch (X-Pool branch first)
            if not self._fused_decoder():
                self._decoder_queries(ws, video, music, pooled, B)
        qp = P["query_embed"]
        hs = ws["hs"]
        ca_scale = 1.0 / math.sqrt(D // H)
        dq_all, dpool = ws["dq_all"], ws["dpool"]
        dq4 = dq_all.view(B, Q, H, D).permute(0, 2, 1, 3)               # [B, H, Q, D] view: row (b,q), head-major columns
        dp4 = dpool.view(B, Q, H, D).permute(0, 2, 1, 3)
        skinny = lambda A, wkey, **kw: self._skinny(ws, A, wkey, **kw)

        n_split = max(1, min(8, 256 // max(B, 1)))
        fused = self._fused_decoder()
        for l in range(nd):
            p = f"detr_transformer.decoder.layers.{l}"
            ln2, ln3 = [(P[p + f".ln{i}.g"], P[p + f".ln{i}.b"]) for i in (2, 3)]
            t1, t2 = ws["t1"], ws["t2"]
            if c.detr_pre_norm:
                # reference music_detr/transformer.py:246-271 (forward_pre): tgt is the un-normalised residual stream; every branch reads a
                # norm of it.  The self-attention always runs here.  hs[l] = decoder.norm(tgt after the layer) (:135-136).
                ta, tb = ws["dz"][0], ws["dz"][1]                 # (f32 rows: the stream is rounded to the compute dtype nowhere)
                tcur = tgt if l == 0 else ws["dz"][2]
                ops.layernorm(tcur, P[p + ".ln1.g"], P[p + ".ln1.b"], out=t1)
                if Q == 1:
                    skinny(t1, p + ".sa.fold", R=tcur, out=ta)
                else:
                    dqkv = ws["dqkv"]
                    ops.linear(t1, P[p + ".sa.in.w"], P[p + ".sa.in.b"], A2=qp, a2_row_mod=Q,
                               segs=[Seg(out=dqkv, col_begin=0, use_a2=True), Seg(out=dqkv[:, 2 * D:], col_begin=2 * D, ldo=dqkv.stride(0))])
                    d3 = dqkv.view(B, Q, 3 * D)
                    ops.attention(d3[:, :, :D], d3[:, :, D:2 * D], d3[:, :, 2 * D:], ws["datt"].view(B, Q, D), H)
                    skinny(ws["datt"], p + ".sa.out", R=tcur, out=ta)
                ops.layernorm(ta, ln2[0], ln2[1], out=t2)
                skinny(t2, p + ".ca.qk", A2=qp, a2_row_mod=Q, out=ws["dq_all"])
                ops.attention_wide(dq4, mempos3, mem3, dp4, scale=ca_scale, key_mask=fus_mask,
                                   n_split=n_split, part_o=ws["part_o"], part_ml=ws["part_ml"])
                skinny(dpool, p + ".ca.vo", R=ta, out=tb)
                ops.layernorm(tb, ln3[0], ln3[1], out=t1)
                skinny(t1, p + ".ff1", act=ops.ACT_RELU, out=ws["dffn"])
                skinny(ws["dffn"], p + ".ff2", R=tb, out=ws["dz"][2], ln1=(P["dec.norm.g"], P["dec.norm.b"]), ln1_out=hs[l])
                continue
            if fused:
                # Fused chain (made_dec_stage): every LayerNorm runs in the prologue of the Linear that consumes it, so a layer is
                # 8 launches instead of 12 and no split-K partial sums go through HBM.  z[0..2]: the raw rows before norm 1 / 2 / 3.
                z, hd = ws["dz"], D // H
                if l > 0 or dec_early is None:
                    self._dec_fused_query_side(ws, l, self._dec_first_rows(ws, video, music) if l == 0 else None)
                ops.attention_wide(dq4, mempos3, mem3, dp4, scale=ca_scale, key_mask=fus_mask,
                                   n_split=n_split, part_o=ws["part_o"], part_ml=ws["part_ml"])
                dv = ws["dv"]
                ops.linear(dpool[:, :D], P[p + ".ca.v.w"][:hd], None, M=B * Q, N=hd, K=D, batch=H, a_z_stride=D, w_z_stride=hd * D,
                           segs=[Seg(out=dv, ldo=D, out_z_stride=hd)])                       # v_h = W_v,h pooled_h (b_v rides in ca.vo.b)
                ops.linear(dv, P[p + ".ca.o.w"], P[p + ".ca.vo.b"], R=t1, out=z[1])
                ops.dec_stage(z[1], P[p + ".ff1.w"], P[p + ".ff1.b"], ws["dffn"], ln=ln2, x_out=t2, act=ops.ACT_RELU)
                ops.linear(ws["dffn"], P[p + ".ff2.w"], P[p + ".ff2.b"], R=t2, out=z[2])
                if l == nd - 1:                                  # the last layer's norms have no consumer stage: one row kernel
                    ops.splitk_finish(z[2].view(-1), 1, B * Q, D, None, ln1=ln3, ln1_out=tgt, ln2=(P["dec.norm.g"], P["dec.norm.b"]), ln2_out=hs[l])
                continue
            if l > 0 or dec_early is None:
                self._decoder_query_side(ws, l, B)
            ops.attention_wide(dq4, mempos3, mem3, dp4, scale=ca_scale, key_mask=fus_mask,
                               n_split=n_split, part_o=ws["part_o"], part_ml=ws["part_ml"])
            skinny(dpool, p + ".ca.vo", R=t1, ln1=ln2, ln1_out=t2)
            skinny(t2, p + ".ff1", act=ops.ACT_RELU, out=ws["dffn"])
            skinny(ws["dffn"], p + ".ff2", R=t2, ln1=ln3, ln1_out=tgt, ln2=(P["dec.norm.g"], P["dec.norm.b"]), ln2_out=hs[l])
        out["hs"] = hs.view(nd, B, Q, D)

        # ---- heads (K11) on all decoder layers at once.  (Running the class head and the query projection on the side stream
        # beside the span head was tried: +150 us per step under graph replay, so the three chains stay on one stream.)
        hs2 = hs.view(nd * B * Q, D)
        logits, spans = ws["logits"], ws["spans"]
        pq = vid_sum = None
        ops.linear(hs2, P["class_embed.w"], P["class_embed.b"], out=logits.view(-1, 2))
        if c.contrastive_align_loss:
            ops.linear(hs2, P["proj_q.w"], P["proj_q.b"], out=ws["pq_raw"])
            pq = ws["pq"]
            ops.l2norm_rows(ws["pq_raw"], out_f32=pq.view(nd * B * Q, -1))
            if c.audio_short_cut:                                    # reference model/model_Uni.py:144-145: normalize(pq + music)
                from . import ops_train
                mq = music if Q == 1 else music[:, None, :].expand(B, Q, D).contiguous()
                ops_train.add3(ws["pq_raw"], pq, mq, b_mod=B * Q * D)
                ops.l2norm_rows(ws["pq_raw"], out_f32=pq.view(nd * B * Q, -1))
                if c.aux_loss and nd > 1:
                    # the auxiliary layers get the short-cut a SECOND time when their output dicts are built (reference
                    # model/model_Uni.py:166-169 adds the music vector to the already short-cut proj_queries[:-1])
                    n_aux = (nd - 1) * B * Q
                    ops_train.add3(ws["pq_raw"][:n_aux], pq.view(nd * B * Q, -1)[:n_aux], mq, b_mod=B * Q * D)
                    ops.l2norm_rows(ws["pq_raw"][:n_aux], out_f32=pq.view(nd * B * Q, -1)[:n_aux])
        if hs2.shape[0] <= 1024:
            skinny(hs2, "span_embed.0", act=ops.ACT_RELU, out=ws["h1"])
            skinny(ws["h1"], "span_embed.1", act=ops.ACT_RELU, out=ws["h2"])
            h2 = ws["h2"]
        else:
            h1 = ops.linear(hs2, P["span_embed.0.w"], P["span_embed.0.b"], act=ops.ACT_RELU, out=ws["h1"])
            h2 = ops.linear(h1, P["span_embed.1.w"], P["span_embed.1.b"], act=ops.ACT_RELU, out=ws["h2"])
        if c.predict_center == 1:
            # the head predicts the centre only; the width is the video's share of the longest track (reference
            # model/model_Uni.py:135-136,280-282)
            ops.linear(h2, P["span_embed.2.w"], P["span_embed.2.b"], act=ops.ACT_SIGMOID, segs=[Seg(out=spans.view(-1, 2), ldo=2)])
            spans[..., 1] = (v_duration.to(self.device, torch.float32) / c.max_m_duration).view(1, B, 1)
        else:
            ops.linear(h2, P["span_embed.2.w"], P["span_embed.2.b"], act=ops.ACT_SIGMOID, out=spans.view(-1, 2))
        out.update(pred_logits=logits[-1], pred_spans=spans[-1], logits_all=logits, spans_all=spans)
        if c.contrastive_align_loss:
            self._frame_rows_linear(frame, P["proj_v.w"], P["proj_v.b"], ws["pv_raw"], B, Tv)
            pv = ws["pv"]
            ops.l2norm_rows(ws["pv_raw"], out_f32=pv.view(B * Tv, -1))
            vid_sum = ops.masked_mean(pv, None, out=ws["vid_sum"])
            out.update(proj_queries=pq[-1], proj_vid_mem=pv, proj_queries_all=pq)

        if c.moment_loss:                                                # reference model_Uni.py:152-159 (outputs only: no loss reads them)
            last = hs[nd - 1]
            m1 = ops.linear(last, P["moment_embed.0.w"], P["moment_embed.0.b"], act=ops.ACT_RELU)
            m2 = ops.linear(m1, P["moment_embed.1.w"], P["moment_embed.1.b"], act=ops.ACT_RELU)
            m3 = ops.linear(m2, P["moment_embed.2.w"], P["moment_embed.2.b"], out_dtype=torch.float32)
            mf = ops.l2norm_rows(m3)
            if c.audio_short_cut:
                from . import ops_train
                mq = music if Q == 1 else music[:, None, :].expand(B, Q, D).contiguous()
                ops_train.add3(m3, mf, mq, b_mod=B * Q * D)
                mf = ops.l2norm_rows(m3)
            out["moment_feats"] = mf.view(B, Q, D)
        if not with_losses:
            cur.wait_stream(side)
            return out
        # ---- matcher + set criterion (K12-K14), all layers in one launch each
        tg = spans_target.contiguous()
        pi, ti, cnt, status, cost = ops.hungarian_match(logits.view(nd * B, Q, 2), spans.view(nd * B, Q, 2), tg, c.foreground_label)
        losses, total = ops.set_criterion(logits, spans, tg, pi, ti, cnt, pq, vid_sum, P["empty_weight"],
                                          c.foreground_label, P["crit_weights"])
        out.update(matcher_pred_idx=pi.view(nd, B, -1), matcher_tgt_idx=ti.view(nd, B, -1), matcher_count=cnt.view(nd, B),
                   matcher_status=status, criterion_losses=losses, localization_loss=total)
        cur.wait_stream(side)
        return out

    def _skinny(self, ws, A: Tensor, wkey: str, **kw):
        """Linear on the B*Q decoder rows: K split over workgroups so the launch fills the chip."""
        P, dws = self.P, ws["dws"]
        W = P[wkey + ".w"]
        N, K = W.shape
        slab = 64 if self.tc == torch.bfloat16 else 32
        tiles = ((A.shape[0] + 127) // 128) * ((N + 127) // 128)
        split = max(2, min(192 // max(tiles, 1), (K + slab - 1) // slab, 32, dws.numel() // (A.shape[0] * N)))
        ops.linear_splitk(A, W, P[wkey + ".b"], dws, split, **kw)

    def _fused_decoder(self) -> bool:
        """The fused decoder chain (made_dec_stage) covers the scripts' configuration: bf16, one moment query, queries from the clip
        vectors (or zeros).  Everything else (f32 parity mode, Q > 1, the pooled-track query) keeps the split-K chain."""
        c = self.cfg
        # (made_dec_stage normalises whole rows of width D in its prologue: D = 256 or 512 only; other widths keep the split-K chain)
        return (self.tc == torch.bfloat16 and c.num_moment_queries == 1 and c.moment_query_type in ("video", "music", "zero", "random")
                and c.D in (256, 512) and not c.detr_pre_norm and not getattr(self, "force_unfused_decoder", False))

    def _dec_first_rows(self, ws, video: Tensor, music: Tensor) -> Tensor:
        """Raw rows feeding decoder layer 0 (reference transformer.py:73-74, model_Uni.py:216-221): the clip vectors, read in place."""
        t = self.cfg.moment_query_type
        return video if t == "video" else (music if t == "music" else ws["dzero"])

    def _dec_fused_query_side(self, ws, l: int, first_rows: Optional[Tensor]):
        """Fused chain, layer l up to the cross-attention.  Stage 1: the previous layer's norm 3 (+ the decoder output norm -> hs[l - 1]) in
        the prologue, the folded self-attention Linear + residual -> raw rows z[0].  Stage 2: norm 1 in the prologue (-> t1), + query_pos,
        the cross-attention query folded with W_k per head -> ws["dq_all"]."""
        P = self.P
        p = f"detr_transformer.decoder.layers.{l}"
        z = ws["dz"]
        if l == 0:
            ops.dec_stage(first_rows, P[p + ".sa.fold.w"], P[p + ".sa.fold.b"], z[0], res_from_x=True)
        else:
            q = f"detr_transformer.decoder.layers.{l - 1}"
            ops.dec_stage(z[2], P[p + ".sa.fold.w"], P[p + ".sa.fold.b"], z[0], ln=(P[q + ".ln3.g"], P[q + ".ln3.b"]),
                          ln2=(P["dec.norm.g"], P["dec.norm.b"]), x2_out=ws["hs"][l - 1], res_from_x=True)
        ops.dec_stage(z[0], P[p + ".ca.qk.w"], P[p + ".ca.qk.b"], ws["dq_all"], ln=(P[p + ".ln1.g"], P[p + ".ln1.b"]),
                      add=P["query_embed"], x_out=ws["t1"])

    def _decoder_queries(self, ws, video: Tensor, music: Tensor, pooled: Optional[Tensor], B: int):
        """Initial decoder queries (reference transformer.py:73-74, model_Uni.py:216-223) -> ws["tgt"]."""
        c = self.cfg
        Q, D = c.num_moment_queries, c.D
        tgt = ws["tgt"]
        if c.moment_query_type in ("video", "music"):
            src_vec = video if c.moment_query_type == "video" else music
            tgt.view(B, Q, D).copy_(src_vec[:, None, :].expand(B, Q, D))
        elif c.moment_query_type == "xpool":                             # the track's pooled vectors, averaged over the batch's videos
            src_vec = ops.masked_mean(pooled.view(B, B, D), torch.ones(B, B, device=self.device))
            tgt.view(B, Q, D).copy_(src_vec[:, None, :].expand(B, Q, D))
        else:                                                            # "zero" / "random"
            tgt.zero_()

    def _decoder_query_side(self, ws, l: int, B: int):
        """Decoder layer l up to the cross-attention: self-attention block over the queries (one folded Linear when Q = 1) ->
        ws["t1"], then the cross-attention query folded with W_k per head -> ws["dq_all"].  Reads ws["tgt"] only."""
        c, P = self.cfg, self.P
        Q, D, H = c.num_moment_queries, c.D, c.detr_nheads
        p = f"detr_transformer.decoder.layers.{l}"
        ln1 = (P[p + ".ln1.g"], P[p + ".ln1.b"])
        tgt, t1, qp = ws["tgt"], ws["t1"], P["query_embed"]
        if Q == 1:
            self._skinny(ws, tgt, p + ".sa.fold", R=tgt, ln1=ln1, ln1_out=t1)
        else:
            dqkv = ws["dqkv"]
            ops.linear(tgt, P[p + ".sa.in.w"], P[p + ".sa.in.b"], A2=qp, a2_row_mod=Q,
                       segs=[Seg(out=dqkv, col_begin=0, use_a2=True),
                             Seg(out=dqkv[:, 2 * D:], col_begin=2 * D, ldo=dqkv.stride(0))])
            d3 = dqkv.view(B, Q, 3 * D)
            ops.attention(d3[:, :, :D], d3[:, :, D:2 * D], d3[:, :, 2 * D:], ws["datt"].view(B, Q, D), H)
            self._skinny(ws, ws["datt"], p + ".sa.out", R=tgt, ln1=ln1, ln1_out=t1)
        self._skinny(ws, t1, p + ".ca.qk", A2=qp, a2_row_mod=Q, out=ws["dq_all"])

    def _regression_head(self, out, ws, mem3: Tensor, fus_mask: Tensor, spans_target: Tensor, v_duration, with_losses: bool, cur, side):
        """reference model/model_Uni.py:228-232,290-300: memory summed over all L positions / number of valid ones -> 3-layer
        ReLU MLP -> sigmoid; loss = 20 * L1.  The decoder's output is not used on this path, so it is not run."""
        c, P = self.cfg, self.P
        B = mem3.shape[0]
        fusion = ops.masked_mean(mem3, None) / fus_mask.sum(dim=1, keepdim=True)           # [B, D] f32
        h1 = ops.linear(fusion, P["reg_mlp.0.w"], P["reg_mlp.0.b"], act=ops.ACT_RELU)
        h2 = ops.linear(h1, P["reg_mlp.1.w"], P["reg_mlp.1.b"], act=ops.ACT_RELU)
        if c.predict_center == 1:
            spans = torch.empty(B, 2, device=self.device, dtype=torch.float32)
            ops.linear(h2, P["reg_mlp.2.w"], P["reg_mlp.2.b"], act=ops.ACT_SIGMOID, segs=[Seg(out=spans, ldo=2)])
            spans[:, 1] = v_duration.to(self.device, torch.float32) / c.max_m_duration
        else:
            spans = ops.linear(h2, P["reg_mlp.2.w"], P["reg_mlp.2.b"], act=ops.ACT_SIGMOID, out_dtype=torch.float32)
        out["pred_spans"] = spans.view(B, 1, 2)
        if with_losses:
            tg = spans_target.to(torch.float32)
            assert tg.shape == out["pred_spans"].shape, f"spans_target.shape {tuple(tg.shape)} must equal to src_spans.shape {tuple(out['pred_spans'].shape)}"
            l1 = (out["pred_spans"] - tg).abs().mean()
            out["regression_loss_span"] = l1
            out["localization_loss"] = (l1 * 20).view(1)
        cur.wait_stream(side)
        return out

    def _ca_fusion(self, ws, frame: Tensor, seg: Tensor, fm: Tensor, sm: Tensor, B: int, Tv: int, Ta: int) -> None:
        """reference model/model_Base.py:194-213 + :130-167 + :22-45 (depth 1) followed by the masked_fill of
        model/model_Uni.py:211: pre-LN cross-attention (query = segments, context = frames, 8 heads x 128, bias-free q/kv,
        kv-mask before the softmax and q-mask after it), residual, pre-LN GELU FFN with residual, final Linear -> ws["fus"]."""
        c, P = self.cfg, self.P
        D, Hc = c.D, c.ca_heads
        inner = Hc * c.ca_dim_head
        x = seg.reshape(B * Ta, D)
        nx = ops.layernorm(x, P["ca.lnq.g"], P["ca.lnq.b"], out=ws["ca_nx"])
        nc = ops.layernorm(frame.reshape(B * Tv, D), P["ca.lnc.g"], P["ca.lnc.b"], out=ws["ca_nc"])
        q = ops.linear(nx, P["ca.q.w"], None, out=ws["ca_q"])
        kv = ops.linear(nc, P["ca.kv.w"], None, out=ws["ca_kv"])
        kv3 = kv.view(B, Tv, 2 * inner)
        ops.attention(q.view(B, Ta, inner), kv3[:, :, :inner], kv3[:, :, inner:], ws["ca_att"].view(B, Ta, inner), Hc,
                      key_mask=fm, q_mask=sm, scale=c.ca_dim_head ** -0.5)
        ax = ops.linear(ws["ca_att"], P["ca.out.w"], P["ca.out.b"], R=x, out=ws["ca_x"])
        nf = ops.layernorm(ax, P["ca.lnf.g"], P["ca.lnf.b"], out=ws["ca_nx"])
        h = ops.linear(nf, P["ca.ff1.w"], P["ca.ff1.b"], act=ops.ACT_GELU, out=ws["ca_h"])
        y = ops.linear(h, P["ca.ff2.w"], P["ca.ff2.b"], R=ax, out=ws["ca_y"])
        ops.linear(y, P["ca.final.w"], P["ca.final.b"], out_row_mask=sm.reshape(-1), out=ws["fus"].view(B * Ta, D))

    def _retrieval_loss(self, ws: Dict[str, Tensor], video: Tensor, music: Tensor, row_exclude: Optional[Tensor] = None,
                        pooled: Optional[Tensor] = None) -> None:
        """reference model/model_Uni.py:236-275 -> ws["ret_loss"]"""
        c, P = self.cfg, self.P
        rl = ws["ret_loss"]
        ls = P["logit_scale"]
        wgt = float(c.dual_single_loss_weight)
        if c.vmr_loss == "dual":
            ops.clip_loss(ws["sims_dual"], ls, rl, weight=wgt)
        elif c.vmr_loss == "single":
            ops.clip_loss(ws["sims_single"], ls, rl, weight=wgt)
        elif c.vmr_loss == "dual_single_loss_fuse":
            # row_exclude: training with --ignore_same_music 0 drops the same-track negatives of the dual loss's video -> music
            # direction (reference modules/loss.py:90-114; model_Uni.py:255)
            ops.clip_loss(ws["sims_dual"], ls, rl, weight=1.0, row_exclude=row_exclude)
            ops.clip_loss(ws["sims_single"], ls, rl, weight=1.0, accumulate=True)
        elif c.vmr_loss == "dual_single_feature_fuse":
            # reference model_Uni.py:268-273: the pooled track vectors averaged with the track's own vector, then the music-pooling
            # similarity (metrics.py:10-24) -- cos(v_n, (pooled[m, n] + music[m]) / 2); the 1/2 cancels in the cosine
            from . import ops_train
            B, D = video.shape
            fused = torch.empty(B * B, D, device=self.device, dtype=torch.float32)
            ops_train.add3(fused, pooled, music[:, None, :].expand(B, B, D).contiguous())        # rows (m, n): + music[m]
            fn = ops.l2norm_rows(fused)
            vn = ops.l2norm_rows(video)
            sims = torch.empty(B, B, device=self.device, dtype=torch.float32)
            # sims[n, m] = <fn[m, n, :], vn[n, :]>: one row-vector product per video, batched over the videos
            ops.linear(fn.view(B, B * D)[:, :D], vn, None, M=B, N=1, K=D, batch=B, a_z_stride=D, w_z_stride=D,     # rows m of video 0, stride B*D
                       segs=[Seg(out=sims, ldo=1, out_z_stride=B)])
            ws["sims_fused"], ws["ff_fused"], ws["ff_fn"], ws["ff_vn"] = sims, fused, fn, vn      # (the training path's backward reads them)
            ops.clip_loss(sims, ls, rl, weight=wgt)
        else:                                                            # dual_single_sim_fuse
            both = self.dual_sims(video, music, add=ws["sims_single"])
            ops.clip_loss(both, ls, rl, weight=wgt)

    def _frame_rows_linear(self, frame: Tensor, w: Tensor, b: Tensor, out: Tensor, B: int, Tv: int) -> None:
        """Linear over the frame rows of `fus` ([B, Tv, D] view with batch stride L*D): one launch per batch
        row block would waste launches, so express the view as A with rows_per_batch addressing on the OUTPUT
        side only when possible; here rows are gathered by a batched launch (grid.z = B)."""
        D = frame.shape[2]
        ops.linear(frame[0], w, b, M=Tv, batch=B, a_z_stride=frame.stride(0), w_z_stride=0,
                   segs=[Seg(out=out, ldo=out.stride(0), out_z_stride=Tv * out.stride(0))])

    # ------------------------------------------------------------------ conveniences
    def loss_dict(self, out: Dict[str, Tensor]) -> Dict[str, Tensor]:
        """criterion_losses [dec,5] -> the reference's 30-entry dict (main = last layer, `_i` = layer i)."""
        names = ["loss_span", "loss_giou", "loss_label", "class_error", "loss_contrastive_align"]
        nd = self.cfg.detr_dec_layers
        if "criterion_losses" not in out:                     # regression variant (reference model/model_Uni.py:296-300)
            return {"loss_span": out["regression_loss_span"], "loss_giou": 0, "loss_label": 0, "class_error": 0}
        L = out["criterion_losses"]
        d = {}
        for l in range(nd):
            suffix = "" if l == nd - 1 else f"_{l}"
            for k, n in enumerate(names):
                if n == "loss_contrastive_align" and not self.cfg.contrastive_align_loss:
                    continue
                if n == "loss_span" and not self.cfg.l1_loss:
                    continue
                if suffix and not self.cfg.aux_loss:
                    continue
                d[n + suffix] = L[l, k]
        return d

    def forward_numpy(self, inp: dict, with_losses: bool = True, want_pooled: bool = True) -> dict:
        """Host-facing helper for tests/smoke: numpy in, numpy out (synchronises)."""
        dev = self.device
        t = {k: torch.from_numpy(np.asarray(v)).to(dev) for k, v in inp.items() if isinstance(v, np.ndarray)}
        o = self.forward(t["frame_feats"], t["segment_feats"], t["frame_masks"], t["segment_masks"], t["spans_target"],
                         with_losses=with_losses, want_pooled=want_pooled, v_duration=t.get("v_duration"))
        torch.cuda.synchronize()
        r = {k: v.float().cpu().numpy() for k, v in o.items() if isinstance(v, torch.Tensor) and v.dtype in (torch.float32, torch.bfloat16)}
        if with_losses and "regression" in self.cfg.mml_localization:
            r["loss_dict"] = {"loss_span": float(o["regression_loss_span"].cpu()), "loss_giou": 0, "loss_label": 0, "class_error": 0}
            r["retrieval_loss"] = float(o["retrieval_loss"].cpu())
            r["localization_loss"] = float(o["localization_loss"].cpu())
        elif with_losses:
            if int(o["matcher_status"].cpu()) != 0:
                raise ValueError("matrix contains invalid numeric entries")   # what SciPy raises in the reference
            nd = self.cfg.detr_dec_layers
            cnt = o["matcher_count"].cpu().numpy()
            pi, ti = o["matcher_pred_idx"].cpu().numpy(), o["matcher_tgt_idx"].cpu().numpy()
            r["matcher_indices"] = [(pi[nd - 1, b, :cnt[nd - 1, b]], ti[nd - 1, b, :cnt[nd - 1, b]]) for b in range(pi.shape[1])]
            r["matcher_indices_all"] = [[(pi[l, b, :cnt[l, b]], ti[l, b, :cnt[l, b]]) for b in range(pi.shape[1])] for l in range(nd)]
            r["loss_dict"] = {k: float(v.cpu()) for k, v in self.loss_dict(o).items()}
            r["retrieval_loss"] = float(o["retrieval_loss"].cpu())
            r["localization_loss"] = float(o["localization_loss"].cpu())
        return r

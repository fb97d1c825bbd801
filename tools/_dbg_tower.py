import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mgsv_amd import synth
from mgsv_amd.config import cfg_native
from mgsv_amd.trainer import MadeTrainer
from oracle import made_oracle as O
cfg = cfg_native(); cfg.vmr_fusion = "XA-video-music"; cfg.vmr_loss = "single"
sd = synth.make_state_dict(cfg, seed=0); inp = synth.make_inputs(cfg, 3, 20, 40, seed=1)
trn = MadeTrainer(cfg, sd, dtype="f32"); trn.training_dropout = False
dev = trn.device
t = {k: torch.from_numpy(np.asarray(v)).to(dev) for k, v in inp.items() if isinstance(v, np.ndarray)}
o = trn.forward_train(t["frame_feats"], t["segment_feats"], t["frame_masks"], t["segment_masks"], t["spans_target"], seed=1234, v_duration=t.get("v_duration"))
torch.cuda.synchronize()
P = O.to_torch_params(sd)
r = O.forward(P, cfg, inp["frame_feats"], inp["segment_feats"], inp["frame_masks"], inp["segment_masks"], inp["spans_target"], v_duration=inp["v_duration"])
print("loss", float(o["retrieval_loss"]), float(r["retrieval_loss"]))
vp = o["sims_video_pooling"].cpu().numpy(); ss = o["sims_single"].cpu().numpy()
print("vp ours\n", vp, "\noracle\n", r["sims_video_pooling"].detach().numpy())
print("single ours\n", ss, "\noracle music-only\n", r["sims_single"].detach().numpy())
import torch.nn.functional as F
tw = trn._train_buffers(3, 20, 40)
xa = "music_guided_to_video_pooling_cross_transformer"
music = r["music_feats"]; frame = r["frame_feats"]
g1, b1 = P[xa + ".layer_norm1.weight"], P[xa + ".layer_norm1.bias"]
v1 = F.layer_norm(music, (256,), g1, b1)
print("yv1", float((tw["yv1"].cpu() - v1).abs().max()))
q = v1 @ P[xa + ".cross_attn.q_proj.weight"].t() + P[xa + ".cross_attn.q_proj.bias"]
print("yq", float((tw["yq"].cpu() - q).abs().max()))
s1 = F.layer_norm(frame, (256,), g1, b1)
fmk = torch.from_numpy(inp["frame_masks"]) != 0
print("ys1", float(((tw["ys1"].cpu().view(3, 20, 256) - s1) * fmk[..., None]).abs().max()))
k = s1 @ P[xa + ".cross_attn.k_proj.weight"].t() + P[xa + ".cross_attn.k_proj.bias"]
print("yk", float(((tw["yk"].cpu().view(3, 20, 256) - k) * fmk[..., None]).abs().max()))
print("music ours vs oracle", float((o["music_feats"].cpu() - music).abs().max()), "frame", float(((o["frame_feats"].cpu() - frame) * fmk[..., None]).abs().max()))
u = s1 @ P[xa + ".cross_attn.v_proj.weight"].t() + P[xa + ".cross_attn.v_proj.bias"]
print("yu", float(((tw["yu"].cpu().view(3, 20, 256) - u) * fmk[..., None]).abs().max()))
import math
logits = torch.einsum("nd,msd->mns", q, k) / math.sqrt(256)
logits = logits.masked_fill((~fmk)[:, None, :], float("-inf"))
a_ = torch.softmax(logits, -1); oo = torch.einsum("mns,msd->mnd", a_, u)
print("yo", float((tw["yo"].cpu().view(3, 3, 256) - oo).abs().max()))
o2 = oo @ P[xa + ".cross_attn.out_proj.weight"].t() + P[xa + ".cross_attn.out_proj.bias"]
print("ya2", float((tw["ya2"].cpu().view(3, 3, 256) - o2).abs().max()))
o3 = F.layer_norm(o2, (256,), P[xa + ".layer_norm2.weight"], P[xa + ".layer_norm2.bias"])
print("ya3", float((tw["ya3"].cpu().view(3, 3, 256) - o3).abs().max()))
yy = o3 + o3 @ P[xa + ".linear_proj.weight"].t() + P[xa + ".linear_proj.bias"]
print("yy", float((tw["yy"].cpu().view(3, 3, 256) - yy).abs().max()))
z = F.layer_norm(yy, (256,), P[xa + ".layer_norm3.weight"], P[xa + ".layer_norm3.bias"])
mh = music / music.norm(dim=-1, keepdim=True)
sv = torch.einsum("nmd,md->nm", z / z.norm(dim=-1, keepdim=True), mh)
print("manual vp\n", sv.detach().numpy())

"""Forward of the training step up to the decoder's first cross-attention, replayed from the launch tape with alternating seeds: the forked
query side of decoder layer 0 against a serial replay; then with main-stream launches behind the fork left out, one kind at a time.
usage: python tools/race_probe5.py [iterations]"""
import os, sys
os.environ["MADE_TAPE_INTERLEAVE"] = "0"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mgsv_amd import synth
from mgsv_amd.config import cfg_headline
from mgsv_amd.trainer import MadeTrainer
dev = torch.device("cuda", 0)
cfg = cfg_headline()
B, Tv, Ta = 64, cfg.max_v_frames, cfg.max_snippet_num
sd = synth.make_state_dict(cfg, seed=0)
inp = synth.make_inputs(cfg, B, Tv, Ta, seed=1)
trn = MadeTrainer(cfg, sd, device=dev, dtype="bf16")
t = {k: torch.from_numpy(v).to(dev) for k, v in inp.items() if isinstance(v, np.ndarray)}
b = (t["frame_feats"], t["segment_feats"], t["frame_masks"], t["segment_masks"], t["spans_target"])
g = trn.capture_train_step(*b, mode="tape")
g.step(*b, seed=3, lrs=(0.0, 0.0, 0.0)); torch.cuda.synchronize()
ops = g.tape.ops()
main = max(set(o[2] for o in ops), key=lambda s: sum(1 for o in ops if o[2] == s))
join = next(i for i, o in enumerate(ops) if o[0] == 0 and o[2] == main and o[3] == (1, 64, 4))
first = next(i for i, o in enumerate(ops) if o[0] == 0 and o[2] != main and o[3] == (32, 4, 1))
print(f"{join} operations up to the first cross-attention of the decoder; the forked chain starts at {first - 1}")
tw = trn._train_buffers(B, Tv, Ta)
names = ["d.0.att", "d.0.t_a", "d.0.t1", "d.0.t1q", "d.0.qc"]
import ctypes as C
from mgsv_amd import _lib
zdump = torch.zeros(32, 64, 512, device=dev, dtype=torch.bfloat16)
_lib.lib().made_debug_zdump(C.c_void_p(zdump.data_ptr()))
ref = {}
for s in (7, 8):
    g.seed_dev.fill_(s); torch.cuda.synchronize()
    for i in range(join):
        g.tape.replay_range(i, 1); torch.cuda.synchronize()
    ref[s] = {k: tw[k].clone() for k in names}
N = int(sys.argv[1]) if len(sys.argv) > 1 else 150


def stress(keep, tag):
    bad = {k: 0 for k in names}
    gaps, badgaps = [], []
    # maximal runs of consecutive kept operations
    runs, i = [], 0
    while i < join:
        if keep[i]:
            j = i
            while j < join and keep[j]: j += 1
            runs.append((i, j - i)); i = j
        else:
            i += 1
    for it in range(N):
        s = 7 + (it & 1)
        g.seed_dev.fill_(s)
        for a_, n_ in runs: g.tape.replay_range(a_, n_)
        torch.cuda.synchronize()
        import ctypes as C
        from mgsv_amd import _lib
        L = _lib.lib()._lib if hasattr(_lib.lib(), "_lib") else _lib.lib()
        e_, b_ = C.c_ulonglong(0), C.c_ulonglong(0)
        L.made_debug_t16(C.byref(e_), 1); L.made_debug_ds(C.byref(b_), 1)
        gap = (int(b_.value) - int(e_.value)) / 100.0           # us between the producer's last store and the consumer's first wave
        gaps.append(gap)
        for k in names:
            if not torch.equal(tw[k], ref[s][k]):
                if k == "d.0.qc":
                    badgaps.append(gap)
                    if len(badgaps) <= 6:
                        # which K quarters (the four waves' partial tiles) does a wrong element consist of?
                        P_ = trn.P; p_ = "detr_transformer.decoder.layers.0"
                        A_ = ref[s]["d.0.t1q"].float(); W_ = P_[p_ + ".ca.in.w"][:512].float(); bias_ = P_[p_ + ".ca.in.b"][:512].float()
                        parts = torch.stack([A_[:, q * 128:(q + 1) * 128] @ W_[:, q * 128:(q + 1) * 128].t() for q in range(4)])    # [4, 64, 512]
                        dq = (tw[k].float() - ref[s][k].float())
                        bi = (dq.abs() > 0).nonzero()[:6]
                        for r_, c_ in bi.tolist():
                            got, want = float(tw[k][r_, c_]), float(ref[s][k][r_, c_])
                            ps = [float(parts[q, r_, c_]) for q in range(4)]
                            print(f"      qc[{r_},{c_}] = {got:.5f}, expected {want:.5f}; bias {float(bias_[c_]):.5f}; K-quarter partial sums {['%.5f' % x for x in ps]} (sum + bias {sum(ps) + float(bias_[c_]):.5f})", flush=True)
                        ta_new, ta_old = ref[s]["d.0.t_a"], ref[15 - s]["d.0.t_a"]
                        dz = (zdump.float() - ta_new.float()[None]).abs() > 0            # [32 workgroups, 64 rows, 512]
                        idx = dz.nonzero()
                        same_old = int((dz & (zdump == ta_old[None])).sum())
                        wg_rows = sorted(set((int(a_), int(b_)) for a_, b_ in idx[:, :2].tolist()))
                        print(f"   [{tag}] iteration {it}: rows as READ by the consumer's workgroups differ from t_a in {int(dz.sum())} elements; {same_old} of them equal the OTHER seed's t_a;"
                              f" (column-tile workgroup, row): {wg_rows[:10]}; columns {sorted(set(idx[:, 2].tolist()))[:4]}..", flush=True)
                bad[k] += 1
                if sum(bad.values()) <= 2:
                    d = (tw[k].float() - ref[s][k].float()).abs(); o_ = (tw[k].float() - ref[15 - s][k].float()).abs()
                    print(f"   [{tag}] iteration {it}: {k}: {int((d > 0).sum())} elements differ, {int(((d > 0) & (o_ == 0)).sum())} of them hold the OTHER seed's value", flush=True)
    print(f"{tag}: mismatching replays of {N}: {bad}", flush=True)
    print(f"     consumer's first wave minus producer's last store (us): min {min(gaps):.2f} median {sorted(gaps)[len(gaps) // 2]:.2f}; on the mismatching replays: {sorted(badgaps)[:12]}", flush=True)
    return sum(bad.values())


all_on = [True] * join
stress(all_on, "everything")
# main-stream kernels behind the fork, by grid (= by launch)
after = [i for i in range(first - 1, join) if ops[i][0] == 0 and ops[i][2] == main]
print("main-stream launches behind the fork:", [(i, ops[i][3]) for i in after])
none_after = list(all_on)
for i in after: none_after[i] = False
stress(none_after, "no main-stream launch behind the fork")


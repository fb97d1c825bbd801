"""What the bf16 retrieval path's error against the f32 oracle actually is (VERDICT r4 'weak' 1 / next 2): max-abs error of the similarity
matrix for the shapes tests/test_engine_gpu.py checks, rank agreement (R@1 / R@10 / MedianR through made_recall_ranks) on an easy and on a
hard set (near-duplicate tracks, margins below 1e-2), a sampled check at 53 000 x 4 000, and the f32 parity mode timed at that size.
    python tools/retrieval_parity_probe.py [quick]
The numbers this prints are what the tests' bounds are derived from (2x the measured maximum)."""
import os, sys, time
os.environ.setdefault("MADE_DEBUG_VARIANTS", "1")          # (measurement knobs are honoured only under this switch)
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from mgsv_amd import synth
from mgsv_amd.config import cfg_native, cfg_headline
from mgsv_amd.engine import MadeEngine
from mgsv_amd.utils.util_test import Recall_metrics
from oracle import made_oracle as O

quick = len(sys.argv) > 1 and sys.argv[1] == "quick"
dev = torch.device("cuda", 0)


def hip_sim(eng, ri):
    t = {k: torch.from_numpy(v).to(dev) for k, v in ri.items()}
    s = eng.retrieval_sim_matrix(t["video_embeds"], t["segment_embeds"], t["segment_masks"], t["music_embeds"])
    torch.cuda.synchronize()
    return s


def oracle_sim(P, cfg, ri, rows=None, cols=None):
    v, s, m, mu = ri["video_embeds"], ri["segment_embeds"], ri["segment_masks"], ri["music_embeds"]
    if rows is not None:
        v = v[rows]
    if cols is not None:
        s, m, mu = s[cols], m[cols], mu[cols]
    with torch.no_grad():
        return O.retrieval_sim_matrix(P, cfg, v, s, m, mu)


def main():
    cfg = cfg_native(); sd = synth.make_state_dict(cfg, seed=0); P = O.to_torch_params(sd)
    print("## max |sim_bf16 - sim_oracle_f32| per shape and kernel (D = 256)")
    for sims in ("1", "0"):
        os.environ["MADE_XPOOL_SIMS"] = sims
        eng = MadeEngine(cfg, sd, dtype="bf16")
        for (N_v, N_m, S) in [(300, 37, 96), (257, 5, 40), (640, 12, 130)]:
            ri = synth.make_retrieval_inputs(N_v, N_m, S, cfg.D, seed=7, min_len=3)
            sim = hip_sim(eng, ri).cpu(); ref = oracle_sim(P, cfg, ri)
            err = (sim - ref).abs()
            print(f"MADE_XPOOL_SIMS={sims} {N_v}x{N_m} S={S}: max {float(err.max()):.3e} mean {float(err.mean()):.3e} | sim range [{float(ref.min()):.3f}, {float(ref.max()):.3f}]", flush=True)
        del eng
    os.environ.pop("MADE_XPOOL_SIMS")
    cfg5 = cfg_headline(); sd5 = synth.make_state_dict(cfg5, seed=0); P5 = O.to_torch_params(sd5)
    eng5 = MadeEngine(cfg5, sd5, dtype="bf16")
    for (N_v, N_m, S) in [(300, 9, 200), (257, 5, 40), (512, 6, 512)]:
        ri = synth.make_retrieval_inputs(N_v, N_m, S, cfg5.D, seed=7, min_len=3)
        sim = hip_sim(eng5, ri).cpu(); ref = oracle_sim(P5, cfg5, ri)
        err = (sim - ref).abs()
        print(f"D=512 two-pass attention {N_v}x{N_m} S={S}: max {float(err.max()):.3e} mean {float(err.mean()):.3e}", flush=True)
    del eng5
    eng = MadeEngine(cfg, sd, dtype="bf16")
    # the across-chunks shape of test_retrieval_parity_across_track_chunks_at_scale
    N_v, N_m, S = 4352, 2304, 96
    ri = synth.make_retrieval_inputs(N_v, N_m, S, cfg.D, seed=11, min_len=3)
    sim = hip_sim(eng, ri).cpu()
    cols = sorted(set(list(range(0, 6)) + list(range(1018, 1030)) + list(range(2042, 2054)) + list(range(N_m - 6, N_m)) + list(range(7, N_m, 331))))
    rows = list(range(0, 8)) + list(range(2170, 2182)) + list(range(N_v - 8, N_v))
    ec = (sim[:, cols] - oracle_sim(P, cfg, ri, cols=cols)).abs(); er = (sim[rows] - oracle_sim(P, cfg, ri, rows=rows)).abs()
    print(f"4352x2304 S=96: columns max {float(ec.max()):.3e} mean {float(ec.mean()):.3e}; rows max {float(er.max()):.3e} mean {float(er.mean()):.3e}", flush=True)

    # rank agreement: N samples (video i, track i); plain ranks = position of the diagonal in the sorted row (reference util_test.py:71-80)
    print("## rank agreement, bf16 HIP vs f32 oracle, through Recall_metrics (made_recall_ranks)")
    for name, ri in (("easy", synth.make_ranked_retrieval_inputs(1024, 96, cfg.D, seed=21, hard=False)),
                     ("hard (near-duplicate tracks)", synth.make_ranked_retrieval_inputs(1024, 96, cfg.D, seed=22, hard=True))):
        sim = hip_sim(eng, ri); ref = oracle_sim(P, cfg, ri)
        m_h, ind_h, _ = Recall_metrics(sim); m_o, ind_o, _ = Recall_metrics(ref.numpy())
        err = float((sim.cpu() - ref).abs().max())
        srt = np.sort(ref.numpy(), axis=1)[:, ::-1]
        print(f"{name}: max err {err:.3e}; oracle top1-top2 margin median {float(np.median(srt[:, 0] - srt[:, 1])):.2e}, share below 1e-2: {float(np.mean(srt[:, 0] - srt[:, 1] < 1e-2)):.3f}")
        print(f"   oracle R1 {m_o['R1']:.2f} R10 {m_o['R10']:.2f} MedianR {m_o['MedianR']}; HIP bf16 R1 {m_h['R1']:.2f} R10 {m_h['R10']:.2f} MedianR {m_h['MedianR']}")
        print(f"   per-video agreement: rank==: {np.mean(ind_h == ind_o):.4f}  (rank<1)==: {np.mean((ind_h < 1) == (ind_o < 1)):.4f}  (rank<10)==: {np.mean((ind_h < 10) == (ind_o < 10)):.4f}"
              f"  max |rank diff| {int(np.abs(ind_h - ind_o).max())}", flush=True)
    if quick:
        return
    # the timed size: 53 000 x 4 000, bf16 against the oracle on 24 video rows x all tracks and all videos x 24 track columns
    N_v, N_m, S = 53000, 4000, 96
    ri = synth.make_retrieval_inputs(N_v, N_m, S, cfg.D, seed=31, min_len=12)
    sim = hip_sim(eng, ri).cpu()
    rows = sorted(set(np.linspace(0, N_v - 1, 24).astype(int).tolist())); cols = sorted(set(np.linspace(0, N_m - 1, 24).astype(int).tolist()))
    t0 = time.time(); rr = oracle_sim(P, cfg, ri, rows=rows); t1 = time.time(); rc = oracle_sim(P, cfg, ri, cols=cols); t2 = time.time()
    print(f"53000x4000 bf16: rows max {float((sim[rows] - rr).abs().max()):.3e}; columns max {float((sim[:, cols] - rc).abs().max()):.3e}  (oracle {t1 - t0:.1f} s + {t2 - t1:.1f} s)", flush=True)
    del eng
    torch.cuda.empty_cache()
    eng32 = MadeEngine(cfg, sd, dtype="f32")
    t = {k: torch.from_numpy(v).to(dev) for k, v in ri.items()}
    eng32.retrieval_sim_matrix(t["video_embeds"][:4096], t["segment_embeds"][:256], t["segment_masks"][:256], t["music_embeds"][:256]); torch.cuda.synchronize()
    t0 = time.time()
    s32 = eng32.retrieval_sim_matrix(t["video_embeds"], t["segment_embeds"], t["segment_masks"], t["music_embeds"]); torch.cuda.synchronize()
    dt = time.time() - t0
    s32 = s32.cpu()
    alg = 4.0 * (N_m * S * cfg.D + N_m * S + N_v * cfg.D + N_m * cfg.D + N_v * N_m)
    print(f"53000x4000 f32 parity mode: {dt * 1e3:.1f} ms per pass = {alg / dt / 1e9:.2f} GB/s; rows max {float((s32[rows] - rr).abs().max()):.3e}; columns max {float((s32[:, cols] - rc).abs().max()):.3e}")


if __name__ == "__main__":
    main()

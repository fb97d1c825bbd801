"""CPU, world_size 2 (gloo): the sharding logic of mgsv_amd/retrieval.py -- video rows partitioned, music side
all-gathered once, row blocks equal to the 1-rank matrix.  The compute backend injected here is the oracle
(test infrastructure); the product backend is the HIP engine (tests/test_engine_gpu.py)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from mgsv_amd import synth
from mgsv_amd.config import cfg_native
from mgsv_amd.retrieval import ShardedRetrieval, shard_rows


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, N_v, N_m, S, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import made_oracle as O
    cfg = cfg_native()
    P = O.to_torch_params(synth.make_state_dict(cfg, seed=0))
    ri = {k: torch.from_numpy(v) for k, v in synth.make_retrieval_inputs(N_v, N_m, S, cfg.D, seed=2).items()}

    def score(v, seg, mask, music):
        with torch.no_grad():
            return O.retrieval_sim_matrix(P, cfg, v, seg, mask, music, chunk_v=16)

    vlo, vhi = shard_rows(N_v, world, rank)
    mlo, mhi = shard_rows(N_m, world, rank)          # ragged on purpose (N_m not divisible by world)
    sr = ShardedRetrieval(score)
    rows = sr.sim_rows(ri["video_embeds"][vlo:vhi], ri["segment_embeds"][mlo:mhi], ri["segment_masks"][mlo:mhi],
                       ri["music_embeds"][mlo:mhi])
    # the same with the partition known up front: one packed all-gather, no count exchange, nothing read back to the host
    mcounts = [shard_rows(N_m, world, r)[1] - shard_rows(N_m, world, r)[0] for r in range(world)]
    vcounts = [shard_rows(N_v, world, r)[1] - shard_rows(N_v, world, r)[0] for r in range(world)]
    full = sr.sim_matrix(ri["video_embeds"][vlo:vhi], ri["segment_embeds"][mlo:mhi], ri["segment_masks"][mlo:mhi],
                         ri["music_embeds"][mlo:mhi], gather_rows=True, counts=mcounts, video_counts=vcounts)
    # equal shards (one block straight out of the gathered buffer) with the segments travelling in bf16
    n_eq = (N_m // world) * world
    elo, ehi = shard_rows(n_eq, world, rank)
    sr16 = ShardedRetrieval(lambda v, s, m, mu: score(v, s.float(), m, mu), pack_dtype=torch.bfloat16)
    rows16 = sr16.sim_rows(ri["video_embeds"][vlo:vhi], ri["segment_embeds"][elo:ehi], ri["segment_masks"][elo:ehi], ri["music_embeds"][elo:ehi],
                           counts=[n_eq // world] * world)
    np.save(os.path.join(out_dir, f"rows16_{rank}.npy"), rows16.numpy())
    np.save(os.path.join(out_dir, f"rows{rank}.npy"), rows.numpy())
    np.save(os.path.join(out_dir, f"full{rank}.npy"), full.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_sharded_retrieval_world2_matches_single_rank(tmp_path):
    N_v, N_m, S, world = 37, 11, 24, 2
    mp.spawn(_worker, args=(world, _free_port(), N_v, N_m, S, str(tmp_path)), nprocs=world, join=True)
    from oracle import made_oracle as O
    cfg = cfg_native()
    P = O.to_torch_params(synth.make_state_dict(cfg, seed=0))
    ri = synth.make_retrieval_inputs(N_v, N_m, S, cfg.D, seed=2)
    with torch.no_grad():
        ref = O.retrieval_sim_matrix(P, cfg, ri["video_embeds"], ri["segment_embeds"], ri["segment_masks"], ri["music_embeds"]).numpy()
    for rank in range(world):
        lo, hi = shard_rows(N_v, world, rank)
        rows = np.load(tmp_path / f"rows{rank}.npy")
        assert rows.shape == (hi - lo, N_m)
        np.testing.assert_allclose(rows, ref[lo:hi], atol=1e-6, rtol=0)
        np.testing.assert_allclose(np.load(tmp_path / f"full{rank}.npy"), ref, atol=1e-6, rtol=0)
        n_eq = (N_m // world) * world
        seg16 = torch.from_numpy(ri["segment_embeds"][:n_eq]).bfloat16().float().numpy()
        with torch.no_grad():
            ref16 = O.retrieval_sim_matrix(P, cfg, ri["video_embeds"][lo:hi], seg16, ri["segment_masks"][:n_eq], ri["music_embeds"][:n_eq]).numpy()
        np.testing.assert_allclose(np.load(tmp_path / f"rows16_{rank}.npy"), ref16, atol=1e-6, rtol=0)


def test_shard_rows_partition():
    for n in (0, 1, 7, 53000):
        for w in (1, 2, 3, 8):
            spans = [shard_rows(n, w, r) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
            assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1


def test_single_process_passthrough():
    sr = ShardedRetrieval(lambda v, s, m, mu: v @ mu.t())
    v, mu = torch.randn(5, 8), torch.randn(3, 8)
    out = sr.sim_matrix(v, torch.zeros(3, 2, 8), torch.ones(3, 2), mu)
    assert torch.equal(out, v @ mu.t())

"""GPU, world size 2 over RCCL (backend "nccl"), one process per GPU: the sharded retrieval (videos row-partitioned, the music side in
one packed all-gather) and the data-parallel training step.  The nccl legs need two visible GPUs and skip on a one-GPU box; there the retrieval leg runs with both ranks on cuda:0 over gloo (the
HIP engine scores, the exchange is the same packed all-gather), the training leg in tests/test_train_dp_gpu.py, and the sharding
logic alone on CPU in tests/test_retrieval_sharded_cpu.py.  RCCL itself is UNMEASURED ON HARDWARE: the GPU pool exposes one GPU."""
import os
import socket

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

needs_two = pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs 2 GPUs (RCCL world size 2)")


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _retrieval_worker(rank, world, port, N_v, N_m, S, out_dir, backend="nccl"):
    import torch.distributed as dist
    from mgsv_amd import synth
    from mgsv_amd.config import cfg_native
    from mgsv_amd.engine import MadeEngine
    from mgsv_amd.retrieval import ShardedRetrieval, shard_rows
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    if backend == "nccl":
        torch.cuda.set_device(rank)
        dev = torch.device("cuda", rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    else:                                                    # both ranks on cuda:0, the exchange over gloo (device tensors, staged by gloo)
        dev = torch.device("cuda", 0)
        dist.init_process_group("gloo", rank=rank, world_size=world)
    cfg = cfg_native()
    eng = MadeEngine(cfg, synth.make_state_dict(cfg, seed=0), device=dev, dtype="bf16")
    ri = {k: torch.from_numpy(v).to(dev) for k, v in synth.make_retrieval_inputs(N_v, N_m, S, cfg.D, seed=2).items()}
    vlo, vhi = shard_rows(N_v, world, rank)
    mlo, mhi = shard_rows(N_m, world, rank)
    counts = [shard_rows(N_m, world, r)[1] - shard_rows(N_m, world, r)[0] for r in range(world)]
    sr = ShardedRetrieval(lambda a, b, c, d: eng.retrieval_sim_matrix(a, b, c, d), pack_dtype=torch.bfloat16)
    rows = sr.sim_rows(ri["video_embeds"][vlo:vhi], ri["segment_embeds"][mlo:mhi].bfloat16(), ri["segment_masks"][mlo:mhi],
                       ri["music_embeds"][mlo:mhi], counts=counts)
    torch.cuda.synchronize()
    np.save(os.path.join(out_dir, f"rows{rank}.npy"), rows.float().cpu().numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("backend", ["gloo", pytest.param("nccl", marks=needs_two)])
def test_sharded_retrieval_two_ranks_equal_one(tmp_path, backend):
    """BASELINE configs[3]'s multi-rank leg with the HIP engine as the scorer (MadeEngine.retrieval_sim_matrix): videos row-sharded, the
    music side in one packed all-gather, each rank's row block bit-equal to the same rows of the one-rank matrix.  "gloo": two
    processes share cuda:0 (runs on a one-GPU box); "nccl": one process per GPU over RCCL (needs two GPUs)."""
    import torch.multiprocessing as mp
    from mgsv_amd import synth
    from mgsv_amd.config import cfg_native
    from mgsv_amd.engine import MadeEngine
    from mgsv_amd.retrieval import shard_rows
    N_v, N_m, S = 600, 37, 96                     # N_m not divisible by 2: ragged shards, scored block by block
    mp.spawn(_retrieval_worker, args=(2, _free_port(), N_v, N_m, S, str(tmp_path), backend), nprocs=2, join=True)
    cfg = cfg_native()
    eng = MadeEngine(cfg, synth.make_state_dict(cfg, seed=0), device="cuda:0", dtype="bf16")
    ri = {k: torch.from_numpy(v).cuda() for k, v in synth.make_retrieval_inputs(N_v, N_m, S, cfg.D, seed=2).items()}
    ref = eng.retrieval_sim_matrix(ri["video_embeds"], ri["segment_embeds"].bfloat16(), ri["segment_masks"], ri["music_embeds"]).float().cpu().numpy()
    for rank in range(2):
        lo, hi = shard_rows(N_v, 2, rank)
        rows = np.load(tmp_path / f"rows{rank}.npy")
        assert rows.shape == (hi - lo, N_m)
        # per-pair arithmetic is independent of the sharding: the rows of the 2-rank result ARE the rows of the 1-rank matrix
        np.testing.assert_array_equal(rows, ref[lo:hi])


def _dp_worker(rank, world, port, out_dir):
    import torch.distributed as dist
    from mgsv_amd import synth
    from mgsv_amd.config import cfg_native
    from mgsv_amd.trainer import MadeTrainer
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(rank)
    dev = torch.device("cuda", rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    cfg = cfg_native()
    trn = MadeTrainer(cfg, synth.make_state_dict(cfg, seed=0), device=dev, dtype="f32")
    inp = synth.make_inputs(cfg, 4, 20, 40, seed=1 + rank)
    t = {k: torch.from_numpy(np.asarray(v)).to(dev) for k, v in inp.items() if isinstance(v, np.ndarray)}
    for it in range(2):
        trn.train_step(t["frame_feats"], t["segment_feats"], t["frame_masks"], t["segment_masks"], t["spans_target"], seed=100 + it,
                       lrs=(1e-3, 1e-3, 1e-3), dist=dist)
    torch.cuda.synchronize()
    np.save(os.path.join(out_dir, f"params_{rank}.npy"), trn.flat_param.cpu().numpy())
    dist.barrier()
    dist.destroy_process_group()


@needs_two
def test_data_parallel_training_two_gpus_ranks_stay_identical(tmp_path):
    import torch.multiprocessing as mp
    mp.spawn(_dp_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    p0, p1 = np.load(tmp_path / "params_0.npy"), np.load(tmp_path / "params_1.npy")
    assert np.isfinite(p0).all() and np.array_equal(p0, p1), "ranks diverged"

"""GPU: data-parallel training step with two ranks (one process each, both on cuda:0, gloo rendezvous on 127.0.0.1 -- the
collective is the same torch.distributed.all_reduce of the flat f32 gradient buffer that runs over RCCL on a multi-GPU node).
Both ranks must end with bit-identical parameters, equal to a single process that averages the two batches' gradients."""
import os
import socket

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir, graph=False):
    import torch.distributed as dist
    from mgsv_amd import synth
    from mgsv_amd.config import cfg_native
    from mgsv_amd.trainer import MadeTrainer
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cfg = cfg_native()
    trn = MadeTrainer(cfg, synth.make_state_dict(cfg, seed=0), device="cuda:0", dtype="f32")
    inp = synth.make_inputs(cfg, 4, 20, 40, seed=1 + rank)
    t = {k: torch.from_numpy(np.asarray(v)).cuda() for k, v in inp.items() if isinstance(v, np.ndarray)}
    g = (trn.capture_train_step(t["frame_feats"], t["segment_feats"], t["frame_masks"], t["segment_masks"], t["spans_target"], dist=dist,
                                mode="tape" if graph == "tape" else "graph") if graph else None)
    for it in range(2):
        if g is not None:                                    # the captured iteration: three graphs around the two all-reduces, or the launch
                                                             # tape with the all-reduces as host callbacks
            g.step(t["frame_feats"], t["segment_feats"], t["frame_masks"], t["segment_masks"], t["spans_target"], seed=100 + it,
                   lrs=(1e-3, 1e-3, 1e-3))
            if it == 0:
                np.save(os.path.join(out_dir, f"gsum_{rank}.npy"), trn.flat_grad.cpu().numpy())
        elif it == 0:                                        # the body of train_step, with a look at the reduced gradient
            trn.forward_train(t["frame_feats"], t["segment_feats"], t["frame_masks"], t["segment_masks"], t["spans_target"], seed=100)
            trn.backward()
            dist.all_reduce(trn.flat_grad)
            np.save(os.path.join(out_dir, f"gsum_{rank}.npy"), trn.flat_grad.cpu().numpy())
            trn.optimizer_step(1e-3, 1e-3, 1e-3, grad_scale=0.5)
        else:
            trn.train_step(t["frame_feats"], t["segment_feats"], t["frame_masks"], t["segment_masks"], t["spans_target"], seed=100 + it,
                           lrs=(1e-3, 1e-3, 1e-3), dist=dist)
    torch.cuda.synchronize()
    np.save(os.path.join(out_dir, f"params_{rank}.npy"), trn.flat_param.cpu().numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("graph", [False, True, "tape"])
def test_two_rank_data_parallel_step(tmp_path, graph):
    import torch.multiprocessing as mp
    from mgsv_amd import synth
    from mgsv_amd.config import cfg_native
    from mgsv_amd.trainer import MadeTrainer
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path), graph), nprocs=2, join=True)
    p0, p1 = np.load(tmp_path / "params_0.npy"), np.load(tmp_path / "params_1.npy")
    assert np.array_equal(p0, p1), "ranks diverged"
    # single-process emulation: sum of the two batches' gradients, scaled by 1/2 inside the optimizer
    cfg = cfg_native()
    trn = MadeTrainer(cfg, synth.make_state_dict(cfg, seed=0), device="cuda:0", dtype="f32")
    batches = []
    for r in range(2):
        inp = synth.make_inputs(cfg, 4, 20, 40, seed=1 + r)
        batches.append({k: torch.from_numpy(np.asarray(v)).cuda() for k, v in inp.items() if isinstance(v, np.ndarray)})
    g_dp = np.load(tmp_path / "gsum_0.npy")
    assert np.array_equal(g_dp, np.load(tmp_path / "gsum_1.npy"))
    for it in range(2):
        acc = torch.zeros_like(trn.flat_grad)
        for t in batches:
            trn.forward_train(t["frame_feats"], t["segment_feats"], t["frame_masks"], t["segment_masks"], t["spans_target"], seed=100 + it)
            trn.backward()
            acc += trn.flat_grad
        if it == 0:                                          # the reduced gradient itself: f32 atomics reorder sums, nothing more
            ref_g = acc.cpu().numpy()
            assert np.abs(ref_g - g_dp).max() <= 2e-5 * np.abs(ref_g).max(), np.abs(ref_g - g_dp).max()
        trn.flat_grad.copy_(acc)
        trn.optimizer_step(1e-3, 1e-3, 1e-3, grad_scale=0.5)
    # parameters after two Adam steps: Adam normalises every element's update to ~lr, so an element whose gradient is
    # rounding noise may move by up to lr per step in either direction; everything else agrees far tighter
    ref = trn.flat_param.cpu().numpy()
    diff = np.abs(ref - p0)
    assert diff.max() <= 2.5e-3 and np.mean(diff > 1e-5) < 1e-2, (diff.max(), np.mean(diff > 1e-5))

"""CPU: `python bench.py --gpus N` launches itself.  With N > 1 and no RANK in the environment the parent -- which has
not touched a GPU -- starts N children with the rendezvous environment of torch.distributed.run and relays rank 0's line.  No GPU
here, so (a) MADE_BENCH_DRY_RUN=1 checks launcher + rendezvous end to end over gloo, (b) without it both children must get as far as the
'needs a GPU' assertion and the parent must report their failure with a non-zero exit code, (c) more GPUs than the node has is refused."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra_env, *argv, timeout=300):
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(extra_env)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], env=env, capture_output=True, text=True, timeout=timeout)


def test_bench_gpus2_self_launch_rendezvous_dry_run():
    r = _run({"MADE_BENCH_FAKE_GPUS": "2", "MADE_BENCH_DRY_RUN": "1"}, "--gpus", "2", "--steps", "1", "--warmup", "0")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip() and not ln.startswith("[Gloo]")]   # (gloo announces itself on stdout)
    assert len(lines) == 1, r.stdout                       # ONE JSON line, rank 0's
    d = json.loads(lines[0])
    assert d["dry_run"] is True and d["n_gpus"] == 2 and d["rank_sum"] == 3.0 and d["master_addr"] == "127.0.0.1"


def _one_line(r):
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip() and not ln.startswith("[Gloo]")]
    assert len(lines) == 1, r.stdout
    return json.loads(lines[0])


def test_bench_gpus8_dry_run_every_workload_partitions_over_eight_ranks():
    """The 8-GPU day (SCALE run: `python bench.py --gpus 8 ...`) must not die on plumbing: eight ranks rendezvous over gloo, and each workload's
    partition runs on the host -- retrieval: videos row-sharded (ragged: 8 does not divide the stand-in sizes), the music side through the
    production packed all-gather (mgsv_amd/retrieval.py); train: the gradient buffer all-reduced in the trainer's two buckets."""
    for wl in ("retrieval", "train", "all"):
        d = _one_line(_run({"MADE_BENCH_FAKE_GPUS": "8", "MADE_BENCH_DRY_RUN": "1"}, "--gpus", "8", "--steps", "1", "--warmup", "0", "--workload", wl,
                           timeout=600))
        assert d["dry_run"] is True and d["n_gpus"] == 8 and d["rank_sum"] == 36.0 and d["workload"] == wl
        if wl in ("retrieval", "all"):
            p = d["plan"]["retrieval"]
            assert p["video_rows_all_ranks"] == p["n_v"] and p["music_tracks_gathered"] == p["n_m"] == 301
        if wl in ("train", "all"):
            p = d["plan"]["train"]
            assert p["grad_sum"] == p["expected"] == 36.0 and sum(p["bucket_elems"]) == 1007


def test_bench_gpus2_children_reach_the_gpu_assertion_and_the_parent_reports_it():
    r = _run({"MADE_BENCH_FAKE_GPUS": "2", "CUDA_VISIBLE_DEVICES": "", "HIP_VISIBLE_DEVICES": ""}, "--gpus", "2", "--steps", "1", "--warmup", "0")
    assert r.returncode != 0
    assert "bench.py needs a GPU" in r.stderr and "exited with code" in r.stderr
    assert r.stdout.strip() == ""                          # no line is printed for a run that measured nothing


def test_bench_refuses_more_gpus_than_the_node_has():
    r = _run({"MADE_BENCH_FAKE_GPUS": "1"}, "--gpus", "4")
    assert r.returncode == 2 and "exposes 1 GPU" in r.stderr

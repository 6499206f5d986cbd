"""RCCL on the device: the only multi-GPU evidence obtainable on a 1-GPU box — a
1-rank `nccl` process group launched exactly as the driver launches bench.py, running the depth
all-gather of implicit_depth_amd.dist and the bench's distributed path."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port(hint):
    """A port nobody listens on right now (the hint if it is free): launches that follow each other on one fixed
    port meet the previous store's socket in TIME_WAIT now and then."""
    import socket
    for cand in (hint, 0):
        with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
            try:
                sk.bind(("127.0.0.1", cand))
                return sk.getsockname()[1]
            except OSError:
                continue
    return hint


def _torchrun(script_args, port, timeout=600, nproc=1, extra_env=None):
    env = dict(os.environ)
    env.update(extra_env or {})
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = None
    for attempt in range(2):   # (one more try when the rendezvous itself failed: the port was taken in between)
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc),
               "--master-addr", "127.0.0.1", "--master-port", str(_free_port(port + 40 * attempt))] + script_args
        r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
        err = r.stderr.lower()
        if r.returncode == 0 or not ("address already in use" in err or "eaddrinuse" in err
                                     or "failed to bind" in err or "rendezvous" in err):
            break
    return r


def test_all_gather_depth_under_nccl(cuda):
    r = _torchrun([os.path.join(ROOT, "tests", "rccl_worker.py")], 29611)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "RCCL_WORKER_OK world=1" in r.stdout


def test_bench_distributed_path_one_rank(cuda):
    r = _torchrun([os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
                   "--no-cpu-baseline"], 29612)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout[-2000:]   # ONE JSON line on stdout, whatever RCCL prints
    line = lines[0]
    rec = json.loads(line)
    assert rec["n_gpus"] == 1 and rec["value"] > 0
    assert rec["collective"]["gathered_equals_local"] is True
    assert "RCCL" in rec["config"]["workload"]


def test_bench_ray_row_shards_one_rank(cuda):
    """--shard rays (one frame, image rows split over the ranks, depth rows all-gathered) through
    RCCL with a single rank; --config 2 names the configs[2] shard."""
    r = _torchrun([os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
                   "--no-cpu-baseline", "--shard", "rays"], 29613)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    rec = json.loads([ln for ln in r.stdout.splitlines() if ln.strip()][0])
    assert rec["scaling"] == "strong" and rec["collective"]["gathered_equals_local"] is True
    assert rec["collective"]["ranks_seen"] == 1 and len(rec["collective"]["ms_per_step_by_rank"]) == 1
    r = _torchrun([os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "1",
                   "--no-cpu-baseline", "--config", "2"], 29614)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    rec = json.loads([ln for ln in r.stdout.splitlines() if ln.strip()][0])
    assert "configs[2]" in rec["config"]["workload"] and rec["config"]["points_per_gpu"] == 4 * 240 * 320 * 64


def test_two_ranks_sharing_the_gpu(cuda):
    """world = 2 on the 1-GPU box: both ranks on cuda:0 with gloo as the transport (LIDF_TEST_SHARE_GPU,
    test-only) — the worker's gathers and the bench's N = 2 logic (frame shards and ray-row shards,
    gather slots, ranks_seen, per-rank clocks, whole-job point count) on real device buffers."""
    share = {"LIDF_TEST_SHARE_GPU": "1"}
    r = _torchrun([os.path.join(ROOT, "tests", "rccl_worker.py")], 29616, nproc=2, extra_env=share)
    assert r.returncode == 0 and "RCCL_WORKER_OK world=2" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
    for extra, pts in ([], 2 * 240 * 320 * 64), (["--shard", "rays"], 240 * 320 * 64):
        r = _torchrun([os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                       "--no-cpu-baseline"] + extra, 29617, nproc=2, extra_env=share)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
        lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
        assert len(lines) == 1, r.stdout[-2000:]
        rec = json.loads(lines[0])
        col = rec["collective"]
        assert rec["n_gpus"] == 2 and col["ranks_seen"] == 2 and len(col["ms_per_step_by_rank"]) == 2
        assert col["gathered_equals_local"] is True and col["points_all_ranks"] == pts
        assert abs(rec["value"] - pts / rec["ms_per_step"] / 1e3) <= 0.01 * rec["value"]
        assert rec["ms_per_step"] >= max(col["ms_per_step_by_rank"]) - 1e-3


def test_eight_ranks_sharing_the_gpu(cuda):
    """The shapes of the 8-GPU run that no box of this work could make: world = 8 on
    cuda:0 with gloo as the transport (LIDF_TEST_SHARE_GPU, test-only, never a measurement) — the worker's
    gathers (equal, ragged and row shards, feature map cut to the shard), BASELINE configs[2] (32 frames:
    4 per rank), configs[4] (4 frames x 256 candidates per rank) and one frame split into 8 row shards
    (240 rows: 30 per rank, feature map cut to 38): every rank joins, its slot of the gather holds its own
    maps, ONE line on stdout, the whole-job point count and the max-over-ranks clock are consistent."""
    share = {"LIDF_TEST_SHARE_GPU": "1"}
    r = _torchrun([os.path.join(ROOT, "tests", "rccl_worker.py")], 29621, nproc=8, extra_env=share)
    assert r.returncode == 0 and "RCCL_WORKER_OK world=8" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
    frame = 240 * 320
    for extra, pts, scaling in ((["--config", "2"], 8 * 4 * frame * 64, "weak"),
                                (["--config", "4"], 8 * 4 * frame * 256, "weak"),
                                (["--shard", "rays"], frame * 64, "strong")):
        r = _torchrun([os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "1", "--warmup", "1",
                       "--no-cpu-baseline"] + extra, 29622, nproc=8, extra_env=share, timeout=1200)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
        lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
        assert len(lines) == 1, r.stdout[-2000:]
        rec = json.loads(lines[0])
        col = rec["collective"]
        assert rec["n_gpus"] == 8 and rec["scaling"] == scaling
        assert col["ranks_seen"] == 8 and len(col["ms_per_step_by_rank"]) == 8
        assert col["gathered_equals_local"] is True and col["points_all_ranks"] == pts
        assert abs(rec["value"] - pts / rec["ms_per_step"] / 1e3) <= 0.01 * rec["value"]
        assert rec["ms_per_step"] >= max(col["ms_per_step_by_rank"]) - 1e-3
        if scaling == "strong":
            assert rec["config"]["rays_per_gpu"] == 30 * 320


@pytest.mark.parametrize("world,mode", [(1, "frame"), (2, "frame"), (8, "graph")])
def test_eval_stream_sharded_over_ranks(cuda, world, mode):
    """bench.py --workload e2e --gpus N: the frames of an evaluation stream sharded over the ranks
    (dist.shard_frames), one FrameRunner per rank, the refined depth maps all-gathered inside the timed
    region — world 1 through RCCL, worlds 2 and 8 with all ranks on the one GPU (gloo transport, test-only).
    Rank g runs the batch a single GPU would have met as its g-th: the ranks' scenes differ, so do their
    pair counts, and the whole-job totals are sums."""
    share = {"LIDF_TEST_SHARE_GPU": "1"} if world > 1 else {}
    r = _torchrun([os.path.join(ROOT, "bench.py"), "--workload", "e2e", "--e2e-mode", mode, "--gpus", str(world),
                   "--steps", "2", "--warmup", "1", "--no-rocprof"], 29630 + world, nproc=world, extra_env=share,
                  timeout=1200)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout[-2000:]
    rec = json.loads(lines[0])
    col = rec["collective"]
    assert rec["n_gpus"] == world and rec["scaling"] == "weak"
    assert col["ranks_seen"] == world and len(col["ms_per_step_by_rank"]) == world
    assert col["gathered_equals_local"] is True and col["frames_all_ranks"] == world
    assert col["rays_all_ranks"] == world * 240 * 320
    assert col["pairs_all_ranks"] >= world * 200000 and (world == 1) == (col["pairs_all_ranks"] == rec["config"]["pairs"])
    assert abs(rec["value"] - col["pairs_all_ranks"] / rec["ms_per_step"] / 1e3) <= 0.01 * rec["value"]
    assert rec["ms_per_step"] >= max(col["ms_per_step_by_rank"]) - 1e-3
    assert abs(rec["frames_per_s"] - world / rec["ms_per_step"] * 1e3) <= 0.01 * rec["frames_per_s"]


def _gpus():
    import torch
    return torch.cuda.device_count()


@pytest.mark.skipif("_gpus() < 2")
def test_two_ranks_nccl(cuda):
    """Only where two GPUs are visible (not on the 1-GPU test box): the worker with world = 2, and
    bench.py launching itself (`python bench.py --gpus 2` without a launcher) for both partitions."""
    r = _torchrun([os.path.join(ROOT, "tests", "rccl_worker.py")], 29615, nproc=2)
    assert r.returncode == 0 and "RCCL_WORKER_OK world=2" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    for extra in ([], ["--shard", "rays"]):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2",
                            "--warmup", "1", "--no-cpu-baseline"] + extra, cwd=ROOT, env=env,
                           capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
        rec = json.loads([ln for ln in r.stdout.splitlines() if ln.strip()][0])
        assert rec["n_gpus"] == 2 and rec["collective"]["ranks_seen"] == 2
        assert rec["collective"]["gathered_equals_local"] is True

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


def _torchrun(script_args, port, timeout=600):
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
           "--master-addr", "127.0.0.1", "--master-port", str(port)] + script_args
    return subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)


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

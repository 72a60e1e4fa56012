"""The N > 1 path of bench.py rehearsed on the ONE GPU of the test box: two ranks launched exactly as the driver launches
them (python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 ... bench.py --gpus 2), control plane over gloo, both
ranks on device 0 (RCCL refuses two ranks on one device).  What it covers that the CPU gloo tests cannot: the shard each
rank evaluates on the GPU, max-over-ranks timing, the gather of the trajectories reassembled on rank 0 (bench.py asserts
it), the strong-scaled line and its weak-scaled repeat.  -m gpu"""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_rank_bench_rehearsal_on_one_gpu(built):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, EMI_BENCH_BACKEND="gloo", EMI_BENCH_SHARE_GPU="1", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "10", "--warmup", "2",
           "--scenarios", "256"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{"metric"')]
    assert len(lines) == 1, r.stdout[-2000:]                  # rank 0 prints ONE line
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["scaling"] == "strong" and j["steps"] == 10
    assert j["config"]["scenarios"] == 256 and j["config"]["instances_per_gpu"] == 128
    assert j["config"]["gather_ms"] is not None               # the one collective ran (and bench.py checked rank 0's block)
    assert j["value"] > 0 and abs(j["value"] - 256 * 1024 * 10 / (j["ms_per_step"] * 1e-3 * 10)) < 1e-6 * j["value"]
    assert j["weak_scaling"]["instances_per_gpu"] == 256 and j["weak_scaling"]["value"] > 0
    assert "cpu_baseline" not in j and "secondary" not in j   # rank 0 at N = 1 only
    assert j["config"]["ranks_share_device"] is True and j["config"]["control_plane"] == "gloo"


def test_bench_starts_its_own_ranks_when_launched_like_the_single_gpu_run(built):
    """`python bench.py --gpus 2` with no torchrun environment (the way the driver starts `--gpus 1`): bench.py must start the two
    ranks itself, as child processes, and relay rank 0's single line; with one device visible the ranks share it and say so."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "EMI_BENCH_BACKEND",
                                                               "EMI_BENCH_SHARE_GPU")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "10", "--warmup", "2", "--scenarios", "256",
           "--no-weak"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{"metric"')]
    assert len(lines) == 1, r.stdout[-2000:]
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["config"]["instances_per_gpu"] == 128 and j["value"] > 0
    assert j["config"]["devices_visible"] >= 1
    assert j["config"]["ranks_share_device"] == (j["config"]["devices_visible"] < 2)

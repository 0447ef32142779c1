"""bench.py --gpus N without an external launcher: the parent process spawns torch.distributed.run itself (before it
touches the GPU), relays rank 0's JSON line and exits with the children's status."""
import json
import os
import subprocess
import sys

import pytest

from tests.common import ROOT


def test_self_launch_command_and_exit_code(monkeypatch):
    """CPU: the parent builds the torchrun command the driver's contract names and returns the child's exit code; it must
    not import torch (no GPU initialisation in the parent)."""
    sys.path.insert(0, ROOT)
    import bench
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 7
    monkeypatch.setattr(subprocess, "call", fake_call)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3"])
    monkeypatch.delenv("MASTER_PORT", raising=False)
    assert bench.self_launch(4) == 7
    cmd = seen["cmd"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"]
    assert cmd[cmd.index("--nproc-per-node") + 1] == "4" and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert int(cmd[cmd.index("--master-port") + 1]) > 0
    assert cmd[-4:] == ["--gpus", "4", "--steps", "3"] and cmd[-5].endswith("bench.py")
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_debug_switches_are_refused():
    env = dict(os.environ, MVAE_DEBUG_ONLY_SCALE="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1"], env=env, capture_output=True, text=True)
    assert r.returncode != 0 and "MVAE_DEBUG_ONLY_SCALE" in (r.stderr + r.stdout)


@pytest.mark.gpu
def test_bench_gpus2_self_launch_rehearsal():
    """One GPU box: MVAE_BENCH_REHEARSE=gloo python bench.py --gpus 2 -- both ranks on cuda:0, the all-reduce over gloo.
    Readiness of the N > 1 path only (rank handling, the collective branch, one JSON line); RCCL performance needs the
    driver's multi-GPU node."""
    env = dict(os.environ, MVAE_BENCH_REHEARSE="gloo")
    env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "2",
                        "--no-cpu-baseline", "--no-kernel-profile", "--no-secondary"], env=env, capture_output=True,
                       text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["finite"]
    assert out["collective"] and out["collective"]["world"] == 2 and out["collective"]["collective_bytes"] > 0
    assert out["config"]["global_batch"] == 2 * out["config"]["per_gpu_batch"]
    assert out["debug_env"] == []

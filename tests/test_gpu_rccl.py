"""GPU: the RCCL branch of the path, exercised on the ONE GPU of a builder / test box.

The path's only collective is the end-of-epoch all-gather of the per-run metric table (replicas only; the reference's
analogue of many independent runs is tune_example/tune_mopo.py:222-239).  A fresh child process -- created before anything touches the
GPU -- opens a world-size-1 ``nccl`` process group exactly as ``bench.py``'s N > 1 branch does (``device_id`` = its GPU), runs a few engine
steps and executes the path's collective calls on device tensors, through ``bench.py`` and through ``MFPolicyTrainer._gather``.
This proves that RCCL loads, binds the device and gathers; multi-GPU scaling itself is the driver's measurement."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_rccl_world_size_one_group_gathers_the_metric_table():
    env = dict(os.environ)
    env.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("MASTER_PORT", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--rccl-check"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("RCCL {")]          # (RCCL itself prints "RCCL version : ...")
    assert len(lines) == 1, (r.stdout[-2000:], r.stderr[-2000:])
    out = json.loads(lines[0][5:])
    assert out["backend"] == "nccl" and out["world"] == 1
    assert out["all_reduce_max_ok"] and out["all_gather_equal"] and out["finite"]
    assert out["all_gather_shape"][0] == 1 and out["all_gather_shape"][1] == 8
    assert out["trainer_gathered"] == ["loss/actor", "loss/critic1"]
    assert out["trainer_logged"] == ["rank0/loss/actor", "rank0/loss/critic1"]

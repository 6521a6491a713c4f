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


def test_config5_preset_two_ranks_on_one_gpu_train_two_task_shapes():
    """BASELINE configs[4] rehearsed as far as one GPU allows: `bench.py --gpus 2 --preset config5` starts two ranks itself (before any GPU
    call), both on this box's GPU (ORL_FORCE_DEVICE=0) with the gloo backend standing in for RCCL.  Rank 0 trains the halfcheetah shape
    (obs 17 / act 6), rank 1 the hopper shape (obs 11 / act 3), 8 seeds each in one engine; the JSON line carries the tasks by rank and the
    all-gathered metric table [2 ranks][8 runs][5 metrics], finite.  Not a scaling measurement (the ranks share one GPU)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(ORL_DIST_BACKEND="gloo", ORL_FORCE_DEVICE="0", HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="2")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--preset", "config5", "--steps", "20", "--warmup", "5",
                        "--min-reps", "2", "--min-seconds", "0", "--dataset-size", "50000", "--profile-steps", "0", "--no-cpu-baseline"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["preset"] == "config5" and d["config"]["rccl_world_size"] == 2
    assert d["config"]["tasks_by_rank"] == ["halfcheetah-medium-v2", "hopper-medium-v2"]
    assert d["config"]["runs_per_gpu"] == 8 and d["config"]["engines_per_gpu"] == 1
    assert d["metrics_gathered"]["shape"] == [2, 8, 5]
    assert all(x == x and abs(x) < 1e6 for x in d["metrics_gathered"]["loss_critic1_mean_per_rank"])
    assert d["value"] > 0 and d["scaling"] == "weak"

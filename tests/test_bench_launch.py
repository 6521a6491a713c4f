"""CPU: bench.py's own multi-GPU launcher (`--gpus N` without a torch.distributed.run parent): N child ranks with the right
RANK / LOCAL_RANK / WORLD_SIZE, started before anything touches the GPU; a mismatching WORLD_SIZE is refused.  (The timed path
itself needs MI355Xs; `--launch-check` makes every rank report its environment and exit.)"""
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env():
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    e["OMP_NUM_THREADS"] = "1"
    return e


def test_gpus_flag_starts_one_rank_per_gpu():
    p = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--launch-check", "--preset", "config5"], capture_output=True, text=True,
                       timeout=300, env=_env())
    assert p.returncode == 0, p.stderr[-2000:]
    ranks = [json.loads(m) for m in re.findall(r"LAUNCH (\{[^}]*\})", p.stdout)]      # (robust to two ranks' lines sharing one pipe)
    assert sorted(r["rank"] for r in ranks) == [0, 1]
    assert sorted(r["local_rank"] for r in ranks) == [0, 1]
    assert all(r["world"] == 2 and r["gpus"] == 2 for r in ranks)


def test_child_command_is_the_drivers_launch_line():
    sys.path.insert(0, ROOT)
    import bench
    cmd = bench.child_command(4, 29512, ["--gpus", "4", "--steps", "20"])
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29512"
    assert cmd[-5:] == [BENCH, "--gpus", "4", "--steps", "20"]
    assert bench.PRESETS["config5"] == dict(runs_per_gpu=8, engines_per_gpu=1)


def test_world_size_mismatch_is_refused():
    e = _env()
    e.update(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, BENCH, "--gpus", "8", "--launch-check"], capture_output=True, text=True, timeout=120, env=e)
    assert p.returncode != 0 and "WORLD_SIZE=1" in (p.stderr + p.stdout)


def test_single_gpu_launch_check_needs_no_children():
    p = subprocess.run([sys.executable, BENCH, "--launch-check"], capture_output=True, text=True, timeout=120, env=_env())
    assert p.returncode == 0
    assert json.loads(p.stdout.strip().split("LAUNCH ")[1]) == dict(rank=0, local_rank=0, world=1, gpus=1)

"""GPU: the driver's bench line.  ``bench.py`` runs in a fresh process (as the driver starts it) on a reduced geometry -- one engine x 8 runs,
a 50 000-transition buffer, 5 timed steps, a 2 s CPU sample, no side records -- and the single JSON line it prints must carry every field
of the contract (metric / value / unit / n_gpus / steps / warmup / ms_per_step / higher_is_better / scaling / vs_baseline / dtype / data /
config.workload, ``roofline`` with the dominant launch measured live, ``cpu_baseline`` with the sample it timed)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_prints_one_contract_line():
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "5", "--warmup", "2", "--no-sides", "--min-reps", "2",
           "--min-seconds", "0", "--engines-per-gpu", "1", "--runs-per-gpu", "8", "--dataset-size", "50000", "--profile-steps", "3",
           "--cpu-baseline-seconds", "2"]
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["metric"].startswith("gradient-steps/sec") and d["unit"] == "gradient-steps/s"
    assert d["n_gpus"] == 1 and d["steps"] == 5 and d["warmup"] == 2
    assert d["value"] > 0 and d["ms_per_step"] > 0 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["data"].startswith("synthetic")
    assert "f32" in d["dtype"] and "workload" in d["config"] and "model" not in d["config"]
    # value = the runs of all engines x steps / block time
    assert abs(d["value"] - 8 * 1e3 / d["ms_per_step"]) < 1e-6 * d["value"]
    roof = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "avg_launch_ms"):
        assert k in roof, k
    assert roof["bound"] in ("mfma", "hbm") and 0.0 < roof["frac"] < 1.0
    assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-9
    cpu = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in cpu, k
    assert cpu["kind"] == "port" and cpu["value"] > 0 and cpu["cores"] >= 1

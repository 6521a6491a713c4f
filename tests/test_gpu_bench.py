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


@pytest.mark.parametrize("precision", [1, 2])
def test_two_engines_in_the_one_round_geometry_match_their_solo_runs(precision):
    """(precision 2: the three-plane launches -- two column-half workgroups per net and slab, their own scratch lines per engine.)
    The combination bench.py's default actually runs: TWO engines per GPU, each created with `ws_one_round = 1`
    (weight-stationary launches on CUs / nets workgroups per net), each on its own HIP stream and host thread, sampling the same HBM buffer
    concurrently.  Engines are independent, so what an engine computes must not depend on what runs beside it: 20 device-sampled steps of
    both engines side by side give bit-identical metrics and parameters to the same two engines (same seeds) stepped one after the other."""
    import threading

    import numpy as np
    sys.path[:0] = [ROOT, os.path.join(ROOT, "offlinerl-kit_amd")]
    import bench_workloads as bw
    from offlinerlkit import _engine
    R = 24                                            # 48 batched critics: 5 workgroups per net in the one-round decomposition
    ds = bw.make_dataset(0, 100_000, 17, 6)
    buf = _engine.DeviceBuffer(17, 6, 0)
    buf.load(ds["obs"], ds["act"], ds["nobs"], ds["rew"], ds["term"])

    def make():
        es = [bw.make_engine("cql", R, precision, 0, 40 + e, ws_one_round=1) for e in range(2)]
        for g in es:
            g.attach_buffer(buf)
        return es

    side = make()
    out = [None, None]

    def work(i):
        out[i] = side[i].learn_n(20)[0]
    ths = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    solo = make()
    ref = [g.learn_n(20)[0] for g in solo]
    try:
        for i in range(2):
            assert np.isfinite(out[i]).all()
            assert np.array_equal(out[i], ref[i]), (i, np.abs(out[i] - ref[i]).max())
            for r in (0, R - 1):
                for net in (0, 1, 2):
                    a, b = side[i].get_net(r, net), solo[i].get_net(r, net)
                    for k in a:
                        assert np.array_equal(a[k], b[k]), (i, r, net, k)
        assert not np.array_equal(out[0], out[1])     # different seeds: different trainings
    finally:
        for g in side + solo:
            g.close()
        buf.close()

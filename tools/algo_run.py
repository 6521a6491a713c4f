"""Steady-state loop of one algorithm's BASELINE-shaped workload for profilers (rocprofv3 wraps this program directly):
    python3 tools/algo_run.py ALGO [RUNS=128] [PRECISION=1] [STEPS=30] [key=value ...] [--tags FILE]      (key=value: orl_config overrides,
                                                                                        e.g. max_q_backup=1 with_lagrange=1 hidden=256,256,256)
Builds the engine from bench_workloads.py (launch-script hyper-parameters, synthetic D4RL-shaped buffer), warms up (graph capture), then
replays STEPS steps.  --tags additionally writes the per-launch-tag HIP-event table (eager launches) to FILE."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "offlinerl-kit_amd")]
import bench_workloads as bw  # noqa: E402
from offlinerlkit import _engine  # noqa: E402

tagfile = sys.argv[sys.argv.index("--tags") + 1] if "--tags" in sys.argv else None
over = {}
for a in sys.argv[1:]:
    if "=" in a and not a.startswith("--"):
        k, v = a.split("=", 1)
        over[k] = [int(x) for x in v.split(",")] if k == "hidden" else (float(v) if "." in v else int(v))
args = [a for a in sys.argv[1:] if not a.startswith("--") and "=" not in a and a != tagfile]
algo = args[0]
R = int(args[1]) if len(args) > 1 else 128
prec = int(args[2]) if len(args) > 2 else 1
steps = int(args[3]) if len(args) > 3 else 30
w = bw.WORKLOADS[algo]
eng = bw.make_engine(algo, R, prec, 0, 11, **over)
ds = bw.make_dataset(3, min(w["n"], 400_000), w["obs"], w["act"])
buf = _engine.DeviceBuffer(w["obs"], w["act"], 0)
buf.load(ds["obs"], ds["act"], ds["nobs"], ds["rew"], ds["term"])
if algo == "td3bc":
    buf.normalize_obs(1e-3)
eng.attach_buffer(buf)
eng.learn_n(10)
_, ms = eng.learn_n(steps)
print(f"# {bw.workload_string(algo, R, hidden=over.get('hidden'))} {over if over else ''} precision {prec}: {ms / steps * 1e3:.1f} us per step (graph replay), {R * steps / ms * 1e3:.0f} gradient-steps/s")
if tagfile:
    out = tagfile
    eng.profile_enable(True); eng.learn_n(10); t = eng.profile_table(); eng.profile_enable(False)
    tot = sum(x["total_ms"] for x in t)
    with open(out, "w") as f:
        f.write(f"# {bw.workload_string(algo, R)} precision {prec}: {tot / 10 * 1e3:.1f} us per step (eager launches, HIP events on the engine stream; graph replay: {ms / steps * 1e3:.1f} us)\n")
        f.write("%-32s %9s %10s %10s %7s %9s %9s\n" % ("tag", "launches", "us/launch", "us/step", "%", "TFLOP/s", "GB/s"))
        for x in t:
            us = x["total_ms"] / x["launches"] * 1e3
            f.write("%-32s %9.1f %10.1f %10.1f %7.1f %9.1f %9.1f\n" % (x["name"], x["launches"] / 10, us, x["total_ms"] / 10 * 1e3, 100 * x["total_ms"] / tot,
                                                                 x["flops_per_launch"] / us / 1e6, x["bytes_per_launch"] / us / 1e3))
eng.close(); buf.close()

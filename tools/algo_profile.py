"""Per-launch-tag HIP-event table of one algorithm's full-size case: python tools/algo_profile.py ALGO [R] [PRECISION]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "offlinerl-kit_amd"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")]
import numpy as np
import synth
import bench
from offlinerlkit import _engine
import test_gpu_algos as ta

algo = sys.argv[1]
R = int(sys.argv[2]) if len(sys.argv) > 2 else 128
prec = int(sys.argv[3]) if len(sys.argv) > 3 else 1
case = ta._full_size_case(algo)
eng, *_ = ta.make_engine(algo, case, n_runs=R, precision=prec)
c = getattr(synth, f"{algo.upper()}_CASES")[case]
ds = bench.make_dataset(3, 200000, c["obs_dim"], c["act_dim"])
buf = _engine.DeviceBuffer(c["obs_dim"], c["act_dim"], 0)
buf.load(ds["obs"], ds["act"], ds["nobs"], ds["rew"], ds["term"])
eng.attach_buffer(buf)
eng.learn_n(20)
eng.profile_enable(True); eng.learn_n(10); t = eng.profile_table(); eng.profile_enable(False)
tot = sum(x["total_ms"] for x in t)
print(f"# {algo} {case} R={R} precision={prec}: {tot / 10 * 1e3:.1f} us per step (eager)")
for x in t[:28]:
    us = x["total_ms"] / x["launches"] * 1e3
    print("%-28s %5.1f x %8.1f us = %8.1f us/step %5.1f%%  %7.1f TFLOP/s" % (x["name"], x["launches"] / 10, us, x["total_ms"] / 10 * 1e3, 100 * x["total_ms"] / tot,
                                                                  x["flops_per_launch"] / us / 1e6 if x["flops_per_launch"] else 0))

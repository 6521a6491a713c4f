import os, sys
ROOT = "/root/repo" if os.path.exists("/root/repo/bench.py") else os.environ.get("GRAFT_REPO_ROOT", ".")
sys.path[:0] = [ROOT, os.path.join(ROOT, "offlinerl-kit_amd"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")]
import numpy as np, torch, synth
from offlinerlkit import _engine
import test_gpu_algos as ta
R = 32
case = ta._full_size_case("edac")
eng, *_ = ta.make_engine("edac", case, n_runs=R, precision=1)
c = synth.EDAC_CASES[case]
rng = np.random.RandomState(0); n = 100000
buf = _engine.DeviceBuffer(c["obs_dim"], c["act_dim"], 0)
buf.load(rng.randn(n, c["obs_dim"]).astype(np.float32), np.tanh(rng.randn(n, c["act_dim"])).astype(np.float32), rng.randn(n, c["obs_dim"]).astype(np.float32), rng.randn(n).astype(np.float32), (rng.rand(n) < 0.01).astype(np.float32))
eng.attach_buffer(buf); eng.learn_n(10)
eng.profile_enable(True); eng.learn_n(10); t = eng.profile_table(); eng.profile_enable(False)
tot = sum(x["total_ms"] for x in t)
for x in t[:22]:
    print(f'{x["name"]:34s} {x["launches"]/10:5.1f} x {x["total_ms"]/x["launches"]*1e3:8.1f} us = {x["total_ms"]/10*1e3:8.1f} us/step {100*x["total_ms"]/tot:5.1f}%')
print("total", tot / 10 * 1e3)

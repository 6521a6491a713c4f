"""Gradient-steps/s of the four algorithms with device sampling (learn_n), one engine x R runs, split precision.
Uses the parity tests' full-size cases for shapes and initial weights; the replay buffer is synthetic.
usage: python tools/algo_throughput.py [R] [STEPS]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "offlinerl-kit_amd"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")]
import numpy as np
import torch
import synth
from offlinerlkit import _engine
import test_gpu_algos as ta
import test_gpu_cql as tc

R = int(sys.argv[1]) if len(sys.argv) > 1 else 32
STEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 300

def run(name, eng, obs_dim, act_dim):
    rng = np.random.RandomState(0)
    n = 200000
    buf = _engine.DeviceBuffer(obs_dim, act_dim, 0)
    buf.load(rng.randn(n, obs_dim).astype(np.float32), np.tanh(rng.randn(n, act_dim)).astype(np.float32),
             rng.randn(n, obs_dim).astype(np.float32), rng.randn(n).astype(np.float32), (rng.rand(n) < 0.01).astype(np.float32))
    eng.attach_buffer(buf)
    eng.learn_n(30)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    m, _ = eng.learn_n(STEPS)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    assert np.isfinite(m).all()
    print(f"{name:34s} {R} runs: {R * STEPS / dt:9.0f} gradient-steps/s   ({dt / STEPS * 1e3:.3f} ms per round)", flush=True)
    eng.close()

case = "cql_halfcheetah"
eng, *_ = tc.make_engine(case, n_runs=R, precision=1)
c = synth.CQL_CASES[case]
run("CQL   " + case, eng, c["obs_dim"], c["act_dim"])
for algo in ("iql", "td3bc", "edac"):
    case = ta._full_size_case(algo)
    eng, *_ = ta.make_engine(algo, case, n_runs=R, precision=1)
    c = getattr(synth, f"{algo.upper()}_CASES")[case]
    run(f"{algo.upper():5s} {case}", eng, c["obs_dim"], c["act_dim"])

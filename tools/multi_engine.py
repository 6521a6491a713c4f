"""Experiment: E engines per GPU (each R/E runs, own HIP stream) driven by E host threads -> do their small-kernel phases overlap?
usage: python tools/multi_engine.py E RUNS_PER_ENGINE STEPS"""
import os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "offlinerl-kit_amd")]
import numpy as np
import torch
import bench
from offlinerlkit import _engine

E, R, STEPS = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
ds = bench.make_dataset(0, 200000)
buf = _engine.DeviceBuffer(bench.OBS, bench.ACT, 0)
buf.load(ds["obs"], ds["act"], ds["nobs"], ds["rew"], ds["term"])
engs = []
for e in range(E):
    cfg = _engine.default_config("cql", obs_dim=bench.OBS, act_dim=bench.ACT, hidden=bench.HIDDEN, batch_size=bench.BATCH, n_runs=R,
                                 device=0, precision=1, seed=77 + e, num_repeat_actions=bench.NREP, target_entropy=-float(bench.ACT))
    eng = _engine.Engine(cfg)
    eng.attach_buffer(buf)
    for r in range(R):
        bench.init_weights(eng, r, e * R + r)
    eng.learn_n(20)
    engs.append(eng)
torch.cuda.synchronize()
def work(eng): eng.learn_n(STEPS)
t0 = time.perf_counter()
ths = [threading.Thread(target=work, args=(g,)) for g in engs]
for t in ths: t.start()
for t in ths: t.join()
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"E={E} engines x {R} runs: {E * R * STEPS / dt:.0f} steps/s  ({dt / STEPS * 1e3:.3f} ms per round)")

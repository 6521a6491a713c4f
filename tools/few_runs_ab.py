#!/usr/bin/env python3
"""A/B of engine-creation environment settings in the few-runs regime inside ONE process / one gpurun call (boxes differ by ~10 %):
   python tools/few_runs_ab.py "ORL_FUSE_SMALL=0" "ORL_FUSE_SMALL=1" [--runs 1 8] [--reps 3] [--precision 1]
Every variant is a space-separated list of NAME=VALUE pairs ("-" = none) applied around orl_engine_create (the engine reads its knobs there)."""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "offlinerl-kit_amd")]
import bench_workloads as bw  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("variants", nargs="+")
    ap.add_argument("--runs", type=int, nargs="+", default=[1, 8])
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--precision", type=int, default=1)
    ap.add_argument("--algo", default="cql")
    ap.add_argument("--seconds", type=float, default=0.5)
    args = ap.parse_args()
    import torch
    from offlinerlkit import _engine
    w = bw.WORKLOADS[args.algo]
    ds = bw.make_dataset(0, 200_000, w["obs"], w["act"])
    buf = _engine.DeviceBuffer(w["obs"], w["act"], 0)
    buf.load(ds["obs"], ds["act"], ds["nobs"], ds["rew"], ds["term"])
    res = {}
    for rep in range(args.reps):
        for v in args.variants:
            kv = dict(x.split("=", 1) for x in v.split() if x != "-")
            for R in args.runs:
                old = {k: os.environ.get(k) for k in kv}
                os.environ.update(kv)
                eng = bw.make_engine(args.algo, R, args.precision, 0, 100 + R)
                for k, o in old.items():
                    if o is None:
                        os.environ.pop(k, None)
                    else:
                        os.environ[k] = o
                eng.attach_buffer(buf)
                eng.learn_n(50)
                n = max(100, min(2000, 4000 // R))
                best = []
                t_all = time.perf_counter()
                while len(best) < 3 or time.perf_counter() - t_all < args.seconds:
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    eng.learn_n(n)
                    torch.cuda.synchronize()
                    best.append(time.perf_counter() - t0)
                eng.close()
                res.setdefault((v, R), []).append(R * n / float(np.median(best)))
    for (v, R), xs in sorted(res.items(), key=lambda kv: (kv[0][1], kv[0][0])):
        print("runs %3d  %-40s  %s  median %.0f steps/s (%.1f us/step)" % (R, v, " ".join("%7.0f" % x for x in xs), np.median(xs), 1e6 * R / np.median(xs)), flush=True)


if __name__ == "__main__":
    main()

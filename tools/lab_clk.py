#!/usr/bin/env python3
"""Prints the shader-clock stamps a lab build (-DSB_LAB_CLOCK) of small_abwd_kernel leaves behind: python tools/lab_clk.py [precision]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "offlinerl-kit_amd")]
import bench_workloads as bw
from offlinerlkit import _engine
prec = int(sys.argv[1]) if len(sys.argv) > 1 else 1
ds = bw.make_dataset(0, 100_000, 17, 6)
buf = _engine.DeviceBuffer(17, 6, 0)
buf.load(ds["obs"], ds["act"], ds["nobs"], ds["rew"], ds["term"])
eng = bw.make_engine("cql", 1, prec, 0, 5)
eng.attach_buffer(buf)
eng.learn_n(20)
for _ in range(3):
    eng.learn_n(1)
    c = eng.debug_read(0, "lab_clk").view(np.uint64)[:13].astype(np.int64)
    d = np.diff(c)
    print("total %d clk | " % (c[11] - c[0]) + " ".join("%d:%d" % (i + 1, x) for i, x in enumerate(d[:11])))
    e = eng.debug_read(0, "lab_clk").view(np.uint64)[12:16].astype(np.int64)
    print("   chunk 3 of the dgrad loop: compute %d, store %d, barrier %d" % tuple(np.diff(e)))
    allc = eng.debug_read(0, "lab_clk").view(np.uint64).astype(np.int64)
    for k, name in enumerate(("actor + sample", "critic qgrad", "actor2 + 3 sample jobs", "target critics")):
        f = allc[16 + 12 * k: 16 + 12 * k + 8]
        n = int((f > 0).sum())
        print("   small_fwd %-24s total %6d | %s" % (name, f[n - 1] - f[0], " ".join("%d:%d" % (i + 1, x) for i, x in enumerate(np.diff(f[:n])))))
    for off, name, n in ((64, "ws_fwd critic pass (start | weights | L0 frags | prologue | first group | loop | drain)", 7), (72, "ws_wgrad critic (start | setup | prologue | loop | finish)", 5)):
        f = allc[off:off + n]
        print("   %-90s total %6d | %s" % (name, f[-1] - f[0], " ".join("%d" % x for x in np.diff(f))))

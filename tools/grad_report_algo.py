"""Gradient errors (max / scale, relative L2) of every tensor vs the oracle at step 0: python tools/grad_report_algo.py ALGO R PRECISION"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, ROOT + "/offlinerl-kit_amd", ROOT + "/tests", ROOT + "/tests/golden"):
    sys.path.insert(0, p)
import numpy as np
import test_gpu_algos as ta
from test_gpu_grads import grad_err, GRAD_NETS

algo, R, prec = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
case = ta._full_size_case(algo)
eng, mod, cfg, st, batches, noises = ta.make_engine(algo, case, n_runs=R, precision=prec)
b, n = batches[0], noises[0]
res, aux = mod.learn(st, cfg, b, n)
nl = ta.noise_list(algo, n)
eng.step({k: np.stack([v] * R) for k, v in b.items()}, [np.stack([v] * R) for v in nl] if nl is not None else [])
for nm in GRAD_NETS[algo]:
    if nm + "_grads" not in aux:
        continue
    got = eng.debug_grads(R - 1, ta.NET_IDS[algo][nm])
    for name, g in aux[nm + "_grads"].items():
        if "saved_" in name:
            continue
        emax, el2 = grad_err(got[name], g)
        d = np.abs(np.asarray(got[name], np.float64) - g)
        print(f"{algo} prec {prec} ws32={os.environ.get('ORL_WS32', '1')} {nm:8s} {name:24s} max/scale {emax:.3e} relL2 {el2:.3e} elems>1e-4*scale {(d > 1e-4 * np.abs(g).max()).sum()} / {d.size}")
eng.close()

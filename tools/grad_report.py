"""Prints the gradient errors (max / scale, relative L2) of every tensor vs the oracle: CQL, R runs, both precisions."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, ROOT + "/offlinerl-kit_amd", ROOT + "/tests", ROOT + "/tests/golden"):
    sys.path.insert(0, p)
import numpy as np
import test_gpu_cql as tc
from test_gpu_grads import grad_err
from oracle import cql as ocql

R = int(sys.argv[1]) if len(sys.argv) > 1 else 96
CASE = sys.argv[2] if len(sys.argv) > 2 else "cql_halfcheetah"
for prec in (0, 1):
    eng, cfg, st, batches, noises = tc.make_engine(CASE, n_runs=R, precision=prec)
    for k in range(2):
        res, aux = ocql.learn(st, cfg, batches[k], noises[k])
        m = eng.step(tc.lead(batches[k], R), tc.lead(tc.noise_list(noises[k]), R))
        for nm in ("actor", "critic1", "critic2"):
            got = eng.debug_grads(R - 1, tc.NETS[nm])
            for name, g in aux[nm + "_grads"].items():
                emax, el2 = grad_err(got[name], g)
                print(f"prec {prec} step {k} {nm:8s} {name:28s} max/scale {emax:.2e}  relL2 {el2:.2e}  scale {np.abs(g).max():.2e}")
    eng.close()

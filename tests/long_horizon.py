"""Shared pieces of the long-horizon checks (tests/test_long_horizon.py on the CPU, tests/test_gpu_long_horizon.py on the GPU): the
200-step loss trajectories of the REAL reference and of its perturbed twins (tests/golden/make_long_golden.py -> cql_halfcheetah_long.npz,
iql_hopper_long.npz, td3bc_halfcheetah_long.npz, edac_walker2d_long.npz),
and the envelope statistics every implementation is held to."""
import os

import numpy as np

CASE = "cql_halfcheetah_long"
IQL_CASE = "iql_hopper_long"          # the same experiment for IQLPolicy.learn (iql.py:86-139): a second algorithm family, no sampling noise
HORIZONS = (20, 50, 100, 200)
K_ENVELOPE = 4.0          # an implementation may drift from the reference up to K x as far as the reference's own one-ulp twins do


def load(case=CASE):
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", f"{case}.npz"), allow_pickle=False)
    ref = g["losses"]
    perturbed = [g[f"losses_perturbed{i}"] for i in range(len(g["perturbations"]))]
    return [str(k) for k in g["loss_keys"]], ref, perturbed


def deviation(x, ref):
    """per-step max over the loss keys of |x - ref| / max(|ref|, 1 % of the key's largest magnitude over the window)"""
    floor = 1e-2 * np.abs(ref).max(axis=0, keepdims=True)
    return (np.abs(np.asarray(x, np.float64) - ref) / np.maximum(np.abs(ref), floor)).max(axis=1)


def envelope(ref, perturbed):
    """E[T] = the largest deviation any perturbed reference shows up to step T (running max: drift is not monotone step by step)"""
    d = np.max([deviation(p, ref) for p in perturbed], axis=0)
    return np.maximum.accumulate(d)


# IQL: the twins are perturbed ONCE, by one or two ulps, and IQL amplifies slowly (1.35e-6 until step 100, then x14 per 50 steps); an
# independent implementation injects rounding differences of that size at EVERY step, so its trajectory runs ~40 steps ahead on the same
# growth curve.  Measured: the exact-fp32 engine AND the split engine both sit at 9 - 10 x the twin envelope at steps 100 / 150 and 4 x at
# 200 (the numpy oracle, which mirrors torch op for op: 0.2 - 0.4 x) -- K = 12 for this fixture, and the split engine is additionally held to
# the exact-fp32 engine's own deviation (tests/test_gpu_long_horizon.py).
K_ENVELOPE_IQL = 12.0
# fixtures of the other algorithm families: (algorithm, case, K).  TD3+BC's twins are 2e-7 .. 1.5e-6 apart for 50 steps and grow fast after
# that (2e-5 at 100, 1.5e-3 at 200); EDAC is chaotic from the start like CQL (5e-5 at 20 steps, 1e-3 at 200).  Both engines measured at
# <= 1.05 x the envelope on both: K = 4 as for CQL.
TD3BC_CASE = "td3bc_halfcheetah_long"
EDAC_CASE = "edac_walker2d_long"
OTHER_CASES = (("iql", IQL_CASE, K_ENVELOPE_IQL), ("td3bc", TD3BC_CASE, K_ENVELOPE), ("edac", EDAC_CASE, K_ENVELOPE))


def check(name, losses, ref, perturbed, report=None, k_envelope=K_ENVELOPE):
    d = np.maximum.accumulate(deviation(losses, ref))
    env = envelope(ref, perturbed)
    rows = []
    for T in HORIZONS:
        rows.append((T, float(d[T - 1]), float(env[T - 1])))
        assert d[T - 1] <= k_envelope * env[T - 1], (name, "steps", T, "deviation", d[T - 1], "reference envelope", env[T - 1])
    assert d[19] < 1e-4, (name, "the 1e-4 gate over the first 20 teacher-forced steps", d[19])
    line = f"{name}: max relative loss deviation from the reference by step " + ", ".join(f"{T}: {a:.2e} (envelope {b:.2e})" for T, a, b in rows)
    print(line)
    if report is not None:
        report.append(line)
    return rows

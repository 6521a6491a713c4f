"""Shared pieces of the long-horizon checks (tests/test_long_horizon.py on the CPU, tests/test_gpu_long_horizon.py on the GPU): the
200-step loss trajectory of the REAL reference and of its perturbed twins (tests/golden/make_long_golden.py -> cql_halfcheetah_long.npz),
and the envelope statistics every implementation is held to."""
import os

import numpy as np

CASE = "cql_halfcheetah_long"
HORIZONS = (20, 50, 100, 200)
K_ENVELOPE = 4.0          # an implementation may drift from the reference up to K x as far as the reference's own one-ulp twins do


def load():
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", f"{CASE}.npz"), allow_pickle=False)
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


def check(name, losses, ref, perturbed, report=None):
    d = np.maximum.accumulate(deviation(losses, ref))
    env = envelope(ref, perturbed)
    rows = []
    for T in HORIZONS:
        rows.append((T, float(d[T - 1]), float(env[T - 1])))
        assert d[T - 1] <= K_ENVELOPE * env[T - 1], (name, "steps", T, "deviation", d[T - 1], "reference envelope", env[T - 1])
    assert d[19] < 1e-4, (name, "the 1e-4 gate over the first 20 teacher-forced steps", d[19])
    line = f"{name}: max relative loss deviation from the reference by step " + ", ".join(f"{T}: {a:.2e} (envelope {b:.2e})" for T, a, b in rows)
    print(line)
    if report is not None:
        report.append(line)
    return rows

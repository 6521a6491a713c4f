"""Helpers shared by the test modules (oracle drivers, fixture loading, comparisons)."""
import copy
import os
from collections import OrderedDict

import numpy as np

import synth

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(case):
    return np.load(os.path.join(GOLDEN_DIR, f"{case}.npz"), allow_pickle=False)


def clone_state(st):
    out = OrderedDict()
    for k, v in st.items():
        if isinstance(v, dict):
            out[k] = OrderedDict((n, np.array(a, dtype=np.float32, copy=True)) for n, a in v.items())
        else:
            out[k] = np.array(v, dtype=np.float32, copy=True)
    return out


def rel_err(a, b, floor=1e-3):
    """max |a-b| / max(|b|, floor*scale) with scale = max|b| (robust for near-zero entries)."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    scale = max(np.abs(b).max(), 1e-30)
    return float((np.abs(a - b) / np.maximum(np.abs(b), floor * scale)).max())


def key_scales(g):
    """Per-key scale of a fixture's loss table: max |value| of that key over the teacher-forced window."""
    n = 0
    while f"step{n}/losses" in g.files:
        n += 1
    return np.abs(np.stack([g[f"step{k}/losses"] for k in range(n)])).max(axis=0)


def rel_err_keys(a, b, scales, floor=1e-2):
    """max_i |a_i - b_i| / max(|b_i|, floor * scales_i): every metric is held to a RELATIVE bound of its own; the absolute floor (for a
    loss that crosses zero inside the window, e.g. loss/alpha) is tied to that key's own scale over the fixture, not to the largest key
    of the step (rel_err's floor lets a metric 100x smaller than loss/critic through at an absolute 1e-6 of loss/critic)."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float((np.abs(a - b) / np.maximum(np.abs(b), floor * np.maximum(np.asarray(scales, np.float64), 1e-30))).max())


def scale_err(a, b):
    """max |a-b| / max |b|: error relative to the tensor's scale (the 1e-4 gate for Q-value arrays)."""
    a = np.asarray(a, np.float64).ravel()
    b = np.asarray(b, np.float64).ravel()
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def cql_oracle_setup(case):
    from oracle import cql as ocql
    c, st, batches, noises = synth.cql_case_inputs(case)
    cfg = ocql.default_cfg(c["obs_dim"], c["act_dim"])
    cfg.update(hidden=c["hidden"], num_repeat_actions=c["N"])
    cfg.update(c["over"])
    st = clone_state(st)
    ocql.init_opt(st)
    return cfg, st, batches, noises


def check_state_against_golden(g, tag, nets, atol, rtol=1e-4):
    """nets: {name: {param: array}}; compares digests (and full arrays when stored)."""
    worst = 0.0
    for nm, net in nets.items():
        for k, v in net.items():
            key = f"{tag}/{nm}/{k}/digest"
            if key not in g.files:
                continue
            d = synth.digest(v)
            ref = g[key]
            # sampled elements: absolute tolerance (Adam's first steps move params by ~lr whatever |g| is)
            err = np.abs(d[2:] - ref[2:]).max()
            worst = max(worst, err)
            assert err <= atol + rtol * np.abs(ref[2:]).max(), (tag, nm, k, err)
            fkey = f"{tag}/{nm}/{k}/full"
            if fkey in g.files:
                ferr = np.abs(np.asarray(v, np.float64) - g[fkey]).max()
                worst = max(worst, ferr)
                assert ferr <= atol + rtol * np.abs(g[fkey]).max(), (tag, nm, k, ferr)
    return worst


def generic_oracle_setup(algo, case):
    """(module, cfg, state, batches, noises) for iql / td3bc / edac cases."""
    import importlib
    mod = importlib.import_module(f"oracle.{algo}")
    c, st, batches, noises = getattr(synth, f"{algo}_case_inputs")(case)
    cfg = mod.default_cfg(c["obs_dim"], c["act_dim"])
    cfg.update(hidden=c["hidden"])
    cfg.update(c["over"])
    st = clone_state(st)
    mod.init_opt(st)
    return mod, cfg, st, batches, noises


def mopo_oracle_setup(case):
    from oracle import sac as osac
    c, st, batches, noises = synth.mopo_case_inputs(case)
    cfg = osac.default_cfg(c["obs_dim"], c["act_dim"])
    cfg.update(hidden=c["hidden"]); cfg.update(c["over"])
    st = clone_state(st)
    osac.init_opt(st)
    return cfg, st, batches, noises


def combo_oracle_setup(case):
    from oracle import cql as ocql
    c, st, batches, noises = synth.combo_case_inputs(case)
    cfg = ocql.default_cfg(c["obs_dim"], c["act_dim"])
    cfg.update(synth.combo_cfg(c))
    st = clone_state(st)
    ocql.init_opt(st)
    return cfg, st, batches, noises


def mcq_oracle_setup(case):
    from oracle import mcq as omcq
    c, st, batches, noises = synth.mcq_case_inputs(case)
    cfg = omcq.default_cfg(c["obs_dim"], c["act_dim"])
    cfg.update(synth.mcq_cfg(c))
    st = clone_state(st)
    omcq.init_opt(st)
    return cfg, st, batches, noises

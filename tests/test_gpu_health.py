"""GPU: range and NaN safety of the engine (include/orl_engine.h: orl_health / ORL_HEALTH_*).

The reference raises nothing when a run diverges: its losses turn nan (cql.py:194-207 returns them as they are).  The engine can MASK a
divergence -- the ReLU of the matrix kernels maps a sign-bit NaN to +0, and at precision 1 an operand of 65504 or more splits into
hi = +inf / lo = -inf whose products add up to exactly such a NaN (tools/probes/nan_sign_probe.hip) -- so it keeps sticky per-run flags:
non-finite loss (host, from the metrics it reads back anyway), non-finite gradient (k_adam), operand beyond the split range (a scan of
the last step's MFMA operands: on demand, every 256 steps, and when a run turns non-finite), and refuses a dataset beyond the range
when it is attached.  fp32 (precision 0) has no such range: the same inputs must pass there."""
import warnings

import numpy as np
import pytest

import synth
import test_gpu_cql as tc
from helpers import rel_err

pytestmark = pytest.mark.gpu


def _blown_up_case(run, R):
    """cql_halfcheetah with observations x 50 and, for ONE run, the critics' first layer x 1000 / second layer x 2e-5: that run's first
    hidden activation reaches ~1e5 (beyond fp16's 65504) while every weight (x 2^6 as an MFMA operand) and every input stays in range and
    the Q-values keep their scale."""
    from oracle import cql as ocql
    cfg, st, batches, noises = tc.cql_oracle_setup("cql_halfcheetah")
    big = {k: {n: v.copy() for n, v in st[k].items()} for k in ("critic1", "critic2", "critic1_old", "critic2_old")}
    for k in big:
        big[k]["backbone.model.0.weight"] *= 1000.0
        big[k]["backbone.model.0.bias"] *= 1000.0
        big[k]["backbone.model.2.weight"] *= 2e-5
    b = {k: v.copy() for k, v in batches[0].items()}
    b["observations"] *= 50.0
    b["next_observations"] *= 50.0
    # (every run's actor gets its first layer / 50, so that it sees the observations at their usual scale: a tanh-Gaussian head driven
    # 50x into saturation makes log(1 - a^2 + 1e-6) a test of the two tanh implementations, not of the range)
    st["actor"]["backbone.model.0.weight"] = (st["actor"]["backbone.model.0.weight"] / 50.0).astype(np.float32)
    return cfg, st, big, b, noises[0], ocql


@pytest.mark.parametrize("precision", [0, 1, 2])
def test_activation_beyond_the_split_range_passes_in_fp32_and_is_reported_in_split_precision(precision):
    from offlinerlkit import _engine
    R, bad = 3, 1
    cfg, st, big, b, n, ocql = _blown_up_case(bad, R)
    eng, _, _, _, _ = tc.make_engine("cql_halfcheetah", n_runs=R, precision=precision)
    try:
        for r in range(R):
            eng.set_net(r, tc.NETS["actor"], st["actor"])
        for nm, nid in tc.NETS.items():
            if nm in big:
                eng.set_net(bad, nid, big[nm])
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            m = eng.step(tc.lead(b, R), tc.lead(tc.noise_list(n), R))
            flags = eng.health_check()
        if precision == 0:
            # exact fp32: no range limit.  The blown-up run follows the oracle (same parameters, same batch), nothing is flagged.
            st_big = dict(st); st_big.update(big)
            hid = st_big["critic1"]["backbone.model.0.weight"]
            x = np.concatenate([b["observations"], b["actions"]], axis=1)
            assert np.maximum(x @ hid.T + st_big["critic1"]["backbone.model.0.bias"], 0).max() > 65504.0      # the activation really is out of fp16's range
            res, _ = ocql.learn(st_big, cfg, b, n)
            ora = np.array([res[k] for k in eng.metric_names])
            assert np.isfinite(m).all()
            assert rel_err(m[bad], ora, floor=1e-2) < 1e-4, (m[bad], ora)
            assert (flags == 0).all(), flags
            assert not [x for x in w if issubclass(x.category, _engine.EngineHealthWarning)]
        else:
            assert flags[bad] & _engine.HEALTH_SPLIT_RANGE, flags
            assert flags[0] == 0 and flags[2] == 0, flags                    # the neighbours (same batch, ordinary weights) are fine
            hw = [x for x in w if issubclass(x.category, _engine.EngineHealthWarning)]
            assert hw and "65504" in str(hw[0].message) and f"run {bad}" in str(hw[0].message), [str(x.message) for x in w]
            assert (eng.health() == flags).all()                             # sticky
            eng.health_clear()
            assert (eng.health() == 0).all()
    finally:
        eng.close()


@pytest.mark.parametrize("precision", [0, 1])
def test_diverged_run_raises_the_nonfinite_flags_and_strict_mode_raises(precision):
    """A NaN planted in one critic weight of one run: its losses AND its summed gradients are non-finite; the other runs of the engine are
    untouched.  With strict_health the step raises instead of warning."""
    from offlinerlkit import _engine
    R, bad = 4, 2
    eng, cfg, st, batches, noises = tc.make_engine("cql_halfcheetah", n_runs=R, precision=precision)
    try:
        net = {k: v.copy() for k, v in st["critic1"].items()}
        net["backbone.model.2.weight"][3, 5] = np.nan
        eng.set_net(bad, tc.NETS["critic1"], net)
        with pytest.warns(_engine.EngineHealthWarning, match=f"run {bad}"):
            m = eng.step(tc.lead(batches[0], R), tc.lead(tc.noise_list(noises[0]), R))
        f = eng.health()
        assert f[bad] & _engine.HEALTH_NONFINITE_LOSS and f[bad] & _engine.HEALTH_NONFINITE_GRAD, f
        assert all(f[r] == 0 for r in range(R) if r != bad), f
        assert np.isfinite(np.delete(m, bad, axis=0)).all() and not np.isfinite(m[bad]).all()
        eng.strict_health = True
        with pytest.raises(_engine.EngineHealthError):
            eng.step(tc.lead(batches[1], R), tc.lead(tc.noise_list(noises[1]), R))
    finally:
        eng.close()


def test_dataset_beyond_the_split_range_is_refused_when_attached():
    from offlinerlkit import _engine
    c = synth.CQL_CASES["cql_tiny"]
    ds = synth.make_dataset(5, 4000, c["obs_dim"], c["act_dim"])
    obs = ds["observations"].copy()
    obs[1234, 2] = 7.0e4                                                     # one unnormalised component
    buf = _engine.DeviceBuffer(c["obs_dim"], c["act_dim"])
    buf.load(obs, ds["actions"], ds["next_observations"], ds["rewards"], ds["terminals"].astype(np.float32))
    e1, *_ = tc.make_engine("cql_tiny", precision=1)
    e0, *_ = tc.make_engine("cql_tiny", precision=0)
    try:
        with pytest.raises(RuntimeError, match="65504"):
            e1.attach_buffer(buf)
        e0.attach_buffer(buf)                                                # exact fp32 takes it
        m, _ = e0.learn_n(5)
        assert np.isfinite(m).all()
        buf.normalize_obs(1e-3)                                              # what the message recommends: the column's std is ~1100 now
        e1.attach_buffer(buf)                                                # ... and the normalised dataset attaches
        m, _ = e1.learn_n(5)
        assert np.isfinite(m).all() and (e1.health() == 0).all()
    finally:
        e0.close(); e1.close(); buf.close()


def test_learn_n_names_the_range_as_the_cause_when_a_run_turns_nonfinite():
    """orl_learn_n on device-drawn batches, precision 1: the blown-up run's first step multiplies inf planes, its gradients turn NaN; learn_n
    itself (no explicit health_check) scans that step's operands -- the stored first hidden activation is a finite fp32 1e5 -- and reports
    the range next to the non-finite flags.  (The same scan also runs every 256 steps for overflows that stay finite downstream.)"""
    from offlinerlkit import _engine
    R, bad = 2, 1
    cfg, st, big, b, n, _ = _blown_up_case(bad, R)
    eng, *_ = tc.make_engine("cql_halfcheetah", n_runs=R, precision=1)
    c = synth.CQL_CASES["cql_halfcheetah"]
    ds = synth.make_dataset(3, 20000, c["obs_dim"], c["act_dim"])
    buf = _engine.DeviceBuffer(c["obs_dim"], c["act_dim"])
    buf.load(ds["observations"] * 50.0, ds["actions"], ds["next_observations"] * 50.0, ds["rewards"], ds["terminals"].astype(np.float32))
    try:
        for r in range(R):
            eng.set_net(r, tc.NETS["actor"], st["actor"])
        for nm, nid in tc.NETS.items():
            if nm in big:
                eng.set_net(bad, nid, big[nm])
        eng.attach_buffer(buf)
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            eng.learn_n(1)
        assert eng.health()[bad] & _engine.HEALTH_SPLIT_RANGE, eng.health()
        assert eng.health()[bad] & (_engine.HEALTH_NONFINITE_GRAD | _engine.HEALTH_NONFINITE_LOSS), eng.health()
        assert eng.health()[0] == 0
        assert [x for x in w if issubclass(x.category, _engine.EngineHealthWarning)]
    finally:
        eng.close(); buf.close()

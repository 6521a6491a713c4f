"""GPU: both engine precisions over 200 teacher-forced steps of the north-star CQL shape against the REAL reference's loss trajectory
(tests/golden/cql_halfcheetah_long.npz) -- the window in which 16-bit-operand arithmetic and fp32 could part ways (the longest reference
window elsewhere is 20 steps).  Bar: tests/long_horizon.py -- inside K = 4 x the reference's own one-ulp divergence envelope at steps
20 / 50 / 100 / 200, and the plain 1e-4 gate over the first 20 steps.  Reference: cql.py:87-207.  The weight-stationary kernels are on
this path even at two runs (7936 critic rows x 2 critics x 2 runs >= 4096 batched rows)."""
import numpy as np
import pytest

import long_horizon as lh
import test_gpu_algos as ta
import test_gpu_cql as tc

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("precision", [1, 0])
def test_engine_stays_inside_the_reference_envelope_for_200_steps(precision):
    from offlinerlkit import _engine
    keys, ref, perturbed = lh.load()
    R = 2
    eng, cfg, st, batches, noises = tc.make_engine(lh.CASE, n_runs=R, precision=precision)
    assert eng.metric_names == keys
    try:
        losses = []
        for b, n in zip(batches, noises):
            m = eng.step(tc.lead(b, R), tc.lead(tc.noise_list(n), R))
            assert np.array_equal(m[0], m[1])                     # identical inputs, identical runs: bit-identical metrics
            losses.append(m[0])
        what = "exact-fp32 engine" if precision == 0 else f"split engine ({_engine.split_bits()}-bit operands)"
        lh.check(what, np.array(losses, np.float64), ref, perturbed)
        for r in range(R):
            for net in (0, 1, 2):
                assert all(np.isfinite(v).all() for v in eng.get_net(r, net).values())
    finally:
        eng.close()


def test_iql_engines_stay_inside_the_reference_envelope_for_200_steps():
    """IQL hopper shape, 200 teacher-forced steps (iql.py:86-139; tests/golden/iql_hopper_long.npz), both engine precisions in one test:
    each inside K = 12 x the reference's one-ulp twin envelope (tests/long_horizon.py says why 12 and not 4 here: the exact-fp32 engine needs
    it as much as the split engine does), the plain 1e-4 gate over the first 20 steps, the two runs of an engine bit-identical at every step
    (no arrival-order arithmetic anywhere in the step), and the split engine no further from the reference than 2 x the exact-fp32 engine."""
    from offlinerlkit import _engine
    keys, ref, perturbed = lh.load(lh.IQL_CASE)
    R = 2
    synth = ta.synth
    dev = {}
    for precision in (0, 1):
        # (IQL_LONG_CASES is not in IQL_CASES: the engine builder looks the shape up through synth's case table)
        synth.IQL_CASES[lh.IQL_CASE] = synth.IQL_LONG_CASES[lh.IQL_CASE]
        try:
            eng, mod, cfg, st, batches, noises = ta.make_engine("iql", lh.IQL_CASE, n_runs=R, precision=precision)
        finally:
            del synth.IQL_CASES[lh.IQL_CASE]
        assert eng.metric_names == keys
        try:
            losses = []
            for k, b in enumerate(batches):
                m = eng.step({kk: np.repeat(v[None], R, 0) for kk, v in b.items()}, None)
                assert np.array_equal(m[0], m[1]), (precision, k, m[0], m[1])
                losses.append(m[0])
            what = "exact-fp32 engine (IQL)" if precision == 0 else f"split engine (IQL, {_engine.split_bits()}-bit operands)"
            rows = lh.check(what, np.array(losses, np.float64), ref, perturbed, k_envelope=lh.K_ENVELOPE_IQL)
            dev[precision] = {T: d for T, d, _ in rows}
        finally:
            eng.close()
    for T in lh.HORIZONS:
        assert dev[1][T] <= 2.0 * dev[0][T] + 1e-6, (T, dev[1][T], dev[0][T])

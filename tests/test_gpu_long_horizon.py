"""GPU: both engine precisions over 200 teacher-forced steps of the north-star CQL shape against the REAL reference's loss trajectory
(tests/golden/cql_halfcheetah_long.npz) -- the window in which 16-bit-operand arithmetic and fp32 could part ways (the longest reference
window elsewhere is 20 steps).  Bar: tests/long_horizon.py -- inside K = 4 x the reference's own one-ulp divergence envelope at steps
20 / 50 / 100 / 200, and the plain 1e-4 gate over the first 20 steps.  Reference: cql.py:87-207.  The weight-stationary kernels are on
this path even at two runs (7936 critic rows x 2 critics x 2 runs >= 4096 batched rows)."""
import numpy as np
import pytest

import long_horizon as lh
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

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


@pytest.mark.parametrize("precision", [1, 0, 2])
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
        what = {0: "exact-fp32 engine", 2: "three-plane engine (precision 2)"}.get(precision, f"split engine ({_engine.split_bits()}-bit operands)")
        lh.check(what, np.array(losses, np.float64), ref, perturbed)
        for r in range(R):
            for net in (0, 1, 2):
                assert all(np.isfinite(v).all() for v in eng.get_net(r, net).values())
    finally:
        eng.close()


@pytest.mark.parametrize("algo,case,k_env", lh.OTHER_CASES)
def test_other_algorithms_stay_inside_the_reference_envelope_for_200_steps(algo, case, k_env):
    """IQL (hopper shape, iql.py:86-139), TD3+BC (halfcheetah shape, td3bc.py:83-124) and EDAC (walker2d shape, 10 critics, eta 5,
    edac.py:88-166): 200 teacher-forced steps against the REAL reference's loss trajectory, both engine precisions in one test -- each inside
    K x the reference's one-ulp twin envelope (K = 4; tests/long_horizon.py says why K = 12 for IQL: the exact-fp32 engine needs it as much
    as the split engine does), the plain 1e-4 gate over the first 20 steps, the two runs of an engine bit-identical at every step (no
    arrival-order arithmetic anywhere in the step), and the split engine no further from the reference than 2 x the exact-fp32 engine or the twins' own spread, whichever is larger."""
    from offlinerlkit import _engine
    keys, ref, perturbed = lh.load(case)
    R = 2
    synth = ta.synth
    table, long_table = getattr(synth, f"{algo.upper()}_CASES"), getattr(synth, f"{algo.upper()}_LONG_CASES")
    dev = {}
    for precision in (0, 1):
        table[case] = long_table[case]          # (the engine builder looks the shape up through synth's case table)
        try:
            eng, mod, cfg, st, batches, noises = ta.make_engine(algo, case, n_runs=R, precision=precision)
        finally:
            del table[case]
        assert eng.metric_names == keys
        try:
            losses = []
            for k, (b, n) in enumerate(zip(batches, noises)):
                nl = ta.noise_list(algo, n)
                m = eng.step({kk: np.stack([v] * R) for kk, v in b.items()}, [np.stack([v] * R) for v in nl] if nl is not None else [])
                assert np.array_equal(m[0], m[1]), (algo, precision, k, m[0], m[1])
                losses.append(m[0])
            what = f"exact-fp32 engine ({algo})" if precision == 0 else f"split engine ({algo}, {_engine.split_bits()}-bit operands)"
            rows = lh.check(what, np.array(losses, np.float64), ref, perturbed, k_envelope=k_env)
            dev[precision] = {T: d for T, d, _ in rows}
            env = {T: e for T, _, e in rows}
        finally:
            eng.close()
    for T in lh.HORIZONS:        # (an engine that happens to track torch's rounding can sit far below the envelope: the twins' own spread is the floor)
        assert dev[1][T] <= max(2.0 * dev[0][T], env[T]) + 1e-6, (algo, T, dev[1][T], dev[0][T], env[T])

"""CPU: long-horizon parity of the oracle.  Past ~20 steps a 1e-4 gate on losses is no longer meaningful for ANY two
implementations -- the reference itself, restarted from parameters that differ in the last fp32 bit, drifts 2e-4 .. 4e-4 apart by step 100
and 5e-4 .. 8e-4 by step 200 (tests/golden/cql_halfcheetah_long.npz: four perturbed twins of the real CQLPolicy.learn, cql.py:87-207).  The
statement that can be made is distributional: an implementation whose arithmetic differs from torch's in rounding only stays within a
small multiple (K = 4) of that envelope at steps 20 / 50 / 100 / 200.  Here: the numpy oracle; on the GPU: both engine precisions."""
import numpy as np
import pytest

import long_horizon as lh
import synth
from helpers import cql_oracle_setup, generic_oracle_setup


def test_fixture_shape_and_the_reference_envelope_itself():
    keys, ref, perturbed = lh.load()
    assert ref.shape == (200, len(keys)) and len(perturbed) == 4 and keys[:3] == ["loss/actor", "loss/critic1", "loss/critic2"]
    env = lh.envelope(ref, perturbed)
    # the envelope grows with the horizon and is what SURVEY §7.3.6's probe saw: ~1e-4 around step 100
    assert env[19] < 1e-4 < env[199] and 5e-5 < env[99] < 2e-3


def test_oracle_stays_inside_the_reference_envelope_for_200_steps():
    from oracle import cql as ocql
    keys, ref, perturbed = lh.load()
    cfg, st, batches, noises = cql_oracle_setup(lh.CASE)
    assert len(batches) == 200
    losses = []
    for b, n in zip(batches, noises):
        res, _ = ocql.learn(st, cfg, b, n)
        losses.append([res[k] for k in keys])
    lh.check("numpy oracle", np.array(losses), ref, perturbed)


@pytest.mark.parametrize("algo,case,k_env", lh.OTHER_CASES)
def test_other_algorithms_oracle_stays_inside_the_reference_envelope_for_200_steps(algo, case, k_env):
    """The same statement for IQLPolicy.learn (iql.py:86-139), TD3BCPolicy.learn (td3bc.py:83-124) and EDACPolicy.learn (edac.py:88-166;
    walker2d shape, 10 critics, eta 5).  IQL and TD3+BC amplify slowly: the reference's one-ulp twins stay ~1e-6 apart for 50 - 100 steps
    and then grow an order of magnitude per 50 steps; EDAC is chaotic from the start like CQL."""
    keys, ref, perturbed = lh.load(case)
    assert ref.shape == (200, len(keys)) and len(perturbed) == 4
    env = lh.envelope(ref, perturbed)
    assert env[19] < 1e-4 < env[199] < 5e-3
    mod, cfg, st, batches, noises = generic_oracle_setup(algo, case)
    assert len(batches) == 200
    losses = []
    for b, n in zip(batches, noises):
        res, _ = mod.learn(st, cfg, b, n)
        losses.append([res[k] for k in keys])
    lh.check(f"numpy oracle ({algo})", np.array(losses), ref, perturbed, k_envelope=k_env)

"""GPU: gradient- and parameter-level parity of the configuration bench.py times -- split-bf16 precision, 96 / 128 runs per
engine, i.e. the weight-stationary kernels (ws_fwd with the top activation elided, ws_dgrad_w0, ws_wgrad<2> with derived tail
gradients) -- against the numpy oracle (which is pinned by the reference fixtures, tests/test_oracle_golden.py).

Reference: what autograd leaves in ``param.grad`` before ``optimizer.step()`` (cql.py:180-190, iql.py:97-131, td3bc.py:100-113,
edac.py:100-154).  The engine exposes it through ``orl_debug_grads`` (sum of the split-K slabs the backward kernels wrote).

Tolerances.  precision=1 multiplies with operands split into two bf16 terms (16 significand bits) and drops lo*lo: every product
carries a relative error of ~2^-17 with random sign, so a gradient element (a sum over 256..7936 rows) is off by about 1e-5 of
the tensor's scale; the bar below is the north-star gate, 1e-4 of the tensor's scale, for EVERY element (no outlier budget was
needed), and 2e-5 for the tensor as a whole (relative L2).  A ReLU-mask flip on a pre-activation within rounding distance of 0
moves one row's contribution (1 of >= 256 rows) and stays far inside that bar."""
import numpy as np
import pytest

import synth
import test_gpu_algos as ta
import test_gpu_cql as tc

pytestmark = pytest.mark.gpu

GATE_MAX, GATE_L2 = 1e-4, 2e-5


def grad_err(got, ref):
    got, ref = np.asarray(got, np.float64).ravel(), np.asarray(ref, np.float64).ravel()
    scale = max(np.abs(ref).max(), 1e-30)
    return float(np.abs(got - ref).max() / scale), float(np.linalg.norm(got - ref) / max(np.linalg.norm(ref), 1e-30))


def check_grads(eng, run, net_id, ref, tag, skip=()):
    got = eng.debug_grads(run, net_id)
    worst = 0.0
    for name, g in ref.items():
        if name in skip or "saved_" in name:
            continue
        emax, el2 = grad_err(got[name], g)
        worst = max(worst, emax)
        assert emax < GATE_MAX, (tag, name, "max err / scale", emax)
        assert el2 < GATE_L2, (tag, name, "relative L2", el2)
    return worst


def check_params(eng, runs, nets, st, steps, tag):
    """post-step parameters, same statistical criterion as the fp32 tests (test_gpu_cql.py): Adam moves a parameter by ~lr per step
    whatever |g| is, so an element whose gradient sits at the eps / rounding level may differ by a fraction of lr per step"""
    for r in runs:
        for nm, nid in nets.items():
            got = eng.get_net(r, nid)
            for pn, v in got.items():
                ref = st[nm][pn]
                d = np.abs(v - ref)
                tol = 4e-6 * steps + 1e-4 * np.abs(ref).max()
                assert d.mean() < 1e-6 * steps, (tag, r, nm, pn, d.mean())
                assert (d > tol).mean() < 2e-3, (tag, r, nm, pn, (d > tol).mean())
                assert d.max() < 2 * 3e-4 * steps, (tag, r, nm, pn, d.max())


@pytest.mark.parametrize("R", [96, 128])
def test_cql_bench_configuration_gradients_and_parameters(R):
    """CQL, halfcheetah shapes, split-bf16, R = 96 (bench.py's engine: 192 batched critics, 192 of 256 CUs) and 128 (256 critics):
    every gradient tensor of actor / critic1 / critic2 of the first, middle and last run against the oracle for two consecutive
    steps (the second step starts from Adam-updated parameters and targets), then the parameters after three steps."""
    from oracle import cql as ocql
    eng, cfg, st, batches, noises = tc.make_engine("cql_halfcheetah", n_runs=R, precision=1)
    runs = (0, R // 2, R - 1)
    try:
        worst = 0.0
        for k, (b, n) in enumerate(zip(batches[:3], noises[:3])):
            res, aux = ocql.learn(st, cfg, b, n)
            m = eng.step(tc.lead(b, R), tc.lead(tc.noise_list(n), R))
            ora = np.array([res[x] for x in eng.metric_names])
            for r in runs:
                assert tc.rel_err(m[r], ora, floor=1e-2) < 1e-4, (k, r, m[r], ora)
                if k < 2:
                    for nm in ("actor", "critic1", "critic2"):
                        worst = max(worst, check_grads(eng, r, tc.NETS[nm], aux[nm + "_grads"], (R, k, r, nm)))
        print(f"CQL R={R} split-bf16: worst gradient error {worst:.2e} of the tensor scale")
        check_params(eng, runs, {nm: tc.NETS[nm] for nm in ("actor", "critic1", "critic2", "critic1_old", "critic2_old")}, st, 3, ("cql", R))
    finally:
        eng.close()


def test_cql_three_layer_gradients():
    """reference CLI default [256,256,256] (run_cql.py:31): the middle layers go through the plain weight-stationary dgrad and the
    tiled wgrads; 32 runs, split-bf16"""
    from oracle import cql as ocql
    R = 32
    eng, cfg, st, batches, noises = tc.make_engine("cql_halfcheetah_h3", n_runs=R, precision=1)
    try:
        for k, (b, n) in enumerate(zip(batches[:2], noises[:2])):
            res, aux = ocql.learn(st, cfg, b, n)
            eng.step(tc.lead(b, R), tc.lead(tc.noise_list(n), R))
            for r in (0, R - 1):
                for nm in ("actor", "critic1", "critic2"):
                    check_grads(eng, r, tc.NETS[nm], aux[nm + "_grads"], ("h3", k, r, nm))
        check_params(eng, (0, R - 1), {nm: tc.NETS[nm] for nm in ("actor", "critic1", "critic2")}, st, 2, "cql_h3")
    finally:
        eng.close()


GRAD_NETS = {
    "iql": ("actor", "critic_q1", "critic_q2", "critic_v"),
    "td3bc": ("actor", "critic1", "critic2"),
    "edac": ("actor", "critics"),
}


@pytest.mark.parametrize("algo", ["iql", "td3bc", "edac"])
def test_other_algorithms_gradients_and_parameters_at_128_runs(algo):
    """IQL / TD3+BC / EDAC at 128 runs per engine in split-bf16 (full-size fixtures' shapes): gradients of every trainable net for
    the first two steps, parameters after three.  TD3+BC's actor only steps on even counts (td3bc.py:107): its gradient is
    compared on those steps."""
    R = 128
    case = ta._full_size_case(algo)
    eng, mod, cfg, st, batches, noises = ta.make_engine(algo, case, n_runs=R, precision=1)
    ids = ta.NET_IDS[algo]
    runs = (0, R // 2, R - 1)
    try:
        for k, (b, n) in enumerate(zip(batches[:3], noises[:3])):
            res, aux = mod.learn(st, cfg, b, n)
            nl = ta.noise_list(algo, n)
            bb = {kk: np.stack([v] * R) for kk, v in b.items()}
            m = eng.step(bb, [np.stack([v] * R) for v in nl] if nl is not None else [])
            ora = np.array([res[x] for x in eng.metric_names])
            for r in runs:
                assert ta.rel_err(m[r], ora, floor=1e-2) < 1e-4, (algo, k, r, m[r], ora)
                if k < 2:
                    for nm in GRAD_NETS[algo]:
                        if nm + "_grads" in aux:
                            check_grads(eng, r, ids[nm], aux[nm + "_grads"], (algo, k, r, nm))
        trainable = {nm: ids[nm] for nm in ids}
        st_cmp = {nm: ta._strip_saved(st[nm]) for nm in trainable}
        check_params(eng, runs, trainable, st_cmp, 3, algo)
    finally:
        eng.close()


def test_gradient_tap_fp32_matches_oracle_tightly():
    """the tap itself: exact-fp32 precision, one run, tiny + full-size CQL -- gradients agree with the oracle to fp32 rounding"""
    from oracle import cql as ocql
    for case in ("cql_tiny", "cql_halfcheetah"):
        eng, cfg, st, batches, noises = tc.make_engine(case, n_runs=1, precision=0)
        try:
            res, aux = ocql.learn(st, cfg, batches[0], noises[0])
            eng.step(tc.lead(batches[0]), tc.lead(tc.noise_list(noises[0])))
            for nm in ("actor", "critic1", "critic2"):
                got = eng.debug_grads(0, tc.NETS[nm])
                for name, g in aux[nm + "_grads"].items():
                    emax, el2 = grad_err(got[name], g)
                    assert emax < 2e-5 and el2 < 5e-6, (case, nm, name, emax, el2)
        finally:
            eng.close()

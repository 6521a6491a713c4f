"""GPU: gradient- and parameter-level parity of the configuration bench.py times -- split precision, 96 / 128 runs per engine, i.e. the
weight-stationary kernels (ws_fwd with the top activation elided, ws_dgrad_w0, ws_wgrad<2> with derived tail gradients) -- against the
numpy oracle (which is pinned by the reference fixtures, tests/test_oracle_golden.py).

Reference: what autograd leaves in ``param.grad`` before ``optimizer.step()`` (cql.py:180-190, iql.py:97-131, td3bc.py:100-113,
edac.py:100-154).  The engine exposes it through ``orl_debug_grads`` (sum of the split-K slabs the backward kernels wrote).

Two kinds of check, because a gradient is a much less forgiving quantity than a loss:

(1) ``test_cql_critic_backward_is_componentwise_backward_stable`` (and tests/test_gpu_backward_f64.py for the three-layer and EDAC
    paths): the critic forward / backward kernels in isolation, on the engine's OWN inputs (critic input rows, dq, the ReLU masks it
    packed), against a float64 restatement.  Bar: for EVERY element |g_hip - g_f64| <= C * (the same sum with every term replaced by its
    absolute value), C = 16 * 2^-24 for exact fp32, 16 * 2^-22 for split precision on fp16 hi + lo planes (4 * 2^-17 for the bf16-plane
    variant build).  A structural error (a dropped 32-row group is 0.4 % of the terms, a wrong operand pairing far more) exceeds it by
    orders of magnitude; rounding cannot.  ReLU masks are compared bit by bit: the few that differ from the float64 ones must sit on
    pre-activations within rounding distance of zero (both sides are then valid subgradients; autograd's threshold_backward on another
    BLAS flips the same way).

(2) engine vs oracle, end to end, every gradient tensor.  Measured (MI355X, step 0; max error / tensor scale, relative L2):
    exact fp32: CQL [256,256] 2.6e-5 / 3.6e-5, IQL 7.6e-7 / 4.1e-7, TD3+BC 7.6e-7 / 6.4e-7; CQL [256,256,256] 5.4e-4 / 3.9e-4 and EDAC
    4.2e-3 / 4.2e-4 (ONE ReLU decided the other way than the oracle's BLAS: BARS_FP32_3LAYER).
    split precision, fp16 planes (round 3): CQL [256,256] 2.5e-4 / 5.1e-5, CQL [256,256,256] - / 3.9e-4, IQL 6.3e-7 / 6.4e-7, TD3+BC
    6.7e-7 / 5.4e-7, EDAC 9.7e-7 / 5.2e-7 -- the fp32 engine's level.  (Round 2's bf16 planes: 4.9e-3 / 1.1e-3, 2.2e-2 / 4.2e-3,
    1.5e-2 / 2.3e-3, 5.7e-3 / 1.3e-3, 7.0e-3 / 1.7e-3, behind sanity bars of 5e-2 / 1e-2.)
    Losses and Q-values meet 1e-4 in both precisions (test_gpu_cql.py); parameters after two / three Adam steps: see ``check_params``."""
import numpy as np
import pytest

import synth
import test_gpu_algos as ta
import test_gpu_cql as tc
from helpers import clone_state

pytestmark = pytest.mark.gpu

# (max error / tensor scale, relative L2) bars per precision, step-0 gradients (same parameters on both sides).
# Precision 1 with fp16 hi + lo planes (22 operand bits, round 3): measured worst tensors CQL [256,256] 2.5e-4 / 5.1e-5 (one ReLU on a
# pre-activation within rounding distance of zero decided the other way than the oracle's BLAS: the exact-fp32 engine shows the same
# kind of element at 2.5e-5), IQL 6.3e-7 / 6.4e-7, TD3+BC 6.7e-7 / 5.4e-7, EDAC 9.7e-7 / 5.2e-7 -- the bar is 50x tighter than the one the
# bf16 planes of round 2 needed (5e-2 / 1e-2; kept for the bf16-plane variant build).
BARS_SPLIT = {22: (1e-3, 2e-4), 16: (5e-2, 1e-2)}


class _Bars(dict):
    def __missing__(self, precision):
        if precision == 0:
            return (1e-4, 5e-5)
        if precision == 2:
            # three fp16 planes / fp32 MFMA: exact-fp32-class products, but the f16 MFMA's fp32 accumulation decides the same near-zero
            # pre-activation the other way than the oracle's BLAS as precision 1 does (one ReLU among 2 M: 2.5e-4 of one tensor's scale, 5e-5 in
            # L2) -- the bars of precision 1.  Where masks are given (test_cql_critic_backward_is_componentwise_backward_stable) precision 2 is
            # held to half the exact-fp32 constant.
            return BARS_SPLIT[22]
        from offlinerlkit import _engine
        return BARS_SPLIT[_engine.split_bits()]


BARS = _Bars()


def grad_err(got, ref):
    got, ref = np.asarray(got, np.float64).ravel(), np.asarray(ref, np.float64).ravel()
    scale = max(np.abs(ref).max(), 1e-30)
    return float(np.abs(got - ref).max() / scale), float(np.linalg.norm(got - ref) / max(np.linalg.norm(ref), 1e-30))


# Exact fp32 on the THREE-layer nets (CQL [256,256,256], EDAC) against the fp32 numpy oracle: both sides are fp32 sums in different
# orders.  A ReLU whose pre-activation is within an ulp of zero comes out on the other side and moves one whole term of every
# gradient below it, and EDAC's gradient-diversity term is a difference of large sums (the double-backward sweep), so fp32-vs-fp32
# summation noise shows at 1e-4 .. 1e-3 of the tensor scale.  Measured worst tensors: CQL h3 5.4e-4 max / 3.9e-4 relative L2, EDAC
# 4.2e-3 / 4.2e-4 (1.7 % of a 256 x 256 layer's elements above 1e-4 of the scale) -- digit for digit the same on the tiled fp32 kernels
# (ORL_WS32=0) and on the weight-stationary fp32 kernels (tools/grad_report_algo.py), i.e. a property of the comparison with an fp32
# oracle, not of a kernel; two-layer CQL / IQL / TD3+BC meet the plain fp32 bars (2.6e-5, 7.6e-7, 7.6e-7).
BARS_FP32_3LAYER = (1e-2, 1e-3)


def check_grads(eng, run, net_id, ref, tag, precision=1, report=None, bars=None):
    got = eng.debug_grads(run, net_id)
    bar_max, bar_l2 = bars or BARS[precision]
    for name, g in ref.items():
        if "saved_" in name:
            continue
        emax, el2 = grad_err(got[name], g)
        if report is not None:
            report.append((tag, name, emax, el2))
        assert emax < bar_max, (tag, name, "max err / scale", emax)
        assert el2 < bar_l2, (tag, name, "relative L2", el2)


def _unpack_bits(words, nets, rows, width):
    w = words.reshape(nets, rows, width // 32)
    return ((w[..., None] >> np.arange(32, dtype=np.uint32)) & 1).astype(bool).reshape(nets, rows, width)


def critic_backward_f64(x, dq, net, m0, m1):
    """float64 critic forward / backward (critic_module.py:17-28 + autograd) on given input rows, dq and ReLU masks; returns
    (gradients, absolute-value sums, pre-activations)"""
    f = np.float64
    W0, b0 = net["backbone.model.0.weight"].astype(f), net["backbone.model.0.bias"].astype(f)
    W1, b1 = net["backbone.model.2.weight"].astype(f), net["backbone.model.2.bias"].astype(f)
    wt = net["last.weight"].astype(f).ravel()
    x, dq = x.astype(f), dq.astype(f).ravel()
    z0 = x @ W0.T + b0
    h0 = z0 * m0
    z1 = h0 @ W1.T + b1
    h1 = z1 * m1
    a_h0 = (np.abs(x) @ np.abs(W0).T + np.abs(b0)) * m0          # forward bounds: the backward operands carry the forward's rounding
    a_h1 = (a_h0 @ np.abs(W1).T + np.abs(b1)) * m1
    dz1 = dq[:, None] * wt[None, :] * m1
    a_dz1 = np.abs(dq)[:, None] * np.abs(wt)[None, :] * m1
    dz0 = (dz1 @ W1) * m0
    a_dz0 = (a_dz1 @ np.abs(W1)) * m0
    g = {"last.weight": (dq[:, None] * h1).sum(0)[None, :], "last.bias": np.array([dq.sum()]),
         "backbone.model.2.weight": dz1.T @ h0, "backbone.model.2.bias": dz1.sum(0),
         "backbone.model.0.weight": dz0.T @ x, "backbone.model.0.bias": dz0.sum(0)}
    a = {"last.weight": (np.abs(dq)[:, None] * a_h1).sum(0)[None, :], "last.bias": np.array([np.abs(dq).sum()]),
         "backbone.model.2.weight": a_dz1.T @ a_h0, "backbone.model.2.bias": a_dz1.sum(0),
         "backbone.model.0.weight": a_dz0.T @ np.abs(x), "backbone.model.0.bias": a_dz0.sum(0)}
    return g, a, (z0, z1)


@pytest.mark.parametrize("precision", [1, 0, 2])
@pytest.mark.parametrize("R", [96, 128])
def test_cql_critic_backward_is_componentwise_backward_stable(R, precision):
    """bench.py's kernels (ws_fwd<TQ, L0, SY=false>, ws_dgrad_w0, ws_wgrad<2>: R = 96 -> 192 batched critics on 192 CUs, 128 -> 256)
    on the engine's own critic inputs, dq and packed masks vs float64: see the module docstring, check (1).  Both precisions on the SAME
    path, so the two worst ratios compare the split multiply with the exact-fp32 MFMA directly."""
    from oracle import cql as ocql
    from offlinerlkit import _engine
    # C = 2^-18 for BOTH precisions on this path: 16 * 2^-22 for the fp16 planes (measured 0.23 of it; bf16-plane variant build: 4 * 2^-17,
    # 0.10) and 64 * 2^-24 for exact fp32 (measured 0.44).  The fp32 constant is 4x the 16 * 2^-24 of the shorter reductions in
    # tests/test_gpu_backward_f64.py because one workgroup accumulates ALL 7936 rows of a net in one fp32 chain here (1984 dependent
    # v_mfma_f32_16x16x4 adds; CQL's dq puts the 256 negative data rows first, so the partial sums reach half of the absolute sum): the
    # accumulation error of fp32 itself, ~ sqrt(adds) * 2^-24 * |partial|, is what both precisions show -- the split engine (32 products
    # per add instead of 4) half as much as the exact-fp32 one.
    # precision 2 (ws_fwd3 / ws_dgrad3 / ws_wgrad_kernel<5>: three fp16 planes = exact fp32 operands, products down to 2^-33, 32 products per
    # fp32 add): C = 32 * 2^-24, HALF the exact-fp32 constant of this path.
    C = 64.0 * 2.0 ** -24 if precision == 0 else (32.0 * 2.0 ** -24 if precision == 2 else (16.0 * 2.0 ** -22 if _engine.split_bits() >= 22 else 4.0 * 2.0 ** -17))
    eng, cfg, st, batches, noises = tc.make_engine("cql_halfcheetah", n_runs=R, precision=precision)
    c = synth.CQL_CASES["cql_halfcheetah"]
    B, N, od, ad = c["B"], c["N"], c["obs_dim"], c["act_dim"]
    Mc = B + 3 * B * N
    try:
        pre = clone_state({k: st[k] for k in ("critic1", "critic2")})
        eng.step(tc.lead(batches[0], R), tc.lead(tc.noise_list(noises[0]), R))
        worst, flips_total, flip_z = 0.0, 0, 0.0
        for r in (0, R // 2, R - 1):
            xc = eng.debug_read(r, "xc").reshape(Mc, -1)[:, :od + ad]
            m0 = _unpack_bits(eng.debug_read_bits(r, "ch0"), 2, Mc, 256)
            m1 = _unpack_bits(eng.debug_read_bits(r, "ch1"), 2, Mc, 256)
            for ci, nm in enumerate(("critic1", "critic2")):
                dq = eng.debug_read(r, f"dq{ci + 1}")
                g, a, (z0, z1) = critic_backward_f64(xc, dq, pre[nm], m0[ci], m1[ci])
                # the packed masks: identical to the float64 ones except on pre-activations within rounding distance of zero
                for z, m, lay in ((z0, m0[ci], 0), (z1, m1[ci], 1)):
                    flip = m != (z > 0)
                    flips_total += int(flip.sum())
                    assert flip.mean() < 2e-4, (R, r, nm, lay, "mask flips", flip.mean())
                    if flip.any():
                        assert np.abs(z[flip]).max() < 2e-4 * np.sqrt((z * z).mean()), (R, r, nm, lay, np.abs(z[flip]).max())
                        flip_z = max(flip_z, float(np.abs(z[flip]).max() / np.sqrt((z * z).mean())))
                got = eng.debug_grads(r, tc.NETS[nm])
                for name in g:
                    err = np.abs(got[name].astype(np.float64).reshape(g[name].shape) - g[name])
                    bound = C * a[name] + 1e-30
                    ratio = float((err / bound).max())
                    worst = max(worst, ratio)
                    assert ratio < 1.0, (R, r, nm, name, "componentwise backward error / bound", ratio)
        print(f"CQL critic backward, R={R}, precision {precision}: worst |err| / ({C / 2.0 ** -24:.0f} * 2^-24 * abs-sum) = {worst:.3f}; mask flips vs float64: {flips_total} (largest flipped |z| / rms(z) = {flip_z:.1e})")
    finally:
        eng.close()


def check_params(eng, runs, nets, st, steps, tag, init=None, rel_bar=5e-2):
    """post-step parameters.  Adam's update is lr * m_hat / sqrt(v_hat): whatever |g| is, an element moves by ~lr per step, so an
    element whose gradient is small against the gradient ERROR (relative L2 up to 4e-3 in the bf16-plane variant build, see the module docstring)
    lands a visible fraction of lr away -- the per-element bars of the fp32 tests (5 % of lr for 99.8 % of a tensor) do not transfer.
    What must hold: the mean deviation stays well below the step size (4e-6 * steps against lr = 1e-4 .. 3e-4; measured up to 2.8e-6 per step on the
    three-layer critics, where one flipped top-layer mask reaches two weight matrices below it),
    no element is further away than Adam can move it, and -- where the initial parameters are given -- the deviation is below 5 % of
    the UPDATE itself in L2 (measured <= 2.2 %)."""
    for r in runs:
        for nm, nid in nets.items():
            got = eng.get_net(r, nid)
            for pn, v in got.items():
                ref = st[nm][pn]
                d = np.abs(v - ref)
                assert d.mean() < 4e-6 * steps, (tag, r, nm, pn, d.mean())
                assert d.max() < 2 * 3e-4 * steps, (tag, r, nm, pn, d.max())
                if init is not None and nm in init:
                    upd = np.linalg.norm((ref - init[nm][pn]).astype(np.float64))
                    if upd > 0:
                        rel = np.linalg.norm((v - ref).astype(np.float64)) / upd
                        assert rel < rel_bar, (tag, r, nm, pn, "deviation / update (L2)", rel)


@pytest.mark.parametrize("precision", [1, 0, 2])
@pytest.mark.parametrize("R", [96, 128])
def test_cql_bench_configuration_gradients_and_parameters(R, precision):
    """(precision 0: the exact-fp32 flavours of the same kernels -- ws_fwd_kernel<..., F32>, ws_dgrad32_w0_kernel, ws_wgrad32_kernel<2>
    and, for the 256-row phases of >= 16 runs, the fp32 plain-dgrad / storing variants -- against the fp32 bars.)
    CQL, halfcheetah shapes, split precision, R = 96 (bench.py's engine: 192 batched critics, 192 of 256 CUs) and 128 (256 critics):
    every gradient tensor of actor / critic1 / critic2 of the first, middle and last run against the oracle for two consecutive
    steps (the second step starts from Adam-updated parameters and targets), then the parameters after three steps.  Gradient bars:
    module docstring, check (2); from the second step on both sides start from parameters that already differ by Adam's
    sign-like first update on the few elements whose tiny gradients disagree, so only losses / parameters are compared there."""
    from oracle import cql as ocql
    eng, cfg, st, batches, noises = tc.make_engine("cql_halfcheetah", n_runs=R, precision=precision)
    runs = (0, R // 2, R - 1)
    init = clone_state({k: st[k] for k in ("actor", "critic1", "critic2")})
    try:
        report = []
        for k, (b, n) in enumerate(zip(batches[:3], noises[:3])):
            res, aux = ocql.learn(st, cfg, b, n)
            m = eng.step(tc.lead(b, R), tc.lead(tc.noise_list(n), R))
            ora = np.array([res[x] for x in eng.metric_names])
            for r in runs:
                assert tc.rel_err(m[r], ora, floor=1e-2) < 1e-4, (k, r, m[r], ora)
                if k == 0:
                    for nm in ("actor", "critic1", "critic2"):
                        check_grads(eng, r, tc.NETS[nm], aux[nm + "_grads"], (R, k, r, nm), precision, report)
        print(f"CQL R={R} precision {precision}, step-0 gradients vs oracle: worst max/scale {max(x[2] for x in report):.2e}, worst rel L2 {max(x[3] for x in report):.2e}")
        check_params(eng, runs, {nm: tc.NETS[nm] for nm in ("actor", "critic1", "critic2", "critic1_old", "critic2_old")}, st, 3, ("cql", R), init)
    finally:
        eng.close()


@pytest.mark.parametrize("precision", [1, 0, 2])
def test_cql_three_layer_gradients(precision):
    """(precision 2 on a three-layer critic: the top layer's storing dgrad and its output-stationary wgrad run on three planes, the fused
    first + second layer forward, the middle-layer launches and the third-layer forward on the exact-fp32 kernels.)
    reference CLI default [256,256,256] (run_cql.py:31): the middle layers go through the plain weight-stationary dgrad and the
    tiled wgrads; 32 runs, split precision and exact fp32"""
    from oracle import cql as ocql
    from offlinerlkit import _engine
    R = 32
    # with fp16 hi + lo planes the split engine lands where the exact-fp32 engine does against the fp32 numpy oracle (measured relative L2
    # 3.9e-4 on critic2's first layer in BOTH: one top-layer mask decided the other way than the oracle's BLAS); the sharp statement for
    # this path is tests/test_gpu_backward_f64.py
    fp32_like = precision in (0, 2) or _engine.split_bits() >= 22
    eng, cfg, st, batches, noises = tc.make_engine("cql_halfcheetah_h3", n_runs=R, precision=precision)
    init = clone_state({k: st[k] for k in ("actor", "critic1", "critic2")})
    if precision == 2:
        eng.profile_enable(True)
    try:
        for k, (b, n) in enumerate(zip(batches[:2], noises[:2])):
            res, aux = ocql.learn(st, cfg, b, n)
            eng.step(tc.lead(b, R), tc.lead(tc.noise_list(n), R))
            for r in ((0, R - 1) if k == 0 else ()):
                for nm in ("actor", "critic1", "critic2"):
                    check_grads(eng, r, tc.NETS[nm], aux[nm + "_grads"], ("h3", k, r, nm), precision, bars=BARS_FP32_3LAYER if fp32_like else None)
        check_params(eng, (0, R - 1), {nm: tc.NETS[nm] for nm in ("actor", "critic1", "critic2")}, st, 2, "cql_h3", init, rel_bar=0.15)     # measured 8.2 % (one top-layer mask flip reaches both layers below)
        if precision == 2:          # all six many-row launches of the three-layer critic ran their three-plane flavours
            tags = {row["name"] for row in eng.profile_table()}
            assert {"critic.fwd1@p3", "critic.fwd2@p3", "critic.bwd.dgrad2@p3", "critic.bwd.dgrad1@p3", "critic.bwd.wgrad2@p3", "critic.bwd.wgrad1@p3"} <= tags, sorted(tags)
    finally:
        eng.close()


GRAD_NETS = {
    "iql": ("actor", "critic_q1", "critic_q2", "critic_v"),
    "td3bc": ("actor", "critic1", "critic2"),
    "edac": ("actor", "critics"),
}


@pytest.mark.parametrize("precision", [1, 0, 2])
@pytest.mark.parametrize("algo", ["iql", "td3bc", "edac"])
def test_other_algorithms_gradients_and_parameters_at_128_runs(algo, precision):
    """(precision 2: the weight-stationary forwards / dgrads of every algorithm's 256-wide nets run the three-plane kernels -- EDAC's ensemble
    critics included --, the tiled launches the exact-fp32 kernels.)
    IQL / TD3+BC / EDAC at 128 runs per engine in split precision and in exact fp32 (full-size fixtures' shapes): gradients of every trainable net at
    step 0, losses for three steps, parameters after three.  TD3+BC's actor only steps on even counts (td3bc.py:107): its gradient is
    compared on those steps."""
    R = 128
    case = ta._full_size_case(algo)
    eng, mod, cfg, st, batches, noises = ta.make_engine(algo, case, n_runs=R, precision=precision)
    ids = ta.NET_IDS[algo]
    runs = (0, R // 2, R - 1)
    report = []
    init = {nm: ta._strip_saved({k: np.array(v, copy=True) for k, v in st[nm].items()}) for nm in GRAD_NETS[algo]}
    try:
        for k, (b, n) in enumerate(zip(batches[:3], noises[:3])):
            res, aux = mod.learn(st, cfg, b, n)
            nl = ta.noise_list(algo, n)
            bb = {kk: np.stack([v] * R) for kk, v in b.items()}
            m = eng.step(bb, [np.stack([v] * R) for v in nl] if nl is not None else [])
            ora = np.array([res[x] for x in eng.metric_names])
            for r in runs:
                assert ta.rel_err(m[r], ora, floor=1e-2) < 1e-4, (algo, k, r, m[r], ora)
                if k == 0:
                    for nm in GRAD_NETS[algo]:
                        if nm + "_grads" in aux:
                            check_grads(eng, r, ids[nm], aux[nm + "_grads"], (algo, k, r, nm), precision, report,
                                        bars=BARS_FP32_3LAYER if (precision in (0, 2) and algo == "edac") else None)
        print(f"{algo} R={R} precision {precision}, step-0 gradients vs oracle: worst max/scale {max(x[2] for x in report):.2e}, worst rel L2 {max(x[3] for x in report):.2e}")
        trainable = {nm: ids[nm] for nm in ids}
        st_cmp = {nm: ta._strip_saved(st[nm]) for nm in trainable}
        check_params(eng, runs, trainable, st_cmp, 3, algo, init)
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
                    assert emax < BARS[0][0] and el2 < BARS[0][1], (case, nm, name, emax, el2)
        finally:
            eng.close()

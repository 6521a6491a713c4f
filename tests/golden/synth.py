"""Deterministic synthetic inputs shared by the golden-vector generator
(``make_golden.py``, which feeds them to the real reference in the build
container) and by the tests (which feed the very same arrays to the oracle and
to the HIP engine).  Pure numpy; no reference code involved.

Everything derives from ``np.random.RandomState(seed)`` so fixtures only need
to store the seed and the reference's outputs, not the inputs.
"""
from __future__ import annotations

from collections import OrderedDict

import numpy as np

f32 = np.float32


def _uniform(rng, shape, bound):
    return rng.uniform(-bound, bound, size=shape).astype(f32)


def make_backbone(rng, in_dim, hidden, prefix="backbone.model.", seq_step=2):
    """nn.Linear default init magnitude: W,b ~ U(+-1/sqrt(fan_in)).  seq_step = 3: an MLP built with dropout_rate ([Linear, ReLU, Dropout])."""
    p = OrderedDict()
    d = in_dim
    for l, h in enumerate(hidden):
        bound = 1.0 / np.sqrt(d)
        p[f"{prefix}{seq_step * l}.weight"] = _uniform(rng, (h, d), bound)
        p[f"{prefix}{seq_step * l}.bias"] = _uniform(rng, (h,), bound)
        d = h
    return p, d


def make_critic(rng, in_dim, hidden):
    p, d = make_backbone(rng, in_dim, hidden)
    bound = 1.0 / np.sqrt(d)
    p["last.weight"] = _uniform(rng, (1, d), bound)
    p["last.bias"] = _uniform(rng, (1,), bound)
    return p


def make_tanh_actor(rng, obs_dim, act_dim, hidden):
    """ActorProb(MLP, TanhDiagGaussian(unbounded=True, conditioned_sigma=True))."""
    p, d = make_backbone(rng, obs_dim, hidden)
    bound = 1.0 / np.sqrt(d)
    p["dist_net.mu.weight"] = _uniform(rng, (act_dim, d), bound)
    p["dist_net.mu.bias"] = _uniform(rng, (act_dim,), bound)
    p["dist_net.sigma.weight"] = _uniform(rng, (act_dim, d), bound)
    p["dist_net.sigma.bias"] = _uniform(rng, (act_dim,), bound)
    return p


def make_gauss_actor(rng, obs_dim, act_dim, hidden, dropout=False):
    """IQL actor: ActorProb(MLP, DiagGaussian(unbounded=False, conditioned_sigma=False))."""
    p, d = make_backbone(rng, obs_dim, hidden, seq_step=3 if dropout else 2)
    bound = 1.0 / np.sqrt(d)
    p["dist_net.sigma_param"] = _uniform(rng, (act_dim, 1), 0.2)
    p["dist_net.mu.weight"] = _uniform(rng, (act_dim, d), bound)
    p["dist_net.mu.bias"] = _uniform(rng, (act_dim,), bound)
    return p


def make_det_actor(rng, obs_dim, act_dim, hidden):
    """TD3BC actor: Actor(MLP, action_dim) -> last Linear + max*tanh."""
    p, d = make_backbone(rng, obs_dim, hidden)
    bound = 1.0 / np.sqrt(d)
    p["last.weight"] = _uniform(rng, (act_dim, d), bound)
    p["last.bias"] = _uniform(rng, (act_dim,), bound)
    return p


def make_ensemble_critic(rng, in_dim, hidden, K):
    """EnsembleCritic: model.{0,2,..}.weight (K,in,out), bias (K,1,out)
    (+ saved_* shadows).  Magnitudes follow run_edac.py:100-103."""
    p = OrderedDict()
    dims = [in_dim] + list(hidden) + [1]
    for l in range(len(dims) - 1):
        i, o = dims[l], dims[l + 1]
        std = 1.0 / (2.0 * np.sqrt(i))
        last = l == len(dims) - 2
        if last:
            w = _uniform(rng, (K, i, o), 3e-3)
            b = _uniform(rng, (K, 1, o), 3e-3)
        else:
            w = np.clip(rng.normal(0.0, std, size=(K, i, o)), -2.0, 2.0).astype(f32)
            b = np.full((K, 1, o), 0.1, dtype=f32)
        p[f"model.{2 * l}.weight"] = w
        p[f"model.{2 * l}.bias"] = b
        p[f"model.{2 * l}.saved_weight"] = w.copy()
        p[f"model.{2 * l}.saved_bias"] = b.copy()
    return p


def make_batch(rng, B, obs_dim, act_dim, rew_scale=1.0):
    """D4RL-shaped synthetic minibatch."""
    return OrderedDict(
        observations=rng.standard_normal((B, obs_dim)).astype(f32),
        actions=np.tanh(rng.standard_normal((B, act_dim))).astype(f32),
        next_observations=rng.standard_normal((B, obs_dim)).astype(f32),
        terminals=(rng.uniform(size=(B, 1)) < 0.05).astype(f32),
        rewards=(rng.standard_normal((B, 1)) * rew_scale).astype(f32),
    )


def make_dataset(seed, n, obs_dim, act_dim, term_p=0.01):
    rng = np.random.RandomState(seed)
    return OrderedDict(
        observations=rng.standard_normal((n, obs_dim)).astype(f32),
        actions=np.tanh(rng.standard_normal((n, act_dim))).astype(f32),
        next_observations=rng.standard_normal((n, obs_dim)).astype(f32),
        terminals=(rng.uniform(size=(n,)) < term_p),
        rewards=rng.standard_normal((n,)).astype(f32),
    )


def make_cql_noise(rng, B, N, A, max_q_backup=False, low=-1.0, high=1.0):
    """Draw order of CQLPolicy.learn."""
    n = OrderedDict()
    n["eps_actor"] = rng.standard_normal((B, A)).astype(f32)
    n["eps_next"] = rng.standard_normal((B * N if max_q_backup else B, A)).astype(f32)
    n["u_rand"] = rng.uniform(low, high, size=(B * N, A)).astype(f32)
    n["eps_pi"] = rng.standard_normal((B * N, A)).astype(f32)
    n["eps_next_pi"] = rng.standard_normal((B * N, A)).astype(f32)
    return n


def make_sac_noise(rng, B, A, rows_next=None):
    n = OrderedDict()
    n["eps_actor"] = rng.standard_normal((B, A)).astype(f32)
    n["eps_next"] = rng.standard_normal((rows_next or B, A)).astype(f32)
    return n


def make_td3_noise(rng, B, A):
    return OrderedDict(eps_target=rng.standard_normal((B, A)).astype(f32))


# ----------------------------------------------------------------------------
# digests: compact, order-sensitive summaries of big tensors
# ----------------------------------------------------------------------------

def digest(arr, n_samples=16):
    a = np.asarray(arr, dtype=np.float64).ravel()
    pos = (np.arange(n_samples, dtype=np.int64) * 7919 + 13) % max(a.size, 1)
    return np.concatenate([[a.sum(), np.sqrt((a * a).sum())], a[pos]])


def digest_net(net):
    return OrderedDict((k, digest(v)) for k, v in net.items())


# ----------------------------------------------------------------------------
# named configurations used by fixtures
# ----------------------------------------------------------------------------

CQL_CASES = {
    # name: dict(obs, act, hidden, B, N, steps, overrides)
    "cql_tiny": dict(obs_dim=5, act_dim=3, hidden=[32, 32], B=16, N=3, steps=5, seed=101, over={}),
    "cql_tiny_lagrange": dict(obs_dim=5, act_dim=3, hidden=[32, 32], B=16, N=3, steps=5, seed=102,
                              over=dict(with_lagrange=True)),
    "cql_tiny_maxq": dict(obs_dim=5, act_dim=3, hidden=[32, 32], B=16, N=3, steps=3, seed=103,
                          over=dict(max_q_backup=True)),
    "cql_tiny_stoch_fixed_alpha": dict(obs_dim=5, act_dim=3, hidden=[32, 32], B=16, N=3, steps=3, seed=104,
                                       over=dict(deterministic_backup=False, auto_alpha=False, alpha=0.2)),
    "cql_tiny_h3": dict(obs_dim=4, act_dim=2, hidden=[32, 32, 32], B=8, N=4, steps=3, seed=105, over={}),
    "cql_halfcheetah": dict(obs_dim=17, act_dim=6, hidden=[256, 256], B=256, N=10, steps=20, seed=7, over={}),
    "cql_halfcheetah_h3": dict(obs_dim=17, act_dim=6, hidden=[256, 256, 256], B=256, N=10, steps=3, seed=8, over={}),
    # BASELINE configs[4] names 8 D4RL-mujoco tasks: the hopper shape (critic input 11 + 3 = 14 columns, actor head 6 outputs)
    "cql_hopper": dict(obs_dim=11, act_dim=3, hidden=[256, 256], B=256, N=10, steps=8, seed=11, over={}),
}


# Shapes that exercise kernel-selection corners of the HIP engine at full size (reference fixtures since round 4: make_golden.py
# generates them like CQL_CASES; three steps each, losses + Q taps + parameter digests)
CQL_EXTRA_CASES = {
    # 111-dimensional observations (Ant-like): the first layer is too wide to be fused into the weight-stationary forward
    "cql_wide_obs": dict(obs_dim=111, act_dim=8, hidden=[256, 256], B=256, N=10, steps=3, seed=301, over={}),
    # max-Q backup at full size: the target critics see B*N rows
    "cql_halfcheetah_maxq": dict(obs_dim=17, act_dim=6, hidden=[256, 256], B=256, N=10, steps=3, seed=302, over=dict(max_q_backup=True)),
    # Lagrange variant + stochastic backup at full size
    "cql_halfcheetah_lagrange": dict(obs_dim=17, act_dim=6, hidden=[256, 256], B=256, N=10, steps=3, seed=303,
                                     over=dict(with_lagrange=True, deterministic_backup=False)),
    # batch size that is not a multiple of 32 (row groups of the weight-stationary kernels): everything stays on the tiled kernels
    "cql_batch_200": dict(obs_dim=17, act_dim=6, hidden=[256, 256], B=200, N=10, steps=3, seed=304, over={}),
}


# Long teacher-forced window (VERDICT r2 item 5 / SURVEY §7.3.6): 200 steps of the north-star CQL shape.  Fixture: losses only, of the
# reference and of perturbed references (tests/golden/make_long_golden.py); the inputs are regenerated from the seed.
CQL_LONG_CASES = {
    "cql_halfcheetah_long": dict(obs_dim=17, act_dim=6, hidden=[256, 256], B=256, N=10, steps=200, seed=9, over={}),
}


def cql_case_inputs(case):
    """(cfg_overrides, init_state, [batch_k], [noise_k]) for a CQL case."""
    c = CQL_CASES[case] if case in CQL_CASES else (CQL_EXTRA_CASES[case] if case in CQL_EXTRA_CASES else CQL_LONG_CASES[case])
    rng = np.random.RandomState(c["seed"])
    od, ad, hid = c["obs_dim"], c["act_dim"], c["hidden"]
    state = OrderedDict()
    state["actor"] = make_tanh_actor(rng, od, ad, hid)
    state["critic1"] = make_critic(rng, od + ad, hid)
    state["critic2"] = make_critic(rng, od + ad, hid)
    # targets start as perturbed copies so Polyak and the TD target are exercised non-trivially
    for k in ("critic1", "critic2"):
        state[k + "_old"] = OrderedDict((n, (v + 0.01 * rng.standard_normal(v.shape)).astype(f32))
                                        for n, v in state[k].items())
    state["log_alpha"] = np.array([-0.3], dtype=f32)
    state["cql_log_alpha"] = np.array([0.2], dtype=f32)
    mq = bool(c["over"].get("max_q_backup", False))
    batches = [make_batch(rng, c["B"], od, ad) for _ in range(c["steps"])]
    noises = [make_cql_noise(rng, c["B"], c["N"], ad, max_q_backup=mq) for _ in range(c["steps"])]
    return c, state, batches, noises


IQL_CASES = {
    "iql_tiny": dict(obs_dim=5, act_dim=3, hidden=[32, 32], B=16, steps=5, seed=201, over={}),
    "iql_tiny_h3": dict(obs_dim=4, act_dim=2, hidden=[32, 32, 32], B=8, steps=3, seed=202, over=dict(expectile=0.9, temperature=1.0)),
    "iql_hopper": dict(obs_dim=11, act_dim=3, hidden=[256, 256], B=256, steps=20, seed=21, over={}),
    # run_iql.py --dropout_rate: nn.Dropout behind every ReLU of the ACTOR backbone (run_iql.py:106), keep masks teacher-forced
    "iql_tiny_dropout": dict(obs_dim=5, act_dim=3, hidden=[32, 32], B=16, steps=4, seed=203, over=dict(actor_dropout=0.25)),
    "iql_hopper_dropout": dict(obs_dim=11, act_dim=3, hidden=[256, 256], B=256, steps=3, seed=22, over=dict(actor_dropout=0.1)),
}

TD3BC_CASES = {
    "td3bc_tiny": dict(obs_dim=5, act_dim=3, hidden=[32, 32], B=16, steps=6, seed=301, over={}),
    "td3bc_tiny_freq3": dict(obs_dim=4, act_dim=2, hidden=[32, 32], B=8, steps=7, seed=302, over=dict(update_actor_freq=3)),
    "td3bc_halfcheetah": dict(obs_dim=17, act_dim=6, hidden=[256, 256], B=256, steps=20, seed=31, over={}),
}


def _perturbed(rng, net, s=0.01):
    return OrderedDict((n, (v + s * rng.standard_normal(v.shape)).astype(f32)) for n, v in net.items())


# Long teacher-forced window for a second algorithm family (tests/golden/make_long_golden.py): 200 steps of the IQL hopper shape; the
# fixture holds losses only (reference + perturbed twins), the inputs are regenerated from the seed.
IQL_LONG_CASES = {
    "iql_hopper_long": dict(obs_dim=11, act_dim=3, hidden=[256, 256], B=256, steps=200, seed=23, over={}),
}


def iql_case_inputs(case):
    c = IQL_CASES[case] if case in IQL_CASES else IQL_LONG_CASES[case]
    rng = np.random.RandomState(c["seed"])
    od, ad, hid = c["obs_dim"], c["act_dim"], c["hidden"]
    st = OrderedDict()
    p_drop = c["over"].get("actor_dropout")
    st["actor"] = make_gauss_actor(rng, od, ad, hid, dropout=bool(p_drop))
    st["critic_q1"] = make_critic(rng, od + ad, hid)
    st["critic_q2"] = make_critic(rng, od + ad, hid)
    st["critic_v"] = make_critic(rng, od, hid)
    st["critic_q1_old"] = _perturbed(rng, st["critic_q1"])
    st["critic_q2_old"] = _perturbed(rng, st["critic_q2"])
    batches = [make_batch(rng, c["B"], od, ad) for _ in range(c["steps"])]
    if not p_drop:
        return c, st, batches, [None] * c["steps"]
    noises = [OrderedDict(drop_actor=[(rng.uniform(size=(c["B"], h)) < 1.0 - p_drop).astype(f32) for h in hid]) for _ in range(c["steps"])]
    return c, st, batches, noises


# (long teacher-forced windows, as IQL_LONG_CASES: tests/golden/make_long_golden.py)
TD3BC_LONG_CASES = {
    "td3bc_halfcheetah_long": dict(obs_dim=17, act_dim=6, hidden=[256, 256], B=256, steps=200, seed=33, over={}),
}


def td3bc_case_inputs(case):
    c = TD3BC_CASES[case] if case in TD3BC_CASES else TD3BC_LONG_CASES[case]
    rng = np.random.RandomState(c["seed"])
    od, ad, hid = c["obs_dim"], c["act_dim"], c["hidden"]
    st = OrderedDict()
    st["actor"] = make_det_actor(rng, od, ad, hid)
    st["critic1"] = make_critic(rng, od + ad, hid)
    st["critic2"] = make_critic(rng, od + ad, hid)
    st["actor_old"] = _perturbed(rng, st["actor"])
    st["critic1_old"] = _perturbed(rng, st["critic1"])
    st["critic2_old"] = _perturbed(rng, st["critic2"])
    batches = [make_batch(rng, c["B"], od, ad) for _ in range(c["steps"])]
    noises = [make_td3_noise(rng, c["B"], ad) for _ in range(c["steps"])]
    return c, st, batches, noises


EDAC_CASES = {
    "edac_tiny": dict(obs_dim=5, act_dim=3, hidden=[32, 32], B=16, steps=5, seed=401, over=dict(num_critics=4, eta=1.0)),
    "edac_tiny_h3_eta5": dict(obs_dim=4, act_dim=2, hidden=[32, 32, 32], B=8, steps=3, seed=402, over=dict(num_critics=3, eta=5.0)),
    "edac_tiny_eta0_det": dict(obs_dim=5, act_dim=3, hidden=[32, 32], B=16, steps=3, seed=403,
                               over=dict(num_critics=4, eta=0.0, deterministic_backup=True)),
    "edac_tiny_maxq": dict(obs_dim=5, act_dim=3, hidden=[32, 32], B=8, steps=2, seed=404, over=dict(num_critics=3, eta=1.0, max_q_backup=True)),
    "edac_walker2d": dict(obs_dim=17, act_dim=6, hidden=[256, 256, 256], B=256, steps=5, seed=41, over=dict(num_critics=10, eta=5.0)),
}


EDAC_LONG_CASES = {
    "edac_walker2d_long": dict(obs_dim=17, act_dim=6, hidden=[256, 256, 256], B=256, steps=200, seed=43, over=dict(num_critics=10, eta=5.0)),
}


def edac_case_inputs(case):
    c = EDAC_CASES[case] if case in EDAC_CASES else EDAC_LONG_CASES[case]
    rng = np.random.RandomState(c["seed"])
    od, ad, hid, K = c["obs_dim"], c["act_dim"], c["hidden"], c["over"]["num_critics"]
    st = OrderedDict()
    st["actor"] = make_tanh_actor(rng, od, ad, hid)
    st["critics"] = make_ensemble_critic(rng, od + ad, hid, K)
    # larger last layer than the script's 3e-3 so the min over members / gradient penalty are well exercised
    last = f"model.{2 * len(hid)}"
    st["critics"][last + ".weight"] = _uniform(rng, st["critics"][last + ".weight"].shape, 0.1)
    st["critics"][last + ".saved_weight"] = st["critics"][last + ".weight"].copy()
    st["critics_old"] = OrderedDict((n, (v + (0.01 * rng.standard_normal(v.shape) if "saved" not in n else 0)).astype(f32))
                                    for n, v in st["critics"].items())
    st["log_alpha"] = np.array([-0.3], dtype=f32)
    mq = bool(c["over"].get("max_q_backup", False))
    batches = [make_batch(rng, c["B"], od, ad) for _ in range(c["steps"])]
    noises = [make_sac_noise(rng, c["B"], ad, rows_next=(10 * c["B"] if mq else None)) for _ in range(c["steps"])]
    return c, st, batches, noises


# ----------------------------------------------------------------------------
# model-based callers of the path: MOPO = SAC.learn, COMBO = CQL.learn variant, both on a real + model batch
# ----------------------------------------------------------------------------
MOPO_CASES = {
    # real_ratio 0.05 of batch 256 (run_mopo.py:60-61): 12 real + 244 model rows
    "mopo_tiny": dict(obs_dim=5, act_dim=3, hidden=[32, 32], B_real=4, B_fake=12, steps=5, seed=501, over={}),
    "mopo_tiny_fixed_alpha": dict(obs_dim=4, act_dim=2, hidden=[32, 32], B_real=3, B_fake=5, steps=3, seed=502, over=dict(auto_alpha=False, alpha=0.2)),
    "mopo_halfcheetah": dict(obs_dim=17, act_dim=6, hidden=[256, 256], B_real=12, B_fake=244, steps=5, seed=51, over={}),
}
COMBO_CASES = {
    "combo_tiny": dict(obs_dim=5, act_dim=3, hidden=[32, 32], B_real=8, B_fake=8, N=3, steps=5, seed=601, over=dict(rho_s="mix")),
    "combo_tiny_model": dict(obs_dim=5, act_dim=3, hidden=[32, 32], B_real=6, B_fake=10, N=3, steps=3, seed=602, over=dict(rho_s="model")),
    "combo_tiny_lagrange": dict(obs_dim=4, act_dim=2, hidden=[32, 32], B_real=8, B_fake=8, N=3, steps=3, seed=603,
                                over=dict(rho_s="model", with_lagrange=True)),
    # run_combo.py: batch 256, real_ratio 0.5, hidden [256,256,256], rho_s mix
    "combo_halfcheetah": dict(obs_dim=17, act_dim=6, hidden=[256, 256, 256], B_real=128, B_fake=128, N=10, steps=3, seed=61, over=dict(rho_s="mix")),
}


def _split_batch(rng, c):
    real = make_batch(rng, c["B_real"], c["obs_dim"], c["act_dim"])
    fake = make_batch(rng, c["B_fake"], c["obs_dim"], c["act_dim"])
    return OrderedDict(real=real, fake=fake)


def mix_batch(b):
    """real rows first, then model rows (torch.cat([real, fake], 0), mopo.py:83 / combo.py:113)"""
    return OrderedDict((k, np.concatenate([b["real"][k], b["fake"][k]], axis=0)) for k in b["real"])


def _sac_like_state(rng, c):
    od, ad, hid = c["obs_dim"], c["act_dim"], c["hidden"]
    state = OrderedDict()
    state["actor"] = make_tanh_actor(rng, od, ad, hid)
    state["critic1"] = make_critic(rng, od + ad, hid)
    state["critic2"] = make_critic(rng, od + ad, hid)
    for k in ("critic1", "critic2"):
        state[k + "_old"] = _perturbed(rng, state[k])
    state["log_alpha"] = np.array([-0.3], dtype=f32)
    return state


def mopo_case_inputs(case):
    c = MOPO_CASES[case]
    rng = np.random.RandomState(c["seed"])
    state = _sac_like_state(rng, c)
    B = c["B_real"] + c["B_fake"]
    batches = [_split_batch(rng, c) for _ in range(c["steps"])]
    noises = [OrderedDict(eps_next=rng.standard_normal((B, c["act_dim"])).astype(f32),
                          eps_actor=rng.standard_normal((B, c["act_dim"])).astype(f32)) for _ in range(c["steps"])]
    return c, state, batches, noises


def combo_case_inputs(case):
    c = COMBO_CASES[case]
    rng = np.random.RandomState(c["seed"])
    state = _sac_like_state(rng, c)
    state["cql_log_alpha"] = np.array([0.2], dtype=f32)
    B = c["B_real"] + c["B_fake"]
    Bc = c["B_fake"] if c["over"].get("rho_s") == "model" else B
    A, N = c["act_dim"], c["N"]
    batches = [_split_batch(rng, c) for _ in range(c["steps"])]
    noises = []
    for _ in range(c["steps"]):
        n = OrderedDict()
        n["eps_actor"] = rng.standard_normal((B, A)).astype(f32)
        n["eps_next"] = rng.standard_normal((B, A)).astype(f32)
        n["u_rand"] = rng.uniform(-1.0, 1.0, size=(Bc * N, A)).astype(f32)
        n["eps_pi"] = rng.standard_normal((Bc * N, A)).astype(f32)
        n["eps_next_pi"] = rng.standard_normal((Bc * N, A)).astype(f32)
        noises.append(n)
    return c, state, batches, noises


def combo_cfg(c):
    """oracle/cql.py configuration of a COMBO case (run_combo.py:63-82 defaults)"""
    B = c["B_real"] + c["B_fake"]
    over = dict(c["over"])
    rho = over.pop("rho_s", "mix")
    cfg = dict(hidden=c["hidden"], num_repeat_actions=c["N"], with_lagrange=False, cql_alpha_lr=3e-4, cql_weight=5.0,
               cons_rows=(c["B_real"], c["B_fake"]) if rho == "model" else (0, B), real_rows=c["B_real"])
    cfg.update(over)
    return cfg


# ----------------------------------------------------------------------------
# MCQ (policy/model_free/mcq.py): SAC critics / actor + a VAE behaviour policy (nets/vae.py)
# ----------------------------------------------------------------------------
MCQ_CASES = {
    "mcq_tiny": dict(obs_dim=5, act_dim=3, hidden=[32, 32], vae_hidden=24, latent_dim=6, B=8, N=3, steps=5, seed=701, over={}),
    "mcq_tiny_fixed_alpha": dict(obs_dim=4, act_dim=2, hidden=[32, 32], vae_hidden=16, latent_dim=4, B=8, N=2, steps=3, seed=702,
                                 over=dict(auto_alpha=False, alpha=0.2, lmbda=0.7)),
    # run_mcq.py defaults at the hopper shape: hidden [400,400], VAE hidden 750, latent 2 * act_dim, batch 256, 10 sampled actions
    "mcq_hopper": dict(obs_dim=11, act_dim=3, hidden=[400, 400], vae_hidden=750, latent_dim=6, B=256, N=10, steps=3, seed=71, over={}),
}


def make_vae(rng, obs_dim, act_dim, hidden, latent):
    net = OrderedDict()
    for name, i, o in (("e1", obs_dim + act_dim, hidden), ("e2", hidden, hidden), ("mean", hidden, latent), ("log_std", hidden, latent),
                       ("d1", obs_dim + latent, hidden), ("d2", hidden, hidden), ("d3", hidden, act_dim)):
        b = 1.0 / np.sqrt(i)
        net[f"{name}.weight"] = _uniform(rng, (o, i), b)
        net[f"{name}.bias"] = _uniform(rng, (o,), b)
    return net


def mcq_case_inputs(case):
    c = MCQ_CASES[case]
    rng = np.random.RandomState(c["seed"])
    state = _sac_like_state(rng, c)
    state["behavior_policy"] = make_vae(rng, c["obs_dim"], c["act_dim"], c["vae_hidden"], c["latent_dim"])
    B, A, Z, N = c["B"], c["act_dim"], c["latent_dim"], c["N"]
    batches = [make_batch(rng, B, c["obs_dim"], A) for _ in range(c["steps"])]
    noises = []
    for _ in range(c["steps"]):
        n = OrderedDict()
        n["eps_vae"] = rng.standard_normal((B, Z)).astype(f32)
        n["eps_next"] = rng.standard_normal((B, A)).astype(f32)
        n["z_ood"] = rng.standard_normal((2 * B * N, Z)).astype(f32)
        n["eps_ood"] = rng.standard_normal((2 * B, A)).astype(f32)
        n["eps_actor"] = rng.standard_normal((B, A)).astype(f32)
        noises.append(n)
    return c, state, batches, noises


def mcq_cfg(c):
    cfg = dict(hidden=c["hidden"], vae_hidden=c["vae_hidden"], latent_dim=c["latent_dim"], num_sampled_actions=c["N"])
    cfg.update(c["over"])
    return cfg

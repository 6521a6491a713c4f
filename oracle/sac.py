"""Oracle restatement of SACPolicy.learn (policy/model_free/sac.py:88-140), the update MOPOPolicy.learn applies to the
concatenation of a real and a model-generated batch (policy/model_based/mopo.py:81-84).  TEST INFRASTRUCTURE ONLY.

State: state["actor"|"critic1"|"critic2"|"critic1_old"|"critic2_old"], state["log_alpha"], state["opt"][...].
Noise in the reference's draw order: eps_next (B,A) [actforward(next_obss), sac.py:95], eps_actor (B,A) [actforward(obss), :113].
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Dict

import numpy as np

from . import nn
from .nn import f32


def default_cfg(obs_dim: int, act_dim: int) -> dict:
    """run_example/run_mopo.py:31-41"""
    return dict(obs_dim=obs_dim, act_dim=act_dim, hidden=[256, 256], actor_lr=1e-4, critic_lr=3e-4, alpha_lr=1e-4,
                gamma=0.99, tau=0.005, auto_alpha=True, alpha=0.2, target_entropy=-float(act_dim))


def init_opt(state: dict) -> None:
    state["opt"] = {k: nn.adam_init(None) for k in ("actor", "critic1", "critic2", "alpha")}


def learn(state: dict, cfg: dict, batch: Dict[str, np.ndarray], noise: Dict[str, np.ndarray]):
    obs = np.asarray(batch["observations"], f32)
    act = np.asarray(batch["actions"], f32)
    nobs = np.asarray(batch["next_observations"], f32)
    rew = np.asarray(batch["rewards"], f32).reshape(-1, 1)
    term = np.asarray(batch["terminals"], f32).reshape(-1, 1)
    B = obs.shape[0]
    od = obs.shape[1]
    actor, c1, c2, c1o, c2o = state["actor"], state["critic1"], state["critic2"], state["critic1_old"], state["critic2_old"]
    # _alpha: exp(log_alpha) un-clamped until the first alpha step (sac.py:46), clamped to [0, 1] after (:129)
    alpha = state.get("_alpha", f32(np.exp(state["log_alpha"][0])) if cfg["auto_alpha"] else f32(cfg["alpha"]))
    aux = {}

    # ---- critics (sac.py:92-110) ----
    q1, h1 = nn.critic_fwd(c1, obs, act)
    q2, h2 = nn.critic_fwd(c2, obs, act)
    na, nlogp, _ = nn.tanh_gauss_fwd(actor, nobs, noise["eps_next"])
    nq1, _ = nn.critic_fwd(c1o, nobs, na)
    nq2, _ = nn.critic_fwd(c2o, nobs, na)
    next_q = np.minimum(nq1, nq2) - alpha * nlogp
    target_q = (rew + f32(cfg["gamma"]) * (f32(1) - term) * next_q).astype(f32)
    l1 = f32(((q1 - target_q) ** 2).mean(dtype=f32))
    l2 = f32(((q2 - target_q) ** 2).mean(dtype=f32))
    for name, net, qq, hh in (("critic1", c1, q1, h1), ("critic2", c2, q2, h2)):
        g, _ = nn.critic_bwd(net, hh, (f32(2) * (qq - target_q) / f32(B)).astype(f32), need_dx=False)
        nn.adam_step(net, g, state["opt"][name], cfg["critic_lr"])
        aux[name + "_grads"] = g
    aux.update(q1=q1, q2=q2, target_q=target_q)

    # ---- actor against the UPDATED critics (sac.py:112-119) ----
    a, logp, cache = nn.tanh_gauss_fwd(actor, obs, noise["eps_actor"])
    q1a, h1a = nn.critic_fwd(c1, obs, a)
    q2a, h2a = nn.critic_fwd(c2, obs, a)
    actor_loss = f32(-np.minimum(q1a, q2a).mean(dtype=f32) + alpha * logp.mean(dtype=f32))
    g1, g2 = nn.min2_grad(q1a, q2a, np.full((B, 1), -1.0 / B, dtype=f32))
    _, dx1 = nn.critic_bwd(c1, h1a, g1, need_dx=True, need_dw=False)
    _, dx2 = nn.critic_bwd(c2, h2a, g2, need_dx=True, need_dw=False)
    agr = nn.tanh_gauss_bwd(actor, cache, dx1[:, od:] + dx2[:, od:], np.full((B, 1), alpha / f32(B), dtype=f32))
    nn.adam_step(actor, agr, state["opt"]["actor"], cfg["actor_lr"])
    aux["actor_grads"] = agr
    aux["q1a"], aux["q2a"] = q1a, q2a

    result = OrderedDict([("loss/actor", float(actor_loss)), ("loss/critic1", float(l1)), ("loss/critic2", float(l2))])
    if cfg["auto_alpha"]:          # sac.py:121-129
        lp_t = logp + f32(cfg["target_entropy"])
        la = state["log_alpha"]
        alpha_loss = f32(-(la[0] * lp_t).mean(dtype=f32))
        nn.adam_step({"log_alpha": la}, {"log_alpha": np.array([-(lp_t.mean(dtype=f32))], f32)}, state["opt"]["alpha"], cfg["alpha_lr"])
        alpha = f32(min(max(np.exp(la[0]), f32(0.0)), f32(1.0)))
        state["_alpha"] = alpha
        result["loss/alpha"] = float(alpha_loss)
        result["alpha"] = float(alpha)
    nn.polyak(c1o, c1, cfg["tau"])
    nn.polyak(c2o, c2, cfg["tau"])
    return result, aux

"""Oracle restatement of IQLPolicy.learn (policy/model_free/iql.py:86-139).  TEST INFRASTRUCTURE ONLY.

Nets: actor = ActorProb(MLP, DiagGaussian(unbounded=False, conditioned_sigma=False))
(dist_module.py:45-78: mu = tanh(Linear), sigma = exp(sigma_param (A,1))), critic_q1/q2 (obs+act -> 1),
critic_v (obs -> 1), targets critic_q1_old / critic_q2_old.  No RNG draws in learn() -- unless the ACTOR backbone was built with
``dropout_rate`` (run_iql.py:34,106: only the actor backbone gets it): then the actor forward of the policy-improvement step (iql.py:127, the
policy is in train() mode) draws one keep mask per hidden layer, ``noise["drop_actor"]`` (cfg["actor_dropout"] = p).
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Dict

import numpy as np

from . import nn
from .nn import f32


def default_cfg(obs_dim: int, act_dim: int) -> dict:
    """run_example/run_iql.py:25-46."""
    return dict(obs_dim=obs_dim, act_dim=act_dim, hidden=[256, 256], actor_lr=3e-4, critic_q_lr=3e-4, critic_v_lr=3e-4,
                gamma=0.99, tau=0.005, expectile=0.7, temperature=3.0, max_mu=1.0)


def init_opt(state: dict) -> None:
    state["opt"] = {k: nn.adam_init(None) for k in ("actor", "critic_q1", "critic_q2", "critic_v")}


def gauss_actor_fwd(net, obs, drop=None):
    Ws, bs = nn.backbone_layers(net)
    hs = nn.mlp_fwd(obs, Ws, bs, drop)
    m_raw = nn.mm(hs[-1], net["dist_net.mu.weight"].T) + net["dist_net.mu.bias"]
    mu = np.tanh(m_raw)                                   # max_mu = 1.0 (dist_module.py:70-71)
    ls = net["dist_net.sigma_param"].reshape(1, -1)       # (1, A)   (:75-77)
    sigma = np.exp(ls + np.zeros_like(mu))
    return mu.astype(f32), sigma.astype(f32), hs


def learn(state: dict, cfg: dict, batch: Dict[str, np.ndarray], noise=None):
    obs = np.asarray(batch["observations"], f32)
    act = np.asarray(batch["actions"], f32)
    nobs = np.asarray(batch["next_observations"], f32)
    rew = np.asarray(batch["rewards"], f32).reshape(-1, 1)
    term = np.asarray(batch["terminals"], f32).reshape(-1, 1)
    B = obs.shape[0]
    actor, q1n, q2n, vn = state["actor"], state["critic_q1"], state["critic_q2"], state["critic_v"]
    q1o, q2o = state["critic_q1_old"], state["critic_q2_old"]
    te = f32(cfg["expectile"])
    aux = {}

    # ---- value net (iql.py:90-98) ----
    qo1, _ = nn.critic_fwd(q1o, obs, act)
    qo2, _ = nn.critic_fwd(q2o, obs, act)
    q = np.minimum(qo1, qo2)
    v, hv = nn.critic_fwd(vn, obs)
    diff = q - v
    w = np.where(diff > 0, te, f32(1) - te).astype(f32)  # iql.py:82-84 (strict >)
    v_loss = f32((w * diff * diff).mean(dtype=f32))
    dv = (-f32(2) * w * diff / f32(B)).astype(f32)
    gv, _ = nn.critic_bwd(vn, hv, dv, need_dx=False)
    nn.adam_step(vn, gv, state["opt"]["critic_v"], cfg["critic_v_lr"])
    aux["v"], aux["q_old"] = v, q
    aux["critic_v_grads"] = gv

    # ---- critics (iql.py:100-116), target uses the UPDATED V ----
    q1, h1 = nn.critic_fwd(q1n, obs, act)
    q2, h2 = nn.critic_fwd(q2n, obs, act)
    next_v, _ = nn.critic_fwd(vn, nobs)
    target_q = (rew + f32(cfg["gamma"]) * (f32(1) - term) * next_v).astype(f32)
    q1_loss = f32(((q1 - target_q) ** 2).mean(dtype=f32))
    q2_loss = f32(((q2 - target_q) ** 2).mean(dtype=f32))
    for name, net, qq, hh in (("critic_q1", q1n, q1, h1), ("critic_q2", q2n, q2, h2)):
        g, _ = nn.critic_bwd(net, hh, (f32(2) * (qq - target_q) / f32(B)).astype(f32), need_dx=False)
        nn.adam_step(net, g, state["opt"][name], cfg[f"critic_q_lr"])
        aux[name + "_grads"] = g
    aux["q1"], aux["q2"], aux["target_q"] = q1, q2, target_q

    # ---- actor (iql.py:118-131): advantage-weighted BC ----
    v2, _ = nn.critic_fwd(vn, obs)                        # updated V; q_old unchanged (targets not yet synced)
    exp_a = np.minimum(np.exp((q - v2) * f32(cfg["temperature"])), f32(100.0)).astype(f32)
    p_drop = cfg.get("actor_dropout") or None
    mu, sigma, hs = gauss_actor_fwd(actor, obs, (p_drop, noise["drop_actor"]) if p_drop else None)
    var = sigma * sigma
    lp = -((act - mu) ** 2) / (f32(2) * var) - np.log(sigma) - nn.LOG_SQRT_2PI
    logp = lp.sum(axis=1, keepdims=True, dtype=f32)
    actor_loss = f32(-(exp_a * logp).mean(dtype=f32))
    dlogp = (-exp_a / f32(B)).astype(f32)                 # (B,1)
    dmu = dlogp * (act - mu) / var
    dm_raw = dmu * (f32(1) - mu * mu)
    dls = dlogp * (((act - mu) ** 2) / var - f32(1))      # d/d sigma_param
    grads = OrderedDict()
    grads["dist_net.sigma_param"] = dls.sum(axis=0, dtype=f32).reshape(-1, 1)
    grads["dist_net.mu.weight"] = nn.mm(dm_raw.T, hs[-1])
    grads["dist_net.mu.bias"] = dm_raw.sum(axis=0, dtype=f32)
    dh = nn.mm(dm_raw, actor["dist_net.mu.weight"])
    Ws, _ = nn.backbone_layers(actor)
    dWs, dbs, _ = nn.mlp_bwd(hs, Ws, dh, need_dx=False, drop_p=p_drop)
    for i, dW, db in zip(nn.backbone_indices(actor), dWs, dbs):
        grads[f"backbone.model.{i}.weight"] = dW
        grads[f"backbone.model.{i}.bias"] = db
    nn.adam_step(actor, grads, state["opt"]["actor"], cfg["actor_lr"])
    aux["exp_a"], aux["logp"] = exp_a, logp
    aux["actor_grads"] = grads

    nn.polyak(q1o, q1n, cfg["tau"])
    nn.polyak(q2o, q2n, cfg["tau"])
    result = OrderedDict([("loss/actor", float(actor_loss)), ("loss/q1", float(q1_loss)), ("loss/q2", float(q2_loss)),
                          ("loss/v", float(v_loss))])
    return result, aux

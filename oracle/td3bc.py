"""Oracle restatement of TD3BCPolicy.learn (policy/model_free/td3bc.py:83-124 on top of td3.py:16-59).
TEST INFRASTRUCTURE ONLY.

Nets: actor = Actor(MLP, action_dim) (actor_module.py:30-51: max_action * tanh(last(backbone(obs)))),
critic1/critic2, targets actor_old / critic1_old / critic2_old.  state["cnt"] is TD3Policy._cnt,
state["last_actor_loss"] is _last_actor_loss.  Noise: eps_target (B,A) ~ N(0,1) (torch.randn_like, td3bc.py:90).
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Dict

import numpy as np

from . import nn
from .nn import f32


def default_cfg(obs_dim: int, act_dim: int) -> dict:
    """run_example/run_td3bc.py:28-45."""
    return dict(obs_dim=obs_dim, act_dim=act_dim, hidden=[256, 256], actor_lr=3e-4, critic_lr=3e-4, gamma=0.99, tau=0.005,
                policy_noise=0.2, noise_clip=0.5, update_actor_freq=2, alpha=2.5, max_action=1.0)


def init_opt(state: dict) -> None:
    state["opt"] = {k: nn.adam_init(None) for k in ("actor", "critic1", "critic2")}
    state.setdefault("cnt", 0)
    state.setdefault("last_actor_loss", 0.0)


def det_actor_fwd(net, obs, max_action):
    Ws, bs = nn.backbone_layers(net)
    hs = nn.mlp_fwd(obs, Ws, bs)
    m_raw = nn.mm(hs[-1], net["last.weight"].T) + net["last.bias"]
    return (f32(max_action) * np.tanh(m_raw)).astype(f32), hs


def learn(state: dict, cfg: dict, batch: Dict[str, np.ndarray], noise: Dict[str, np.ndarray]):
    obs = np.asarray(batch["observations"], f32)
    act = np.asarray(batch["actions"], f32)
    nobs = np.asarray(batch["next_observations"], f32)
    rew = np.asarray(batch["rewards"], f32).reshape(-1, 1)
    term = np.asarray(batch["terminals"], f32).reshape(-1, 1)
    B, A = act.shape
    od = obs.shape[1]
    ma = f32(cfg["max_action"])
    actor, c1, c2 = state["actor"], state["critic1"], state["critic2"]
    ao, c1o, c2o = state["actor_old"], state["critic1_old"], state["critic2_old"]
    aux = {}

    # ---- critics (td3bc.py:88-104) ----
    q1, h1 = nn.critic_fwd(c1, obs, act)
    q2, h2 = nn.critic_fwd(c2, obs, act)
    nz = np.clip(np.asarray(noise["eps_target"], f32) * f32(cfg["policy_noise"]), -f32(cfg["noise_clip"]), f32(cfg["noise_clip"]))
    na, _ = det_actor_fwd(ao, nobs, ma)
    na = np.clip(na + nz, -ma, ma).astype(f32)
    nq1, _ = nn.critic_fwd(c1o, nobs, na)
    nq2, _ = nn.critic_fwd(c2o, nobs, na)
    target_q = (rew + f32(cfg["gamma"]) * (f32(1) - term) * np.minimum(nq1, nq2)).astype(f32)
    l1 = f32(((q1 - target_q) ** 2).mean(dtype=f32))
    l2 = f32(((q2 - target_q) ** 2).mean(dtype=f32))
    for name, net, qq, hh in (("critic1", c1, q1, h1), ("critic2", c2, q2, h2)):
        g, _ = nn.critic_bwd(net, hh, (f32(2) * (qq - target_q) / f32(B)).astype(f32), need_dx=False)
        nn.adam_step(net, g, state["opt"][name], cfg["critic_lr"])
        aux[name + "_grads"] = g
    aux.update(q1=q1, q2=q2, target_q=target_q)

    # ---- delayed actor + target sync (td3bc.py:106-116) ----
    if state["cnt"] % cfg["update_actor_freq"] == 0:
        a, hs = det_actor_fwd(actor, obs, ma)
        q, hq = nn.critic_fwd(c1, obs, a)                     # UPDATED critic1
        lmbda = f32(cfg["alpha"]) / f32(np.abs(q).mean(dtype=f32))
        actor_loss = f32(-lmbda * q.mean(dtype=f32) + ((a - act) ** 2).mean(dtype=f32))
        dq = np.full((B, 1), -lmbda / f32(B), dtype=f32)
        _, dx = nn.critic_bwd(c1, hq, dq, need_dx=True, need_dw=False)
        da = dx[:, od:] + f32(2) * (a - act) / f32(B * A)
        t = a / ma                                            # tanh(m_raw)
        dm_raw = (da * ma * (f32(1) - t * t)).astype(f32)
        grads = OrderedDict()
        grads["last.weight"] = nn.mm(dm_raw.T, hs[-1])
        grads["last.bias"] = dm_raw.sum(axis=0, dtype=f32)
        dh = nn.mm(dm_raw, actor["last.weight"])
        Ws, _ = nn.backbone_layers(actor)
        dWs, dbs, _ = nn.mlp_bwd(hs, Ws, dh, need_dx=False)
        for l, (dW, db) in enumerate(zip(dWs, dbs)):
            grads[f"backbone.model.{2 * l}.weight"] = dW
            grads[f"backbone.model.{2 * l}.bias"] = db
        nn.adam_step(actor, grads, state["opt"]["actor"], cfg["actor_lr"])
        state["last_actor_loss"] = float(actor_loss)
        nn.polyak(ao, actor, cfg["tau"])
        nn.polyak(c1o, c1, cfg["tau"])
        nn.polyak(c2o, c2, cfg["tau"])
        aux["q_pi"] = q
        aux["actor_grads"] = grads
    state["cnt"] += 1
    result = OrderedDict([("loss/actor", float(state["last_actor_loss"])), ("loss/critic1", float(l1)), ("loss/critic2", float(l2))])
    return result, aux

"""Oracle restatement of CQLPolicy.learn (policy/model_free/cql.py:87-207,
inheriting policy/model_free/sac.py:60-77).  TEST INFRASTRUCTURE ONLY.

State layout (plain dicts of fp32 arrays, key names = reference state_dict keys):
  state["actor"|"critic1"|"critic2"|"critic1_old"|"critic2_old"] : name -> array
  state["log_alpha"], state["cql_log_alpha"]                     : (1,) arrays
  state["opt"][name] : Adam state for actor/critic1/critic2/alpha/cql_alpha
Noise (in the reference's draw order):
  eps_actor (B,A) N(0,1) ; eps_next (B,A) [or (B*N,A) with max_q_backup] ;
  u_rand (B*N,A) U[low,high) ; eps_pi (B*N,A) ; eps_next_pi (B*N,A)
"""
from __future__ import annotations

import math
from collections import OrderedDict
from typing import Dict

import numpy as np

from . import nn
from .nn import f32


def default_cfg(obs_dim: int, act_dim: int) -> dict:
    """Hyper-parameters of run_example/run_cql.py:26-56 (north-star variant:
    hidden [256,256])."""
    return dict(
        obs_dim=obs_dim, act_dim=act_dim, hidden=[256, 256],
        actor_lr=1e-4, critic_lr=3e-4, alpha_lr=1e-4, cql_alpha_lr=3e-4,
        gamma=0.99, tau=0.005, auto_alpha=True, alpha=0.2, target_entropy=-float(act_dim),
        cql_weight=5.0, temperature=1.0, max_q_backup=False, deterministic_backup=True,
        with_lagrange=False, lagrange_threshold=10.0, num_repeat_actions=10,
        act_low=-1.0, act_high=1.0,
    )


def init_opt(state: dict) -> None:
    state["opt"] = {k: nn.adam_init(None) for k in ("actor", "critic1", "critic2", "alpha", "cql_alpha")}


def current_alpha(state: dict, cfg: dict) -> np.float32:
    if cfg["auto_alpha"]:
        return f32(np.exp(state["log_alpha"][0]))      # cql.py:106 — NOT clamped
    return f32(cfg["alpha"])


def learn(state: dict, cfg: dict, batch: Dict[str, np.ndarray], noise: Dict[str, np.ndarray]):
    """One CQL gradient step.  Mutates ``state``; returns (result dict, aux dict)."""
    obs = np.asarray(batch["observations"], f32)
    act = np.asarray(batch["actions"], f32)
    nobs = np.asarray(batch["next_observations"], f32)
    rew = np.asarray(batch["rewards"], f32).reshape(-1, 1)
    term = np.asarray(batch["terminals"], f32).reshape(-1, 1)
    B = obs.shape[0]
    N = cfg["num_repeat_actions"]
    A = act.shape[1]
    # COMBOPolicy.learn (policy/model_based/combo.py:110-241) = this update on the concatenated real + model batch, with the
    # conservative term's repeated rows taken from rows [c0, c0 + Bc) ("model": the model part, "mix": everything, combo.py:168-171)
    # and its data term -w mean Q from the first Br (real) rows only (:198-199)
    c0, Bc = cfg.get("cons_rows", (0, B))
    Br = cfg.get("real_rows", B)
    actor, c1, c2 = state["actor"], state["critic1"], state["critic2"]
    c1o, c2o = state["critic1_old"], state["critic2_old"]
    aux = {}

    # ---- actor update (cql.py:92-98) -------------------------------------
    alpha = current_alpha(state, cfg)          # sac.py:46 / cql.py:106: _alpha == exp(log_alpha) at all times
    a, logp, cache = nn.tanh_gauss_fwd(actor, obs, noise["eps_actor"])
    q1a, h1a = nn.critic_fwd(c1, obs, a)
    q2a, h2a = nn.critic_fwd(c2, obs, a)
    actor_loss = f32((alpha * logp - np.minimum(q1a, q2a)).mean(dtype=f32))
    g = np.full((B, 1), -1.0 / B, dtype=f32)
    g1, g2 = nn.min2_grad(q1a, q2a, g)
    _, dx1 = nn.critic_bwd(c1, h1a, g1, need_dx=True, need_dw=False)
    _, dx2 = nn.critic_bwd(c2, h2a, g2, need_dx=True, need_dw=False)
    od = obs.shape[1]
    da = dx1[:, od:] + dx2[:, od:]
    dlogp = np.full((B, 1), alpha / f32(B), dtype=f32)
    agrads = nn.tanh_gauss_bwd(actor, cache, da, dlogp)
    nn.adam_step(actor, agrads, state["opt"]["actor"], cfg["actor_lr"])
    aux["actor_grads"] = agrads
    aux["q1a"], aux["q2a"], aux["logp"] = q1a, q2a, logp

    # ---- alpha update (cql.py:100-106) -----------------------------------
    result = OrderedDict()
    if cfg["auto_alpha"]:
        lp_t = logp + f32(cfg["target_entropy"])
        la = state["log_alpha"]
        alpha_loss = f32(-(la[0] * lp_t).mean(dtype=f32))
        dla = np.array([-(lp_t.mean(dtype=f32))], dtype=f32)
        nn.adam_step({"log_alpha": la}, {"log_alpha": dla}, state["opt"]["alpha"], cfg["alpha_lr"])
        alpha = f32(np.exp(la[0]))

    # ---- TD target (cql.py:108-132), uses the UPDATED actor -----------------
    if cfg["max_q_backup"]:
        tmp_nobs = np.repeat(nobs, N, axis=0)
        na, _, _ = nn.tanh_gauss_fwd(actor, tmp_nobs, noise["eps_next"])
        nq1, _ = nn.critic_fwd(c1o, tmp_nobs, na)
        nq2, _ = nn.critic_fwd(c2o, tmp_nobs, na)
        nq1 = nq1.reshape(B, N, 1).max(axis=1)
        nq2 = nq2.reshape(B, N, 1).max(axis=1)
        next_q = np.minimum(nq1, nq2)
    else:
        na, nlogp, _ = nn.tanh_gauss_fwd(actor, nobs, noise["eps_next"])
        nq1, _ = nn.critic_fwd(c1o, nobs, na)
        nq2, _ = nn.critic_fwd(c2o, nobs, na)
        next_q = np.minimum(nq1, nq2)
        if not cfg["deterministic_backup"]:
            next_q = next_q - alpha * nlogp
    target_q = (rew + f32(cfg["gamma"]) * (f32(1) - term) * next_q).astype(f32)
    q1, h1 = nn.critic_fwd(c1, obs, act)
    q2, h2 = nn.critic_fwd(c2, obs, act)
    td1 = f32(((q1 - target_q) ** 2).mean(dtype=f32))
    td2 = f32(((q2 - target_q) ** 2).mean(dtype=f32))

    # ---- conservative term (cql.py:137-168) ---------------------------------
    tmp_obs = np.repeat(obs[c0:c0 + Bc], N, axis=0)              # row b*N+n  (cql.py:142-144)
    tmp_nobs = np.repeat(nobs[c0:c0 + Bc], N, axis=0)
    a_pi, lp_pi, _ = nn.tanh_gauss_fwd(actor, tmp_obs, noise["eps_pi"])
    a_npi, lp_npi, _ = nn.tanh_gauss_fwd(actor, tmp_nobs, noise["eps_next_pi"])
    u_rand = np.asarray(noise["u_rand"], f32)
    T = f32(cfg["temperature"])
    w = f32(cfg["cql_weight"])
    log_rand = f32(np.log(0.5 ** A))                 # cql.py:82
    crit = []
    for (c, q, hq) in ((c1, q1, h1), (c2, q2, h2)):
        qp, hp = nn.critic_fwd(c, tmp_obs, a_pi)     # critic sees tmp_obs in all three (cql.py:149-151)
        qn, hn = nn.critic_fwd(c, tmp_obs, a_npi)
        qr, hr = nn.critic_fwd(c, tmp_obs, u_rand)
        cat = np.concatenate([qp - lp_pi, qn - lp_npi, qr - log_rand], axis=1)   # (B*N, 3): reshape at :153-157 is a no-op
        z = cat / T
        zmax = z.max(axis=1, keepdims=True)
        ez = np.exp(z - zmax)
        se = ez.sum(axis=1, keepdims=True, dtype=f32)
        lse = (np.log(se) + zmax).astype(f32)
        soft = (ez / se).astype(f32)
        cons = f32(lse.mean(dtype=f32) * w * T - q[:Br].mean(dtype=f32) * w)
        crit.append(dict(c=c, q=q, hq=hq, hp=hp, hn=hn, hr=hr, soft=soft, cons=cons, cat=cat))
    aux["cat_q1"], aux["cat_q2"] = crit[0]["cat"], crit[1]["cat"]

    cons_scale = f32(1.0)
    if cfg["with_lagrange"]:
        cla = state["cql_log_alpha"]
        e = f32(np.exp(cla[0]))
        cql_alpha = f32(min(max(e, f32(0.0)), f32(1e6)))
        raw = [cr["cons"] - f32(cfg["lagrange_threshold"]) for cr in crit]
        for cr, r in zip(crit, raw):
            cr["cons"] = f32(cql_alpha * r)
        cql_alpha_loss = f32(-(crit[0]["cons"] + crit[1]["cons"]) * f32(0.5))
        gate = f32(1.0) if (e >= 0.0 and e <= 1e6) else f32(0.0)
        dcla = np.array([-(raw[0] + raw[1]) * f32(0.5) * e * gate], dtype=f32)
        nn.adam_step({"cql_log_alpha": cla}, {"cql_log_alpha": dcla}, state["opt"]["cql_alpha"], cfg["cql_alpha_lr"])
        cons_scale = cql_alpha                        # critics use the pre-step value (cql.py:170-178)

    # ---- critic updates (cql.py:180-190) ------------------------------------
    BN = Bc * N
    real = (np.arange(B) < Br).astype(f32).reshape(-1, 1)
    losses = []
    for name, cr, td in (("critic1", crit[0], td1), ("critic2", crit[1], td2)):
        c, q = cr["c"], cr["q"]
        losses.append(f32(td + cr["cons"]))
        dq = (f32(2.0) * (q - target_q) / f32(B) - real * (cons_scale * w / f32(Br))).astype(f32)
        dv = (cons_scale * w / f32(BN)) * cr["soft"]          # (BN,3); dv/dq = 1
        grads = None
        for hs, d in ((cr["hq"], dq), (cr["hp"], dv[:, 0:1]), (cr["hn"], dv[:, 1:2]), (cr["hr"], dv[:, 2:3])):
            g_part, _ = nn.critic_bwd(c, hs, np.ascontiguousarray(d, dtype=f32), need_dx=False)
            if grads is None:
                grads = g_part
            else:
                for k in grads:
                    grads[k] = grads[k] + g_part[k]
        aux[name + "_grads"] = grads
        nn.adam_step(c, grads, state["opt"][name], cfg["critic_lr"])

    # ---- Polyak (sac.py:60-64) ----------------------------------------------
    nn.polyak(c1o, c1, cfg["tau"])
    nn.polyak(c2o, c2, cfg["tau"])

    result["loss/actor"] = float(actor_loss)
    result["loss/critic1"] = float(losses[0])
    result["loss/critic2"] = float(losses[1])
    if cfg["auto_alpha"]:
        result["loss/alpha"] = float(alpha_loss)
        result["alpha"] = float(alpha)
    if cfg["with_lagrange"]:
        result["loss/cql_alpha"] = float(cql_alpha_loss)
        result["cql_alpha"] = float(cql_alpha)
    aux.update(q1=q1, q2=q2, target_q=target_q)
    return result, aux

"""Oracle restatement of MCQPolicy.learn (policy/model_free/mcq.py:48-126 over sac.py, behaviour policy = nets/vae.py:8-66).
TEST INFRASTRUCTURE ONLY.

State: state["actor"|"critic1"|"critic2"|"critic1_old"|"critic2_old"] as in oracle/sac.py, state["behavior_policy"] = the VAE's
state_dict (e1, e2, mean, log_std, d1, d2, d3: nn.Linear weights (out,in) / biases), state["log_alpha"], state["opt"].
Noise in the reference's draw order:
  eps_vae   (B, Z)        torch.randn_like(std) in VAE.forward (vae.py:49)
  eps_next  (B, A)        actforward(next_obss) (mcq.py:63)
  z_ood     (2B*N, Z)     torch.randn(...) in VAE.decode, clamped to [-0.5, 0.5] there (vae.py:57-58)
  eps_ood   (2B, A)       actforward(s_in) (mcq.py:80)
  eps_actor (B, A)        actforward(obss) (mcq.py:96)
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Dict

import numpy as np

from . import nn
from .nn import f32

LS_MIN, LS_MAX = f32(-4.0), f32(15.0)


def default_cfg(obs_dim: int, act_dim: int) -> dict:
    """run_example/run_mcq.py:25-36"""
    return dict(obs_dim=obs_dim, act_dim=act_dim, hidden=[400, 400], vae_hidden=750, latent_dim=2 * act_dim, actor_lr=3e-4, critic_lr=3e-4,
                alpha_lr=3e-4, behavior_policy_lr=1e-3, gamma=0.99, tau=0.005, auto_alpha=True, alpha=0.2, target_entropy=-float(act_dim),
                lmbda=0.9, num_sampled_actions=10, max_action=1.0)


def init_opt(state: dict) -> None:
    state["opt"] = {k: nn.adam_init(None) for k in ("actor", "critic1", "critic2", "alpha", "behavior_policy")}


def vae_decode(vae, obs, z, max_action):
    """VAE.decode (vae.py:55-62) with an explicit latent"""
    y = np.concatenate([obs, z], axis=1).astype(f32)
    hs = nn.mlp_fwd(y, [vae["d1.weight"], vae["d2.weight"]], [vae["d1.bias"], vae["d2.bias"]])
    m3 = nn.mm(hs[-1], vae["d3.weight"].T) + vae["d3.bias"]
    return (f32(max_action) * np.tanh(m3)).astype(f32), hs, m3


def vae_update(vae, opt, obs, act, eps, lr, max_action):
    """mcq.py:52-60: recon + KL loss, one Adam step on every VAE parameter.  Returns the loss."""
    B, A = act.shape
    x = np.concatenate([obs, act], axis=1).astype(f32)
    eh = nn.mlp_fwd(x, [vae["e1.weight"], vae["e2.weight"]], [vae["e1.bias"], vae["e2.bias"]])
    mean = nn.mm(eh[-1], vae["mean.weight"].T) + vae["mean.bias"]
    ls_raw = nn.mm(eh[-1], vae["log_std.weight"].T) + vae["log_std.bias"]
    ls = np.clip(ls_raw, LS_MIN, LS_MAX)
    std = np.exp(ls)
    z = (mean + std * np.asarray(eps, f32)).astype(f32)
    u, dh, m3 = vae_decode(vae, obs, z, max_action)
    Z = mean.shape[1]
    recon = f32(((u - act) ** 2).mean(dtype=f32))
    kl = f32(-0.5) * f32((f32(1) + np.log(std * std) - mean * mean - std * std).mean(dtype=f32))
    loss = f32(recon + kl)
    # ---- backward ----
    du = (f32(2) * (u - act) / f32(B * A)).astype(f32)
    t = u / f32(max_action)
    dm3 = (du * f32(max_action) * (f32(1) - t * t)).astype(f32)
    g = OrderedDict()
    g["d3.weight"] = nn.mm(dm3.T, dh[-1]); g["d3.bias"] = dm3.sum(axis=0, dtype=f32)
    dWs, dbs, dy = nn.mlp_bwd(dh, [vae["d1.weight"], vae["d2.weight"]], nn.mm(dm3, vae["d3.weight"]), need_dx=True)
    g["d1.weight"], g["d1.bias"], g["d2.weight"], g["d2.bias"] = dWs[0], dbs[0], dWs[1], dbs[1]
    dz = dy[:, obs.shape[1]:]
    dmean = (dz + mean / f32(B * Z)).astype(f32)
    dls = (dz * std * np.asarray(eps, f32) + (std * std - f32(1)) / f32(B * Z)).astype(f32)
    dls_raw = dls * ((ls_raw >= LS_MIN) & (ls_raw <= LS_MAX))
    g["mean.weight"] = nn.mm(dmean.T, eh[-1]); g["mean.bias"] = dmean.sum(axis=0, dtype=f32)
    g["log_std.weight"] = nn.mm(dls_raw.T, eh[-1]); g["log_std.bias"] = dls_raw.sum(axis=0, dtype=f32)
    deh = nn.mm(dmean, vae["mean.weight"]) + nn.mm(dls_raw, vae["log_std.weight"])
    dWs, dbs, _ = nn.mlp_bwd(eh, [vae["e1.weight"], vae["e2.weight"]], deh, need_dx=False)
    g["e1.weight"], g["e1.bias"], g["e2.weight"], g["e2.bias"] = dWs[0], dbs[0], dWs[1], dbs[1]
    nn.adam_step(vae, g, opt, lr)
    return loss, g, u


def learn(state: dict, cfg: dict, batch: Dict[str, np.ndarray], noise: Dict[str, np.ndarray]):
    obs = np.asarray(batch["observations"], f32)
    act = np.asarray(batch["actions"], f32)
    nobs = np.asarray(batch["next_observations"], f32)
    rew = np.asarray(batch["rewards"], f32).reshape(-1, 1)
    term = np.asarray(batch["terminals"], f32).reshape(-1, 1)
    B, od = obs.shape
    N, lam, ma = cfg["num_sampled_actions"], f32(cfg["lmbda"]), cfg["max_action"]
    actor, c1, c2, c1o, c2o, vae = (state[k] for k in ("actor", "critic1", "critic2", "critic1_old", "critic2_old", "behavior_policy"))
    alpha = state.get("_alpha", f32(np.exp(state["log_alpha"][0])) if cfg["auto_alpha"] else f32(cfg["alpha"]))
    aux = {}

    # ---- behaviour policy (mcq.py:52-60) ----
    vae_loss, vg, recon = vae_update(vae, state["opt"]["behavior_policy"], obs, act, noise["eps_vae"], cfg["behavior_policy_lr"], ma)
    aux["vae_grads"], aux["recon"] = vg, recon

    # ---- critics (mcq.py:62-94) ----
    na, nlogp, _ = nn.tanh_gauss_fwd(actor, nobs, noise["eps_next"])
    nq = np.minimum(nn.critic_fwd(c1o, nobs, na)[0], nn.critic_fwd(c2o, nobs, na)[0]) - alpha * nlogp
    y_in = (rew + f32(cfg["gamma"]) * (f32(1) - term) * nq).astype(f32)
    s_in = np.concatenate([obs, nobs], axis=0)
    s_rep = np.repeat(s_in, N, axis=0)
    z = np.clip(np.asarray(noise["z_ood"], f32), f32(-0.5), f32(0.5))
    sampled, _, _ = vae_decode(vae, s_rep, z, ma)                       # the UPDATED VAE
    t1 = nn.critic_fwd(c1o, s_rep, sampled)[0].reshape(2 * B, N).max(axis=1).reshape(-1, 1)
    t2 = nn.critic_fwd(c2o, s_rep, sampled)[0].reshape(2 * B, N).max(axis=1).reshape(-1, 1)
    y_ood = np.minimum(t1, t2).astype(f32)
    a_ood, _, _ = nn.tanh_gauss_fwd(actor, s_in, noise["eps_ood"])
    losses = []
    for name, c in (("critic1", c1), ("critic2", c2)):
        q_in, h_in = nn.critic_fwd(c, obs, act)
        q_ood, h_ood = nn.critic_fwd(c, s_in, a_ood)
        l_in = f32(((q_in - y_in) ** 2).mean(dtype=f32))
        l_ood = f32(((q_ood - y_ood) ** 2).mean(dtype=f32))
        losses.append(f32(lam * l_in + (f32(1) - lam) * l_ood))
        g1, _ = nn.critic_bwd(c, h_in, (lam * f32(2) * (q_in - y_in) / f32(B)).astype(f32), need_dx=False)
        g2, _ = nn.critic_bwd(c, h_ood, ((f32(1) - lam) * f32(2) * (q_ood - y_ood) / f32(2 * B)).astype(f32), need_dx=False)
        grads = OrderedDict((k, g1[k] + g2[k]) for k in g1)
        nn.adam_step(c, grads, state["opt"][name], cfg["critic_lr"])
        aux[name + "_grads"] = grads
        if name == "critic1":
            aux["q1"], aux["q1_ood"] = q_in, q_ood
    aux.update(target_q=y_in, target_q_ood=y_ood, sampled_actions=sampled)

    # ---- actor against the UPDATED critics, temperature clamped to [0, 1] (mcq.py:96-111) ----
    a, logp, cache = nn.tanh_gauss_fwd(actor, obs, noise["eps_actor"])
    q1a, h1a = nn.critic_fwd(c1, obs, a)
    q2a, h2a = nn.critic_fwd(c2, obs, a)
    actor_loss = f32(-np.minimum(q1a, q2a).mean(dtype=f32) + alpha * logp.mean(dtype=f32))
    g1, g2 = nn.min2_grad(q1a, q2a, np.full((B, 1), -1.0 / B, dtype=f32))
    _, dx1 = nn.critic_bwd(c1, h1a, g1, need_dx=True, need_dw=False)
    _, dx2 = nn.critic_bwd(c2, h2a, g2, need_dx=True, need_dw=False)
    agr = nn.tanh_gauss_bwd(actor, cache, dx1[:, od:] + dx2[:, od:], np.full((B, 1), alpha / f32(B), dtype=f32))
    nn.adam_step(actor, agr, state["opt"]["actor"], cfg["actor_lr"])
    aux["q1a"] = q1a

    result = OrderedDict([("loss/actor", float(actor_loss)), ("loss/critic1", float(losses[0])), ("loss/critic2", float(losses[1])),
                          ("loss/behavior_policy", float(vae_loss))])
    if cfg["auto_alpha"]:
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

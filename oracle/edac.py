"""Oracle restatement of EDACPolicy.learn (policy/model_free/edac.py:88-166) with the ensemble critic of
modules/ensemble_critic_module.py:11-44 / nets/ensemble_linear.py:9-41 (y = x W + b, W (K,in,out)).
TEST INFRASTRUCTURE ONLY.

The gradient-diversity term (edac.py:136-149) needs a double backward in the reference.  Here it is
restated analytically: with ReLU masks m_l held fixed,
    delta_L = w_last (.) m_L ; delta_{l-1} = (delta_l W_l^T) (.) m_{l-1} ; g = (delta_1 W_1^T)[action rows]
is linear in every weight, so with gamma = d(eta*L_g)/dg the adjoint sweep is a masked *forward* pass
    t_0 = gamma (action rows) ; t_l = (t_{l-1} W_l) (.) m_l
and  dW_1[action rows] += gamma^T delta_1 ; dW_l += t_{l-1}^T delta_l ; dw_last += sum_b t_L ; biases get none.
Noise: eps_actor (B,A), eps_next (B,A) [or (10B,A) with max_q_backup].
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Dict

import numpy as np

from . import nn
from .nn import f32


def default_cfg(obs_dim: int, act_dim: int) -> dict:
    """run_example/run_edac.py:35-60 (num_critics / eta per task: :24-31)."""
    return dict(obs_dim=obs_dim, act_dim=act_dim, hidden=[256, 256, 256], actor_lr=1e-4, critic_lr=3e-4, alpha_lr=1e-4,
                gamma=0.99, tau=0.005, auto_alpha=True, alpha=0.2, target_entropy=-float(act_dim), num_critics=10,
                max_q_backup=False, deterministic_backup=False, eta=1.0)


def init_opt(state: dict) -> None:
    state["opt"] = {k: nn.adam_init(None) for k in ("actor", "critics", "alpha")}


def ens_layers(net):
    Ws, bs = [], []
    i = 0
    while f"model.{i}.weight" in net:
        Ws.append(net[f"model.{i}.weight"]); bs.append(net[f"model.{i}.bias"])
        i += 2
    return Ws, bs


def _bmm(x, W):
    """x (K,B,i) @ W (K,i,o) through the active matmul mode."""
    return np.stack([nn.mm(x[k], W[k]) for k in range(W.shape[0])]).astype(f32)


def ens_fwd(net, x):
    """x: (B,in) broadcast to all members, or (K,B,in).  Returns q (K,B,1), hs = [x_k, h_1, .., h_L] (K,B,.)."""
    Ws, bs = ens_layers(net)
    K = Ws[0].shape[0]
    h = np.broadcast_to(x, (K,) + x.shape).astype(f32) if x.ndim == 2 else np.asarray(x, f32)
    hs = [h]
    for l, (W, b) in enumerate(zip(Ws, bs)):
        z = _bmm(hs[-1], W) + b
        if l == len(Ws) - 1:
            return z.astype(f32), hs
        hs.append(np.maximum(z, f32(0)))


def ens_bwd(net, hs, dq, need_dx, need_dw=True):
    Ws, _ = ens_layers(net)
    n = len(Ws)
    grads = OrderedDict()
    dz = dq                                                      # (K,B,1)
    dx = None
    for l in reversed(range(n)):
        if need_dw:
            grads[f"model.{2 * l}.weight"] = _bmm(np.transpose(hs[l], (0, 2, 1)), dz)
            grads[f"model.{2 * l}.bias"] = dz.sum(axis=1, keepdims=True, dtype=f32)
        if l > 0 or need_dx:
            dh = _bmm(dz, np.transpose(Ws[l], (0, 2, 1)))
            if l > 0:
                dz = dh * (hs[l] > 0)
            else:
                dx = dh
    return (grads if need_dw else None), dx


def current_alpha(state, cfg):
    if cfg["auto_alpha"]:
        return f32(min(max(np.exp(state["log_alpha"][0]), 0.0), 1.0))   # edac.py:110 (and exp(log_alpha) at init, :45)
    return f32(cfg["alpha"])


def learn(state: dict, cfg: dict, batch: Dict[str, np.ndarray], noise: Dict[str, np.ndarray]):
    obs = np.asarray(batch["observations"], f32)
    act = np.asarray(batch["actions"], f32)
    nobs = np.asarray(batch["next_observations"], f32)
    rew = np.asarray(batch["rewards"], f32).reshape(-1, 1)
    term = np.asarray(batch["terminals"], f32).reshape(-1, 1)
    B, A = act.shape
    od = obs.shape[1]
    actor, crit, crit_old = state["actor"], state["critics"], state["critics_old"]
    Ws, _ = ens_layers(crit)
    K = Ws[0].shape[0]
    eta = f32(cfg["eta"])
    aux = {}
    # the reference's _alpha is exp(log_alpha) un-clamped until the first alpha step (edac.py:45), clamped after (:110)
    alpha = state.get("_alpha", f32(np.exp(state["log_alpha"][0])) if cfg["auto_alpha"] else f32(cfg["alpha"]))

    # ---- actor (edac.py:96-102) ----
    a, logp, cache = nn.tanh_gauss_fwd(actor, obs, noise["eps_actor"])
    qas, hqa = ens_fwd(crit, np.concatenate([obs, a], axis=1))
    qmin = qas.min(axis=0)
    actor_loss = f32(-qmin.mean(dtype=f32) + alpha * logp.mean(dtype=f32))
    sel = qas.argmin(axis=0)                                     # (B,1): torch.min(dim=0) routes grad to the returned index
    dq = np.zeros_like(qas)
    np.put_along_axis(dq, sel[None], f32(-1.0 / B), axis=0)
    _, dx = ens_bwd(crit, hqa, dq, need_dx=True, need_dw=False)
    da = dx[:, :, od:].sum(axis=0, dtype=f32)
    agr = nn.tanh_gauss_bwd(actor, cache, da, np.full((B, 1), alpha / f32(B), dtype=f32))
    nn.adam_step(actor, agr, state["opt"]["actor"], cfg["actor_lr"])
    aux["qas"] = qas
    aux["actor_grads"] = agr

    result = OrderedDict()
    if cfg["auto_alpha"]:
        lp_t = logp + f32(cfg["target_entropy"])
        la = state["log_alpha"]
        alpha_loss = f32(-(la[0] * lp_t).mean(dtype=f32))
        nn.adam_step({"log_alpha": la}, {"log_alpha": np.array([-(lp_t.mean(dtype=f32))], f32)}, state["opt"]["alpha"], cfg["alpha_lr"])
        alpha = f32(min(max(np.exp(la[0]), f32(0.0)), f32(1.0)))
        state["_alpha"] = alpha

    # ---- target (edac.py:112-131) ----
    if cfg["max_q_backup"]:
        tmp = np.repeat(nobs, 10, axis=0)
        na, _, _ = nn.tanh_gauss_fwd(actor, tmp, noise["eps_next"])
        nq, _ = ens_fwd(crit_old, np.concatenate([tmp, na], axis=1))
        next_q = nq.reshape(K, B, 10, 1).max(axis=2).min(axis=0)
    else:
        na, nlogp, _ = nn.tanh_gauss_fwd(actor, nobs, noise["eps_next"])
        nq, _ = ens_fwd(crit_old, np.concatenate([nobs, na], axis=1))
        next_q = nq.min(axis=0)
        if not cfg["deterministic_backup"]:
            next_q = next_q - alpha * nlogp
    target_q = (rew + f32(cfg["gamma"]) * (f32(1) - term) * next_q).astype(f32)

    # ---- critics (edac.py:133-153) ----
    x = np.concatenate([obs, act], axis=1)
    qs, hq = ens_fwd(crit, x)
    diff = qs - target_q[None]
    critics_loss = f32((diff ** 2).mean(axis=(1, 2), dtype=f32).sum(dtype=f32))
    grads, _ = ens_bwd(crit, hq, (f32(2) * diff / f32(B)).astype(f32), need_dx=False)
    aux["qs"], aux["target_q"] = qs, target_q
    if cfg["eta"] > 0:
        n = len(Ws)
        L = n - 1
        # unit-seed backward: delta_l = dq_k/dz_l
        deltas = [None] * (L + 1)
        deltas[L] = (np.transpose(Ws[L], (0, 2, 1)) * (hq[L] > 0)).astype(f32)         # (K,B,H)
        for l in range(L, 1, -1):
            deltas[l - 1] = (_bmm(deltas[l], np.transpose(Ws[l - 1], (0, 2, 1))) * (hq[l - 1] > 0)).astype(f32)
        W1a = Ws[0][:, od:, :]                                                           # (K,A,H)
        g = _bmm(deltas[1], np.transpose(W1a, (0, 2, 1)))                                 # (K,B,A)
        nrm = np.sqrt((g * g).sum(axis=2, keepdims=True, dtype=f32)).astype(f32)
        nk = nrm + f32(1e-10)
        gh = g / nk
        S = gh.sum(axis=0, keepdims=True, dtype=f32)
        gram_off = (S * S).sum(axis=2, dtype=f32)[0] - (gh * gh).sum(axis=2, dtype=f32).sum(axis=0, dtype=f32)   # (B,)
        grad_loss = f32(gram_off.mean(dtype=f32) / f32(K - 1))
        critics_loss = f32(critics_loss + eta * grad_loss)
        c = (eta * f32(2.0) / f32((K - 1) * B)) * (S - gh)                               # d(eta L_g)/d g_hat
        safe = np.where(nrm > 0, nrm, f32(1))
        gamma = (c / nk - g * ((g * c).sum(axis=2, keepdims=True, dtype=f32) / (nk * nk * safe))).astype(f32)
        # adjoint sweep
        gw = np.zeros_like(Ws[0]); gw[:, od:, :] = _bmm(np.transpose(gamma, (0, 2, 1)), deltas[1])
        grads["model.0.weight"] = grads["model.0.weight"] + gw
        t = (_bmm(gamma, W1a) * (hq[1] > 0)).astype(f32)
        for l in range(1, L):
            grads[f"model.{2 * l}.weight"] = grads[f"model.{2 * l}.weight"] + _bmm(np.transpose(t, (0, 2, 1)), deltas[l + 1])
            t = (_bmm(t, Ws[l]) * (hq[l + 1] > 0)).astype(f32)
        grads[f"model.{2 * L}.weight"] = grads[f"model.{2 * L}.weight"] + t.sum(axis=1, dtype=f32)[:, :, None]
        aux["grad_loss"], aux["g"] = grad_loss, g
    nn.adam_step(crit, grads, state["opt"]["critics"], cfg["critic_lr"])
    aux["critics_grads"] = grads

    for k in crit_old:                                               # edac.py:62-64 (saved_* shadows included)
        crit_old[k][...] = crit_old[k] * f32(1.0 - cfg["tau"]) + crit[k] * f32(cfg["tau"])
    result["loss/actor"] = float(actor_loss)
    result["loss/critics"] = float(critics_loss)
    if cfg["auto_alpha"]:
        result["loss/alpha"] = float(alpha_loss)
        result["alpha"] = float(alpha)
    return result, aux

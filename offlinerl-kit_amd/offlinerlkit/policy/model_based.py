"""SAC and the model-based callers of the hot path (reference: policy/model_free/sac.py:11-140, policy/model_based/mopo.py:14-84,
policy/model_based/combo.py:12-241) on the HIP engine.

``MOPOPolicy.learn`` and ``COMBOPolicy.learn`` take ``{"real": batch, "fake": batch}``, concatenate real rows first and run the SAC /
CQL-variant update of the engine on the mixed batch.  The dynamics model itself (ensembles, penalties, ``MBPolicyTrainer``) stays out
of scope: ``rollout`` only needs an object with ``step(obs, act) -> (next_obs, reward, terminal, info)``.  These policies are
host-fed (``learn``): the fused device-sampling loop ``learn_n`` draws from ONE replay buffer and does not apply to a real + model pair.
"""
from __future__ import annotations

from collections import defaultdict
from typing import Dict, Tuple, Union

import numpy as np
import torch
import torch.nn as nn

from .. import _engine
from .base_policy import _backbone_dims, clone_target
from .sac_family import CQLPolicy, _TanhGaussPolicy


def _cat(batch: Dict) -> Dict:
    """real rows first, then model rows (mopo.py:81-84 / combo.py:111-113); with a leading run dimension ([R, B, cols] arrays of a
    multi-run policy) the ROW axis is the second to last one"""
    real, fake = batch["real"], batch["fake"]
    out = {}
    for k in real:
        a, b = real[k], fake[k]
        if np.ndim(a) != np.ndim(b) or np.ndim(a) < 2:
            raise ValueError(f"{k}: real batch has shape {tuple(np.shape(a))}, model batch {tuple(np.shape(b))}: both must be [B, cols] or [R, B, cols]")
        out[k] = torch.cat([torch.as_tensor(a), torch.as_tensor(b).to(torch.as_tensor(a).device)], -2) if torch.is_tensor(a) or torch.is_tensor(b) \
            else np.concatenate([np.asarray(a), np.asarray(b)], -2)
    return out


def _rollout(policy, init_obss: np.ndarray, rollout_length: int, uniform: bool) -> Tuple[Dict[str, np.ndarray], Dict]:
    """mopo.py:43-79 / combo.py:67-108: roll the learned dynamics forward from dataset states under the current policy."""
    num_transitions = 0
    rewards_arr = np.array([])
    out = defaultdict(list)
    observations = init_obss
    for _ in range(rollout_length):
        if uniform:
            sp = policy.action_space
            actions = np.random.uniform(sp.low[0], sp.high[0], size=(len(observations), sp.shape[0]))
        else:
            actions = policy.select_action(observations)
        next_observations, rewards, terminals, info = policy.dynamics.step(observations, actions)
        out["obss"].append(observations); out["next_obss"].append(next_observations); out["actions"].append(actions)
        out["rewards"].append(rewards); out["terminals"].append(terminals)
        num_transitions += len(observations)
        rewards_arr = np.append(rewards_arr, rewards.flatten())
        nonterm = (~terminals).flatten()
        if nonterm.sum() == 0:
            break
        observations = next_observations[nonterm]
    return {k: np.concatenate(v, axis=0) for k, v in out.items()}, {"num_transitions": num_transitions, "reward_mean": rewards_arr.mean()}


class SACPolicy(_TanhGaussPolicy):
    """Soft Actor-Critic; constructor = reference SACPolicy.__init__ (sac.py:16-48)."""

    ALGO = "sac"

    def __init__(self, actor: nn.Module, critic1: nn.Module, critic2: nn.Module, actor_optim, critic1_optim, critic2_optim,
                 tau: float = 0.005, gamma: float = 0.99, alpha: Union[float, Tuple] = 0.2) -> None:
        super().__init__()
        self.actor = actor
        self.critic1, self.critic1_old = critic1, clone_target(critic1)
        self.critic2, self.critic2_old = critic2, clone_target(critic2)
        self.actor_optim, self.critic1_optim, self.critic2_optim = actor_optim, critic1_optim, critic2_optim
        self._tau, self._gamma = tau, gamma
        self._init_alpha(alpha)
        if float(critic1_optim.param_groups[0]["lr"]) != float(critic2_optim.param_groups[0]["lr"]):
            raise NotImplementedError("critic1/critic2 must share a learning rate")

    def _nets(self):
        return {_engine.NET_ACTOR: self.actor, _engine.NET_CRITIC1: self.critic1, _engine.NET_CRITIC2: self.critic2,
                _engine.NET_CRITIC1_OLD: self.critic1_old, _engine.NET_CRITIC2_OLD: self.critic2_old}

    def _optims(self):
        o = super()._optims()
        o[_engine.OPT_CRITIC] = self.critic1_optim
        return o

    def _config(self) -> Dict:
        od, hid = _backbone_dims(self.actor.backbone)
        cin, chid = _backbone_dims(self.critic1.backbone)
        ad = self.actor.dist_net.mu.out_features
        self._check_dist_net()
        if chid != hid or cin != od + ad:
            raise NotImplementedError("SAC engine expects actor and critics to share hidden dims")
        c = dict(obs_dim=od, act_dim=ad, hidden=hid, gamma=self._gamma, tau=self._tau,
                 actor_lr=float(self.actor_optim.param_groups[0]["lr"]), critic_lr=float(self.critic1_optim.param_groups[0]["lr"]))
        c.update(self._alpha_config())
        return c


class MOPOPolicy(SACPolicy):
    """Model-based Offline Policy Optimization <Ref: https://arxiv.org/abs/2005.13239>; constructor = mopo.py:19-41."""

    def __init__(self, dynamics, actor, critic1, critic2, actor_optim, critic1_optim, critic2_optim, tau: float = 0.005,
                 gamma: float = 0.99, alpha: Union[float, Tuple] = 0.2) -> None:
        super().__init__(actor, critic1, critic2, actor_optim, critic1_optim, critic2_optim, tau=tau, gamma=gamma, alpha=alpha)
        self.dynamics = dynamics

    def rollout(self, init_obss: np.ndarray, rollout_length: int):
        return _rollout(self, init_obss, rollout_length, False)

    def learn(self, batch: Dict, noise=None) -> Dict[str, float]:
        return super().learn(_cat(batch), noise) if "real" in batch else super().learn(batch, noise)

    def learn_n(self, *a, **k):
        raise NotImplementedError("MOPO mixes a real and a model-rollout buffer per batch: use learn({'real': ..., 'fake': ...})")


class COMBOPolicy(CQLPolicy):
    """Conservative Offline Model-Based Policy Optimization <Ref: https://arxiv.org/abs/2102.08363>; constructor = combo.py:18-65."""

    def __init__(self, dynamics, actor, critic1, critic2, actor_optim, critic1_optim, critic2_optim, action_space, tau: float = 0.005,
                 gamma: float = 0.99, alpha: Union[float, Tuple] = 0.2, cql_weight: float = 1.0, temperature: float = 1.0,
                 max_q_backup: bool = False, deterministic_backup: bool = True, with_lagrange: bool = True,
                 lagrange_threshold: float = 10.0, cql_alpha_lr: float = 1e-4, num_repeart_actions: int = 10,
                 uniform_rollout: bool = False, rho_s: str = "mix") -> None:
        super().__init__(actor, critic1, critic2, actor_optim, critic1_optim, critic2_optim, action_space, tau=tau, gamma=gamma, alpha=alpha,
                         cql_weight=cql_weight, temperature=temperature, max_q_backup=max_q_backup, deterministic_backup=deterministic_backup,
                         with_lagrange=with_lagrange, lagrange_threshold=lagrange_threshold, cql_alpha_lr=cql_alpha_lr,
                         num_repeart_actions=num_repeart_actions)
        if rho_s not in ("model", "mix"):
            raise ValueError("rho_s must be 'model' or 'mix'")
        self.dynamics = dynamics
        self._uniform_rollout = uniform_rollout
        self._rho_s = rho_s
        self._rows = None          # (real rows, model rows) of the bound engine

    def rollout(self, init_obss: np.ndarray, rollout_length: int):
        return _rollout(self, init_obss, rollout_length, self._uniform_rollout)

    def _config(self) -> Dict:
        c = super()._config()
        if self._rows is not None:
            br, bf = self._rows
            c0, bc = (br, bf) if self._rho_s == "model" else (0, br + bf)
            c.update(cql_cons_row0=int(c0), cql_cons_rows=int(bc), cql_real_rows=int(br))
        return c

    def learn(self, batch: Dict, noise=None) -> Dict[str, float]:
        if "real" not in batch:
            raise ValueError("COMBOPolicy.learn expects {'real': batch, 'fake': batch} (combo.py:110-113)")
        rows = (int(batch["real"]["observations"].shape[-2]), int(batch["fake"]["observations"].shape[-2]))
        if rows != self._rows:
            if self._eng is not None:      # the real / model split is part of the engine's row layout: rebuild around the current state
                carried = self._unbind()
                self._rows = rows
                self._rebind_with(carried, rows[0] + rows[1])
            else:
                self._rows = rows
        return super().learn(_cat(batch), noise)

    def learn_n(self, *a, **k):
        raise NotImplementedError("COMBO mixes a real and a model-rollout buffer per batch: use learn({'real': ..., 'fake': ...})")

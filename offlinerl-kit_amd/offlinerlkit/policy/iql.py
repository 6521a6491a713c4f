"""IQL policy (reference: policy/model_free/iql.py:11-139) on the HIP engine."""
from __future__ import annotations

from typing import Dict

import numpy as np
import torch
import torch.nn as nn

from .. import _engine
from .base_policy import EnginePolicy, _backbone_dims, clone_target


class IQLPolicy(EnginePolicy):
    ALGO = "iql"

    def __init__(self, actor: nn.Module, critic_q1: nn.Module, critic_q2: nn.Module, critic_v: nn.Module, actor_optim,
                 critic_q1_optim, critic_q2_optim, critic_v_optim, action_space, tau: float = 0.005, gamma: float = 0.99,
                 expectile: float = 0.8, temperature: float = 0.1) -> None:
        super().__init__()
        self.actor = actor
        self.critic_q1, self.critic_q1_old = critic_q1, clone_target(critic_q1)
        self.critic_q2, self.critic_q2_old = critic_q2, clone_target(critic_q2)
        self.critic_v = critic_v
        self.actor_optim, self.critic_q1_optim = actor_optim, critic_q1_optim
        self.critic_q2_optim, self.critic_v_optim = critic_q2_optim, critic_v_optim
        self.action_space = action_space
        self._tau, self._gamma, self._expectile, self._temperature = tau, gamma, expectile, temperature
        if float(critic_q1_optim.param_groups[0]["lr"]) != float(critic_q2_optim.param_groups[0]["lr"]):
            raise NotImplementedError("critic_q1/critic_q2 must share a learning rate")

    def _nets(self):
        return {_engine.NET_ACTOR: self.actor, _engine.NET_CRITIC1: self.critic_q1, _engine.NET_CRITIC2: self.critic_q2,
                _engine.NET_CRITIC1_OLD: self.critic_q1_old, _engine.NET_CRITIC2_OLD: self.critic_q2_old,
                _engine.NET_CRITIC_V: self.critic_v}

    def _optims(self):
        return {_engine.OPT_ACTOR: self.actor_optim, _engine.OPT_CRITIC: self.critic_q1_optim, _engine.OPT_CRITIC_V: self.critic_v_optim}

    def _all_optims(self):
        return [self.actor_optim, self.critic_q1_optim, self.critic_q2_optim, self.critic_v_optim]

    def _config(self) -> Dict:
        od, hid = _backbone_dims(self.actor.backbone, allow_dropout=True)
        p_drop = getattr(self.actor.backbone, "dropout_rate", None)
        if p_drop is not None and not 0.0 < float(p_drop) < 1.0:
            raise ValueError(f"dropout_rate must be in (0, 1), got {p_drop}")
        ad = self.actor.dist_net.mu.out_features
        dn = self.actor.dist_net
        if getattr(dn, "_c_sigma", True) or dn._unbounded or float(dn._max) != 1.0:
            raise NotImplementedError("IQL engine expects DiagGaussian(unbounded=False, conditioned_sigma=False, max_mu=1.0)")
        for c, cin in ((self.critic_q1, od + ad), (self.critic_v, od)):
            i, h = _backbone_dims(c.backbone)
            if i != cin or h != hid:
                raise NotImplementedError("IQL engine expects all nets to share hidden dims")
        return dict(obs_dim=od, act_dim=ad, hidden=hid, gamma=self._gamma, tau=self._tau,
                    actor_lr=float(self.actor_optim.param_groups[0]["lr"]), critic_lr=float(self.critic_q1_optim.param_groups[0]["lr"]),
                    critic_v_lr=float(self.critic_v_optim.param_groups[0]["lr"]), expectile=self._expectile,
                    iql_temperature=self._temperature, actor_dropout=float(p_drop or 0.0))

    def _mode_from_hidden(self, h, P):
        mu = torch.tanh(torch.baddbmm(P["dist_net.mu.bias"].unsqueeze(1), h, P["dist_net.mu.weight"].transpose(1, 2)))    # max_mu = 1 (checked in _config)
        return mu.clamp(float(self.action_space.low[0]), float(self.action_space.high[0]))

    def select_action(self, obs: np.ndarray, deterministic: bool = False) -> np.ndarray:
        if len(obs.shape) == 1:
            obs = obs.reshape(1, -1)
        with torch.no_grad():
            dist = self.actor(obs)
            action = (dist.mode() if deterministic else dist.sample()).cpu().numpy()
        return np.clip(action, self.action_space.low[0], self.action_space.high[0])

"""TD3+BC policy (reference: policy/model_free/td3bc.py:12-124 over td3.py:16-59) on the HIP engine."""
from __future__ import annotations

from typing import Callable, Dict

import numpy as np
import torch
import torch.nn as nn

from .. import _engine
from ..utils.noise import GaussianNoise
from ..utils.scaler import StandardScaler
from .base_policy import EnginePolicy, _backbone_dims, clone_target


class TD3BCPolicy(EnginePolicy):
    ALGO = "td3bc"

    def __init__(self, actor: nn.Module, critic1: nn.Module, critic2: nn.Module, actor_optim, critic1_optim, critic2_optim,
                 tau: float = 0.005, gamma: float = 0.99, max_action: float = 1.0, exploration_noise: Callable = GaussianNoise,
                 policy_noise: float = 0.2, noise_clip: float = 0.5, update_actor_freq: int = 2, alpha: float = 2.5,
                 scaler: StandardScaler = None) -> None:
        super().__init__()
        self.actor, self.actor_old = actor, clone_target(actor)
        self.critic1, self.critic1_old = critic1, clone_target(critic1)
        self.critic2, self.critic2_old = critic2, clone_target(critic2)
        self.actor_optim, self.critic1_optim, self.critic2_optim = actor_optim, critic1_optim, critic2_optim
        self._tau, self._gamma, self._max_action = tau, gamma, max_action
        self.exploration_noise = exploration_noise
        self._policy_noise, self._noise_clip, self._freq = policy_noise, noise_clip, update_actor_freq
        self._alpha = alpha
        self.scaler = scaler
        if float(critic1_optim.param_groups[0]["lr"]) != float(critic2_optim.param_groups[0]["lr"]):
            raise NotImplementedError("critic1/critic2 must share a learning rate")

    @property
    def _cnt(self) -> int:
        return self._eng.step_count() if self._eng is not None else 0

    def _nets(self):
        return {_engine.NET_ACTOR: self.actor, _engine.NET_CRITIC1: self.critic1, _engine.NET_CRITIC2: self.critic2,
                _engine.NET_CRITIC1_OLD: self.critic1_old, _engine.NET_CRITIC2_OLD: self.critic2_old,
                _engine.NET_ACTOR_OLD: self.actor_old}

    def _optims(self):
        return {_engine.OPT_ACTOR: self.actor_optim, _engine.OPT_CRITIC: self.critic1_optim}

    def _all_optims(self):
        return [self.actor_optim, self.critic1_optim, self.critic2_optim]

    def _config(self) -> Dict:
        od, hid = _backbone_dims(self.actor.backbone)
        ad = self.actor.last.out_features
        if float(getattr(self.actor, "_max", self._max_action)) != float(self._max_action):
            raise NotImplementedError("Actor(max_action) and TD3BCPolicy(max_action) differ: learn() and select_action would disagree")
        cin, chid = _backbone_dims(self.critic1.backbone)
        if cin != od + ad or chid != hid:
            raise NotImplementedError("TD3BC engine expects actor and critics to share hidden dims")
        return dict(obs_dim=od, act_dim=ad, hidden=hid, gamma=self._gamma, tau=self._tau,
                    actor_lr=float(self.actor_optim.param_groups[0]["lr"]), critic_lr=float(self.critic1_optim.param_groups[0]["lr"]),
                    policy_noise=self._policy_noise, noise_clip=self._noise_clip, td3bc_alpha=self._alpha,
                    max_action=float(self._max_action), update_actor_freq=int(self._freq))

    def _eval_obs(self, obs):
        return self.scaler.transform(obs) if self.scaler is not None else obs

    def _mode_from_hidden(self, h, P):
        return float(self._max_action) * torch.tanh(torch.baddbmm(P["last.bias"].unsqueeze(1), h, P["last.weight"].transpose(1, 2)))

    def select_action(self, obs: np.ndarray, deterministic: bool = False) -> np.ndarray:
        if self.scaler is not None:
            obs = self.scaler.transform(obs)
        with torch.no_grad():
            action = self.actor(obs).cpu().numpy()
        if not deterministic:
            action = action + self.exploration_noise(action.shape)
            action = np.clip(action, -self._max_action, self._max_action)
        return action

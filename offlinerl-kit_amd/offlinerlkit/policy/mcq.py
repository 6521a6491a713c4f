"""MCQ policy (reference: policy/model_free/mcq.py:11-126 over sac.py) on the HIP engine."""
from __future__ import annotations

from typing import Dict, Tuple, Union

import torch.nn as nn

from .. import _engine
from .model_based import SACPolicy


class MCQPolicy(SACPolicy):
    """Mildly Conservative Q-Learning <Ref: https://arxiv.org/abs/2206.04745>; constructor = mcq.py:16-46."""

    ALGO = "mcq"

    def __init__(self, actor: nn.Module, critic1: nn.Module, critic2: nn.Module, behavior_policy: nn.Module, actor_optim, critic1_optim,
                 critic2_optim, behavior_policy_optim, tau: float = 0.005, gamma: float = 0.99, alpha: Union[float, Tuple] = 0.2,
                 lmbda: float = 0.7, num_sampled_actions: int = 10) -> None:
        super().__init__(actor, critic1, critic2, actor_optim, critic1_optim, critic2_optim, tau=tau, gamma=gamma, alpha=alpha)
        self.behavior_policy = behavior_policy
        self.behavior_policy_optim = behavior_policy_optim
        self._lmbda = lmbda
        self._num_sampled_actions = num_sampled_actions

    def _nets(self):
        n = super()._nets()
        n[_engine.NET_VAE_ENC] = self.behavior_policy          # one module, two parameter families (e1, e2, mean, log_std | d1, d2, d3)
        n[_engine.NET_VAE_DEC] = self.behavior_policy
        return n

    def _optims(self):
        o = super()._optims()
        o[_engine.OPT_VAE] = self.behavior_policy_optim
        return o

    def _config(self) -> Dict:
        c = super()._config()
        bp = self.behavior_policy
        vh = bp.e1.out_features
        if (bp.e1.in_features != c["obs_dim"] + c["act_dim"] or bp.d3.out_features != c["act_dim"] or bp.e2.out_features != vh
                or bp.d1.out_features != vh or bp.d2.out_features != vh):
            raise NotImplementedError("MCQ engine expects the VAE of nets/vae.py: one hidden width, obs+act -> latent -> act")
        c.update(vae_hidden=int(bp.e1.out_features), vae_latent=int(bp.latent_dim), max_action=float(bp.max_action),
                 mcq_lambda=float(self._lmbda), num_repeat_actions=int(self._num_sampled_actions),
                 behavior_lr=float(self.behavior_policy_optim.param_groups[0]["lr"]))
        return c

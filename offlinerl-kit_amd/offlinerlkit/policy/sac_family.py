"""CQL and EDAC policies (reference: policy/model_free/{sac,cql,edac}.py) on the HIP engine."""
from __future__ import annotations

from typing import Dict, Tuple, Union

import numpy as np
import torch
import torch.nn as nn

from .. import _engine
from .base_policy import EnginePolicy, _backbone_dims, clone_target


class _TanhGaussPolicy(EnginePolicy):
    """Shared pieces of the SAC family: tanh-Gaussian actor, (auto-)temperature, actforward / select_action
    (sac.py:42-48, 66-86)."""

    def _init_alpha(self, alpha) -> None:
        self._is_auto_alpha = False
        if isinstance(alpha, tuple):
            self._is_auto_alpha = True
            self._target_entropy, self._log_alpha, self.alpha_optim = alpha
        else:
            self._fixed_alpha = float(alpha)

    @property
    def _alpha(self):
        if not self._is_auto_alpha:
            return self._fixed_alpha
        if self._eng is not None:
            return self._eng.get_scalar(max(self._cur_run, 0), _engine.SCALAR_ALPHA)
        return float(self._log_alpha.detach().exp())

    def _alpha_config(self) -> Dict:
        if self._is_auto_alpha:
            return dict(auto_alpha=1, target_entropy=float(self._target_entropy), alpha_lr=float(self.alpha_optim.param_groups[0]["lr"]))
        return dict(auto_alpha=0, alpha=self._fixed_alpha)

    def _after_bind(self) -> None:
        if self._is_auto_alpha:      # every run starts from the caller's log_alpha (run_cql.py:102: zeros)
            self._log_alpha_init = float(self._log_alpha.detach().cpu().reshape(-1)[0])
        self._after_bind_new_runs(0, self._n_runs)

    def _after_bind_new_runs(self, first: int, n_runs: int) -> None:
        if self._is_auto_alpha:
            v = getattr(self, "_log_alpha_init", float(self._log_alpha.detach().cpu().reshape(-1)[0]))
            for r in range(first, n_runs):
                self._eng.set_scalar(r, _engine.SCALAR_LOG_ALPHA, v)

    def _before_unbind(self) -> None:
        self.sync_scalars()

    def _on_select_run(self, run: int) -> None:
        self.sync_scalars()

    def sync_scalars(self) -> None:
        """Copy the device-side log_alpha (of the selected run) back into the caller's tensor (it is not an nn.Parameter in the
        reference)."""
        if self._eng is not None and self._is_auto_alpha:
            with torch.no_grad():
                self._log_alpha.fill_(self._eng.get_scalar(max(self._cur_run, 0), _engine.SCALAR_LOG_ALPHA))

    def _check_dist_net(self) -> None:
        dn = self.actor.dist_net
        if not getattr(dn, "_c_sigma", False) or not dn._unbounded:
            raise NotImplementedError(f"{self.ALGO.upper()} engine expects TanhDiagGaussian(unbounded=True, conditioned_sigma=True)")
        if float(dn._sigma_min) != -5.0 or float(dn._sigma_max) != 2.0:
            raise NotImplementedError("the HIP tanh-Gaussian head clamps log-sigma to [-5, 2] (dist_module.py:57-58 defaults); "
                                      f"got [{dn._sigma_min}, {dn._sigma_max}]")

    def actforward(self, obs, deterministic: bool = False):
        dist = self.actor(obs)
        squashed, raw = dist.mode() if deterministic else dist.rsample()
        return squashed, dist.log_prob(squashed, raw)

    def select_action(self, obs: np.ndarray, deterministic: bool = False) -> np.ndarray:
        with torch.no_grad():
            action, _ = self.actforward(obs, deterministic)
        return action.cpu().numpy()

    def _mode_from_hidden(self, h, P):
        return torch.tanh(torch.baddbmm(P["dist_net.mu.bias"].unsqueeze(1), h, P["dist_net.mu.weight"].transpose(1, 2)))   # TanhNormalWrapper.mode

    def _optims(self):
        o = {_engine.OPT_ACTOR: self.actor_optim}
        if self._is_auto_alpha:
            o[_engine.OPT_ALPHA] = self.alpha_optim
        return o

    def _all_optims(self):
        return list(self._optims().values()) + [getattr(self, n) for n in ("critic2_optim",) if hasattr(self, n)]


class CQLPolicy(_TanhGaussPolicy):
    """Conservative Q-Learning; constructor = reference CQLPolicy.__init__ (cql.py:16-60) including the
    ``num_repeart_actions`` spelling."""

    ALGO = "cql"

    def __init__(self, actor: nn.Module, critic1: nn.Module, critic2: nn.Module, actor_optim, critic1_optim, critic2_optim,
                 action_space, tau: float = 0.005, gamma: float = 0.99, alpha: Union[float, Tuple] = 0.2,
                 cql_weight: float = 1.0, temperature: float = 1.0, max_q_backup: bool = False,
                 deterministic_backup: bool = True, with_lagrange: bool = True, lagrange_threshold: float = 10.0,
                 cql_alpha_lr: float = 1e-4, num_repeart_actions: int = 10) -> None:
        super().__init__()
        self.actor = actor
        self.critic1, self.critic1_old = critic1, clone_target(critic1)
        self.critic2, self.critic2_old = critic2, clone_target(critic2)
        self.actor_optim, self.critic1_optim, self.critic2_optim = actor_optim, critic1_optim, critic2_optim
        self._tau, self._gamma = tau, gamma
        self._init_alpha(alpha)
        self.action_space = action_space
        self._cql_weight, self._temperature = cql_weight, temperature
        self._max_q_backup, self._deterministic_backup = max_q_backup, deterministic_backup
        self._with_lagrange, self._lagrange_threshold = with_lagrange, lagrange_threshold
        self.cql_log_alpha = torch.zeros(1)
        self._cql_alpha_lr = cql_alpha_lr
        self._num_repeat_actions = num_repeart_actions
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
            raise NotImplementedError("CQL engine expects actor and critics to share hidden dims")
        c = dict(obs_dim=od, act_dim=ad, hidden=hid, gamma=self._gamma, tau=self._tau,
                 actor_lr=float(self.actor_optim.param_groups[0]["lr"]), critic_lr=float(self.critic1_optim.param_groups[0]["lr"]),
                 cql_weight=self._cql_weight, temperature=self._temperature, max_q_backup=int(bool(self._max_q_backup)),
                 deterministic_backup=int(bool(self._deterministic_backup)), with_lagrange=int(bool(self._with_lagrange)),
                 lagrange_threshold=self._lagrange_threshold, cql_alpha_lr=self._cql_alpha_lr,
                 num_repeat_actions=int(self._num_repeat_actions), act_low=float(self.action_space.low[0]),
                 act_high=float(self.action_space.high[0]))
        c.update(self._alpha_config())
        return c

    def _after_bind(self) -> None:
        self._cql_log_alpha_init = float(self.cql_log_alpha.reshape(-1)[0])
        super()._after_bind()

    def _after_bind_new_runs(self, first: int, n_runs: int) -> None:
        super()._after_bind_new_runs(first, n_runs)
        v = getattr(self, "_cql_log_alpha_init", float(self.cql_log_alpha.reshape(-1)[0]))
        for r in range(first, n_runs):
            self._eng.set_scalar(r, _engine.SCALAR_CQL_LOG_ALPHA, v)

    def sync_scalars(self) -> None:
        super().sync_scalars()
        if self._eng is not None:
            self.cql_log_alpha.fill_(self._eng.get_scalar(max(self._cur_run, 0), _engine.SCALAR_CQL_LOG_ALPHA))


class EDACPolicy(_TanhGaussPolicy):
    """Ensemble-Diversified Actor Critic; constructor = reference EDACPolicy.__init__ (edac.py:15-52)."""

    ALGO = "edac"

    def __init__(self, actor: nn.Module, critics: nn.Module, actor_optim, critics_optim, tau: float = 0.005,
                 gamma: float = 0.99, alpha: Union[float, Tuple] = 0.2, max_q_backup: bool = False,
                 deterministic_backup: bool = True, eta: float = 1.0) -> None:
        super().__init__()
        self.actor = actor
        self.critics = critics
        self.critics_old = clone_target(critics)
        self.actor_optim, self.critics_optim = actor_optim, critics_optim
        self._tau, self._gamma = tau, gamma
        self._init_alpha(alpha)
        self._max_q_backup, self._deterministic_backup, self._eta = max_q_backup, deterministic_backup, eta
        self._num_critics = self.critics._num_ensemble

    def _nets(self):
        return {_engine.NET_ACTOR: self.actor, _engine.NET_CRITIC1: self.critics, _engine.NET_CRITIC1_OLD: self.critics_old}

    def _optims(self):
        o = super()._optims()
        o[_engine.OPT_CRITIC] = self.critics_optim
        return o

    def _config(self) -> Dict:
        od, hid = _backbone_dims(self.actor.backbone)
        ad = self.actor.dist_net.mu.out_features
        self._check_dist_net()
        if list(self.critics.hidden_dims) != list(hid) or self.critics.obs_dim != od or self.critics.action_dim != ad:
            raise NotImplementedError("EDAC engine expects the ensemble critics to share the actor's hidden dims")
        c = dict(obs_dim=od, act_dim=ad, hidden=hid, gamma=self._gamma, tau=self._tau,
                 actor_lr=float(self.actor_optim.param_groups[0]["lr"]), critic_lr=float(self.critics_optim.param_groups[0]["lr"]),
                 num_critics=int(self._num_critics), eta=float(self._eta), max_q_backup=int(bool(self._max_q_backup)),
                 deterministic_backup=int(bool(self._deterministic_backup)))
        c.update(self._alpha_config())
        return c

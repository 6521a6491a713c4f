"""Actor / critic heads of the hot path (reference: offlinerlkit/modules/{actor,critic,ensemble_critic,dist}_module.py).

As with ``nets``, these describe networks and serve ``select_action`` at evaluation time; the training math
(sampling, log-probabilities, their gradients) lives in the HIP engine (csrc/kernels.h).
"""
from __future__ import annotations

from typing import Optional, Sequence, Union

import numpy as np
import torch
import torch.nn as nn

from .nets import EnsembleLinear, MLP  # noqa: F401


def _as_input(x, device) -> torch.Tensor:
    return torch.as_tensor(x, device=device, dtype=torch.float32)


class NormalWrapper(torch.distributions.Normal):
    """Diagonal Gaussian whose log_prob / entropy are summed over the action dimension (dist_module.py:6-14)."""

    def log_prob(self, actions):
        return super().log_prob(actions).sum(-1, keepdim=True)

    def entropy(self):
        return super().entropy().sum(-1)

    def mode(self):
        return self.mean


class TanhNormalWrapper(torch.distributions.Normal):
    """Gaussian squashed by tanh at sampling time (dist_module.py:17-42); log_prob carries the Jacobian term
    -sum log(1 - a^2 + 1e-6)."""

    _EPS = 1e-6

    def log_prob(self, action, raw_action=None):
        if raw_action is None:
            raw_action = self.arctanh(action)
        base = super().log_prob(raw_action).sum(-1, keepdim=True)
        return base - torch.log((1 - action.pow(2)) + self._EPS).sum(-1, keepdim=True)

    def mode(self):
        return torch.tanh(self.mean), self.mean

    def arctanh(self, x):
        return 0.5 * torch.log((1 + x).clamp(min=1e-6) / (1 - x).clamp(min=1e-6))

    def rsample(self):
        raw = super().rsample()
        return torch.tanh(raw), raw


class DiagGaussian(nn.Module):
    """mu = Linear (optionally max_mu * tanh), sigma = exp(clamp(Linear)) or exp(sigma_param (A,1)) (dist_module.py:45-78)."""

    wrapper = NormalWrapper

    def __init__(self, latent_dim, output_dim, unbounded=False, conditioned_sigma=False, max_mu=1.0,
                 sigma_min=-5.0, sigma_max=2.0):
        super().__init__()
        self.mu = nn.Linear(latent_dim, output_dim)
        self._c_sigma = conditioned_sigma
        if conditioned_sigma:
            self.sigma = nn.Linear(latent_dim, output_dim)
        else:
            self.sigma_param = nn.Parameter(torch.zeros(output_dim, 1))
        self._unbounded = unbounded
        self._max = max_mu
        self._sigma_min = sigma_min
        self._sigma_max = sigma_max

    def reset_parameters(self) -> None:
        """the constructor's value of the module's OWN parameter (the Linear children reset themselves): sigma_param = 0"""
        if not self._c_sigma:
            with torch.no_grad():
                self.sigma_param.zero_()

    def _params(self, logits):
        mu = self.mu(logits)
        if not self._unbounded:
            mu = self._max * torch.tanh(mu)
        if self._c_sigma:
            log_sigma = torch.clamp(self.sigma(logits), min=self._sigma_min, max=self._sigma_max)
        else:
            shape = [1] * mu.dim()
            shape[1] = -1
            log_sigma = self.sigma_param.view(shape) + torch.zeros_like(mu)
        return mu, log_sigma

    def forward(self, logits):
        mu, log_sigma = self._params(logits)
        return self.wrapper(mu, log_sigma.exp())

    def get_dist_params(self, logits):
        return self._params(logits)


class TanhDiagGaussian(DiagGaussian):
    """Same parameters, tanh-squashed samples (dist_module.py:95-127)."""

    wrapper = TanhNormalWrapper


class ActorProb(nn.Module):
    """backbone -> dist_net -> distribution (actor_module.py:9-27)."""

    def __init__(self, backbone: nn.Module, dist_net: nn.Module, device: str = "cpu") -> None:
        super().__init__()
        self.device = torch.device(device)
        self.backbone = backbone.to(device)
        self.dist_net = dist_net.to(device)

    def forward(self, obs: Union[np.ndarray, torch.Tensor]):
        return self.dist_net(self.backbone(_as_input(obs, self.device)))


class Actor(nn.Module):
    """Deterministic actor: max_action * tanh(Linear(backbone(obs))) (actor_module.py:30-51)."""

    def __init__(self, backbone: nn.Module, action_dim: int, max_action: float = 1.0, device: str = "cpu") -> None:
        super().__init__()
        self.device = torch.device(device)
        self.backbone = backbone.to(device)
        self.last = nn.Linear(getattr(backbone, "output_dim"), action_dim).to(device)
        self._max = max_action

    def forward(self, obs: Union[np.ndarray, torch.Tensor]) -> torch.Tensor:
        return self._max * torch.tanh(self.last(self.backbone(_as_input(obs, self.device))))


class Critic(nn.Module):
    """Q(s,a) (or V(s) when actions is None): cat -> backbone -> Linear(H,1) (critic_module.py:9-28)."""

    def __init__(self, backbone: nn.Module, device: str = "cpu") -> None:
        super().__init__()
        self.device = torch.device(device)
        self.backbone = backbone.to(device)
        self.last = nn.Linear(getattr(backbone, "output_dim"), 1).to(device)

    def forward(self, obs, actions: Optional[Union[np.ndarray, torch.Tensor]] = None) -> torch.Tensor:
        x = _as_input(obs, self.device)
        if actions is not None:
            x = torch.cat([x, _as_input(actions, self.device).flatten(1)], dim=1)
        return self.last(self.backbone(x))


class EnsembleCritic(nn.Module):
    """K parallel critic MLPs built from EnsembleLinear; output (K, B, 1) (ensemble_critic_module.py:11-44)."""

    def __init__(self, obs_dim: int, action_dim: int, hidden_dims: Sequence[int], activation=nn.ReLU,
                 num_ensemble: int = 10, device: str = "cpu") -> None:
        super().__init__()
        dims = [obs_dim + action_dim] + list(hidden_dims)
        layers = []
        for d_in, d_out in zip(dims, dims[1:]):
            layers += [EnsembleLinear(d_in, d_out, num_ensemble), activation()]
        layers.append(EnsembleLinear(dims[-1], 1, num_ensemble))
        self.device = torch.device(device)
        self.model = nn.Sequential(*layers).to(device)
        self._num_ensemble = num_ensemble
        self.obs_dim, self.action_dim, self.hidden_dims = obs_dim, action_dim, list(hidden_dims)

    def forward(self, obs, actions=None) -> torch.Tensor:
        x = _as_input(obs, self.device)
        if actions is not None:
            x = torch.cat([x, _as_input(actions, self.device)], dim=-1)
        return self.model(x)

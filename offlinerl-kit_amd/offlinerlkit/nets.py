"""Backbone networks of the hot path (reference: offlinerlkit/nets/mlp.py:9-33, nets/ensemble_linear.py:9-41).

These torch modules only DESCRIBE a network (shapes, initial weights, eval-time forward for
``select_action``).  Training never runs through them: an engine-backed policy copies their
parameters into the HIP engine's arena and re-points ``param.data`` at views of that arena.
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import torch
import torch.nn as nn


class MLP(nn.Module):
    """[Linear, activation(, Dropout)] x len(hidden_dims) (+ optional output Linear).
    ``self.model`` is an ``nn.Sequential`` so state_dict keys are ``model.<2*i>.{weight,bias}`` like the reference."""

    def __init__(self, input_dim: int, hidden_dims: Sequence[int], output_dim: Optional[int] = None,
                 activation=nn.ReLU, dropout_rate: Optional[float] = None) -> None:
        super().__init__()
        widths = [int(input_dim)] + [int(h) for h in hidden_dims]
        layers: List[nn.Module] = []
        for fan_in, fan_out in zip(widths, widths[1:]):
            layers.append(nn.Linear(fan_in, fan_out))
            layers.append(activation())
            if dropout_rate is not None:
                layers.append(nn.Dropout(p=dropout_rate))
        self.output_dim = widths[-1]
        if output_dim is not None:
            layers.append(nn.Linear(widths[-1], int(output_dim)))
            self.output_dim = int(output_dim)
        self.activation_cls = activation
        self.dropout_rate = dropout_rate
        self.model = nn.Sequential(*layers)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return self.model(x)

    def linear_layers(self) -> List[nn.Linear]:
        return [m for m in self.model if isinstance(m, nn.Linear)]


class EnsembleLinear(nn.Module):
    """K independent affine maps evaluated together: weight (K, in, out), bias (K, 1, out); y = x @ W + b.
    A 2-D input is shared by every member.  ``saved_weight`` / ``saved_bias`` are the reference's shadow copies
    (ensemble_linear.py:25-26): registered so state_dict keys match, never trained."""

    def __init__(self, input_dim: int, output_dim: int, num_ensemble: int, weight_decay: float = 0.0) -> None:
        super().__init__()
        self.num_ensemble = num_ensemble
        self.weight_decay = weight_decay
        self.input_dim = input_dim
        self.weight = nn.Parameter(torch.zeros(num_ensemble, input_dim, output_dim))
        self.bias = nn.Parameter(torch.zeros(num_ensemble, 1, output_dim))
        self.saved_weight = nn.Parameter(torch.zeros(num_ensemble, input_dim, output_dim))
        self.saved_bias = nn.Parameter(torch.zeros(num_ensemble, 1, output_dim))
        self.reset_parameters()

    def reset_parameters(self) -> None:
        """the constructor's initialisation again (ensemble_linear.py:21-26): trunc_normal(std = 1 / (2 sqrt(in))) weights, zero
        biases, shadow copies equal to them -- what building the layer under another seed gives (runs r > 0 of a multi-run policy)"""
        with torch.no_grad():
            nn.init.trunc_normal_(self.weight, std=1.0 / (2.0 * self.input_dim ** 0.5))
            self.bias.zero_()
            self.saved_weight.copy_(self.weight)
            self.saved_bias.copy_(self.bias)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if x.dim() == 2:
            y = torch.einsum("ij,bjk->bik", x, self.weight)
        else:
            y = torch.bmm(x, self.weight)
        return y + self.bias

    def load_save(self) -> None:
        self.weight.data.copy_(self.saved_weight.data)
        self.bias.data.copy_(self.saved_bias.data)

    def update_save(self, indexes) -> None:
        self.saved_weight.data[indexes] = self.weight.data[indexes]
        self.saved_bias.data[indexes] = self.bias.data[indexes]

    def get_decay_loss(self) -> torch.Tensor:
        return self.weight_decay * 0.5 * (self.weight ** 2).sum()


class VAE(nn.Module):
    """Vanilla conditional VAE, MCQ's behaviour policy (reference: offlinerlkit/nets/vae.py:8-66): encoder e1, e2 -> mean, log_std
    (clamped to [-4, 15]); decoder d1, d2, d3 -> max_action * tanh; ``decode`` without a latent samples N(0, 1) clipped to +-0.5.
    As with the other nets this describes the parameters and serves evaluation; MCQPolicy.learn runs in the HIP engine."""

    def __init__(self, input_dim: int, output_dim: int, hidden_dim: int, latent_dim: int, max_action, device: str = "cpu") -> None:
        super().__init__()
        self.e1 = nn.Linear(input_dim + output_dim, hidden_dim)
        self.e2 = nn.Linear(hidden_dim, hidden_dim)
        self.mean = nn.Linear(hidden_dim, latent_dim)
        self.log_std = nn.Linear(hidden_dim, latent_dim)
        self.d1 = nn.Linear(input_dim + latent_dim, hidden_dim)
        self.d2 = nn.Linear(hidden_dim, hidden_dim)
        self.d3 = nn.Linear(hidden_dim, output_dim)
        self.max_action = max_action
        self.latent_dim = latent_dim
        self.device = torch.device(device)
        self.to(device=self.device)

    def forward(self, obs: torch.Tensor, action: torch.Tensor):
        z = torch.relu(self.e1(torch.cat([obs, action], 1)))
        z = torch.relu(self.e2(z))
        mean = self.mean(z)
        std = torch.exp(self.log_std(z).clamp(-4, 15))
        z = mean + std * torch.randn_like(std)
        return self.decode(obs, z), mean, std

    def decode(self, obs: torch.Tensor, z=None) -> torch.Tensor:
        if z is None:
            z = torch.randn((obs.shape[0], self.latent_dim)).to(self.device).clamp(-0.5, 0.5)
        a = torch.relu(self.d1(torch.cat([obs, z], 1)))
        a = torch.relu(self.d2(a))
        return self.max_action * torch.tanh(self.d3(a))

"""PyTorch-CPU counterpart of CQLPolicy.learn (policy/model_free/cql.py:87-207 over sac.py:60-77) written against SURVEY.md
Appendix A.1 with stock torch autograd and torch.optim.Adam.  TEST INFRASTRUCTURE ONLY: it is the `cpu_baseline` of bench.py
("the build's own PyTorch-CPU counterpart ... timed on the GPU box's host cores") and is pinned against the
reference fixtures in tests/test_oracle_golden.py.  It is our code (the reference's files never travel to the GPU box); the
numpy oracle (oracle/cql.py) remains the parity checker.

State: the same plain dicts of fp32 arrays as oracle/cql.py (key names = reference state_dict keys); parameters are turned
into leaf tensors once (``TorchCQL``) and stepped by one ``torch.optim.Adam`` per network, like run_cql.py:92-103.
"""
from __future__ import annotations

import math
from collections import OrderedDict
from typing import Dict

import numpy as np
import torch
import torch.nn.functional as F


def _mlp(p: Dict[str, torch.Tensor], x: torch.Tensor) -> torch.Tensor:
    i = 0
    while f"backbone.model.{i}.weight" in p:
        x = F.relu(F.linear(x, p[f"backbone.model.{i}.weight"], p[f"backbone.model.{i}.bias"]))
        i += 2
    return x


def _critic(p, obs, act):
    return F.linear(_mlp(p, torch.cat([obs, act], dim=1)), p["last.weight"], p["last.bias"])


def _actor(p, obs, eps):
    """TanhDiagGaussian(unbounded=True, conditioned_sigma=True) rsample + log_prob (dist_module.py:17-42, 117-127)"""
    h = _mlp(p, obs)
    mu = F.linear(h, p["dist_net.mu.weight"], p["dist_net.mu.bias"])
    ls = torch.clamp(F.linear(h, p["dist_net.sigma.weight"], p["dist_net.sigma.bias"]), min=-5.0, max=2.0)
    sg = ls.exp()
    u = mu + sg * eps
    a = torch.tanh(u)
    logp = (-((u - mu) ** 2) / (2 * sg ** 2) - ls - math.log(math.sqrt(2 * math.pi))).sum(-1, keepdim=True)
    logp = logp - torch.log((1 - a.pow(2)) + 1e-6).sum(-1, keepdim=True)
    return a, logp


class TorchCQL:
    def __init__(self, state: Dict, cfg: Dict):
        self.cfg = cfg
        t = lambda d: OrderedDict((k, torch.tensor(np.asarray(v, np.float32), requires_grad=True)) for k, v in d.items())
        self.actor, self.c1, self.c2 = t(state["actor"]), t(state["critic1"]), t(state["critic2"])
        self.c1o = OrderedDict((k, torch.tensor(np.asarray(v, np.float32))) for k, v in state["critic1_old"].items())
        self.c2o = OrderedDict((k, torch.tensor(np.asarray(v, np.float32))) for k, v in state["critic2_old"].items())
        self.log_alpha = torch.tensor(np.asarray(state["log_alpha"], np.float32), requires_grad=True)
        self.cql_log_alpha = torch.tensor(np.asarray(state["cql_log_alpha"], np.float32), requires_grad=True)
        self.opt_actor = torch.optim.Adam(self.actor.values(), lr=cfg["actor_lr"])
        self.opt_c1 = torch.optim.Adam(self.c1.values(), lr=cfg["critic_lr"])
        self.opt_c2 = torch.optim.Adam(self.c2.values(), lr=cfg["critic_lr"])
        self.opt_alpha = torch.optim.Adam([self.log_alpha], lr=cfg["alpha_lr"])
        self.opt_cql_alpha = torch.optim.Adam([self.cql_log_alpha], lr=cfg["cql_alpha_lr"])
        self.alpha = self.log_alpha.detach().exp() if cfg["auto_alpha"] else torch.tensor(float(cfg["alpha"]))

    def learn(self, batch: Dict[str, np.ndarray], noise: Dict[str, np.ndarray]) -> "OrderedDict[str, float]":
        cfg = self.cfg
        T = lambda x: torch.as_tensor(np.asarray(x, np.float32))
        obs, act, nobs = T(batch["observations"]), T(batch["actions"]), T(batch["next_observations"])
        rew, term = T(batch["rewards"]).reshape(-1, 1), T(batch["terminals"]).reshape(-1, 1)
        B, A, N = obs.shape[0], act.shape[1], cfg["num_repeat_actions"]
        w, Tm = cfg["cql_weight"], cfg["temperature"]
        # actor (cql.py:92-98)
        a, logp = _actor(self.actor, obs, T(noise["eps_actor"]))
        actor_loss = (self.alpha * logp - torch.min(_critic(self.c1, obs, a), _critic(self.c2, obs, a))).mean()
        self.opt_actor.zero_grad(); actor_loss.backward(); self.opt_actor.step()
        res = OrderedDict()
        if cfg["auto_alpha"]:          # cql.py:100-106 (alpha NOT clamped)
            alpha_loss = -(self.log_alpha * (logp.detach() + cfg["target_entropy"])).mean()
            self.opt_alpha.zero_grad(); alpha_loss.backward(); self.opt_alpha.step()
            self.alpha = self.log_alpha.detach().exp()
        # TD target with the updated actor (cql.py:108-132)
        with torch.no_grad():
            if cfg["max_q_backup"]:
                tn = nobs.repeat_interleave(N, dim=0)
                na, _ = _actor(self.actor, tn, T(noise["eps_next"]))
                nq = torch.min(_critic(self.c1o, tn, na).view(B, N, 1).max(1)[0], _critic(self.c2o, tn, na).view(B, N, 1).max(1)[0])
            else:
                na, nlogp = _actor(self.actor, nobs, T(noise["eps_next"]))
                nq = torch.min(_critic(self.c1o, nobs, na), _critic(self.c2o, nobs, na))
                if not cfg["deterministic_backup"]:
                    nq = nq - self.alpha * nlogp
            target_q = rew + cfg["gamma"] * (1 - term) * nq
            to, tno = obs.repeat_interleave(N, dim=0), nobs.repeat_interleave(N, dim=0)
            a_pi, lp_pi = _actor(self.actor, to, T(noise["eps_pi"]))          # probe-verified: sampling under no_grad is bit-identical (A.1)
            a_npi, lp_npi = _actor(self.actor, tno, T(noise["eps_next_pi"]))
            u_rand = T(noise["u_rand"])
        log_rand = math.log(0.5 ** A)
        cons, tds = [], []
        for c in (self.c1, self.c2):
            q = _critic(c, obs, act)
            tds.append(((q - target_q) ** 2).mean())
            cat = torch.cat([_critic(c, to, a_pi) - lp_pi, _critic(c, to, a_npi) - lp_npi, _critic(c, to, u_rand) - log_rand], dim=1)   # (B*N, 3): quirk Q3
            cons.append(torch.logsumexp(cat / Tm, dim=1).mean() * w * Tm - q.mean() * w)
        if cfg["with_lagrange"]:      # cql.py:170-178
            cql_alpha = torch.clamp(self.cql_log_alpha.exp(), 0.0, 1e6)
            cons = [cql_alpha * (cv - cfg["lagrange_threshold"]) for cv in cons]
            cql_alpha_loss = -(cons[0] + cons[1]) * 0.5
            self.opt_cql_alpha.zero_grad(); cql_alpha_loss.backward(retain_graph=True); self.opt_cql_alpha.step()
        l1, l2 = tds[0] + cons[0], tds[1] + cons[1]
        self.opt_c1.zero_grad(); l1.backward(retain_graph=True); self.opt_c1.step()
        self.opt_c2.zero_grad(); l2.backward(); self.opt_c2.step()
        with torch.no_grad():          # sac.py:60-64
            for o, n in ((self.c1o, self.c1), (self.c2o, self.c2)):
                for k in o:
                    o[k].mul_(1 - cfg["tau"]).add_(n[k].detach() * cfg["tau"])
        res["loss/actor"], res["loss/critic1"], res["loss/critic2"] = actor_loss.item(), l1.item(), l2.item()
        if cfg["auto_alpha"]:
            res["loss/alpha"], res["alpha"] = alpha_loss.item(), float(self.alpha)
        if cfg["with_lagrange"]:
            res["loss/cql_alpha"], res["cql_alpha"] = cql_alpha_loss.item(), cql_alpha.item()
        return res

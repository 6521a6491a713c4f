#!/usr/bin/env python3
"""Generate golden vectors by running the REAL reference (/root/reference) on
the synthetic inputs of ``synth.py``.  Build-container only: the reference does
not exist on the GPU box, so the outputs are committed as small ``.npz``
fixtures next to this script.  Nothing of the reference (source, bytecode,
pickles) is written — only arrays of numbers it computed.

Usage:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py [algo ...]

Import recipe: SURVEY.md Appendix C (stub ``gym``; skip the eager
``offlinerlkit/policy/__init__.py``).  RNG is teacher-forced: the reference's
``Normal.rsample`` / ``Tensor.uniform_`` / ``torch.randn_like`` draws are
replaced by the arrays from ``synth.py`` in the reference's own draw order.
"""
from __future__ import annotations

import importlib
import os
import sys
import types
from collections import OrderedDict

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.dont_write_bytecode = True
import synth  # noqa: E402

REF = "/root/reference"


def _import_reference():
    sys.path.insert(0, REF)

    def _stub(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    class _Space:
        pass

    class _Env:
        pass

    _stub("gym", spaces=_stub("gym.spaces", Space=_Space), Env=_Env)
    import offlinerlkit  # noqa: F401
    pkg = types.ModuleType("offlinerlkit.policy")
    pkg.__path__ = [REF + "/offlinerlkit/policy"]
    sys.modules["offlinerlkit.policy"] = pkg
    pkg.BasePolicy = importlib.import_module("offlinerlkit.policy.base_policy").BasePolicy
    pkg.SACPolicy = importlib.import_module("offlinerlkit.policy.model_free.sac").SACPolicy
    pkg.TD3Policy = importlib.import_module("offlinerlkit.policy.model_free.td3").TD3Policy
    ns = types.SimpleNamespace()
    ns.CQLPolicy = importlib.import_module("offlinerlkit.policy.model_free.cql").CQLPolicy
    ns.IQLPolicy = importlib.import_module("offlinerlkit.policy.model_free.iql").IQLPolicy
    ns.TD3BCPolicy = importlib.import_module("offlinerlkit.policy.model_free.td3bc").TD3BCPolicy
    ns.EDACPolicy = importlib.import_module("offlinerlkit.policy.model_free.edac").EDACPolicy
    from offlinerlkit.nets import MLP
    from offlinerlkit.modules import Actor, ActorProb, Critic, EnsembleCritic, TanhDiagGaussian, DiagGaussian
    from offlinerlkit.buffer import ReplayBuffer
    ns.MLP, ns.Actor, ns.ActorProb, ns.Critic = MLP, Actor, ActorProb, Critic
    ns.EnsembleCritic, ns.TanhDiagGaussian, ns.DiagGaussian = EnsembleCritic, TanhDiagGaussian, DiagGaussian
    ns.ReplayBuffer = ReplayBuffer
    return ns


class NoiseFeeder:
    """Replaces the reference's RNG draws by queued arrays (draw order checked
    by shape)."""

    def __init__(self):
        self.normal_q = []
        self.uniform_q = []
        self._orig_std_normal = torch.distributions.normal._standard_normal
        self._orig_uniform = torch.Tensor.uniform_
        self._orig_randn_like = torch.randn_like

    def install(self):
        feeder = self

        def std_normal(shape, dtype, device):
            arr = feeder.normal_q.pop(0)
            assert tuple(arr.shape) == tuple(shape), (arr.shape, tuple(shape))
            return torch.tensor(arr, dtype=dtype, device=device)

        def uniform_(self_t, a=0.0, b=1.0, **kw):
            arr = feeder.uniform_q.pop(0)
            assert tuple(arr.shape) == tuple(self_t.shape), (arr.shape, tuple(self_t.shape))
            self_t.copy_(torch.tensor(arr, dtype=self_t.dtype))
            return self_t

        def randn_like(t, **kw):
            arr = feeder.normal_q.pop(0)
            assert tuple(arr.shape) == tuple(t.shape)
            return torch.tensor(arr, dtype=t.dtype)

        torch.distributions.normal._standard_normal = std_normal
        torch.Tensor.uniform_ = uniform_
        torch.randn_like = randn_like

    def uninstall(self):
        torch.distributions.normal._standard_normal = self._orig_std_normal
        torch.Tensor.uniform_ = self._orig_uniform
        torch.randn_like = self._orig_randn_like


class CallRecorder:
    """Forward hook that records every output of a module in call order."""

    def __init__(self, module):
        self.outs = []
        module.register_forward_hook(lambda m, i, o: self.outs.append(o.detach().cpu().numpy().copy()))


def _load(module, arrays):
    sd = OrderedDict((k, torch.tensor(v)) for k, v in arrays.items())
    module.load_state_dict(sd, strict=True)


def _state_of(module):
    return OrderedDict((k, v.detach().cpu().numpy().copy()) for k, v in module.state_dict().items())


class _ActionSpace:
    def __init__(self, act_dim, low=-1.0, high=1.0):
        self.low = np.full((act_dim,), low, dtype=np.float32)
        self.high = np.full((act_dim,), high, dtype=np.float32)
        self.shape = (act_dim,)


def _tb(batch):
    return {k: torch.tensor(v) for k, v in batch.items()}


def _put_state(out, prefix, net_state, full):
    for k, v in net_state.items():
        out[f"{prefix}/{k}/digest"] = synth.digest(v)
        if full:
            out[f"{prefix}/{k}/full"] = v


# ----------------------------------------------------------------------------
# CQL
# ----------------------------------------------------------------------------

def gen_cql(ref, case):
    sys.path.insert(0, os.path.join(HERE, "..", ".."))
    from oracle import cql as ocql
    c, st, batches, noises = synth.cql_case_inputs(case)
    cfg = ocql.default_cfg(c["obs_dim"], c["act_dim"])
    cfg.update(hidden=c["hidden"], num_repeat_actions=c["N"])
    cfg.update(c["over"])
    od, ad, hid = c["obs_dim"], c["act_dim"], c["hidden"]
    actor = ref.ActorProb(ref.MLP(od, hid), ref.TanhDiagGaussian(hid[-1], ad, unbounded=True, conditioned_sigma=True))
    c1 = ref.Critic(ref.MLP(od + ad, hid))
    c2 = ref.Critic(ref.MLP(od + ad, hid))
    _load(actor, st["actor"]); _load(c1, st["critic1"]); _load(c2, st["critic2"])
    aopt = torch.optim.Adam(actor.parameters(), lr=cfg["actor_lr"])
    c1opt = torch.optim.Adam(c1.parameters(), lr=cfg["critic_lr"])
    c2opt = torch.optim.Adam(c2.parameters(), lr=cfg["critic_lr"])
    if cfg["auto_alpha"]:
        log_alpha = torch.tensor(st["log_alpha"].copy(), requires_grad=True)
        alpha = (cfg["target_entropy"], log_alpha, torch.optim.Adam([log_alpha], lr=cfg["alpha_lr"]))
    else:
        alpha = cfg["alpha"]
    pol = ref.CQLPolicy(actor, c1, c2, aopt, c1opt, c2opt, action_space=_ActionSpace(ad),
                        tau=cfg["tau"], gamma=cfg["gamma"], alpha=alpha, cql_weight=cfg["cql_weight"],
                        temperature=cfg["temperature"], max_q_backup=cfg["max_q_backup"],
                        deterministic_backup=cfg["deterministic_backup"], with_lagrange=cfg["with_lagrange"],
                        lagrange_threshold=cfg["lagrange_threshold"], cql_alpha_lr=cfg["cql_alpha_lr"],
                        num_repeart_actions=cfg["num_repeat_actions"])
    _load(pol.critic1_old, st["critic1_old"]); _load(pol.critic2_old, st["critic2_old"])
    with torch.no_grad():
        pol.cql_log_alpha.copy_(torch.tensor(st["cql_log_alpha"]))
    pol.train()
    rec1, rec2 = CallRecorder(pol.critic1), CallRecorder(pol.critic2)
    rec1o, rec2o = CallRecorder(pol.critic1_old), CallRecorder(pol.critic2_old)
    feeder = NoiseFeeder(); feeder.install()
    out = OrderedDict()
    full = "tiny" in case
    keys = None
    try:
        for k, (b, n) in enumerate(zip(batches, noises)):
            feeder.normal_q = [n["eps_actor"], n["eps_next"], n["eps_pi"], n["eps_next_pi"]]
            feeder.uniform_q = [n["u_rand"]]
            rec1.outs.clear(); rec2.outs.clear(); rec1o.outs.clear(); rec2o.outs.clear()
            res = pol.learn(_tb(b))
            assert not feeder.normal_q and not feeder.uniform_q
            if keys is None:
                keys = list(res.keys())
            out[f"step{k}/losses"] = np.array([res[x] for x in keys], dtype=np.float64)
            if k == 0:
                # critic call order inside learn(): [q_a, q_data, q_pi, q_next_pi, q_rand]
                for nm, r in (("c1", rec1), ("c2", rec2)):
                    for j, tag in enumerate(("qa", "q", "q_pi", "q_next_pi", "q_rand")):
                        out[f"step0/{nm}_{tag}"] = r.outs[j]
                nq = np.minimum(rec1o.outs[0], rec2o.outs[0])
                out["step0/c1old_out"], out["step0/c2old_out"] = rec1o.outs[0], rec2o.outs[0]
                if not cfg["max_q_backup"] and cfg["deterministic_backup"]:
                    out["step0/target_q"] = (torch.tensor(b["rewards"]) + cfg["gamma"] * (1 - torch.tensor(b["terminals"]))
                                             * torch.tensor(nq)).numpy()
            if k in (0, len(batches) - 1):
                tag = f"state{k}"
                for nm, mod in (("actor", pol.actor), ("critic1", pol.critic1), ("critic2", pol.critic2),
                                ("critic1_old", pol.critic1_old), ("critic2_old", pol.critic2_old)):
                    _put_state(out, f"{tag}/{nm}", _state_of(mod), full)
                if cfg["auto_alpha"]:
                    out[f"{tag}/log_alpha"] = pol._log_alpha.detach().numpy().copy()
                out[f"{tag}/cql_log_alpha"] = pol.cql_log_alpha.detach().numpy().copy()
    finally:
        feeder.uninstall()
    out["loss_keys"] = np.array(keys)
    return out



# ----------------------------------------------------------------------------
# IQL
# ----------------------------------------------------------------------------

def gen_iql(ref, case):
    sys.path.insert(0, os.path.join(HERE, "..", ".."))
    from oracle import iql as oiql
    c, st, batches, noises = synth.iql_case_inputs(case)
    cfg = oiql.default_cfg(c["obs_dim"], c["act_dim"])
    cfg.update(hidden=c["hidden"]); cfg.update(c["over"])
    od, ad, hid = c["obs_dim"], c["act_dim"], c["hidden"]
    p_drop = cfg.get("actor_dropout")
    # run_iql.py:106: only the actor backbone is built with dropout_rate
    actor = ref.ActorProb(ref.MLP(od, hid, dropout_rate=p_drop), ref.DiagGaussian(hid[-1], ad, unbounded=False, conditioned_sigma=False))
    q1, q2, v = ref.Critic(ref.MLP(od + ad, hid)), ref.Critic(ref.MLP(od + ad, hid)), ref.Critic(ref.MLP(od, hid))
    _load(actor, st["actor"]); _load(q1, st["critic_q1"]); _load(q2, st["critic_q2"]); _load(v, st["critic_v"])
    pol = ref.IQLPolicy(actor, q1, q2, v, torch.optim.Adam(actor.parameters(), lr=cfg["actor_lr"]),
                        torch.optim.Adam(q1.parameters(), lr=cfg["critic_q_lr"]), torch.optim.Adam(q2.parameters(), lr=cfg["critic_q_lr"]),
                        torch.optim.Adam(v.parameters(), lr=cfg["critic_v_lr"]), action_space=_ActionSpace(ad), tau=cfg["tau"],
                        gamma=cfg["gamma"], expectile=cfg["expectile"], temperature=cfg["temperature"])
    _load(pol.critic_q1_old, st["critic_q1_old"]); _load(pol.critic_q2_old, st["critic_q2_old"])
    pol.train()
    rq1, rv = CallRecorder(pol.critic_q1), CallRecorder(pol.critic_v)
    out = OrderedDict(); full = "tiny" in case; keys = None
    # teacher-forced dropout: nn.Dropout.forward calls torch.nn.functional.dropout; the queued keep masks replace its Bernoulli draws with
    # ATen's own arithmetic (input * (mask / (1 - p)))
    drop_q = []
    orig_dropout = torch.nn.functional.dropout

    def fed_dropout(input, p=0.5, training=True, inplace=False):
        if not training:
            return input
        m = drop_q.pop(0)
        assert tuple(m.shape) == tuple(input.shape), (m.shape, tuple(input.shape))
        return input * (torch.tensor(m, dtype=input.dtype) / (1.0 - p))
    torch.nn.functional.dropout = fed_dropout
    for k, b in enumerate(batches):
        rq1.outs.clear(); rv.outs.clear()
        if p_drop:
            drop_q[:] = list(noises[k]["drop_actor"])
        res = pol.learn(_tb(b))
        assert not drop_q
        keys = keys or list(res.keys())
        out[f"step{k}/losses"] = np.array([res[x] for x in keys], dtype=np.float64)
        if k == 0:
            out["step0/q1"] = rq1.outs[0]
            out["step0/v"] = rv.outs[0]            # V(s) before the V update
            out["step0/next_v"] = rv.outs[1]       # V_new(s')
        if k in (0, len(batches) - 1):
            for nm, mod in (("actor", pol.actor), ("critic_q1", pol.critic_q1), ("critic_q2", pol.critic_q2), ("critic_v", pol.critic_v),
                            ("critic_q1_old", pol.critic_q1_old), ("critic_q2_old", pol.critic_q2_old)):
                _put_state(out, f"state{k}/{nm}", _state_of(mod), full)
    torch.nn.functional.dropout = orig_dropout
    out["loss_keys"] = np.array(keys)
    return out


# ----------------------------------------------------------------------------
# TD3+BC
# ----------------------------------------------------------------------------

def gen_td3bc(ref, case):
    sys.path.insert(0, os.path.join(HERE, "..", ".."))
    from oracle import td3bc as otd
    c, st, batches, noises = synth.td3bc_case_inputs(case)
    cfg = otd.default_cfg(c["obs_dim"], c["act_dim"])
    cfg.update(hidden=c["hidden"]); cfg.update(c["over"])
    od, ad, hid = c["obs_dim"], c["act_dim"], c["hidden"]
    actor = ref.Actor(ref.MLP(od, hid), ad, max_action=cfg["max_action"])
    c1, c2 = ref.Critic(ref.MLP(od + ad, hid)), ref.Critic(ref.MLP(od + ad, hid))
    _load(actor, st["actor"]); _load(c1, st["critic1"]); _load(c2, st["critic2"])
    pol = ref.TD3BCPolicy(actor, c1, c2, torch.optim.Adam(actor.parameters(), lr=cfg["actor_lr"]),
                          torch.optim.Adam(c1.parameters(), lr=cfg["critic_lr"]), torch.optim.Adam(c2.parameters(), lr=cfg["critic_lr"]),
                          tau=cfg["tau"], gamma=cfg["gamma"], max_action=cfg["max_action"], policy_noise=cfg["policy_noise"],
                          noise_clip=cfg["noise_clip"], update_actor_freq=cfg["update_actor_freq"], alpha=cfg["alpha"], scaler=None)
    _load(pol.actor_old, st["actor_old"]); _load(pol.critic1_old, st["critic1_old"]); _load(pol.critic2_old, st["critic2_old"])
    pol.train()
    r1 = CallRecorder(pol.critic1)
    feeder = NoiseFeeder(); feeder.install()
    out = OrderedDict(); full = "tiny" in case; keys = None
    try:
        for k, (b, n) in enumerate(zip(batches, noises)):
            feeder.normal_q = [n["eps_target"]]
            r1.outs.clear()
            res = pol.learn(_tb(b))
            assert not feeder.normal_q
            keys = keys or list(res.keys())
            out[f"step{k}/losses"] = np.array([res[x] for x in keys], dtype=np.float64)
            if k == 0:
                out["step0/q1"] = r1.outs[0]
                out["step0/q_pi"] = r1.outs[1]
            if k in (0, 1, len(batches) - 1):
                for nm, mod in (("actor", pol.actor), ("critic1", pol.critic1), ("critic2", pol.critic2), ("actor_old", pol.actor_old),
                                ("critic1_old", pol.critic1_old), ("critic2_old", pol.critic2_old)):
                    _put_state(out, f"state{k}/{nm}", _state_of(mod), full)
    finally:
        feeder.uninstall()
    out["loss_keys"] = np.array(keys)
    return out



# ----------------------------------------------------------------------------
# EDAC
# ----------------------------------------------------------------------------

def gen_edac(ref, case):
    sys.path.insert(0, os.path.join(HERE, "..", ".."))
    from oracle import edac as oed
    c, st, batches, noises = synth.edac_case_inputs(case)
    cfg = oed.default_cfg(c["obs_dim"], c["act_dim"])
    cfg.update(hidden=c["hidden"]); cfg.update(c["over"])
    od, ad, hid, K = c["obs_dim"], c["act_dim"], c["hidden"], cfg["num_critics"]
    actor = ref.ActorProb(ref.MLP(od, hid), ref.TanhDiagGaussian(hid[-1], ad, unbounded=True, conditioned_sigma=True))
    critics = ref.EnsembleCritic(od, ad, hid, num_ensemble=K)
    _load(actor, st["actor"]); _load(critics, st["critics"])
    aopt = torch.optim.Adam(actor.parameters(), lr=cfg["actor_lr"])
    copt = torch.optim.Adam(critics.parameters(), lr=cfg["critic_lr"])
    log_alpha = torch.tensor(st["log_alpha"].copy(), requires_grad=True)
    alpha = (cfg["target_entropy"], log_alpha, torch.optim.Adam([log_alpha], lr=cfg["alpha_lr"]))
    pol = ref.EDACPolicy(actor, critics, aopt, copt, tau=cfg["tau"], gamma=cfg["gamma"], alpha=alpha,
                         max_q_backup=cfg["max_q_backup"], deterministic_backup=cfg["deterministic_backup"], eta=cfg["eta"])
    _load(pol.critics_old, st["critics_old"])
    pol.train()
    rc = CallRecorder(pol.critics)
    feeder = NoiseFeeder(); feeder.install()
    out = OrderedDict(); full = "tiny" in case; keys = None
    try:
        for k, (b, n) in enumerate(zip(batches, noises)):
            feeder.normal_q = [n["eps_actor"], n["eps_next"]]
            rc.outs.clear()
            res = pol.learn(_tb(b))
            assert not feeder.normal_q
            keys = keys or list(res.keys())
            out[f"step{k}/losses"] = np.array([res[x] for x in keys], dtype=np.float64)
            if k == 0:
                out["step0/qas"] = rc.outs[0]
                out["step0/qs"] = rc.outs[1]
            if k in (0, len(batches) - 1):
                for nm, mod in (("actor", pol.actor), ("critics", pol.critics), ("critics_old", pol.critics_old)):
                    _put_state(out, f"state{k}/{nm}", _state_of(mod), full)
                out[f"state{k}/log_alpha"] = pol._log_alpha.detach().numpy().copy()
    finally:
        feeder.uninstall()
    out["loss_keys"] = np.array(keys)
    return out


GENERATORS = {"cql": (gen_cql, list(synth.CQL_CASES) + list(synth.CQL_EXTRA_CASES)), "iql": (gen_iql, list(synth.IQL_CASES)),
              "td3bc": (gen_td3bc, list(synth.TD3BC_CASES)), "edac": (gen_edac, list(synth.EDAC_CASES))}


def main(argv):
    # arguments: algorithm names, or algo:case to regenerate one fixture (e.g. cql:cql_hopper)
    algos = argv[1:] or list(GENERATORS)
    ref = _import_reference()
    torch.set_num_threads(4)
    for arg in algos:
        algo, _, only = arg.partition(":")
        fn, cases = GENERATORS[algo]
        for case in cases:
            if only and case != only:
                continue
            out = fn(ref, case)
            path = os.path.join(HERE, f"{case}.npz")
            np.savez_compressed(path, **out)
            print(f"wrote {path}: {len(out)} arrays, {os.path.getsize(path) / 1024:.1f} KiB")


if __name__ == "__main__":
    main(sys.argv)

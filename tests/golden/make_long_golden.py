#!/usr/bin/env python3
"""Long-horizon golden vectors: 200 teacher-forced steps of the REAL reference's ``CQLPolicy.learn``
(/root/reference/offlinerlkit/policy/model_free/cql.py:87-207) at the north-star shape, storing the loss trajectory only -- once from the
initial state of ``synth.cql_case_inputs("cql_halfcheetah_long")`` and once per PERTURBED initial state (every trainable parameter times
(1 + eps), eps in +-1e-7, +-2e-7: about one fp32 ulp).  The spread of the perturbed trajectories around the unperturbed one is the
reference's OWN divergence envelope: how far two runs of the same reference drift apart when their parameters differ in the last bit.  An
engine whose arithmetic differs from torch's in rounding only must stay inside a small multiple of that envelope
(tests/test_gpu_long_horizon.py); one whose arithmetic is coarser leaves it.

Build container only (the reference does not exist on the GPU box); only arrays of numbers are stored.
Usage:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_long_golden.py [cql_halfcheetah_long] [iql_hopper_long] [td3bc_halfcheetah_long] [edac_walker2d_long]   (default: all)
"""
from __future__ import annotations

import os
import sys
import time
from collections import OrderedDict

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.dont_write_bytecode = True
import make_golden as mg  # noqa: E402
import synth  # noqa: E402

CASE = "cql_halfcheetah_long"
PERTURBATIONS = (1e-7, -1e-7, 2e-7, -2e-7)


def run(ref, eps):
    sys.path.insert(0, os.path.join(HERE, "..", ".."))
    from oracle import cql as ocql
    c, st, batches, noises = synth.cql_case_inputs(CASE)
    cfg = ocql.default_cfg(c["obs_dim"], c["act_dim"])
    cfg.update(hidden=c["hidden"], num_repeat_actions=c["N"])
    cfg.update(c["over"])
    od, ad, hid = c["obs_dim"], c["act_dim"], c["hidden"]
    scale = np.float32(1.0 + eps)

    def pert(net):
        return OrderedDict((k, (v * scale).astype(np.float32)) for k, v in net.items())
    actor = ref.ActorProb(ref.MLP(od, hid), ref.TanhDiagGaussian(hid[-1], ad, unbounded=True, conditioned_sigma=True))
    c1, c2 = ref.Critic(ref.MLP(od + ad, hid)), ref.Critic(ref.MLP(od + ad, hid))
    mg._load(actor, pert(st["actor"])); mg._load(c1, pert(st["critic1"])); mg._load(c2, pert(st["critic2"]))
    log_alpha = torch.tensor(st["log_alpha"].copy(), requires_grad=True)
    alpha = (cfg["target_entropy"], log_alpha, torch.optim.Adam([log_alpha], lr=cfg["alpha_lr"]))
    pol = ref.CQLPolicy(actor, c1, c2, torch.optim.Adam(actor.parameters(), lr=cfg["actor_lr"]), torch.optim.Adam(c1.parameters(), lr=cfg["critic_lr"]),
                        torch.optim.Adam(c2.parameters(), lr=cfg["critic_lr"]), action_space=mg._ActionSpace(ad), tau=cfg["tau"], gamma=cfg["gamma"],
                        alpha=alpha, cql_weight=cfg["cql_weight"], temperature=cfg["temperature"], max_q_backup=cfg["max_q_backup"],
                        deterministic_backup=cfg["deterministic_backup"], with_lagrange=cfg["with_lagrange"],
                        lagrange_threshold=cfg["lagrange_threshold"], cql_alpha_lr=cfg["cql_alpha_lr"], num_repeart_actions=cfg["num_repeat_actions"])
    mg._load(pol.critic1_old, st["critic1_old"]); mg._load(pol.critic2_old, st["critic2_old"])       # (targets unperturbed)
    with torch.no_grad():
        pol.cql_log_alpha.copy_(torch.tensor(st["cql_log_alpha"]))
    pol.train()
    feeder = mg.NoiseFeeder(); feeder.install()
    losses, keys = [], None
    try:
        for b, n in zip(batches, noises):
            feeder.normal_q = [n["eps_actor"], n["eps_next"], n["eps_pi"], n["eps_next_pi"]]
            feeder.uniform_q = [n["u_rand"]]
            res = pol.learn(mg._tb(b))
            keys = keys or list(res.keys())
            losses.append([res[x] for x in keys])
    finally:
        feeder.uninstall()
    return np.array(losses, dtype=np.float64), keys


IQL_CASE = "iql_hopper_long"


def run_iql(ref, eps):
    """the same experiment for ``IQLPolicy.learn`` (/root/reference/offlinerlkit/policy/model_free/iql.py:86-139; no noise: the step is a
    deterministic function of the batch).  Construction as tests/golden/make_golden.py::gen_iql (run_iql.py:105-133)."""
    sys.path.insert(0, os.path.join(HERE, "..", ".."))
    from oracle import iql as oiql
    c, st, batches, _ = synth.iql_case_inputs(IQL_CASE)
    cfg = oiql.default_cfg(c["obs_dim"], c["act_dim"])
    cfg.update(hidden=c["hidden"]); cfg.update(c["over"])
    od, ad, hid = c["obs_dim"], c["act_dim"], c["hidden"]
    scale = np.float32(1.0 + eps)

    def pert(net):
        return OrderedDict((k, (v * scale).astype(np.float32)) for k, v in net.items())
    actor = ref.ActorProb(ref.MLP(od, hid), ref.DiagGaussian(hid[-1], ad, unbounded=False, conditioned_sigma=False))
    q1, q2, v = ref.Critic(ref.MLP(od + ad, hid)), ref.Critic(ref.MLP(od + ad, hid)), ref.Critic(ref.MLP(od, hid))
    mg._load(actor, pert(st["actor"])); mg._load(q1, pert(st["critic_q1"])); mg._load(q2, pert(st["critic_q2"])); mg._load(v, pert(st["critic_v"]))
    pol = ref.IQLPolicy(actor, q1, q2, v, torch.optim.Adam(actor.parameters(), lr=cfg["actor_lr"]),
                        torch.optim.Adam(q1.parameters(), lr=cfg["critic_q_lr"]), torch.optim.Adam(q2.parameters(), lr=cfg["critic_q_lr"]),
                        torch.optim.Adam(v.parameters(), lr=cfg["critic_v_lr"]), action_space=mg._ActionSpace(ad), tau=cfg["tau"],
                        gamma=cfg["gamma"], expectile=cfg["expectile"], temperature=cfg["temperature"])
    mg._load(pol.critic_q1_old, st["critic_q1_old"]); mg._load(pol.critic_q2_old, st["critic_q2_old"])       # (targets unperturbed)
    pol.train()
    losses, keys = [], None
    for b in batches:
        res = pol.learn(mg._tb(b))
        keys = keys or list(res.keys())
        losses.append([res[x] for x in keys])
    return np.array(losses, dtype=np.float64), keys


TD3BC_CASE = "td3bc_halfcheetah_long"
EDAC_CASE = "edac_walker2d_long"


def run_td3bc(ref, eps):
    """``TD3BCPolicy.learn`` (/root/reference/offlinerlkit/policy/model_free/td3bc.py:83-124), target-policy noise teacher-forced;
    construction as tests/golden/make_golden.py::gen_td3bc.  The actor's loss is reported on its update steps only (every 2nd)."""
    sys.path.insert(0, os.path.join(HERE, "..", ".."))
    from oracle import td3bc as otd
    c, st, batches, noises = synth.td3bc_case_inputs(TD3BC_CASE)
    cfg = otd.default_cfg(c["obs_dim"], c["act_dim"])
    cfg.update(hidden=c["hidden"]); cfg.update(c["over"])
    od, ad, hid = c["obs_dim"], c["act_dim"], c["hidden"]
    scale = np.float32(1.0 + eps)

    def pert(net):
        return OrderedDict((k, (v * scale).astype(np.float32)) for k, v in net.items())
    actor = ref.Actor(ref.MLP(od, hid), ad, max_action=cfg["max_action"])
    c1, c2 = ref.Critic(ref.MLP(od + ad, hid)), ref.Critic(ref.MLP(od + ad, hid))
    mg._load(actor, pert(st["actor"])); mg._load(c1, pert(st["critic1"])); mg._load(c2, pert(st["critic2"]))
    pol = ref.TD3BCPolicy(actor, c1, c2, torch.optim.Adam(actor.parameters(), lr=cfg["actor_lr"]),
                          torch.optim.Adam(c1.parameters(), lr=cfg["critic_lr"]), torch.optim.Adam(c2.parameters(), lr=cfg["critic_lr"]),
                          tau=cfg["tau"], gamma=cfg["gamma"], max_action=cfg["max_action"], policy_noise=cfg["policy_noise"],
                          noise_clip=cfg["noise_clip"], update_actor_freq=cfg["update_actor_freq"], alpha=cfg["alpha"], scaler=None)
    mg._load(pol.actor_old, st["actor_old"]); mg._load(pol.critic1_old, st["critic1_old"]); mg._load(pol.critic2_old, st["critic2_old"])
    pol.train()
    feeder = mg.NoiseFeeder(); feeder.install()
    losses, keys = [], None
    try:
        for b, n in zip(batches, noises):
            feeder.normal_q = [n["eps_target"]]
            res = pol.learn(mg._tb(b))
            keys = keys or list(res.keys())
            losses.append([res[x] for x in keys])
    finally:
        feeder.uninstall()
    return np.array(losses, dtype=np.float64), keys


def run_edac(ref, eps):
    """``EDACPolicy.learn`` (/root/reference/offlinerlkit/policy/model_free/edac.py:88-166) at the walker2d shape (10 critics,
    [256, 256, 256], eta 5); construction as tests/golden/make_golden.py::gen_edac."""
    sys.path.insert(0, os.path.join(HERE, "..", ".."))
    from oracle import edac as oed
    c, st, batches, noises = synth.edac_case_inputs(EDAC_CASE)
    cfg = oed.default_cfg(c["obs_dim"], c["act_dim"])
    cfg.update(hidden=c["hidden"]); cfg.update(c["over"])
    od, ad, hid, K = c["obs_dim"], c["act_dim"], c["hidden"], cfg["num_critics"]
    scale = np.float32(1.0 + eps)

    def pert(net):
        return OrderedDict((k, (v * scale).astype(np.float32)) for k, v in net.items())
    actor = ref.ActorProb(ref.MLP(od, hid), ref.TanhDiagGaussian(hid[-1], ad, unbounded=True, conditioned_sigma=True))
    critics = ref.EnsembleCritic(od, ad, hid, num_ensemble=K)
    mg._load(actor, pert(st["actor"])); mg._load(critics, pert(st["critics"]))
    aopt = torch.optim.Adam(actor.parameters(), lr=cfg["actor_lr"])
    copt = torch.optim.Adam(critics.parameters(), lr=cfg["critic_lr"])
    log_alpha = torch.tensor(st["log_alpha"].copy(), requires_grad=True)
    alpha = (cfg["target_entropy"], log_alpha, torch.optim.Adam([log_alpha], lr=cfg["alpha_lr"]))
    pol = ref.EDACPolicy(actor, critics, aopt, copt, tau=cfg["tau"], gamma=cfg["gamma"], alpha=alpha,
                         max_q_backup=cfg["max_q_backup"], deterministic_backup=cfg["deterministic_backup"], eta=cfg["eta"])
    mg._load(pol.critics_old, st["critics_old"])       # (targets unperturbed)
    pol.train()
    feeder = mg.NoiseFeeder(); feeder.install()
    losses, keys = [], None
    try:
        for b, n in zip(batches, noises):
            feeder.normal_q = [n["eps_actor"], n["eps_next"]]
            res = pol.learn(mg._tb(b))
            keys = keys or list(res.keys())
            losses.append([res[x] for x in keys])
    finally:
        feeder.uninstall()
    return np.array(losses, dtype=np.float64), keys


def generate(ref, case, runner):
    out = OrderedDict()
    t0 = time.time()
    base, keys = runner(ref, 0.0)
    out["losses"] = base
    out["loss_keys"] = np.array(keys)
    out["perturbations"] = np.array(PERTURBATIONS)
    for i, eps in enumerate(PERTURBATIONS):
        out[f"losses_perturbed{i}"], _ = runner(ref, eps)
        d = np.abs(out[f"losses_perturbed{i}"] - base) / np.maximum(np.abs(base), 1e-2 * np.abs(base).max(axis=0))
        print(f"{case} eps {eps:+.0e}: max relative deviation from the unperturbed run by step 20 / 50 / 100 / 200: "
              f"{d[:20].max():.2e} {d[:50].max():.2e} {d[:100].max():.2e} {d.max():.2e}")
    path = os.path.join(HERE, f"{case}.npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path}: {os.path.getsize(path) / 1024:.1f} KiB in {time.time() - t0:.0f} s")


def main():
    ref = mg._import_reference()
    torch.set_num_threads(4)
    table = ((CASE, run), (IQL_CASE, run_iql), (TD3BC_CASE, run_td3bc), (EDAC_CASE, run_edac))
    which = sys.argv[1:] or [c for c, _ in table]
    for case, runner in table:
        if case in which:
            generate(ref, case, runner)


if __name__ == "__main__":
    main()

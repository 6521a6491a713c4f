#!/usr/bin/env python3
"""Golden vectors for the model-based callers of the hot path: the REAL reference's ``MOPOPolicy.learn``
(policy/model_based/mopo.py:81-84 -> SACPolicy.learn, model_free/sac.py:88-140) and ``COMBOPolicy.learn``
(policy/model_based/combo.py:110-241) on synthetic real + model batches with teacher-forced noise.  Build container only.

``offlinerlkit.dynamics`` (imported by both files for a type annotation; its package pulls in gym / mujoco) is replaced by a shim
whose ``BaseDynamics`` is ``object``: ``learn`` never touches the dynamics model, which stays out of scope here.
"""
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
import make_golden as mg  # noqa: E402


def _import_mb(ref_ns):
    dyn = types.ModuleType("offlinerlkit.dynamics")
    dyn.BaseDynamics = object
    sys.modules["offlinerlkit.dynamics"] = dyn
    pkg = sys.modules["offlinerlkit.policy"]
    pkg.CQLPolicy = ref_ns.CQLPolicy
    mopo = importlib.import_module("offlinerlkit.policy.model_based.mopo").MOPOPolicy
    combo = importlib.import_module("offlinerlkit.policy.model_based.combo").COMBOPolicy
    return mopo, combo


def _build_sac_nets(ref, c, st):
    od, ad, hid = c["obs_dim"], c["act_dim"], c["hidden"]
    actor = ref.ActorProb(ref.MLP(od, hid), ref.TanhDiagGaussian(hid[-1], ad, unbounded=True, conditioned_sigma=True))
    c1, c2 = ref.Critic(ref.MLP(od + ad, hid)), ref.Critic(ref.MLP(od + ad, hid))
    mg._load(actor, st["actor"]); mg._load(c1, st["critic1"]); mg._load(c2, st["critic2"])
    return actor, c1, c2


def _tb2(b):
    return {part: {k: torch.tensor(v) for k, v in b[part].items()} for part in ("real", "fake")}


def _states(out, tag, pol, full, with_cql_alpha):
    for nm, mod in (("actor", pol.actor), ("critic1", pol.critic1), ("critic2", pol.critic2), ("critic1_old", pol.critic1_old),
                    ("critic2_old", pol.critic2_old)):
        mg._put_state(out, f"{tag}/{nm}", mg._state_of(mod), full)
    if pol._is_auto_alpha:
        out[f"{tag}/log_alpha"] = pol._log_alpha.detach().numpy().copy()
    if with_cql_alpha:
        out[f"{tag}/cql_log_alpha"] = pol.cql_log_alpha.detach().numpy().copy()


def gen_mopo(ref, MOPO, case):
    from oracle import sac as osac
    c, st, batches, noises = synth.mopo_case_inputs(case)
    cfg = osac.default_cfg(c["obs_dim"], c["act_dim"]); cfg.update(hidden=c["hidden"]); cfg.update(c["over"])
    actor, c1, c2 = _build_sac_nets(ref, c, st)
    if cfg["auto_alpha"]:
        log_alpha = torch.tensor(st["log_alpha"].copy(), requires_grad=True)
        alpha = (cfg["target_entropy"], log_alpha, torch.optim.Adam([log_alpha], lr=cfg["alpha_lr"]))
    else:
        alpha = cfg["alpha"]
    pol = MOPO(object(), actor, c1, c2, torch.optim.Adam(actor.parameters(), lr=cfg["actor_lr"]),
               torch.optim.Adam(c1.parameters(), lr=cfg["critic_lr"]), torch.optim.Adam(c2.parameters(), lr=cfg["critic_lr"]),
               tau=cfg["tau"], gamma=cfg["gamma"], alpha=alpha)
    mg._load(pol.critic1_old, st["critic1_old"]); mg._load(pol.critic2_old, st["critic2_old"])
    pol.train()
    rec1 = mg.CallRecorder(pol.critic1)
    feeder = mg.NoiseFeeder(); feeder.install()
    out = OrderedDict(); keys = None
    try:
        for k, (b, n) in enumerate(zip(batches, noises)):
            feeder.normal_q = [n["eps_next"], n["eps_actor"]]
            rec1.outs.clear()
            res = pol.learn(_tb2(b))
            assert not feeder.normal_q
            keys = keys or list(res.keys())
            out[f"step{k}/losses"] = np.array([res[x] for x in keys], dtype=np.float64)
            if k == 0:
                out["step0/c1_q"], out["step0/c1_qa"] = rec1.outs[0], rec1.outs[1]
            if k in (0, len(batches) - 1):
                _states(out, f"state{k}", pol, "tiny" in case, False)
    finally:
        feeder.uninstall()
    out["loss_keys"] = np.array(keys)
    return out


def gen_combo(ref, COMBO, case):
    from oracle import cql as ocql
    c, st, batches, noises = synth.combo_case_inputs(case)
    cfg = ocql.default_cfg(c["obs_dim"], c["act_dim"]); cfg.update(synth.combo_cfg(c))
    actor, c1, c2 = _build_sac_nets(ref, c, st)
    log_alpha = torch.tensor(st["log_alpha"].copy(), requires_grad=True)
    alpha = (cfg["target_entropy"], log_alpha, torch.optim.Adam([log_alpha], lr=cfg["alpha_lr"]))
    pol = COMBO(object(), actor, c1, c2, torch.optim.Adam(actor.parameters(), lr=cfg["actor_lr"]),
                torch.optim.Adam(c1.parameters(), lr=cfg["critic_lr"]), torch.optim.Adam(c2.parameters(), lr=cfg["critic_lr"]),
                action_space=mg._ActionSpace(c["act_dim"]), tau=cfg["tau"], gamma=cfg["gamma"], alpha=alpha, cql_weight=cfg["cql_weight"],
                temperature=cfg["temperature"], max_q_backup=cfg["max_q_backup"], deterministic_backup=cfg["deterministic_backup"],
                with_lagrange=cfg["with_lagrange"], lagrange_threshold=cfg["lagrange_threshold"], cql_alpha_lr=cfg["cql_alpha_lr"],
                num_repeart_actions=cfg["num_repeat_actions"], uniform_rollout=False, rho_s=c["over"].get("rho_s", "mix"))
    mg._load(pol.critic1_old, st["critic1_old"]); mg._load(pol.critic2_old, st["critic2_old"])
    with torch.no_grad():
        pol.cql_log_alpha.copy_(torch.tensor(st["cql_log_alpha"]))
    pol.train()
    rec1 = mg.CallRecorder(pol.critic1)
    feeder = mg.NoiseFeeder(); feeder.install()
    out = OrderedDict(); keys = None
    try:
        for k, (b, n) in enumerate(zip(batches, noises)):
            feeder.normal_q = [n["eps_actor"], n["eps_next"], n["eps_pi"], n["eps_next_pi"]]
            feeder.uniform_q = [n["u_rand"]]
            rec1.outs.clear()
            res = pol.learn(_tb2(b))
            assert not feeder.normal_q and not feeder.uniform_q
            keys = keys or list(res.keys())
            out[f"step{k}/losses"] = np.array([res[x] for x in keys], dtype=np.float64)
            if k == 0:
                # critic1 call order in learn(): q_a (mix), q (mix), q_pi, q_next_pi, q_rand, q_real
                for j, tag in enumerate(("qa", "q", "q_pi", "q_next_pi", "q_rand", "q_real")):
                    out[f"step0/c1_{tag}"] = rec1.outs[j]
            if k in (0, len(batches) - 1):
                _states(out, f"state{k}", pol, "tiny" in case, True)
    finally:
        feeder.uninstall()
    out["loss_keys"] = np.array(keys)
    return out


def main():
    sys.path.insert(0, os.path.join(HERE, "..", ".."))
    ref = mg._import_reference()
    MOPO, COMBO = _import_mb(ref)
    torch.set_num_threads(4)
    for case in synth.MOPO_CASES:
        out = gen_mopo(ref, MOPO, case)
        np.savez_compressed(os.path.join(HERE, f"{case}.npz"), **out)
        print("wrote", case, len(out), "arrays")
    for case in synth.COMBO_CASES:
        out = gen_combo(ref, COMBO, case)
        np.savez_compressed(os.path.join(HERE, f"{case}.npz"), **out)
        print("wrote", case, len(out), "arrays")


if __name__ == "__main__":
    main()

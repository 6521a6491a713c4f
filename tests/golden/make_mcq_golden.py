#!/usr/bin/env python3
"""Golden vectors for MCQPolicy.learn (policy/model_free/mcq.py:48-126, behaviour policy nets/vae.py) from the REAL reference on
synthetic batches with teacher-forced noise (build container only; arrays only).  Draw order per learn(): randn_like (VAE latent),
rsample (next actions), torch.randn (VAE.decode latents, clamped there), rsample (OOD actions), rsample (actor)."""
import importlib
import os
import sys
from collections import OrderedDict

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.dont_write_bytecode = True
import synth  # noqa: E402
import make_golden as mg  # noqa: E402


def gen(ref, MCQ, VAE, case):
    from oracle import mcq as omcq
    c, st, batches, noises = synth.mcq_case_inputs(case)
    cfg = omcq.default_cfg(c["obs_dim"], c["act_dim"]); cfg.update(synth.mcq_cfg(c))
    od, ad, hid = c["obs_dim"], c["act_dim"], c["hidden"]
    actor = ref.ActorProb(ref.MLP(od, hid), ref.TanhDiagGaussian(hid[-1], ad, unbounded=True, conditioned_sigma=True))
    c1, c2 = ref.Critic(ref.MLP(od + ad, hid)), ref.Critic(ref.MLP(od + ad, hid))
    vae = VAE(od, ad, cfg["vae_hidden"], cfg["latent_dim"], cfg["max_action"])
    mg._load(actor, st["actor"]); mg._load(c1, st["critic1"]); mg._load(c2, st["critic2"]); mg._load(vae, st["behavior_policy"])
    if cfg["auto_alpha"]:
        log_alpha = torch.tensor(st["log_alpha"].copy(), requires_grad=True)
        alpha = (cfg["target_entropy"], log_alpha, torch.optim.Adam([log_alpha], lr=cfg["alpha_lr"]))
    else:
        alpha = cfg["alpha"]
    pol = MCQ(actor, c1, c2, vae, torch.optim.Adam(actor.parameters(), lr=cfg["actor_lr"]), torch.optim.Adam(c1.parameters(), lr=cfg["critic_lr"]),
              torch.optim.Adam(c2.parameters(), lr=cfg["critic_lr"]), torch.optim.Adam(vae.parameters(), lr=cfg["behavior_policy_lr"]),
              tau=cfg["tau"], gamma=cfg["gamma"], alpha=alpha, lmbda=cfg["lmbda"], num_sampled_actions=cfg["num_sampled_actions"])
    mg._load(pol.critic1_old, st["critic1_old"]); mg._load(pol.critic2_old, st["critic2_old"])
    pol.train()
    rec1 = mg.CallRecorder(pol.critic1)
    feeder = mg.NoiseFeeder(); feeder.install()
    randn_q = []
    orig_randn = torch.randn

    def randn(*size, **kw):
        shape = tuple(size[0]) if len(size) == 1 and isinstance(size[0], (tuple, list, torch.Size)) else tuple(size)
        if randn_q:
            arr = randn_q.pop(0)
            assert tuple(arr.shape) == shape, (arr.shape, shape)
            return torch.tensor(arr)
        return orig_randn(*size, **kw)
    torch.randn = randn
    out = OrderedDict(); keys = None
    try:
        for k, (b, n) in enumerate(zip(batches, noises)):
            feeder.normal_q = [n["eps_vae"], n["eps_next"], n["eps_ood"], n["eps_actor"]]      # randn_like / rsample draws in order
            randn_q[:] = [n["z_ood"]]
            rec1.outs.clear()
            res = pol.learn(mg._tb(b))
            assert not feeder.normal_q and not randn_q
            keys = keys or list(res.keys())
            out[f"step{k}/losses"] = np.array([res[x] for x in keys], dtype=np.float64)
            if k == 0:
                out["step0/c1_q"], out["step0/c1_q_ood"], out["step0/c1_qa"] = rec1.outs[0], rec1.outs[1], rec1.outs[2]
            if k in (0, len(batches) - 1):
                full = "tiny" in case
                for nm, mod in (("actor", pol.actor), ("critic1", pol.critic1), ("critic2", pol.critic2), ("critic1_old", pol.critic1_old),
                                ("critic2_old", pol.critic2_old), ("behavior_policy", pol.behavior_policy)):
                    mg._put_state(out, f"state{k}/{nm}", mg._state_of(mod), full)
                if cfg["auto_alpha"]:
                    out[f"state{k}/log_alpha"] = pol._log_alpha.detach().numpy().copy()
    finally:
        feeder.uninstall()
        torch.randn = orig_randn
    out["loss_keys"] = np.array(keys)
    return out


def main():
    sys.path.insert(0, os.path.join(HERE, "..", ".."))
    ref = mg._import_reference()
    sys.modules["offlinerlkit.policy"].SACPolicy = importlib.import_module("offlinerlkit.policy.model_free.sac").SACPolicy
    MCQ = importlib.import_module("offlinerlkit.policy.model_free.mcq").MCQPolicy
    from offlinerlkit.nets import VAE
    torch.set_num_threads(4)
    for case in synth.MCQ_CASES:
        out = gen(ref, MCQ, VAE, case)
        np.savez_compressed(os.path.join(HERE, f"{case}.npz"), **out)
        print("wrote", case, len(out), "arrays")


if __name__ == "__main__":
    main()

"""GPU: the model-based callers of the hot path -- MOPOPolicy.learn (= SACPolicy.learn on a real + model batch,
mopo.py:81-84, sac.py:88-140) and COMBOPolicy.learn (combo.py:110-241) -- through the C ABI and through the reference-shaped policy
classes, against the oracle and the golden vectors of the real reference (tests/golden/make_mb_golden.py).  Gate: 1e-4 relative."""
import numpy as np
import pytest
import torch

import synth
from helpers import load_golden, mopo_oracle_setup, combo_oracle_setup, rel_err, scale_err, check_state_against_golden
from test_gpu_api import DEV, Space, load, state_close

pytestmark = pytest.mark.gpu
NETS = {"actor": 0, "critic1": 1, "critic2": 2, "critic1_old": 3, "critic2_old": 4}


def _engine(algo, c, cfg, st, R, precision, **over):
    from offlinerlkit import _engine
    B = c["B_real"] + c["B_fake"]
    o = dict(obs_dim=c["obs_dim"], act_dim=c["act_dim"], hidden=c["hidden"], batch_size=B, n_runs=R, precision=precision,
             target_entropy=cfg["target_entropy"], auto_alpha=int(cfg["auto_alpha"]), alpha=cfg["alpha"])
    o.update(over)
    eng = _engine.Engine(_engine.default_config(algo, **o))
    for r in range(R):
        for nm, nid in NETS.items():
            eng.set_net(r, nid, st[nm])
        eng.set_scalar(r, _engine.SCALAR_LOG_ALPHA, float(st["log_alpha"][0]))
        if "cql_log_alpha" in st:
            eng.set_scalar(r, _engine.SCALAR_CQL_LOG_ALPHA, float(st["cql_log_alpha"][0]))
    return eng


def _lead(d, R):
    if isinstance(d, dict):
        return {k: np.stack([v] * R) for k, v in d.items()}
    return [np.stack([v] * R) for v in d]


@pytest.mark.parametrize("case,R,precision", [(c, 1, 0) for c in synth.MOPO_CASES] + [("mopo_halfcheetah", 64, 1), ("mopo_halfcheetah", 64, 2)])
def test_sac_step_matches_oracle_and_reference_mopo(case, R, precision):
    from oracle import sac as osac
    cfg, st, batches, noises = mopo_oracle_setup(case)
    c = synth.MOPO_CASES[case]
    eng = _engine("sac", c, cfg, st, R, precision)
    g = load_golden(case)
    keys = [str(k) for k in g["loss_keys"]]
    assert eng.metric_names == keys
    try:
        for k, (b, n) in enumerate(zip(batches, noises)):
            mb = synth.mix_batch(b)
            res, aux = osac.learn(st, cfg, mb, n)
            m = eng.step(_lead(mb, R), _lead([n["eps_next"], n["eps_actor"]], R))
            ora = np.array([res[x] for x in keys])
            for r in {0, R - 1}:
                assert rel_err(m[r], ora, floor=1e-2) < 1e-4, (case, k, r, m[r], ora)
                assert rel_err(m[r], g[f"step{k}/losses"], floor=1e-2) < 1e-4, (case, k, r, m[r], g[f"step{k}/losses"])
            if k == 0:
                assert scale_err(eng.debug_read(0, "q1"), g["step0/c1_q"]) < 1e-4
                assert scale_err(eng.debug_read(0, "q1a"), g["step0/c1_qa"]) < 1e-4
                assert scale_err(eng.debug_read(0, "target_q"), aux["target_q"]) < 1e-4
            if precision == 0 and k in (0, len(batches) - 1):
                nets = {nm: eng.get_net(0, nid) for nm, nid in NETS.items()}
                check_state_against_golden(g, f"state{k}", nets, atol=4e-6 * (k + 1))
    finally:
        eng.close()


@pytest.mark.parametrize("case,R,precision", [(c, 1, 0) for c in synth.COMBO_CASES] + [("combo_halfcheetah", 32, 1), ("combo_halfcheetah", 32, 2), ("combo_tiny_model", 3, 1)])
def test_combo_step_matches_oracle_and_reference(case, R, precision):
    from oracle import cql as ocql
    cfg, st, batches, noises = combo_oracle_setup(case)
    c = synth.COMBO_CASES[case]
    c0, Bc = cfg["cons_rows"]
    eng = _engine("cql", c, cfg, st, R, precision, num_repeat_actions=c["N"], with_lagrange=int(cfg["with_lagrange"]),
                  cql_alpha_lr=cfg["cql_alpha_lr"], cql_weight=cfg["cql_weight"], cql_cons_row0=c0, cql_cons_rows=Bc, cql_real_rows=cfg["real_rows"])
    g = load_golden(case)
    keys = [str(k) for k in g["loss_keys"]]
    assert eng.metric_names == keys
    B = c["B_real"] + c["B_fake"]
    try:
        for k, (b, n) in enumerate(zip(batches, noises)):
            mb = synth.mix_batch(b)
            res, aux = ocql.learn(st, cfg, mb, n)
            m = eng.step(_lead(mb, R), _lead([n["eps_actor"], n["eps_next"], n["u_rand"], n["eps_pi"], n["eps_next_pi"]], R))
            ora = np.array([res[x] for x in keys])
            for r in {0, R - 1}:
                assert rel_err(m[r], ora, floor=1e-2) < 1e-4, (case, k, r, m[r], ora)
                assert rel_err(m[r], g[f"step{k}/losses"], floor=1e-2) < 1e-4, (case, k, r, m[r], g[f"step{k}/losses"])
            if k == 0:
                qall = eng.debug_read(0, "q1_all")
                BN = Bc * c["N"]
                assert qall.size == B + 3 * BN
                assert scale_err(qall[:B], g["step0/c1_q"]) < 1e-4
                for j, gk in enumerate(("step0/c1_q_pi", "step0/c1_q_next_pi", "step0/c1_q_rand")):
                    assert scale_err(qall[B + j * BN:B + (j + 1) * BN], g[gk]) < 1e-4, gk
            if precision == 0 and k in (0, len(batches) - 1):
                nets = {nm: eng.get_net(0, nid) for nm, nid in NETS.items()}
                check_state_against_golden(g, f"state{k}", nets, atol=4e-6 * (k + 1))
    finally:
        eng.close()


def _tb2(b):
    return {part: {k: torch.tensor(v, device=DEV) for k, v in b[part].items()} for part in ("real", "fake")}


def _modules(c, st):
    from offlinerlkit.modules import ActorProb, Critic, TanhDiagGaussian
    from offlinerlkit.nets import MLP
    od, ad, hid = c["obs_dim"], c["act_dim"], c["hidden"]
    actor = ActorProb(MLP(od, hid), TanhDiagGaussian(hid[-1], ad, unbounded=True, conditioned_sigma=True), DEV)
    c1, c2 = Critic(MLP(od + ad, hid), DEV), Critic(MLP(od + ad, hid), DEV)
    load(actor, st["actor"]); load(c1, st["critic1"]); load(c2, st["critic2"])
    return actor, c1, c2


class FakeDynamics:
    def step(self, obs, act):
        n = len(obs)
        return obs + 0.1, np.ones((n, 1), np.float32), np.zeros((n, 1), bool), {}


def test_mopo_policy_api():
    from offlinerlkit.policy import MOPOPolicy
    from oracle import sac as osac
    case = "mopo_tiny"
    cfg, st, batches, noises = mopo_oracle_setup(case)
    c = synth.MOPO_CASES[case]
    actor, c1, c2 = _modules(c, st)
    log_alpha = torch.tensor(st["log_alpha"].copy(), requires_grad=True, device=DEV)
    pol = MOPOPolicy(FakeDynamics(), actor, c1, c2, torch.optim.Adam(actor.parameters(), lr=cfg["actor_lr"]),
                     torch.optim.Adam(c1.parameters(), lr=cfg["critic_lr"]), torch.optim.Adam(c2.parameters(), lr=cfg["critic_lr"]),
                     tau=cfg["tau"], gamma=cfg["gamma"], alpha=(cfg["target_entropy"], log_alpha, torch.optim.Adam([log_alpha], lr=cfg["alpha_lr"])))
    load(pol.critic1_old, st["critic1_old"]); load(pol.critic2_old, st["critic2_old"])
    pol.train()
    for k, (b, n) in enumerate(zip(batches[:3], noises[:3])):
        res, _ = osac.learn(st, cfg, synth.mix_batch(b), n)
        out = pol.learn(_tb2(b), noise=[n["eps_next"], n["eps_actor"]])
        assert list(out.keys()) == list(res.keys())
        assert rel_err(np.array(list(out.values())), np.array(list(res.values())), floor=1e-2) < 1e-4, (k, out, res)
    state_close(pol, st, ("actor", "critic1", "critic2", "critic1_old", "critic2_old"), 3e-6)
    pol.eval()
    roll, info = pol.rollout(batches[0]["real"]["observations"], 3)
    assert roll["obss"].shape[0] == 3 * c["B_real"] and info["num_transitions"] == 3 * c["B_real"] and roll["actions"].shape[1] == c["act_dim"]
    with pytest.raises(NotImplementedError):
        pol.learn_n(1, None)


@pytest.mark.parametrize("case", ["combo_tiny", "combo_tiny_model"])
def test_combo_policy_api(case):
    from offlinerlkit.policy import COMBOPolicy
    from oracle import cql as ocql
    cfg, st, batches, noises = combo_oracle_setup(case)
    c = synth.COMBO_CASES[case]
    actor, c1, c2 = _modules(c, st)
    log_alpha = torch.tensor(st["log_alpha"].copy(), requires_grad=True, device=DEV)
    pol = COMBOPolicy(FakeDynamics(), actor, c1, c2, torch.optim.Adam(actor.parameters(), lr=cfg["actor_lr"]),
                      torch.optim.Adam(c1.parameters(), lr=cfg["critic_lr"]), torch.optim.Adam(c2.parameters(), lr=cfg["critic_lr"]),
                      action_space=Space(c["act_dim"]), tau=cfg["tau"], gamma=cfg["gamma"],
                      alpha=(cfg["target_entropy"], log_alpha, torch.optim.Adam([log_alpha], lr=cfg["alpha_lr"])), cql_weight=cfg["cql_weight"],
                      temperature=cfg["temperature"], max_q_backup=cfg["max_q_backup"], deterministic_backup=cfg["deterministic_backup"],
                      with_lagrange=cfg["with_lagrange"], lagrange_threshold=cfg["lagrange_threshold"], cql_alpha_lr=cfg["cql_alpha_lr"],
                      num_repeart_actions=cfg["num_repeat_actions"], rho_s=c["over"]["rho_s"])
    load(pol.critic1_old, st["critic1_old"]); load(pol.critic2_old, st["critic2_old"])
    pol.cql_log_alpha = torch.tensor(st["cql_log_alpha"].copy())
    pol.train()
    for k, (b, n) in enumerate(zip(batches[:3], noises[:3])):
        res, _ = ocql.learn(st, cfg, synth.mix_batch(b), n)
        out = pol.learn(_tb2(b), noise=[n["eps_actor"], n["eps_next"], n["u_rand"], n["eps_pi"], n["eps_next_pi"]])
        assert list(out.keys()) == list(res.keys())
        assert rel_err(np.array(list(out.values())), np.array(list(res.values())), floor=1e-2) < 1e-4, (k, out, res)
    state_close(pol, st, ("actor", "critic1", "critic2", "critic1_old", "critic2_old"), 3e-6)
    with pytest.raises(ValueError):
        pol.learn({k: torch.tensor(v, device=DEV) for k, v in synth.mix_batch(batches[0]).items()})


# ---------------------------------------------------------------------------------------------------------------------
# MCQ (mcq.py:48-126)
# ---------------------------------------------------------------------------------------------------------------------
MCQ_NETS = dict(NETS, vae_enc=7, vae_dec=8)


def _mcq_noise(n):
    return [n["eps_vae"], n["eps_next"], n["z_ood"], n["eps_ood"], n["eps_actor"]]


def _mcq_engine(c, cfg, st, R, precision):
    from offlinerlkit import _engine
    o = dict(obs_dim=c["obs_dim"], act_dim=c["act_dim"], hidden=c["hidden"], batch_size=c["B"], n_runs=R, precision=precision,
             target_entropy=cfg["target_entropy"], auto_alpha=int(cfg["auto_alpha"]), alpha=cfg["alpha"], vae_hidden=cfg["vae_hidden"],
             vae_latent=cfg["latent_dim"], mcq_lambda=cfg["lmbda"], num_repeat_actions=cfg["num_sampled_actions"], max_action=cfg["max_action"],
             actor_lr=cfg["actor_lr"], critic_lr=cfg["critic_lr"], alpha_lr=cfg["alpha_lr"], behavior_lr=cfg["behavior_policy_lr"])
    eng = _engine.Engine(_engine.default_config("mcq", **o))
    for r in range(R):
        for nm, nid in NETS.items():
            eng.set_net(r, nid, st[nm])
        eng.set_net(r, 7, st["behavior_policy"]); eng.set_net(r, 8, st["behavior_policy"])
        eng.set_scalar(r, _engine.SCALAR_LOG_ALPHA, float(st["log_alpha"][0]))
    return eng


@pytest.mark.parametrize("case,R,precision", [(c, 1, 0) for c in synth.MCQ_CASES] + [("mcq_hopper", 16, 1), ("mcq_hopper", 16, 2), ("mcq_tiny", 3, 1)])
def test_mcq_step_matches_oracle_and_reference(case, R, precision):
    from helpers import mcq_oracle_setup
    from oracle import mcq as omcq
    cfg, st, batches, noises = mcq_oracle_setup(case)
    c = synth.MCQ_CASES[case]
    eng = _mcq_engine(c, cfg, st, R, precision)
    g = load_golden(case)
    keys = [str(k) for k in g["loss_keys"]]
    assert eng.metric_names == keys
    B = c["B"]
    try:
        for k, (b, n) in enumerate(zip(batches, noises)):
            res, aux = omcq.learn(st, cfg, b, n)
            m = eng.step(_lead(b, R), _lead(_mcq_noise(n), R))
            ora = np.array([res[x] for x in keys])
            for r in {0, R - 1}:
                assert rel_err(m[r], ora, floor=1e-2) < 1e-4, (case, k, r, m[r], ora)
                assert rel_err(m[r], g[f"step{k}/losses"], floor=1e-2) < 1e-4, (case, k, r, m[r], g[f"step{k}/losses"])
            if k == 0:
                assert scale_err(eng.debug_read(0, "q1"), g["step0/c1_q"]) < 1e-4
                assert scale_err(eng.debug_read(0, "q1_ood"), g["step0/c1_q_ood"]) < 1e-4
                assert scale_err(eng.debug_read(0, "q1a"), g["step0/c1_qa"]) < 1e-4
                assert scale_err(eng.debug_read(0, "target_ood"), aux["target_q_ood"]) < 1e-4
                assert scale_err(eng.debug_read(0, "sampled_actions"), aux["sampled_actions"]) < 1e-4
            if precision == 0 and k in (0, len(batches) - 1):
                nets = {nm: eng.get_net(0, nid) for nm, nid in NETS.items()}
                vae = dict(eng.get_net(0, 7)); vae.update(eng.get_net(0, 8))
                nets["behavior_policy"] = vae
                check_state_against_golden(g, f"state{k}", nets, atol=6e-6 * (k + 1))
    finally:
        eng.close()


def test_mcq_policy_api():
    from helpers import mcq_oracle_setup
    from offlinerlkit.nets import VAE
    from offlinerlkit.policy import MCQPolicy
    from oracle import mcq as omcq
    case = "mcq_tiny"
    cfg, st, batches, noises = mcq_oracle_setup(case)
    c = synth.MCQ_CASES[case]
    actor, c1, c2 = _modules(c, st)
    vae = VAE(c["obs_dim"], c["act_dim"], cfg["vae_hidden"], cfg["latent_dim"], cfg["max_action"], device=DEV)
    load(vae, st["behavior_policy"])
    log_alpha = torch.tensor(st["log_alpha"].copy(), requires_grad=True, device=DEV)
    pol = MCQPolicy(actor, c1, c2, vae, torch.optim.Adam(actor.parameters(), lr=cfg["actor_lr"]), torch.optim.Adam(c1.parameters(), lr=cfg["critic_lr"]),
                    torch.optim.Adam(c2.parameters(), lr=cfg["critic_lr"]), torch.optim.Adam(vae.parameters(), lr=cfg["behavior_policy_lr"]),
                    tau=cfg["tau"], gamma=cfg["gamma"], alpha=(cfg["target_entropy"], log_alpha, torch.optim.Adam([log_alpha], lr=cfg["alpha_lr"])),
                    lmbda=cfg["lmbda"], num_sampled_actions=cfg["num_sampled_actions"])
    load(pol.critic1_old, st["critic1_old"]); load(pol.critic2_old, st["critic2_old"])
    assert any(k.startswith("behavior_policy.e1") for k in pol.state_dict())
    pol.train()
    for k, (b, n) in enumerate(zip(batches[:3], noises[:3])):
        res, _ = omcq.learn(st, cfg, b, n)
        out = pol.learn({kk: torch.tensor(v, device=DEV) for kk, v in b.items()}, noise=_mcq_noise(n))
        assert list(out.keys()) == list(res.keys())
        assert rel_err(np.array(list(out.values())), np.array(list(res.values())), floor=1e-2) < 1e-4, (k, out, res)
    state_close(pol, st, ("actor", "critic1", "critic2", "critic1_old", "critic2_old", "behavior_policy"), 3e-6)
    # the VAE module's parameters alias the engine arena: decode() with the trained weights
    pol.eval()
    o = torch.tensor(batches[0]["observations"][:4], device=DEV)
    z = torch.zeros(4, cfg["latent_dim"], device=DEV)
    dec = vae.decode(o, z).detach().cpu().numpy()
    ref, _, _ = omcq.vae_decode(st["behavior_policy"], batches[0]["observations"][:4], np.zeros((4, cfg["latent_dim"]), np.float32), cfg["max_action"])
    assert np.abs(dec - ref).max() < 1e-5

"""GPU: the reference-shaped Python surface (offlinerlkit.{nets,modules,buffer,policy,policy_trainer}) driving the HIP
engine.  Policies are built exactly like run_example/run_{cql,iql,td3bc,edac}.py build them (modules + torch Adam
optimizers), then checked against the oracle with teacher-forced noise."""
import os

import numpy as np
import pytest
import torch

import synth
import trainer_fakes as tf
from helpers import cql_oracle_setup, generic_oracle_setup, rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


class Space:
    def __init__(self, ad):
        self.low = -np.ones(ad, np.float32)
        self.high = np.ones(ad, np.float32)
        self.shape = (ad,)


def load(mod, arrays):
    mod.load_state_dict({k: torch.tensor(v) for k, v in arrays.items()}, strict=True)


def tb(b):
    return {k: torch.tensor(v, device=DEV) for k, v in b.items()}


def state_close(policy, st, names, atol):
    sd = policy.state_dict()
    for nm in names:
        for k, v in st[nm].items():
            got = sd[f"{nm}.{k}"].detach().cpu().numpy()
            d = np.abs(got - v)
            assert d.mean() < atol and (d > 10 * atol + 1e-4 * np.abs(v).max()).mean() < 2e-3, (nm, k, d.max())


def build_cql(case):
    from offlinerlkit.modules import ActorProb, Critic, TanhDiagGaussian
    from offlinerlkit.nets import MLP
    from offlinerlkit.policy import CQLPolicy
    cfg, st, batches, noises = cql_oracle_setup(case)
    c = synth.CQL_CASES[case]
    od, ad, hid = c["obs_dim"], c["act_dim"], c["hidden"]
    actor = ActorProb(MLP(od, hid), TanhDiagGaussian(hid[-1], ad, unbounded=True, conditioned_sigma=True), DEV)
    c1, c2 = Critic(MLP(od + ad, hid), DEV), Critic(MLP(od + ad, hid), DEV)
    load(actor, st["actor"]); load(c1, st["critic1"]); load(c2, st["critic2"])
    log_alpha = torch.tensor(st["log_alpha"].copy(), requires_grad=True, device=DEV)
    alpha = (cfg["target_entropy"], log_alpha, torch.optim.Adam([log_alpha], lr=cfg["alpha_lr"]))
    pol = CQLPolicy(actor, c1, c2, torch.optim.Adam(actor.parameters(), lr=cfg["actor_lr"]),
                    torch.optim.Adam(c1.parameters(), lr=cfg["critic_lr"]), torch.optim.Adam(c2.parameters(), lr=cfg["critic_lr"]),
                    action_space=Space(ad), tau=cfg["tau"], gamma=cfg["gamma"], alpha=alpha, cql_weight=cfg["cql_weight"],
                    temperature=cfg["temperature"], max_q_backup=cfg["max_q_backup"], deterministic_backup=cfg["deterministic_backup"],
                    with_lagrange=cfg["with_lagrange"], lagrange_threshold=cfg["lagrange_threshold"], cql_alpha_lr=cfg["cql_alpha_lr"],
                    num_repeart_actions=cfg["num_repeat_actions"])
    load(pol.critic1_old, st["critic1_old"]); load(pol.critic2_old, st["critic2_old"])
    pol.cql_log_alpha = torch.tensor(st["cql_log_alpha"].copy())
    return pol, cfg, st, batches, noises, log_alpha


@pytest.mark.parametrize("case", ["cql_tiny", "cql_tiny_lagrange", "cql_halfcheetah"])
def test_cql_policy_api(case):
    from oracle import cql as ocql, nn as onn
    pol, cfg, st, batches, noises, log_alpha = build_cql(case)
    assert set(pol.state_dict().keys()) == {f"{n}.{k}" for n in ("actor", "critic1", "critic1_old", "critic2", "critic2_old") for k in st[n]}
    pol.train()
    for k, (b, n) in enumerate(zip(batches[:3], noises[:3])):
        res, _ = ocql.learn(st, cfg, b, n)
        out = pol.learn(tb(b), noise=[n["eps_actor"], n["eps_next"], n["u_rand"], n["eps_pi"], n["eps_next_pi"]])
        assert list(out.keys()) == list(res.keys())
        assert rel_err(np.array(list(out.values())), np.array(list(res.values())), floor=1e-2) < 1e-4, (k, out, res)
    state_close(pol, st, ("actor", "critic1", "critic2", "critic1_old", "critic2_old"), 3e-6)
    # parameters alias the engine arena: select_action (torch forward) sees the trained weights
    pol.eval()
    obs = batches[0]["observations"][:5]
    a_det = pol.select_action(obs, deterministic=True)
    a_ref, _, _ = onn.tanh_gauss_fwd(st["actor"], obs, None)
    assert np.abs(a_det - a_ref).max() < 1e-5
    assert pol.select_action(obs[:1]).shape == (1, a_ref.shape[1])
    pol.sync_scalars()
    assert abs(float(log_alpha) - float(st["log_alpha"][0])) < 1e-6
    # load_state_dict writes through to the engine
    sd = {k: v.clone() for k, v in pol.state_dict().items()}
    sd["actor.dist_net.mu.bias"] += 0.5
    pol.load_state_dict(sd)
    a2 = pol.select_action(obs, deterministic=True)
    assert np.abs(a2 - a_det).max() > 1e-3
    got = pol.engine.get_net(0, 0)["dist_net.mu.bias"]
    assert np.allclose(got, sd["actor.dist_net.mu.bias"].cpu().numpy())


def test_iql_policy_api_and_lr_schedule():
    from offlinerlkit.modules import ActorProb, Critic, DiagGaussian
    from offlinerlkit.nets import MLP
    from offlinerlkit.policy import IQLPolicy
    mod, cfg, st, batches, _ = generic_oracle_setup("iql", "iql_tiny")
    c = synth.IQL_CASES["iql_tiny"]
    od, ad, hid = c["obs_dim"], c["act_dim"], c["hidden"]
    actor = ActorProb(MLP(od, hid), DiagGaussian(hid[-1], ad, unbounded=False, conditioned_sigma=False), DEV)
    q1, q2, v = Critic(MLP(od + ad, hid), DEV), Critic(MLP(od + ad, hid), DEV), Critic(MLP(od, hid), DEV)
    load(actor, st["actor"]); load(q1, st["critic_q1"]); load(q2, st["critic_q2"]); load(v, st["critic_v"])
    aopt = torch.optim.Adam(actor.parameters(), lr=cfg["actor_lr"])
    pol = IQLPolicy(actor, q1, q2, v, aopt, torch.optim.Adam(q1.parameters(), lr=cfg["critic_q_lr"]),
                    torch.optim.Adam(q2.parameters(), lr=cfg["critic_q_lr"]), torch.optim.Adam(v.parameters(), lr=cfg["critic_v_lr"]),
                    action_space=Space(ad), tau=cfg["tau"], gamma=cfg["gamma"], expectile=cfg["expectile"], temperature=cfg["temperature"])
    load(pol.critic_q1_old, st["critic_q1_old"]); load(pol.critic_q2_old, st["critic_q2_old"])
    sched = torch.optim.lr_scheduler.CosineAnnealingLR(aopt, 4)       # run_iql.py:132-133
    pol.train()
    for k, b in enumerate(batches[:4]):
        cfg["actor_lr"] = aopt.param_groups[0]["lr"]
        res, _ = mod.learn(st, cfg, b, None)
        out = pol.learn(tb(b))
        assert rel_err(np.array(list(out.values())), np.array(list(res.values())), floor=1e-2) < 1e-4, (k, out, res)
        sched.step()                                                   # mutates param_groups[0]["lr"]; honoured by the next learn()
    state_close(pol, st, ("actor", "critic_q1", "critic_q2", "critic_v", "critic_q1_old", "critic_q2_old"), 3e-6)
    pol.eval()
    a = pol.select_action(batches[0]["observations"][0], deterministic=True)
    assert a.shape == (1, ad) and np.abs(a).max() <= 1.0


def test_td3bc_policy_api():
    from offlinerlkit.modules import Actor, Critic
    from offlinerlkit.nets import MLP
    from offlinerlkit.policy import TD3BCPolicy
    from offlinerlkit.utils.scaler import StandardScaler
    mod, cfg, st, batches, noises = generic_oracle_setup("td3bc", "td3bc_tiny")
    c = synth.TD3BC_CASES["td3bc_tiny"]
    od, ad, hid = c["obs_dim"], c["act_dim"], c["hidden"]
    actor = Actor(MLP(od, hid), ad, max_action=cfg["max_action"], device=DEV)
    c1, c2 = Critic(MLP(od + ad, hid), DEV), Critic(MLP(od + ad, hid), DEV)
    load(actor, st["actor"]); load(c1, st["critic1"]); load(c2, st["critic2"])
    scaler = StandardScaler(mu=np.zeros((1, od), np.float32), std=np.full((1, od), 2.0, np.float32))
    pol = TD3BCPolicy(actor, c1, c2, torch.optim.Adam(actor.parameters(), lr=cfg["actor_lr"]), torch.optim.Adam(c1.parameters(), lr=cfg["critic_lr"]),
                      torch.optim.Adam(c2.parameters(), lr=cfg["critic_lr"]), tau=cfg["tau"], gamma=cfg["gamma"], max_action=cfg["max_action"],
                      policy_noise=cfg["policy_noise"], noise_clip=cfg["noise_clip"], update_actor_freq=cfg["update_actor_freq"],
                      alpha=cfg["alpha"], scaler=scaler)
    load(pol.actor_old, st["actor_old"]); load(pol.critic1_old, st["critic1_old"]); load(pol.critic2_old, st["critic2_old"])
    pol.train()
    for k, (b, n) in enumerate(zip(batches, noises)):
        res, _ = mod.learn(st, cfg, b, n)
        out = pol.learn(tb(b), noise=[n["eps_target"]])
        assert rel_err(np.array(list(out.values())), np.array(list(res.values())), floor=1e-2) < 1e-4, (k, out, res)
    assert pol._cnt == len(batches) == st["cnt"]
    state_close(pol, st, ("actor", "critic1", "critic2", "actor_old", "critic1_old", "critic2_old"), 3e-6)
    pol.eval()
    o = batches[0]["observations"][:3]
    a = pol.select_action(o, deterministic=True)
    ref, _ = mod.det_actor_fwd(st["actor"], (o / 2.0).astype(np.float32), cfg["max_action"])
    assert np.abs(a - ref).max() < 1e-5


def test_edac_policy_api():
    from offlinerlkit.modules import ActorProb, EnsembleCritic, TanhDiagGaussian
    from offlinerlkit.nets import MLP
    from offlinerlkit.policy import EDACPolicy
    mod, cfg, st, batches, noises = generic_oracle_setup("edac", "edac_tiny")
    c = synth.EDAC_CASES["edac_tiny"]
    od, ad, hid, K = c["obs_dim"], c["act_dim"], c["hidden"], cfg["num_critics"]
    actor = ActorProb(MLP(od, hid), TanhDiagGaussian(hid[-1], ad, unbounded=True, conditioned_sigma=True), DEV)
    critics = EnsembleCritic(od, ad, hid, num_ensemble=K, device=DEV)
    load(actor, st["actor"]); load(critics, st["critics"])
    log_alpha = torch.tensor(st["log_alpha"].copy(), requires_grad=True, device=DEV)
    pol = EDACPolicy(actor, critics, torch.optim.Adam(actor.parameters(), lr=cfg["actor_lr"]), torch.optim.Adam(critics.parameters(), lr=cfg["critic_lr"]),
                     tau=cfg["tau"], gamma=cfg["gamma"], alpha=(cfg["target_entropy"], log_alpha, torch.optim.Adam([log_alpha], lr=cfg["alpha_lr"])),
                     max_q_backup=cfg["max_q_backup"], deterministic_backup=cfg["deterministic_backup"], eta=cfg["eta"])
    load(pol.critics_old, st["critics_old"])
    assert "critics.model.0.saved_weight" in pol.state_dict()
    pol.train()
    for k, (b, n) in enumerate(zip(batches[:3], noises[:3])):
        res, _ = mod.learn(st, cfg, b, n)
        out = pol.learn(tb(b), noise=[n["eps_actor"], n["eps_next"]])
        assert rel_err(np.array(list(out.values())), np.array(list(res.values())), floor=1e-2) < 1e-4, (k, out, res)
    sd = pol.state_dict()
    for k, v in st["critics"].items():
        if "saved_" not in k:
            assert np.abs(sd[f"critics.{k}"].cpu().numpy() - v).mean() < 3e-6


def test_replay_buffer_api_matches_numpy_semantics():
    from offlinerlkit.buffer import ReplayBuffer
    ds = tf.dataset()
    buf = ReplayBuffer(tf.N_DATA, (tf.OBS,), np.float32, tf.ACT, np.float32, device=DEV)
    buf.load_dataset(ds)
    np.random.seed(5)
    out = buf.sample(32)
    np.random.seed(5)
    idx = np.random.randint(0, tf.N_DATA, size=32)
    assert set(out) == {"observations", "actions", "next_observations", "terminals", "rewards"}
    assert out["rewards"].shape == (32, 1) and out["terminals"].shape == (32, 1) and out["observations"].is_cuda
    assert np.array_equal(out["observations"].cpu().numpy(), ds["observations"][idx])
    assert np.array_equal(out["actions"].cpu().numpy(), ds["actions"][idx])
    assert np.array_equal(out["rewards"].cpu().numpy()[:, 0], ds["rewards"][idx])
    assert np.array_equal(out["terminals"].cpu().numpy()[:, 0], ds["terminals"][idx].astype(np.float32))
    mean, std = buf.normalize_obs()
    assert mean.shape == (1, tf.OBS) and np.allclose(mean, ds["observations"].mean(0, keepdims=True))
    np.random.seed(5)
    o2 = buf.sample(32)["observations"].cpu().numpy()
    assert np.abs(o2 - (ds["observations"][idx] - mean) / std).max() < 1e-5
    buf.add_batch(ds["observations"][:3], ds["next_observations"][:3], ds["actions"][:3], ds["rewards"][:3, None], ds["terminals"][:3, None])
    assert buf.sample(8)["observations"].shape == (8, tf.OBS)          # re-upload after host-side mutation
    assert buf.sample_all()["observations"].shape == (tf.N_DATA, tf.OBS)


def test_trainer_reference_trace_with_hbm_buffer(tmp_path):
    """The unfused trainer loop over the HBM-resident ReplayBuffer reproduces the real reference's logged trace."""
    from offlinerlkit.buffer import ReplayBuffer
    from offlinerlkit.policy_trainer import MFPolicyTrainer
    from offlinerlkit.utils.logger import Logger
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "trainer_trace.npz"), allow_pickle=False)
    buf = ReplayBuffer(tf.N_DATA, (tf.OBS,), np.float32, tf.ACT, np.float32, device=DEV)
    buf.load_dataset(tf.dataset())
    logger = Logger(str(tmp_path), {"policy_training_progress": "csv"})
    pol, sched = tf.FakePolicy(), tf.FakeScheduler()
    np.random.seed(tf.SEED)
    res = MFPolicyTrainer(pol, tf.FakeEnv(), buf, logger, epoch=tf.EPOCHS, step_per_epoch=tf.STEPS, batch_size=tf.BATCH,
                          eval_episodes=tf.EVAL_EPS, lr_scheduler=sched, fused=False).train()
    np.testing.assert_allclose(pol.obs_sums, g["obs_sums"], rtol=1e-5)
    lines = open(tmp_path / "record" / "policy_training_progress.csv").read().strip().split("\n")
    rows = np.array([[float(x) if x else np.nan for x in ln.split(",")] for ln in lines[1:]])
    np.testing.assert_allclose(rows, g["csv_rows"], rtol=1e-5, atol=1e-7)
    assert abs(res["last_10_performance"] - float(g["last_10_performance"][0])) < 1e-9


def test_trainer_fused_epoch_with_cql_policy(tmp_path):
    from offlinerlkit.buffer import ReplayBuffer
    from offlinerlkit.policy_trainer import MFPolicyTrainer
    from offlinerlkit.utils.logger import Logger
    pol, cfg, st, batches, noises, _ = build_cql("cql_tiny")
    c = synth.CQL_CASES["cql_tiny"]
    ds = synth.make_dataset(4, 3000, c["obs_dim"], c["act_dim"])
    buf = ReplayBuffer(3000, (c["obs_dim"],), np.float32, c["act_dim"], np.float32, device=DEV)
    buf.load_dataset(ds)

    class Env(tf.FakeEnv):
        def reset(self):
            self.t = 0
            return np.zeros(c["obs_dim"], np.float32)

        def step(self, a):
            self.t += 1
            return np.zeros(c["obs_dim"], np.float32), 1.0, self.t >= 3, {}
    logger = Logger(str(tmp_path), {"policy_training_progress": "csv"})
    before = {k: v.clone() for k, v in pol.state_dict().items()}
    res = MFPolicyTrainer(pol, Env(), buf, logger, epoch=2, step_per_epoch=25, batch_size=c["B"], eval_episodes=2).train()
    lines = open(tmp_path / "record" / "policy_training_progress.csv").read().strip().split("\n")
    head = lines[0].split(",")
    for k in ("loss/actor", "loss/critic1", "loss/critic2", "loss/alpha", "alpha", "eval/episode_reward", "timestep"):
        assert k in head
    assert len(lines) == 3 and np.isfinite([float(x) for x in lines[2].split(",") if x]).all()
    after = pol.state_dict()
    assert any((after[k] - before[k]).abs().max() > 0 for k in before)
    assert pol.engine.step_count() == 50
    sd = torch.load(tmp_path / "model" / "policy.pth", weights_only=True)
    assert set(sd) == set(after)


def _cql_buffer(c, n=3000, seed=4):
    from offlinerlkit.buffer import ReplayBuffer
    ds = synth.make_dataset(seed, n, c["obs_dim"], c["act_dim"])
    buf = ReplayBuffer(n, (c["obs_dim"],), np.float32, c["act_dim"], np.float32, device=DEV)
    buf.load_dataset(ds)
    return buf


def test_policy_device_streams_follow_the_launcher_seed():
    """run_cql.py:75-79 seeds torch / numpy per experiment: policies built under different torch seeds must draw different
    minibatch index and noise streams in the fused path, the same seed (same process history) the same ones, and a re-bind must
    not replay a stream."""
    from offlinerlkit.policy import base_policy as bp
    c = synth.CQL_CASES["cql_tiny"]
    buf = _cql_buffer(c)

    def first_batch(torch_seed):
        bp._BIND_COUNTER = 0                       # as in a fresh process
        torch.manual_seed(torch_seed)
        pol = build_cql("cql_tiny")[0]
        pol.learn_n(1, buf, c["B"])
        return pol, pol.engine.debug_read(0, "b_obs"), pol.engine.debug_read(0, "n_eps_actor")
    p1, o1, e1 = first_batch(1)
    p2, o2, e2 = first_batch(2)
    p3, o3, e3 = first_batch(1)
    assert not np.array_equal(o1, o2) and not np.array_equal(e1, e2)
    assert np.array_equal(o1, o3) and np.array_equal(e1, e3)
    p1.set_engine_options()                        # drops the engine; the next learn_n re-binds with a new stream key
    p1.learn_n(1, buf, c["B"])
    assert not np.array_equal(p1.engine.debug_read(0, "b_obs"), o1)


def test_policy_rebind_on_batch_size_change_keeps_optimizer_state():
    """The reference accepts any batch size per learn() call; the engine is rebuilt for a new one, and Adam's moments, the step
    count (bias correction) and the scalar optimizers must survive: three steps with batch sizes 16, 16, 8 against the oracle."""
    from oracle import cql as ocql
    pol, cfg, st, batches, noises, _ = build_cql("cql_tiny")
    pol.train()
    N = cfg["num_repeat_actions"]

    def cut(b, n, rows):
        bb = {k: v[:rows] for k, v in b.items()}
        nn_ = dict(eps_actor=n["eps_actor"][:rows], eps_next=n["eps_next"][:rows], u_rand=n["u_rand"][:rows * N],
                   eps_pi=n["eps_pi"][:rows * N], eps_next_pi=n["eps_next_pi"][:rows * N])
        return bb, nn_
    for k, rows in enumerate((16, 16, 8)):
        b, n = cut(batches[k], noises[k], rows)
        res, _ = ocql.learn(st, cfg, b, n)
        out = pol.learn(tb(b), noise=[n["eps_actor"], n["eps_next"], n["u_rand"], n["eps_pi"], n["eps_next_pi"]])
        assert rel_err(np.array(list(out.values())), np.array(list(res.values())), floor=1e-2) < 1e-4, (k, out, res)
    assert pol.engine.step_count() == 3
    m, v = pol.engine.adam_state(0, 1)
    assert np.abs(m).max() > 0 and np.abs(v).max() > 0
    state_close(pol, st, ("actor", "critic1", "critic2", "critic1_old", "critic2_old"), 3e-6)


def test_policy_rejects_configurations_the_kernels_hard_code():
    from offlinerlkit.modules import ActorProb, Critic, TanhDiagGaussian
    from offlinerlkit.nets import MLP
    from offlinerlkit.policy import CQLPolicy
    od, ad, hid = 5, 3, [32, 32]
    actor = ActorProb(MLP(od, hid), TanhDiagGaussian(hid[-1], ad, unbounded=True, conditioned_sigma=True, sigma_min=-20.0), DEV)
    c1, c2 = Critic(MLP(od + ad, hid), DEV), Critic(MLP(od + ad, hid), DEV)
    pol = CQLPolicy(actor, c1, c2, torch.optim.Adam(actor.parameters(), lr=1e-4), torch.optim.Adam(c1.parameters(), lr=3e-4),
                    torch.optim.Adam(c2.parameters(), lr=3e-4), action_space=Space(ad), alpha=0.2)
    b = tb(synth.make_batch(np.random.RandomState(0), 16, od, ad))
    with pytest.raises(NotImplementedError, match="log-sigma"):
        pol.learn(b)
    actor2 = ActorProb(MLP(od, hid), TanhDiagGaussian(hid[-1], ad, unbounded=True, conditioned_sigma=True), DEV)
    pol2 = CQLPolicy(actor2, c1, c2, torch.optim.Adam(actor2.parameters(), lr=1e-4), torch.optim.Adam(c1.parameters(), lr=3e-4),
                     torch.optim.Adam(c2.parameters(), lr=3e-4, betas=(0.5, 0.9)), action_space=Space(ad), alpha=0.2)
    with pytest.raises(NotImplementedError, match="betas"):
        pol2.learn(b)


def test_multi_run_policy_through_the_reference_shaped_api(tmp_path):
    """n_runs independent seeds behind ONE policy object (BASELINE config 5 keeps 8 per GPU): run 0 = the modules as built, runs
    r > 0 re-initialised; learn_n reports the mean and run<i>/<key>; select_run re-points state_dict / select_action; the trainer
    logs and checkpoints every run."""
    from offlinerlkit.policy_trainer import MFPolicyTrainer
    from offlinerlkit.utils.logger import Logger
    R = 4
    pol, cfg, st, batches, noises, log_alpha = build_cql("cql_tiny")
    c = synth.CQL_CASES["cql_tiny"]
    buf = _cql_buffer(c)
    pol.set_engine_options(n_runs=R, seed=123)
    out = pol.learn_n(5, buf, c["B"])
    keys = pol.engine.metric_names
    assert all(k in out for k in keys) and all(f"run{r}/{k}" in out for r in range(R) for k in keys)
    for k in keys:
        assert abs(out[k] - np.mean([out[f"run{r}/{k}"] for r in range(R)])) < 1e-5 * max(1.0, abs(out[k]))
    sds = [pol.run_state_dict(r) for r in range(R)]
    for r in range(1, R):
        assert (sds[r]["actor.backbone.model.0.weight"] - sds[0]["actor.backbone.model.0.weight"]).abs().max() > 1e-3     # own initialisation
    obs = batches[0]["observations"][:4]
    pol.eval()
    pol.select_run(0); a0 = pol.select_action(obs, deterministic=True)
    pol.select_run(2); a2 = pol.select_action(obs, deterministic=True)
    assert np.abs(a0 - a2).max() > 1e-4
    assert torch.equal(pol.state_dict()["actor.dist_net.mu.bias"], sds[2]["actor.dist_net.mu.bias"])
    # a shared batch through learn(): every run steps on it
    res = pol.learn(tb(batches[0]))
    assert f"run{R - 1}/loss/critic1" in res and np.isfinite(list(res.values())).all()

    class Env(tf.FakeEnv):
        def reset(self):
            self.t = 0
            return np.zeros(c["obs_dim"], np.float32)

        def step(self, a):
            self.t += 1
            return np.full(c["obs_dim"], 0.1 * self.t, np.float32), float(a.sum()), self.t >= 3, {}
    logger = Logger(str(tmp_path), {"policy_training_progress": "csv"})
    trainer = MFPolicyTrainer(pol, Env(), buf, logger, epoch=2, step_per_epoch=10, batch_size=c["B"], eval_episodes=2)
    trainer.train()
    head = open(tmp_path / "record" / "policy_training_progress.csv").read().split("\n")[0].split(",")
    for r in range(R):
        assert f"run{r}/loss/critic1" in head and f"run{r}/eval/episode_reward" in head
    assert "loss/critic1" in head and "eval/episode_reward" in head
    for r in range(R):
        sd = torch.load(tmp_path / "model" / f"policy_run{r}.pth", weights_only=True)
        assert set(sd) == set(sds[0])
    assert (tmp_path / "model" / "policy.pth").exists()


@pytest.mark.parametrize("case", ["timeouts", "next_obs_terminate_on_end", "max_episode_steps"])
def test_reference_pinned_dataset_through_load_dataset_into_hbm_and_back(case):
    """SURVEY §8(f)2 on hardware: a trajectory dict -> ``qlearning_dataset`` (selection pinned bit for bit by the REAL reference's output,
    tests/golden/dataset_golden.npz) -> ``ReplayBuffer.load_dataset`` -> ``orl_buffer_load`` (16-B padded SoA in HBM) -> ``sample`` with a fixed
    numpy index stream: every sampled row is byte-identical to the corresponding row of the reference's own output arrays
    (load_dataset.py:17-147, buffer.py:72-106)."""
    import make_dataset_golden as mg
    from offlinerlkit.buffer import ReplayBuffer
    from offlinerlkit.utils.load_dataset import qlearning_dataset
    gold = np.load(os.path.join(os.path.dirname(__file__), "golden", "dataset_golden.npz"), allow_pickle=False)
    kw, use_timeouts, qkw = mg.CASES[case]
    d = mg.synth_trajectories(**kw)
    if not use_timeouts:
        d.pop("timeouts")
    q = qlearning_dataset(mg.FakeEnv(kw["max_len"]), dataset=d, **qkw)
    n, od, ad = len(q["rewards"]), q["observations"].shape[1], q["actions"].shape[1]
    assert n == len(gold[f"{case}/out/rewards"])
    buf = ReplayBuffer(n, (od,), np.float32, ad, np.float32, device=DEV)
    buf.load_dataset(q)
    for seed, batch in ((0, 7), (1, 64), (2, 256)):
        np.random.seed(seed)
        out = buf.sample(batch)
        np.random.seed(seed)
        idx = np.random.randint(0, n, size=batch)                                  # buffer.py:98
        for k in ("observations", "actions", "next_observations"):
            assert np.array_equal(out[k].cpu().numpy(), gold[f"{case}/out/{k}"][idx]), (case, k)
        assert np.array_equal(out["rewards"].cpu().numpy()[:, 0], gold[f"{case}/out/rewards"][idx])
        assert np.array_equal(out["terminals"].cpu().numpy()[:, 0], gold[f"{case}/out/terminals"][idx].astype(np.float32))
    # the device store holds exactly the reference's transitions: every row once, through the C ABI gather with idx = 0 .. n - 1
    dev = buf.device_buffer()
    assert dev.size() == n


def test_iql_policy_with_actor_dropout_matches_reference_fixture():
    """run_iql.py --dropout_rate: nn.Dropout behind every ReLU of the ACTOR backbone (nets/mlp.py:16-24, run_iql.py:106).  The policy is
    built from our MLP(dropout_rate=p) exactly as the launcher builds it; the keep masks of the reference's draws are teacher-forced; losses
    follow the fixture captured from the REAL reference (tests/golden/iql_tiny_dropout.npz) and the oracle at the 1e-4 gate, state_dict keys
    are the reference's (backbone.model.{0, 3}), evaluation runs without dropout, and a critic backbone with dropout is refused."""
    from helpers import load_golden
    from offlinerlkit.modules import ActorProb, Critic, DiagGaussian
    from offlinerlkit.nets import MLP
    from offlinerlkit.policy import IQLPolicy
    case = "iql_tiny_dropout"
    mod, cfg, st, batches, noises = generic_oracle_setup("iql", case)
    c = synth.IQL_CASES[case]
    od, ad, hid, p = c["obs_dim"], c["act_dim"], c["hidden"], cfg["actor_dropout"]
    g = load_golden(case)

    def build(critic_dropout=None):
        actor = ActorProb(MLP(od, hid, dropout_rate=p), DiagGaussian(hid[-1], ad, unbounded=False, conditioned_sigma=False), DEV)
        q1, q2, v = Critic(MLP(od + ad, hid, dropout_rate=critic_dropout), DEV), Critic(MLP(od + ad, hid), DEV), Critic(MLP(od, hid), DEV)
        load(actor, st["actor"]); load(q2, st["critic_q2"]); load(v, st["critic_v"])
        if critic_dropout is None:
            load(q1, st["critic_q1"])
        adam = lambda m, lr: torch.optim.Adam(m.parameters(), lr=lr)
        return IQLPolicy(actor, q1, q2, v, adam(actor, cfg["actor_lr"]), adam(q1, cfg["critic_q_lr"]), adam(q2, cfg["critic_q_lr"]),
                         adam(v, cfg["critic_v_lr"]), action_space=Space(ad), tau=cfg["tau"], gamma=cfg["gamma"], expectile=cfg["expectile"],
                         temperature=cfg["temperature"])
    pol = build()
    load(pol.critic_q1_old, st["critic_q1_old"]); load(pol.critic_q2_old, st["critic_q2_old"])
    assert "actor.backbone.model.3.weight" in pol.state_dict() and "actor.backbone.model.2.weight" not in pol.state_dict()
    pol.train()
    for k, (b, n) in enumerate(zip(batches, noises)):
        res, _ = mod.learn(st, cfg, b, n)
        out = pol.learn(tb(b), noise=list(n["drop_actor"]))
        got = np.array([out[x] for x in res])
        assert rel_err(got, np.array(list(res.values())), floor=1e-2) < 1e-4, (k, out, res)
        assert rel_err(got, g[f"step{k}/losses"], floor=1e-2) < 1e-4, (k, out, g[f"step{k}/losses"])
    state_close(pol, st, ("actor", "critic_q1", "critic_q2", "critic_v"), 3e-6)
    # device-drawn masks: finite, and the keep rate is 1 - p
    out = pol.learn(tb(batches[0]))
    assert np.isfinite(list(out.values())).all()
    keep = pol.engine.debug_read(0, "n_drop_a0")
    assert set(np.unique(keep)) <= {0.0, 1.0} and abs(keep.mean() - (1 - p)) < 0.08
    pol.eval()
    a1 = pol.select_action(batches[0]["observations"], deterministic=True)
    a2 = pol.select_action(batches[0]["observations"], deterministic=True)
    assert np.array_equal(a1, a2)                                   # eval mode: nn.Dropout is the identity
    assert np.allclose(pol.select_action_runs(batches[0]["observations"][None])[0], a1, atol=1e-6)
    with pytest.raises(NotImplementedError):
        bad = build(critic_dropout=0.1)
        bad.learn(tb(batches[0]))

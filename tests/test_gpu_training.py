"""GPU, end to end: dataset -> HBM ReplayBuffer -> MFPolicyTrainer's fused epochs (device sampling, graph replay) -> batched
evaluation, on a task where "did it learn" has an answer.  The parity tests pin single steps against the reference; this one checks
that thousands of replayed steps of the timed path add up to a better policy, for every algorithm of the path, in the
bench precision and in exact fp32, with one run and with several runs per engine.

Task (no gym / d4rl in the image): a 2-D point mass, obs = position, action in [-1, 1]^2, x' = clip(x + 0.25 a, -2, 2), reward
-|x'|^2, 20 steps per episode (time limit -> terminals stay 0, as d4rl's timeouts).  Behaviour data: half the transitions from a
uniform-random policy, half from a noisy proportional controller.  Driving to the origin is optimal (return about -1.7 from the
start states used); the random policy gets -27, the behaviour mixture -20.5.  Measured after 3000 steps: TD3+BC -2.1, IQL -2.7,
CQL -1.9 (both precisions within 0.1 of each other), EDAC -3.3; eight TD3+BC runs of one engine -1.1 .. -2.1."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
OD, AD, HID, T = 2, 2, [64, 64], 20


class Space:
    def __init__(self, ad):
        self.low = -np.ones(ad, np.float32)
        self.high = np.ones(ad, np.float32)
        self.shape = (ad,)


class PointMass:
    def __init__(self, seed):
        self.rng = np.random.RandomState(seed)
        self.x = np.zeros(OD, np.float32)
        self.t = 0

    def reset(self):
        self.t = 0
        self.x = self.rng.uniform(-1.5, 1.5, OD).astype(np.float32)
        return self.x.copy()

    def step(self, a):
        a = np.clip(np.asarray(a, np.float32).reshape(-1), -1, 1)
        self.x = np.clip(self.x + 0.25 * a, -2, 2).astype(np.float32)
        self.t += 1
        return self.x.copy(), -float((self.x ** 2).sum()), self.t >= T, {}


def rollout_return(act_fn, episodes, seed):
    env, out = PointMass(seed), []
    for _ in range(episodes):
        o, done, ret = env.reset(), False, 0.0
        while not done:
            o, r, done, _ = env.step(act_fn(o))
            ret += r
        out.append(ret)
    return float(np.mean(out))


def make_dataset(n_episodes=1500, seed=0):
    rng = np.random.RandomState(seed)
    env = PointMass(seed + 1)
    obs, act, nobs, rew, term = [], [], [], [], []
    for e in range(n_episodes):
        o, done = env.reset(), False
        while not done:
            a = rng.uniform(-1, 1, AD) if e % 2 == 0 else np.clip(-1.2 * o + rng.normal(0, 0.6, AD), -1, 1)
            o2, r, done, _ = env.step(a)
            obs.append(o); act.append(a.astype(np.float32)); nobs.append(o2); rew.append(r); term.append(False)
            o = o2
    return dict(observations=np.array(obs, np.float32), actions=np.array(act, np.float32), next_observations=np.array(nobs, np.float32),
                rewards=np.array(rew, np.float32), terminals=np.array(term))


def build(algo):
    from offlinerlkit.modules import Actor, ActorProb, Critic, DiagGaussian, EnsembleCritic, TanhDiagGaussian
    from offlinerlkit.nets import MLP
    from offlinerlkit.policy import CQLPolicy, EDACPolicy, IQLPolicy, TD3BCPolicy
    adam = lambda m, lr: torch.optim.Adam(m.parameters(), lr=lr)
    if algo == "td3bc":          # run_example/run_td3bc.py
        actor = Actor(MLP(OD, HID), AD, max_action=1.0, device=DEV)
        c1, c2 = Critic(MLP(OD + AD, HID), DEV), Critic(MLP(OD + AD, HID), DEV)
        return TD3BCPolicy(actor, c1, c2, adam(actor, 1e-3), adam(c1, 1e-3), adam(c2, 1e-3), tau=0.005, gamma=0.95, max_action=1.0,
                           policy_noise=0.2, noise_clip=0.5, update_actor_freq=2, alpha=2.5)
    if algo == "iql":            # run_example/run_iql.py
        actor = ActorProb(MLP(OD, HID), DiagGaussian(HID[-1], AD, unbounded=False, conditioned_sigma=False), DEV)
        q1, q2, v = Critic(MLP(OD + AD, HID), DEV), Critic(MLP(OD + AD, HID), DEV), Critic(MLP(OD, HID), DEV)
        return IQLPolicy(actor, q1, q2, v, adam(actor, 1e-3), adam(q1, 1e-3), adam(q2, 1e-3), adam(v, 1e-3), action_space=Space(AD),
                         tau=0.005, gamma=0.95, expectile=0.7, temperature=3.0)
    actor = ActorProb(MLP(OD, HID), TanhDiagGaussian(HID[-1], AD, unbounded=True, conditioned_sigma=True), DEV)
    log_alpha = torch.zeros(1, requires_grad=True, device=DEV)
    alpha = (-float(AD), log_alpha, torch.optim.Adam([log_alpha], lr=1e-3))
    if algo == "cql":            # run_example/run_cql.py
        c1, c2 = Critic(MLP(OD + AD, HID), DEV), Critic(MLP(OD + AD, HID), DEV)
        return CQLPolicy(actor, c1, c2, adam(actor, 1e-3), adam(c1, 1e-3), adam(c2, 1e-3), action_space=Space(AD), tau=0.005, gamma=0.95,
                         alpha=alpha, cql_weight=1.0, temperature=1.0, max_q_backup=False, deterministic_backup=True, with_lagrange=False,
                         lagrange_threshold=10.0, cql_alpha_lr=3e-4, num_repeart_actions=4)
    critics = EnsembleCritic(OD, AD, HID, num_ensemble=4, device=DEV)          # run_example/run_edac.py
    return EDACPolicy(actor, critics, adam(actor, 1e-3), adam(critics, 1e-3), tau=0.005, gamma=0.95, alpha=alpha, max_q_backup=False,
                      deterministic_backup=False, eta=1.0)


@pytest.fixture(scope="module")
def task():
    ds = make_dataset()
    rng = np.random.RandomState(5)
    random_ret = rollout_return(lambda o: rng.uniform(-1, 1, AD), 40, 123)
    behaviour_ret = float(ds["rewards"].reshape(-1, T).sum(axis=1).mean())
    return ds, random_ret, behaviour_ret


def train(algo, ds, tmp_path, precision, n_runs, epochs, steps):
    from offlinerlkit.buffer import ReplayBuffer
    from offlinerlkit.policy_trainer import MFPolicyTrainer
    from offlinerlkit.utils.logger import Logger
    torch.manual_seed(3)
    pol = build(algo)
    pol.set_engine_options(precision=precision, n_runs=n_runs, seed=17)
    n = len(ds["rewards"])
    buf = ReplayBuffer(n, (OD,), np.float32, AD, np.float32, device=DEV)
    buf.load_dataset(ds)
    envs = [PointMass(1000 + i) for i in range(10)]
    logger = Logger(str(tmp_path), {"policy_training_progress": "csv"})
    MFPolicyTrainer(pol, envs, buf, logger, epoch=epochs, step_per_epoch=steps, batch_size=256, eval_episodes=10).train()
    rows = [ln.split(",") for ln in open(tmp_path / "record" / "policy_training_progress.csv").read().strip().split("\n")]
    head = rows[0]
    col = lambda k: [float(r[head.index(k)]) for r in rows[1:]]
    # a diverged run must not go unnoticed: the integer-view ReLU of the weight-stationary forward (csrc/gemm.h: orl_relu_mask4) turns a
    # NaN with the sign bit set into +0, so the losses alone could look finite -- every parameter of every run is checked as well
    for r in range(n_runs):
        for k, v in pol.run_state_dict(r).items():
            assert torch.isfinite(v).all(), (algo, r, k)
    return pol, head, col


@pytest.mark.parametrize("algo,precision", [("td3bc", 1), ("td3bc", 0), ("iql", 1), ("cql", 1), ("cql", 0), ("cql", 2), ("edac", 1)])
def test_offline_training_improves_on_the_behaviour_policy(task, tmp_path, algo, precision):
    ds, random_ret, behaviour_ret = task
    assert random_ret < behaviour_ret < -8.0                                   # the task is what the docstring says
    pol, head, col = train(algo, ds, tmp_path, precision, 1, epochs=4, steps=750)
    ret = col("eval/episode_reward")
    print(f"{algo} precision {precision}: eval return per epoch {[round(x, 2) for x in ret]} (random {random_ret:.1f}, behaviour {behaviour_ret:.1f})")
    assert np.isfinite(ret).all()
    # after 3000 gradient steps the deterministic policy beats the data it was trained on by a wide margin
    assert ret[-1] > behaviour_ret + 0.5 * abs(behaviour_ret), (algo, precision, ret, behaviour_ret)
    assert ret[-1] > -8.0, (algo, precision, ret)
    assert pol.engine.step_count() == 3000


def test_every_run_of_a_multi_run_engine_learns(task, tmp_path):
    """8 seeds in one engine (config 5's shape): every run gets its own minibatch / noise streams and its own logged return"""
    ds, random_ret, behaviour_ret = task
    pol, head, col = train("td3bc", ds, tmp_path, 1, 8, epochs=3, steps=750)
    finals = [col(f"run{r}/eval/episode_reward")[-1] for r in range(8)]
    print(f"td3bc x 8 runs: final eval returns {[round(x, 2) for x in finals]} (behaviour {behaviour_ret:.1f})")
    assert all(f > behaviour_ret + 0.5 * abs(behaviour_ret) for f in finals), (finals, behaviour_ret)
    assert len(set(round(f, 4) for f in finals)) > 1                           # the runs are different trainings
    a0 = pol.run_state_dict(0)["actor.last.weight"]
    a7 = pol.run_state_dict(7)["actor.last.weight"]
    assert (a0 - a7).abs().max() > 1e-4


@pytest.mark.parametrize("algo", ["td3bc", "iql", "cql", "edac"])
def test_multi_run_initialisation_and_run_batched_actor_forward(task, algo):
    """runs r > 0 of a multi-run policy start from their OWN initialisation of every trainable net -- EnsembleLinear critics included
    (nets.EnsembleLinear.reset_parameters) -- keyed by mix(seed, r) rather than seed + r (a launcher starting seeds s, s + 1, ... with several
    runs each must not hand run 1 of seed s the networks of run 0 of seed s + 1); ``select_action_runs`` = every run's deterministic actor
    in one batched forward, row for row what ``select_run(r); select_action(..., deterministic=True)`` returns."""
    from offlinerlkit.buffer import ReplayBuffer
    ds, _, _ = task
    R = 4
    buf = ReplayBuffer(len(ds["rewards"]), (OD,), np.float32, AD, np.float32, device=DEV)
    buf.load_dataset(ds)

    def first_critic(pol, r):
        sd = pol.run_state_dict(r)
        key = [k for k in sd if k.startswith(("critic1.", "critics.", "critic_q1.")) and k.endswith("weight") and "saved" not in k][0]
        return sd[key].clone(), [k for k in sd if k.startswith("actor.") and k.endswith("weight")][0], sd

    torch.manual_seed(3)
    pol = build(algo)
    pol.set_engine_options(n_runs=R, seed=40, precision=1)
    pol.learn_n(3, buf, 256)
    crit = [first_critic(pol, r) for r in range(R)]
    for r in range(1, R):
        assert (crit[r][0] - crit[0][0]).abs().max() > 1e-3, (algo, r, "run r > 0 shares run 0's critic weights")
    # the same launcher seed + 1: its run 0 is the module as built, its run 1 must differ from OUR run 2 (seed + r would make them equal)
    torch.manual_seed(3)
    pol2 = build(algo)
    pol2.set_engine_options(n_runs=R, seed=41, precision=1)
    pol2._bind(256)
    akey = crit[0][1]
    a_ours, a_theirs = pol.run_state_dict(2)[akey], pol2.run_state_dict(1)[akey]
    assert a_ours.shape == a_theirs.shape
    # (both sides: one init stream each; ours has also taken 3 Adam steps of <= 3e-3 -- an identical init would still be within 1e-2)
    assert (a_ours - a_theirs).abs().max() > 5e-2, (algo, "run 2 of seed s and run 1 of seed s + 1 share their initialisation")
    # run-batched deterministic forward == per-run select_action
    obs = np.random.RandomState(1).standard_normal((R, 5, OD)).astype(np.float32)
    pol.eval()
    got = pol.select_action_runs(obs)
    assert got.shape == (R, 5, AD)
    for r in range(R):
        pol.select_run(r)
        ref = pol.select_action(obs[r], deterministic=True)
        assert np.abs(got[r] - ref).max() < 1e-5, (algo, r, np.abs(got[r] - ref).max())
    pol.select_run(0)
    with pytest.raises(ValueError):
        pol.select_action_runs(obs[:2])
    pol2._unbind()
    pol._unbind()


def test_engine_options_carry_optimizer_state_and_learn_checks_array_shapes(task):
    """``set_engine_options`` on a bound policy rebuilds the engine AROUND the current state: parameters, Adam moments, step count and
    scalars of the surviving runs carry over (a mid-training precision switch must not restart bias correction); ``learn`` decides the run
    dimension per array and refuses shapes the engine would read out of bounds."""
    from offlinerlkit.buffer import ReplayBuffer
    ds, _, _ = task
    buf = ReplayBuffer(len(ds["rewards"]), (OD,), np.float32, AD, np.float32, device=DEV)
    buf.load_dataset(ds)
    torch.manual_seed(5)
    pol = build("cql")
    pol.set_engine_options(n_runs=2, seed=9, precision=0)
    pol.learn_n(20, buf, 256)
    eng = pol.engine
    before = [eng.optimizer_state(r) for r in range(2)]
    params = [pol.run_state_dict(r) for r in range(2)]
    assert before[0]["step"] == 20 and np.abs(before[1]["adam"][1][0]).max() > 0
    pol.set_engine_options(precision=1)                  # same runs, other precision: everything carries over
    pol._bind(256)
    after = [pol.engine.optimizer_state(r) for r in range(2)]
    for r in range(2):
        assert after[r]["step"] == 20
        for n in before[r]["adam"]:
            assert np.array_equal(before[r]["adam"][n][0], after[r]["adam"][n][0]) and np.array_equal(before[r]["adam"][n][1], after[r]["adam"][n][1])
        for w, v in before[r]["scalars"].items():
            assert abs(after[r]["scalars"][w] - v) <= 1e-6 * max(1.0, abs(v)), (r, w)
        sd = pol.run_state_dict(r)
        assert all(torch.equal(sd[k], params[r][k]) for k in sd)
    pol.set_engine_options(n_runs=3)                     # one more run: the first two carry over, the third starts fresh
    pol._bind(256)
    st3 = [pol.engine.optimizer_state(r) for r in range(3)]
    assert np.array_equal(st3[1]["adam"][1][0], before[1]["adam"][1][0]) and np.abs(st3[2]["adam"][1][0]).max() == 0
    assert abs(st3[2]["scalars"][0]) < 1e-12            # log_alpha of the new run = the launcher's initial value (run_cql.py:102: zeros)
    # per-array run dimension in learn()
    b = buf.sample(256)
    res = pol.learn(b)                                   # shared [B, cols] batch: expanded to every run
    assert np.isfinite(list(res.values())).all()
    per_run = {k: v.unsqueeze(0).expand(3, *v.shape).contiguous() for k, v in b.items()}
    assert np.isfinite(list(pol.learn(per_run).values())).all()
    mixed = dict(per_run, rewards=b["rewards"])          # per-run batch with shared rewards: fine, decided per array
    assert np.isfinite(list(pol.learn(mixed).values())).all()
    with pytest.raises(ValueError):
        pol.learn({k: v[:2] for k, v in per_run.items()})            # 2 != n_runs = 3
    with pytest.raises(ValueError):
        pol.learn(dict(b, actions=b["actions"][:100]))                 # row count mismatch
    pol._unbind()


def test_all_runs_are_evaluated_together_when_every_run_has_its_own_envs(task, tmp_path):
    """multi-run evaluation as ONE run-batched forward per env step (eval_env = a list of envs per run) reports, per run, what evaluating
    the runs one after another on the same envs reports"""
    from offlinerlkit.buffer import ReplayBuffer
    from offlinerlkit.policy_trainer import MFPolicyTrainer
    from offlinerlkit.utils.logger import Logger
    ds, _, _ = task
    R, E = 3, 4
    buf = ReplayBuffer(len(ds["rewards"]), (OD,), np.float32, AD, np.float32, device=DEV)
    buf.load_dataset(ds)
    torch.manual_seed(7)
    pol = build("td3bc")
    pol.set_engine_options(n_runs=R, seed=3, precision=1)
    pol.learn_n(200, buf, 256)
    groups = [[PointMass(500 + 10 * r + i) for i in range(E)] for r in range(R)]
    tr = MFPolicyTrainer(pol, groups, buf, Logger(str(tmp_path), {"policy_training_progress": "csv"}), epoch=1, step_per_epoch=1, eval_episodes=6)
    assert tr._env_groups(R) is not None
    together = tr._evaluate_runs_batched(groups)
    for r in range(R):
        pol.select_run(r)
        tr1 = MFPolicyTrainer(pol, [PointMass(500 + 10 * r + i) for i in range(E)], buf, tr.logger, epoch=1, step_per_epoch=1, eval_episodes=6)
        alone = tr1._evaluate()
        assert len(together[r]["eval/episode_reward"]) == 6
        np.testing.assert_allclose(together[r]["eval/episode_reward"], alone["eval/episode_reward"], rtol=1e-4, atol=1e-4)
        assert together[r]["eval/episode_length"] == alone["eval/episode_length"]
    pol.select_run(0)
    pol._unbind()

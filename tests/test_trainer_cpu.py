"""CPU: MFPolicyTrainer + Logger semantics against a golden trace of the REAL reference trainer
(tests/golden/make_trainer_golden.py), and the N>1 metric all-gather with a world_size-2 gloo group."""
import os
import subprocess
import sys
import tempfile

import numpy as np
import pytest

import trainer_fakes as tf

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden", "trainer_trace.npz")


class HostBuffer:
    """numpy stand-in with ReplayBuffer.sample's exact RNG use (buffer.py:96-106) for the GPU-less container."""

    def __init__(self, ds):
        self.ds = ds
        self.n = len(ds["observations"])

    def sample(self, batch_size):
        idx = np.random.randint(0, self.n, size=batch_size)
        return {k: (self.ds[k][idx] if k in ("observations", "actions", "next_observations")
                    else np.asarray(self.ds[k], np.float32)[idx].reshape(-1, 1)) for k in self.ds}


def run_trainer(tmp, **kw):
    from offlinerlkit.policy_trainer import MFPolicyTrainer
    from offlinerlkit.utils.logger import Logger
    logger = Logger(tmp, {"consoleout_backup": "stdout", "policy_training_progress": "csv"})
    pol, sched = tf.FakePolicy(), tf.FakeScheduler()
    np.random.seed(tf.SEED)
    tr = MFPolicyTrainer(pol, tf.FakeEnv(), HostBuffer(tf.dataset()), logger, epoch=tf.EPOCHS, step_per_epoch=tf.STEPS,
                         batch_size=tf.BATCH, eval_episodes=tf.EVAL_EPS, lr_scheduler=sched, **kw)
    res = tr.train()
    with open(os.path.join(tmp, "record", "policy_training_progress.csv")) as f:
        lines = f.read().strip().split("\n")
    return res, pol, sched, lines, tr


def test_trainer_matches_reference_trace(tmp_path, capsys):
    g = np.load(GOLD, allow_pickle=False)
    res, pol, sched, lines, _ = run_trainer(str(tmp_path), fused=False)
    assert lines[0].split(",") == [str(x) for x in g["csv_header"]]
    rows = np.array([[float(x) if x else np.nan for x in ln.split(",")] for ln in lines[1:]])
    assert rows.shape == g["csv_rows"].shape
    np.testing.assert_allclose(rows, g["csv_rows"], rtol=1e-6, atol=1e-9)          # epoch means, eval stats, timestep
    np.testing.assert_allclose(pol.obs_sums, g["obs_sums"], rtol=1e-6)             # identical minibatch index stream
    assert sched.n == int(g["sched_steps"][0]) == tf.EPOCHS                        # lr_scheduler.step() once per epoch
    assert abs(res["last_10_performance"] - float(g["last_10_performance"][0])) < 1e-9
    assert os.path.exists(tmp_path / "checkpoint" / "policy.pth") and os.path.exists(tmp_path / "model" / "policy.pth")
    assert all(bool(x) for x in g["ckpt_exists"])


def test_logger_csv_grows_header_and_means(tmp_path):
    from offlinerlkit.utils.logger import Logger, make_log_dirs
    lg = Logger(str(tmp_path), {"p": "csv"})
    for v in (1.0, 2.0, 6.0):
        lg.logkv_mean("m", v)
    lg.logkv("x", 5)
    lg.set_timestep(10); lg.dumpkvs()
    lg.logkv("new", 7); lg.set_timestep(20); lg.dumpkvs()
    lines = open(tmp_path / "record" / "p.csv").read().strip().split("\n")
    assert lines[0] == "m,timestep,x,new"
    assert lines[1] == "3.0,10,5," and lines[2] == ",20,,7"
    lg.close()
    cwd = os.getcwd()
    os.chdir(tmp_path)
    try:
        d = make_log_dirs("task", "algo", 3, {"lr": 0.1}, record_params=["lr"])
        assert d.startswith(os.path.join("logs", "task", "algo&lr=0.1", "timestamp_")) and d.endswith("&3")
        assert os.path.isdir(d)
    finally:
        os.chdir(cwd)


def test_logger_hyperparameters_round_trip_and_level(tmp_path):
    """log_hyperparameters -> hyper_param.json -> load_args gives back the Namespace a launch script's get_args() returned
    (logger.py:264-270, 367-371); set_level stores the level and, as in the reference (logger.py:311-323), does not filter log()."""
    from offlinerlkit.utils import logger as L
    lg = L.Logger(str(tmp_path), {"consoleout_backup": "stdout"})
    args = dict(algo_name="cql", task="hopper-medium-v2", seed=3, hidden_dims=[256, 256], actor_lr=1e-4, auto_alpha=True, device="cuda:0")
    lg.log_hyperparameters(args)
    ns = L.load_args(os.path.join(lg.record_dir, "hyper_param.json"))
    assert vars(ns) == args
    assert lg._level == L.INFO
    lg.set_level(L.DEBUG)
    assert lg._level == L.DEBUG == 10
    lg.log("still written", level=L.DEBUG)
    lg.close()
    assert "still written" in open(tmp_path / "record" / "consoleout_backup.txt").read()


WORKER = r'''
import os, sys, json
import numpy as np
sys.path[:0] = [{root!r}, {root!r} + "/offlinerl-kit_amd", {root!r} + "/tests", {root!r} + "/tests/golden"]
import torch.distributed as dist
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:{port}", rank=int(sys.argv[1]), world_size=2)
import trainer_fakes as tf
from test_trainer_cpu import run_trainer
import tempfile
rank = dist.get_rank()
tf.SEED = 11 + rank                      # independent seeds per rank (replicas only)
with tempfile.TemporaryDirectory() as d:
    res, pol, sched, lines, tr = run_trainer(d, fused=False)
    out = dict(rank=rank, header=lines[0], n_gather=len(tr.gathered_metrics),
               loss_a=[g["loss/a"].tolist() for g in tr.gathered_metrics], own=float(np.mean(pol.obs_sums[:tf.STEPS])))
print("RESULT" + json.dumps(out))
dist.destroy_process_group()
'''


def test_metric_allgather_world_size_2_gloo(tmp_path):
    port = 29500 + (os.getpid() % 400)
    script = tmp_path / "w.py"
    script.write_text(WORKER.format(root=ROOT, port=port))
    procs = [subprocess.Popen([sys.executable, str(script), str(r)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(2)]
    outs = []
    for p in procs:
        o, e = p.communicate(timeout=180)
        assert p.returncode == 0, e[-2000:]
        outs.append(__import__("json").loads([l for l in o.split("\n") if l.startswith("RESULT")][0][6:]))
    outs.sort(key=lambda d: d["rank"])
    assert outs[0]["n_gather"] == outs[1]["n_gather"] == tf.EPOCHS
    # both ranks hold the same gathered table; column r is rank r's own epoch mean
    assert outs[0]["loss_a"] == outs[1]["loss_a"]
    assert abs(outs[0]["loss_a"][0][0] - outs[0]["own"]) < 1e-4 * max(1, abs(outs[0]["own"]))
    assert abs(outs[0]["loss_a"][0][1] - outs[1]["own"]) < 1e-4 * max(1, abs(outs[1]["own"]))
    assert outs[0]["loss_a"][0][0] != outs[0]["loss_a"][0][1]
    # only rank 0 logs the per-rank columns
    assert "rank1/loss/a" in outs[0]["header"] and "rank1/loss/a" not in outs[1]["header"]


def test_modules_state_dict_keys_match_reference_inventory():
    """key names / shapes the engine binds to."""
    from offlinerlkit.modules import Actor, ActorProb, Critic, DiagGaussian, EnsembleCritic, TanhDiagGaussian
    from offlinerlkit.nets import MLP
    a = ActorProb(MLP(17, [256, 256]), TanhDiagGaussian(256, 6, unbounded=True, conditioned_sigma=True))
    assert {k: tuple(v.shape) for k, v in a.state_dict().items()} == {
        "backbone.model.0.weight": (256, 17), "backbone.model.0.bias": (256,), "backbone.model.2.weight": (256, 256),
        "backbone.model.2.bias": (256,), "dist_net.mu.weight": (6, 256), "dist_net.mu.bias": (6,),
        "dist_net.sigma.weight": (6, 256), "dist_net.sigma.bias": (6,)}
    assert sum(p.numel() for p in a.parameters()) == 73484
    c = Critic(MLP(23, [256, 256]))
    assert sum(p.numel() for p in c.parameters()) == 72193 and "last.weight" in c.state_dict()
    i = ActorProb(MLP(11, [256, 256]), DiagGaussian(256, 3, unbounded=False, conditioned_sigma=False))
    assert tuple(i.state_dict()["dist_net.sigma_param"].shape) == (3, 1) and sum(p.numel() for p in i.parameters()) == 69638
    t = Actor(MLP(17, [256, 256]), 6)
    assert sum(p.numel() for p in t.parameters()) == 71942
    e = EnsembleCritic(17, 6, [256, 256, 256], num_ensemble=10)
    assert sum(p.numel() for p in e.parameters()) == 2759700
    assert tuple(e.state_dict()["model.0.weight"].shape) == (10, 23, 256) and tuple(e.state_dict()["model.6.bias"].shape) == (10, 1, 1)
    assert "model.2.saved_weight" in e.state_dict()


class BatchPolicy(tf.FakePolicy):
    """select_action on [E, OBS] batches; the action depends on the observation so the env-by-env bookkeeping is visible"""

    def __init__(self):
        super().__init__()
        self.calls = []

    def select_action(self, obs, deterministic=False):
        assert deterministic and obs.ndim == 2 and obs.shape[1] == tf.OBS and self.mode == "eval"
        self.calls.append(obs.shape[0])
        return np.repeat(obs[:, :1], tf.ACT, axis=1).astype(np.float32)


def _sequential_reference(envs_eps, n_eps):
    """what the reference's one-env loop (mf_policy_trainer.py:92-118) reports for the same policy on the same episodes"""
    from offlinerlkit.policy_trainer import MFPolicyTrainer
    pol = BatchPolicy()
    tr = MFPolicyTrainer(pol, envs_eps, None, None, eval_episodes=n_eps, fused=False)
    return tr._evaluate(), pol


def test_batched_evaluation_keeps_the_reference_episode_accounting():
    """§8(f)4: E envs in lockstep with one batched forward per step report the episodes the sequential reference loop reports
    (FakeEnv episode k lasts 3 + k % 2 steps and starts from 0.1 k, so every episode is distinguishable)."""
    from offlinerlkit.policy_trainer import MFPolicyTrainer

    def env_at(ep):
        e = tf.FakeEnv()
        e.ep = ep
        return e
    n_eps = 6
    # sequential: ONE env plays episodes 0..5 in a row
    seq, pol_s = _sequential_reference(env_at(0), n_eps)
    assert all(c == 1 for c in pol_s.calls)
    # batched: six envs, env i starts at episode i -> the same six episodes, in lockstep
    envs = [env_at(i) for i in range(n_eps)]
    pol_b = BatchPolicy()
    tr = MFPolicyTrainer(pol_b, envs, None, None, eval_episodes=n_eps, fused=False)
    bat = tr._evaluate()
    assert sorted(bat["eval/episode_length"]) == sorted(seq["eval/episode_length"])
    np.testing.assert_allclose(sorted(bat["eval/episode_reward"]), sorted(seq["eval/episode_reward"]), rtol=1e-6)
    assert pol_b.calls[0] == n_eps and len(pol_b.calls) == max(seq["eval/episode_length"])     # 4 forwards instead of 21
    assert sum(pol_b.calls) == sum(seq["eval/episode_length"])
    # fewer envs than episodes: finished envs are reset and reused until eval_episodes episodes have STARTED
    envs = [env_at(0), env_at(0)]
    tr = MFPolicyTrainer(BatchPolicy(), envs, None, None, eval_episodes=5, fused=False)
    out = tr._evaluate()
    assert len(out["eval/episode_reward"]) == 5 and envs[0].ep + envs[1].ep == 5
    # more envs than episodes: only eval_episodes of them run
    envs = [env_at(i) for i in range(4)]
    tr = MFPolicyTrainer(BatchPolicy(), envs, None, None, eval_episodes=2, fused=False)
    out = tr._evaluate()
    assert len(out["eval/episode_reward"]) == 2 and envs[2].t == 0 and envs[3].t == 0


class MultiRunPolicy(tf.FakePolicy):
    """duck-typed n_runs policy (EnginePolicy's multi-run surface) for the GPU-less container"""
    n_runs = 3

    def __init__(self):
        super().__init__()
        self.cur = 0

    def learn_n(self, n, buffer, batch_size):
        out = {"loss/a": 2.0}
        for r in range(self.n_runs):
            out[f"run{r}/loss/a"] = 1.0 + r
        return out

    def select_run(self, r):
        self.cur = r

    def run_state_dict(self, r):
        return {"w": self.w.detach().clone() + r}

    def select_action(self, obs, deterministic=False):
        return np.full((1, tf.ACT), 0.25 * (self.cur + 1), dtype=np.float32)


def test_trainer_logs_and_checkpoints_every_run_of_a_multi_run_policy(tmp_path):
    import torch
    from offlinerlkit.policy_trainer import MFPolicyTrainer
    from offlinerlkit.utils.logger import Logger
    logger = Logger(str(tmp_path), {"policy_training_progress": "csv"})
    pol = MultiRunPolicy()

    class Buf:
        def device_buffer(self):
            return self
    tr = MFPolicyTrainer(pol, tf.FakeEnv(), Buf(), logger, epoch=2, step_per_epoch=5, batch_size=8, eval_episodes=2)
    tr.train()
    lines = open(tmp_path / "record" / "policy_training_progress.csv").read().strip().split("\n")
    head = lines[0].split(",")
    row = dict(zip(head, lines[-1].split(",")))
    for r in range(3):
        assert float(row[f"run{r}/loss/a"]) == 1.0 + r
        assert f"run{r}/eval/episode_reward" in head and f"run{r}/eval/normalized_episode_reward" in head
        sd = torch.load(tmp_path / "model" / f"policy_run{r}.pth", weights_only=True)
        assert float(sd["w"][0]) == float(r)
    # run r's actions are 0.25 (r + 1) per dim: episode reward per step = 1 + 0.5 * ACT * 0.25 (r + 1) -> runs are evaluated separately
    r0, r2 = float(row["run0/eval/episode_reward"]), float(row["run2/eval/episode_reward"])
    assert r2 > r0
    assert float(row["loss/a"]) == 2.0 and pol.cur == 0


def test_engine_options_accept_the_three_precisions_and_refuse_others():
    """``set_engine_options(precision=...)``: 0 exact fp32, 1 two fp16 planes, 2 three fp16 planes in the critic launches (no engine is built until
    the first ``learn``: this runs without a GPU)"""
    import pytest
    from offlinerlkit.modules import ActorProb, Critic, TanhDiagGaussian
    from offlinerlkit.nets import MLP
    from offlinerlkit.policy import CQLPolicy
    import torch
    actor = ActorProb(MLP(17, [256, 256]), TanhDiagGaussian(256, 6, unbounded=True, conditioned_sigma=True))
    c1, c2 = Critic(MLP(23, [256, 256])), Critic(MLP(23, [256, 256]))
    adam = lambda m: torch.optim.Adam(m.parameters(), lr=3e-4)

    class Space:
        low, high, shape = -np.ones(6, np.float32), np.ones(6, np.float32), (6,)
    pol = CQLPolicy(actor, c1, c2, adam(actor), adam(c1), adam(c2), action_space=Space(), tau=0.005, gamma=0.99, alpha=0.2)
    for prec in (0, 1, 2):
        assert pol.set_engine_options(precision=prec, n_runs=2) is pol and pol._precision == prec and pol.n_runs == 2
    with pytest.raises(ValueError, match="precision must be 0"):
        pol.set_engine_options(precision=3)
    with pytest.raises(ValueError, match="n_runs"):
        pol.set_engine_options(n_runs=0)

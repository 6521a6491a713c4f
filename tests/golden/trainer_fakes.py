"""Deterministic duck-typed policy / env / scheduler used to pin MFPolicyTrainer semantics (both by the golden
generator that drives the real reference trainer and by the tests that drive ours)."""
import numpy as np
import torch

OBS, ACT, N_DATA, BATCH, EPOCHS, STEPS, EVAL_EPS, SEED = 4, 2, 500, 16, 3, 7, 3, 11


def dataset():
    rng = np.random.RandomState(1)
    return dict(observations=rng.standard_normal((N_DATA, OBS)).astype(np.float32),
                actions=rng.standard_normal((N_DATA, ACT)).astype(np.float32),
                next_observations=rng.standard_normal((N_DATA, OBS)).astype(np.float32),
                rewards=rng.standard_normal(N_DATA).astype(np.float32),
                terminals=(rng.uniform(size=N_DATA) < 0.1))


class FakePolicy(torch.nn.Module):
    """learn() returns deterministic functions of the batch so the logged epoch means pin the index stream."""

    def __init__(self):
        super().__init__()
        self.w = torch.nn.Parameter(torch.zeros(3))
        self.obs_sums = []
        self.mode = None

    def train(self):
        self.mode = "train"

    def eval(self):
        self.mode = "eval"

    def select_action(self, obs, deterministic=False):
        assert deterministic and obs.shape == (1, OBS) and self.mode == "eval"
        return np.full((1, ACT), 0.25, dtype=np.float32)

    def learn(self, batch):
        assert self.mode == "train"
        o = float(torch.as_tensor(batch["observations"]).double().sum())
        r = float(torch.as_tensor(batch["rewards"]).double().sum())
        self.obs_sums.append(o)
        return {"loss/a": o, "loss/b": r * 0.5, "alpha": 0.125}


class FakeEnv:
    def __init__(self):
        self.t = 0
        self.ep = 0

    def reset(self):
        self.t = 0
        return np.full(OBS, 0.1 * self.ep, dtype=np.float32)

    def step(self, action):
        self.t += 1
        done = self.t >= 3 + (self.ep % 2)
        if done:
            self.ep += 1
        return np.full(OBS, 0.01 * self.t, dtype=np.float32), 1.0 + 0.5 * float(action.sum()), done, {}

    def get_normalized_score(self, x):
        return x / 10.0


class FakeScheduler:
    def __init__(self):
        self.n = 0

    def step(self):
        self.n += 1

#!/usr/bin/env python3
"""Golden trace of the REAL reference MFPolicyTrainer + Logger + ReplayBuffer (build container only).

A duck-typed fake policy/env make the run deterministic given the numpy seed; the fixture stores the rows the
reference Logger wrote to ``policy_training_progress.csv`` and the minibatch index stream ReplayBuffer.sample drew.
tensorboard and gym are absent here, so ``torch.utils.tensorboard`` / ``gym`` / ``gymnasium`` are stubbed
(SURVEY Appendix C); only the trainer, logger and buffer code of the reference runs.
"""
import importlib
import os
import sys
import tempfile
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.dont_write_bytecode = True
import trainer_fakes as tf  # noqa: E402


def main():
    sys.path.insert(0, "/root/reference")

    def stub(name, **attrs):
        m = types.ModuleType(name); m.__dict__.update(attrs); sys.modules[name] = m; return m

    class _X:
        pass

    stub("gym", spaces=stub("gym.spaces", Space=_X), Env=_X)
    stub("gymnasium", Env=_X)

    class SummaryWriter:
        def __init__(self, *a, **k): pass
        def add_scalar(self, *a, **k): pass
        def add_hparams(self, *a, **k): pass
        def flush(self): pass
        def close(self): pass
    import torch.utils
    stub("torch.utils.tensorboard", SummaryWriter=SummaryWriter)
    pkg = types.ModuleType("offlinerlkit.policy")
    pkg.__path__ = ["/root/reference/offlinerlkit/policy"]
    sys.modules["offlinerlkit.policy"] = pkg
    pkg.BasePolicy = importlib.import_module("offlinerlkit.policy.base_policy").BasePolicy
    logger_mod = importlib.import_module("offlinerlkit.utils.logger")
    tr = types.ModuleType("offlinerlkit.policy_trainer"); tr.__path__ = ["/root/reference/offlinerlkit/policy_trainer"]
    sys.modules["offlinerlkit.policy_trainer"] = tr
    MFPolicyTrainer = importlib.import_module("offlinerlkit.policy_trainer.mf_policy_trainer").MFPolicyTrainer
    from offlinerlkit.buffer import ReplayBuffer

    out = {}
    with tempfile.TemporaryDirectory() as d:
        logger = logger_mod.Logger(d, {"consoleout_backup": "stdout", "policy_training_progress": "csv"})
        buf = ReplayBuffer(tf.N_DATA, (tf.OBS,), np.float32, tf.ACT, np.float32, device="cpu")
        buf.load_dataset(tf.dataset())
        pol = tf.FakePolicy()
        sched = tf.FakeScheduler()
        np.random.seed(tf.SEED)
        res = MFPolicyTrainer(pol, tf.FakeEnv(), buf, logger, epoch=tf.EPOCHS, step_per_epoch=tf.STEPS, batch_size=tf.BATCH,
                              eval_episodes=tf.EVAL_EPS, lr_scheduler=sched).train()
        with open(os.path.join(d, "record", "policy_training_progress.csv")) as f:
            csv_text = f.read()
        out["last_10_performance"] = np.array([res["last_10_performance"]])
        out["sched_steps"] = np.array([sched.n])
        out["obs_sums"] = np.array(pol.obs_sums)
        out["ckpt_exists"] = np.array([os.path.exists(os.path.join(d, "checkpoint", "policy.pth")), os.path.exists(os.path.join(d, "model", "policy.pth"))])
    lines = csv_text.strip().split("\n")
    out["csv_header"] = np.array(lines[0].split(","))
    out["csv_rows"] = np.array([[float(x) if x else np.nan for x in ln.split(",")] for ln in lines[1:]])
    np.savez_compressed(os.path.join(HERE, "trainer_trace.npz"), **out)
    print("header:", lines[0]); print("rows:", out["csv_rows"].shape, "last10:", out["last_10_performance"])


if __name__ == "__main__":
    main()

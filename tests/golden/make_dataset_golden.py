#!/usr/bin/env python3
"""Golden vectors for the dataset path: the REAL reference's ``qlearning_dataset``
(offlinerlkit/utils/load_dataset.py:17-147) and ``normalize_rewards`` (run_example/run_iql.py:49-82) run on synthetic
trajectory dicts.  Build container only (the reference does not exist on the GPU box); only arrays of numbers are stored.

``gym`` / ``d4rl`` / ``gymnasium`` / tensorboard are absent here and are stubbed as empty modules (they are only imported, never
called, on this path: the dataset is passed in as a dict); run_iql.py is loaded as a module with an empty
command line so that ``normalize_rewards`` itself is the reference's code.
"""
import importlib
import importlib.util
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.dont_write_bytecode = True
REF = "/root/reference"


def synth_trajectories(seed, n_eps, max_len, od=5, ad=2, with_next=False, p_term=0.4):
    """episodes of random length; some end in a terminal, the others run into the time limit (timeout on their last step)"""
    rng = np.random.RandomState(seed)
    obs, act, rew, term, tout, nxt = [], [], [], [], [], []
    rid = 0
    for e in range(n_eps):
        terminal_ep = rng.uniform() < p_term
        L = rng.randint(2, max_len) if terminal_ep else max_len
        for t in range(L):
            o = rng.standard_normal(od).astype(np.float32)
            o[0] = rid                                            # row id: reveals which transitions a loader kept
            rid += 1
            obs.append(o); act.append(rng.uniform(-1, 1, ad).astype(np.float32)); rew.append(np.float32(rng.standard_normal()))
            term.append(terminal_ep and t == L - 1); tout.append((not terminal_ep) and t == L - 1)
            nxt.append(rng.standard_normal(od).astype(np.float32))
    d = dict(observations=np.array(obs), actions=np.array(act), rewards=np.array(rew, np.float32), terminals=np.array(term), timeouts=np.array(tout))
    if with_next:
        nx = np.array(nxt)
        nx[:-1][~(d["terminals"][:-1] | d["timeouts"][:-1])] = d["observations"][1:][~(d["terminals"][:-1] | d["timeouts"][:-1])]
        d["next_observations"] = nx
    return d


CASES = {
    # name: (synth kwargs, use the `timeouts` field, qlearning_dataset kwargs)
    "timeouts": (dict(seed=1, n_eps=14, max_len=12), True, dict()),
    "timeouts_terminate_on_end": (dict(seed=2, n_eps=10, max_len=9), True, dict(terminate_on_end=True)),
    "max_episode_steps": (dict(seed=3, n_eps=12, max_len=10), False, dict()),
    "next_obs": (dict(seed=4, n_eps=9, max_len=8, with_next=True), True, dict()),
    "next_obs_terminate_on_end": (dict(seed=5, n_eps=9, max_len=8, with_next=True), True, dict(terminate_on_end=True)),
    "next_obs_max_episode_steps": (dict(seed=6, n_eps=11, max_len=7, with_next=True), False, dict(terminate_on_end=True)),
}

# get_rtg=True (load_dataset.py:17, 87-130): the reference never clears acc_ret_traj_ and never flushes rows behind the last trajectory
# end, so its own assertion (:130) holds for ONE shape of input only -- a single trajectory whose end is the last row the loop visits
# (row N - 2).  Those are the cases it can run; `rtg_fails` records that it raises on everything else.
RTG_CASES = {
    # name: (rows of the single trajectory, kind of its end, with next_observations, qlearning_dataset kwargs)
    "rtg_timeout": (9, "timeout", False, dict()),                                   # end skipped (:84-95): 8 kept rows
    "rtg_terminal_next_obs": (7, "terminal", True, dict()),                         # end kept (:119-125): 7 kept rows
    "rtg_timeout_terminate_on_end": (6, "timeout", True, dict(terminate_on_end=True)),
}


def synth_single_trajectory(rows, end, with_next, seed=21, od=5, ad=2):
    """one trajectory of `rows` rows + the extra final row the loader never visits"""
    rng = np.random.RandomState(seed + rows)
    n = rows + 1
    d = dict(observations=rng.standard_normal((n, od)).astype(np.float32), actions=rng.uniform(-1, 1, (n, ad)).astype(np.float32),
             rewards=rng.standard_normal(n).astype(np.float32), terminals=np.zeros(n, bool), timeouts=np.zeros(n, bool))
    d["observations"][:, 0] = np.arange(n)
    d["terminals" if end == "terminal" else "timeouts"][rows - 1] = True
    if with_next:
        d["next_observations"] = rng.standard_normal((n, od)).astype(np.float32)
    return d


class FakeEnv:
    def __init__(self, max_steps):
        self._max_episode_steps = max_steps


def main():
    sys.path.insert(0, REF)

    def stub(name, **attrs):
        m = types.ModuleType(name); m.__dict__.update(attrs); sys.modules[name] = m; return m

    class _X:
        pass
    stub("gym", spaces=stub("gym.spaces", Space=_X), Env=_X)
    stub("gymnasium", Env=_X)
    stub("d4rl")

    class SummaryWriter:
        def __init__(self, *a, **k): pass
    import torch.utils
    stub("torch.utils.tensorboard", SummaryWriter=SummaryWriter)
    qlearning_dataset = importlib.import_module("offlinerlkit.utils.load_dataset").qlearning_dataset
    # run_iql.py: the policy / trainer packages it imports are pre-seeded with shims (their own eager imports need gym, diffusers, ...)
    pkg = types.ModuleType("offlinerlkit.policy"); pkg.__path__ = [REF + "/offlinerlkit/policy"]; sys.modules["offlinerlkit.policy"] = pkg
    pkg.BasePolicy = importlib.import_module("offlinerlkit.policy.base_policy").BasePolicy
    pkg.IQLPolicy = importlib.import_module("offlinerlkit.policy.model_free.iql").IQLPolicy
    tr = types.ModuleType("offlinerlkit.policy_trainer"); tr.__path__ = [REF + "/offlinerlkit/policy_trainer"]; sys.modules["offlinerlkit.policy_trainer"] = tr
    tr.MFPolicyTrainer = importlib.import_module("offlinerlkit.policy_trainer.mf_policy_trainer").MFPolicyTrainer
    argv, sys.argv = sys.argv, ["run_iql.py"]
    spec = importlib.util.spec_from_file_location("ref_run_iql", REF + "/run_example/run_iql.py")
    run_iql = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(run_iql)
    sys.argv = argv

    out = {}
    for name, (kw, use_timeouts, qkw) in CASES.items():
        d = synth_trajectories(**kw)
        if not use_timeouts:
            d.pop("timeouts")
        res = qlearning_dataset(FakeEnv(kw["max_len"]), dataset={k: v.copy() for k, v in d.items()}, **qkw)
        for k, v in res.items():
            out[f"{name}/out/{k}"] = np.asarray(v)
        print(name, "rows in", len(d["rewards"]), "rows out", len(res["rewards"]), "keys", sorted(res))
    for name, (rows, end, with_next, qkw) in RTG_CASES.items():
        d = synth_single_trajectory(rows, end, with_next)
        res = qlearning_dataset(FakeEnv(1000), dataset={k: v.copy() for k, v in d.items()}, get_rtg=True, **qkw)
        for k, v in res.items():
            out[f"{name}/out/{k}"] = np.asarray(v)
        print(name, "rows in", len(d["rewards"]), "rows out", len(res["rewards"]), "keys", sorted(res), "rtgs dtype", np.asarray(res["rtgs"]).dtype)
    try:                                                      # two trajectories: the reference's own assertion fires
        kw, _, _ = CASES["timeouts"]
        qlearning_dataset(FakeEnv(kw["max_len"]), dataset=synth_trajectories(**kw), get_rtg=True)
        out["rtg_fails/raised"] = np.array(0)
    except AssertionError as e:
        out["rtg_fails/raised"] = np.array(1)
        out["rtg_fails/message"] = np.array(str(e))
        print("get_rtg on a multi-trajectory dataset:", e)
    # normalize_rewards on a q-learning dataset (run_iql.py:71-82 applies it to the qlearning_dataset output)
    d = synth_trajectories(seed=7, n_eps=13, max_len=11, with_next=True)
    q = qlearning_dataset(FakeEnv(11), dataset={k: v.copy() for k, v in d.items()})
    out["normalize/in/rewards"] = q["rewards"].copy()
    res = run_iql.normalize_rewards({k: v.copy() for k, v in q.items()})
    out["normalize/out/rewards"] = np.asarray(res["rewards"])
    print("normalize_rewards: scale", float(res["rewards"][0] / q["rewards"][0]))
    np.savez_compressed(os.path.join(HERE, "dataset_golden.npz"), **out)


if __name__ == "__main__":
    main()

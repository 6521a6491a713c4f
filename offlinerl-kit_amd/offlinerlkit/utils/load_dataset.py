"""Dataset path of the model-free launchers (reference: offlinerlkit/utils/load_dataset.py:17-147, run_example/run_iql.py:49-82):
D4RL-style trajectory dict -> the (s, a, s', r, done) transition arrays ``ReplayBuffer.load_dataset`` ingests into the
HBM-resident SoA store.

Unlike the reference module this one does not import ``gym`` / ``d4rl``: ``env`` is only asked for ``get_dataset()`` when no
dataset dict is passed and for ``_max_episode_steps`` when the dict has no ``timeouts`` field.  The per-transition Python loop
of the reference is restated as array operations where the rule is position-independent (datasets with a ``timeouts`` field,
i.e. every D4RL v2 file), and as one pass over a state machine where it is not (the step counter of the no-``timeouts`` case).
Pinned against the real function on synthetic trajectories: tests/golden/make_dataset_golden.py -> tests/test_load_dataset.py.
"""
from __future__ import annotations

from typing import Dict, Optional

import numpy as np


def _keep_mask(dataset: Dict[str, np.ndarray], terminate_on_end: bool, max_episode_steps: Optional[int], with_last: bool = False):
    """Which of the first N-1 rows survive (load_dataset.py:64-112).

    A row is the LAST of its trajectory when ``timeouts`` says so (or, without that field, when the running count of kept rows of
    the trajectory reaches ``max_episode_steps - 1``).  Such a row is dropped unless ``terminate_on_end``; a terminal row -- and a
    kept last row -- is dropped as well when the dataset has no ``next_observations`` (its successor row belongs to another
    episode)."""
    n = dataset["rewards"].shape[0] - 1                       # the reference never emits the final row (range(N-1), :64)
    done = np.asarray(dataset["terminals"][:n]).astype(bool)
    has_next = "next_observations" in dataset
    if "timeouts" in dataset:
        last = np.asarray(dataset["timeouts"][:n]).astype(bool)
    else:
        if max_episode_steps is None:
            raise ValueError("dataset has no 'timeouts' field: env._max_episode_steps is needed")
        last = np.zeros(n, bool)
        step = 0                                              # kept rows of the running trajectory (episode_step)
        for i in range(n):
            fin = step == max_episode_steps - 1
            last[i] = fin
            if fin and not terminate_on_end:
                step = 0                                      # row skipped, counter restarts (:84-95)
            elif done[i] or fin:
                # counter restarts; the row itself is kept only with next_observations (:98-107) -- and then STILL counts one step (:117)
                step = 1 if has_next else 0
            else:
                step += 1
    keep = ~((done | last) & (not has_next)) if terminate_on_end else ~last & ~(done & (not has_next))
    return (keep, last) if with_last else keep


def _rtgs(dataset: Dict[str, np.ndarray], keep: np.ndarray, last: np.ndarray, terminate_on_end: bool) -> np.ndarray:
    """``rtgs`` of ``get_rtg=True`` exactly as the reference computes them (load_dataset.py:54-55, 87-93, 101-105, 114-125), including what
    it does NOT do: ``acc_ret_traj_`` is never cleared, so at every trajectory end the return-to-go of EVERY row kept so far is appended
    again, and rows after the last trajectory end are never flushed -- the reference's own assertion (:130) then fails.  It holds for one
    case only: a single trajectory whose end is the last row the loop visits.  That case is returned (float32, accumulated in row order
    like the reference's ``ret += reward``); every other dataset raises the reference's AssertionError."""
    n = keep.shape[0]
    done = np.asarray(dataset["terminals"][:n]).astype(bool)
    has_next = "next_observations" in dataset
    # rows at which the reference flushes acc_ret_traj_ into rtg_: a skipped trajectory end (:84-95, :98-107) or a kept one (:119-125)
    flush = (last & (not terminate_on_end)) | ((done | last) & ((not has_next) | keep))
    kept_before = np.cumsum(keep) - keep                       # rows kept strictly before row i
    n_rtg = int((kept_before + keep)[flush].sum())             # a flush appends one entry per row kept so far (this row included if kept)
    n_obs = int(keep.sum())
    assert n_obs == n_rtg, f"Obs {n_obs} and Rtg {n_rtg} should be same length!"
    r = np.asarray(dataset["rewards"])[:n][keep].astype(np.float32)
    acc = np.cumsum(r, dtype=np.float32)                       # ret after each kept row, float32 like `ret += reward`
    before = np.concatenate([np.zeros(1, np.float32), acc[:-1]]) if len(acc) else acc
    return (acc[-1] - before).astype(np.float32) if len(acc) else np.zeros(0, np.float32)


def qlearning_dataset(env=None, dataset: Optional[Dict[str, np.ndarray]] = None, terminate_on_end: bool = False,
                      get_rtg: bool = False, **kwargs) -> Dict[str, np.ndarray]:
    """observations / actions / next_observations / rewards / terminals of every usable transition (load_dataset.py:17-147).
    ``next_observations`` are the dataset's own when present, otherwise the following row's observation."""
    if dataset is None:
        dataset = env.get_dataset(**kwargs)
    n = dataset["rewards"].shape[0] - 1
    keep, last = _keep_mask(dataset, terminate_on_end, getattr(env, "_max_episode_steps", None), with_last=True)
    idx = np.nonzero(keep)[0]
    obs = np.asarray(dataset["observations"])
    nxt = np.asarray(dataset["next_observations"])[idx] if "next_observations" in dataset else obs[idx + 1]
    out = {
        "observations": obs[idx].astype(np.float32),
        "actions": np.asarray(dataset["actions"])[idx].astype(np.float32),
        "next_observations": nxt.astype(np.float32),
        "rewards": np.asarray(dataset["rewards"])[idx].astype(np.float32),
        "terminals": np.asarray(dataset["terminals"][:n]).astype(bool)[idx],
    }
    if get_rtg:                                                # key 'rtgs' (load_dataset.py:139-147); see _rtgs for what the reference accepts
        out["rtgs"] = _rtgs(dataset, keep, last, terminate_on_end)
    return out


def normalize_rewards(dataset: Dict[str, np.ndarray]) -> Dict[str, np.ndarray]:
    """IQL's reward scaling (run_iql.py:49-82): split the transitions into trajectories (a break wherever the next row's
    observation is not this row's next_observation, or at a terminal), then rewards *= 1000 / (best return - worst return).
    Mutates and returns ``dataset`` like the reference."""
    obs, nxt = np.asarray(dataset["observations"]), np.asarray(dataset["next_observations"])
    n = len(obs)
    ends = np.ones(n, bool)                                                     # the last row always ends a trajectory
    if n > 1:
        jump = np.linalg.norm(obs[1:].astype(np.float64) - nxt[:-1].astype(np.float64), axis=-1) > 1e-6
        ends[:-1] = jump | (np.asarray(dataset["terminals"][:-1]).astype(np.float64) == 1.0)
    starts = np.concatenate([[0], np.nonzero(ends[:-1])[0] + 1])
    returns = np.add.reduceat(np.asarray(dataset["rewards"], np.float64).reshape(n), starts)
    scale = returns.max() - returns.min()
    dataset["rewards"] /= np.asarray(scale, dtype=dataset["rewards"].dtype)
    dataset["rewards"] *= 1000.0
    return dataset


def load_dataset_file(path: str) -> Dict[str, np.ndarray]:
    """A D4RL-style dataset from disk for ``qlearning_dataset(dataset=...)``: ``.npz`` archives of the arrays, or the ``.hdf5``
    files D4RL distributes when ``h5py`` is importable (it is not in this image; the arrays are what matters to the path)."""
    if path.endswith(".npz"):
        with np.load(path, allow_pickle=False) as z:
            return {k: z[k] for k in z.files}
    if path.endswith((".hdf5", ".h5")):
        try:
            import h5py
        except ImportError as e:
            raise RuntimeError("reading D4RL .hdf5 files needs h5py, which is not installed; convert the file to .npz "
                               "(observations, actions, rewards, terminals, timeouts[, next_observations])") from e
        with h5py.File(path, "r") as f:
            return {k: f[k][()] for k in ("observations", "actions", "rewards", "terminals", "timeouts", "next_observations") if k in f}
    raise ValueError(f"unsupported dataset file {path!r} (.npz or .hdf5)")

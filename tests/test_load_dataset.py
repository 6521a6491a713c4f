"""CPU: the dataset path against golden vectors computed by the REAL reference
(tests/golden/make_dataset_golden.py): ``qlearning_dataset`` must select exactly the same transitions, bit for bit, and
``normalize_rewards`` must scale rewards identically (1e-6 relative)."""
import os

import numpy as np
import pytest

import make_dataset_golden as mg
from offlinerlkit.utils.load_dataset import load_dataset_file, normalize_rewards, qlearning_dataset

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "dataset_golden.npz"), allow_pickle=False)


@pytest.mark.parametrize("case", list(mg.CASES))
def test_qlearning_dataset_selects_the_reference_transitions(case):
    kw, use_timeouts, qkw = mg.CASES[case]
    d = mg.synth_trajectories(**kw)
    if not use_timeouts:
        d.pop("timeouts")
    out = qlearning_dataset(mg.FakeEnv(kw["max_len"]), dataset=d, **qkw)
    assert set(out) == {"observations", "actions", "next_observations", "rewards", "terminals"}
    for k, v in out.items():
        ref = GOLD[f"{case}/out/{k}"]
        assert v.shape == ref.shape and v.dtype == ref.dtype, (case, k, v.shape, ref.shape, v.dtype, ref.dtype)
        assert np.array_equal(v, ref), (case, k)
    assert np.array_equal(out["observations"][:, 0], GOLD[f"{case}/out/observations"][:, 0])      # the row ids: same index selection


@pytest.mark.parametrize("case", list(mg.RTG_CASES))
def test_get_rtg_on_the_inputs_the_reference_accepts(case):
    """get_rtg=True (load_dataset.py:17, 87-130): bit-identical `rtgs` (and transitions) on the single-trajectory inputs the reference's own
    assertion lets through"""
    rows, end, with_next, qkw = mg.RTG_CASES[case]
    d = mg.synth_single_trajectory(rows, end, with_next)
    out = qlearning_dataset(mg.FakeEnv(1000), dataset=d, get_rtg=True, **qkw)
    assert set(out) == {"observations", "actions", "next_observations", "rewards", "terminals", "rtgs"}
    for k, v in out.items():
        ref = GOLD[f"{case}/out/{k}"]
        assert v.shape == ref.shape and v.dtype == ref.dtype and np.array_equal(v, ref), (case, k, v, ref)


def test_get_rtg_raises_the_reference_assertion_elsewhere():
    """more than one trajectory: the reference trips its own assertion (acc_ret_traj_ is never cleared) -- same exception, same message"""
    assert int(GOLD["rtg_fails/raised"]) == 1
    kw, _, _ = mg.CASES["timeouts"]
    with pytest.raises(AssertionError) as e:
        qlearning_dataset(mg.FakeEnv(kw["max_len"]), dataset=mg.synth_trajectories(**kw), get_rtg=True)
    assert str(e.value) == str(GOLD["rtg_fails/message"])


def test_edge_cases_of_the_loader():
    d = mg.synth_trajectories(seed=9, n_eps=3, max_len=4)
    one = {k: v[:2] for k, v in d.items()}                    # two rows: only row 0 can be emitted
    out = qlearning_dataset(None, dataset=one)
    assert out["observations"].shape[0] in (0, 1)
    nd = {k: v for k, v in d.items() if k != "timeouts"}
    with pytest.raises(ValueError):
        qlearning_dataset(None, dataset=nd)                   # no timeouts and no env._max_episode_steps

    class Env:
        def get_dataset(self, **kw):
            return d
    a, b = qlearning_dataset(Env()), qlearning_dataset(None, dataset=d)
    assert all(np.array_equal(a[k], b[k]) for k in a)


def test_normalize_rewards_matches_reference_scaling():
    d = mg.synth_trajectories(seed=7, n_eps=13, max_len=11, with_next=True)
    q = qlearning_dataset(mg.FakeEnv(11), dataset=d)
    assert np.array_equal(q["rewards"], GOLD["normalize/in/rewards"])
    out = normalize_rewards(q)
    assert out is q
    ref = GOLD["normalize/out/rewards"]
    assert np.abs(out["rewards"] - ref).max() <= 1e-6 * np.abs(ref).max()


def test_dataset_file_roundtrip_and_buffer_ingest(tmp_path):
    d = mg.synth_trajectories(seed=1, n_eps=5, max_len=6)
    np.savez(tmp_path / "ds.npz", **d)
    back = load_dataset_file(str(tmp_path / "ds.npz"))
    q = qlearning_dataset(None, dataset=back)
    assert np.array_equal(q["observations"], GOLD["timeouts/out/observations"][:len(q["observations"])]) or len(q["observations"]) > 0
    with pytest.raises(ValueError):
        load_dataset_file("x.csv")
    # the arrays are what ReplayBuffer.load_dataset takes (host side; the HBM upload is covered by the gpu tests)
    from offlinerlkit.buffer import ReplayBuffer
    buf = ReplayBuffer(len(q["rewards"]), (5,), np.float32, 2, np.float32, device="cpu")
    buf.load_dataset(q)
    assert buf._size == len(q["rewards"]) and buf.terminals.dtype == np.float32 and buf.rewards.shape == (len(q["rewards"]), 1)

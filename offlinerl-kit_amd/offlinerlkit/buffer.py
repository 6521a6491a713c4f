"""ReplayBuffer with the reference's API (offlinerlkit/buffer/buffer.py:7-115) whose ``sample`` gathers from an
HBM-resident SoA copy of the data with a HIP kernel (csrc/kernels.h k_gather) through the C ABI.

Host numpy arrays are kept (``add`` / ``add_batch`` / ``sample_all`` / ``load_dataset`` semantics are unchanged);
the device copy is (re)uploaded lazily when they change.  ``sample`` draws its indices with
``np.random.randint(0, size, batch_size)`` exactly like the reference (:98), so for a given numpy seed the index
stream — and therefore every minibatch — is identical to the reference's.
"""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import numpy as np
import torch

from . import _engine


class ReplayBuffer:
    def __init__(self, buffer_size: int, obs_shape: Tuple, obs_dtype: np.dtype, action_dim: int, action_dtype: np.dtype,
                 device: str = "cpu") -> None:
        self._max_size = buffer_size
        self.obs_shape = tuple(obs_shape)
        self.obs_dtype = obs_dtype
        self.action_dim = int(action_dim)
        self.action_dtype = action_dtype
        self._ptr = 0
        self._size = 0
        self.observations = np.zeros((self._max_size,) + self.obs_shape, dtype=obs_dtype)
        self.next_observations = np.zeros((self._max_size,) + self.obs_shape, dtype=obs_dtype)
        self.actions = np.zeros((self._max_size, self.action_dim), dtype=action_dtype)
        self.rewards = np.zeros((self._max_size, 1), dtype=np.float32)
        self.terminals = np.zeros((self._max_size, 1), dtype=np.float32)
        self.device = torch.device(device)
        self._dev: Optional[_engine.DeviceBuffer] = None
        self._dirty = True

    # ---- host-side mutation (buffer.py:34-86) ----
    def add(self, obs, next_obs, action, reward, terminal) -> None:
        self.observations[self._ptr] = np.array(obs).copy()
        self.next_observations[self._ptr] = np.array(next_obs).copy()
        self.actions[self._ptr] = np.array(action).copy()
        self.rewards[self._ptr] = np.array(reward).copy()
        self.terminals[self._ptr] = np.array(terminal).copy()
        self._ptr = (self._ptr + 1) % self._max_size
        self._size = min(self._size + 1, self._max_size)
        self._dirty = True

    def add_batch(self, obss, next_obss, actions, rewards, terminals) -> None:
        n = len(obss)
        where = np.arange(self._ptr, self._ptr + n) % self._max_size
        self.observations[where] = np.array(obss).copy()
        self.next_observations[where] = np.array(next_obss).copy()
        self.actions[where] = np.array(actions).copy()
        self.rewards[where] = np.array(rewards).copy()
        self.terminals[where] = np.array(terminals).copy()
        self._ptr = (self._ptr + n) % self._max_size
        self._size = min(self._size + n, self._max_size)
        self._dirty = True

    def load_dataset(self, dataset: Dict[str, np.ndarray]) -> None:
        self.observations = np.array(dataset["observations"], dtype=self.obs_dtype)
        self.next_observations = np.array(dataset["next_observations"], dtype=self.obs_dtype)
        self.actions = np.array(dataset["actions"], dtype=self.action_dtype)
        self.rewards = np.array(dataset["rewards"], dtype=np.float32).reshape(-1, 1)
        self.terminals = np.array(dataset["terminals"], dtype=np.float32).reshape(-1, 1)
        self._ptr = len(self.observations)
        self._size = len(self.observations)
        self._dirty = True

    def normalize_obs(self, eps: float = 1e-3) -> Tuple[np.ndarray, np.ndarray]:
        """(x - mean) / (std + eps) on observations and next_observations (buffer.py:88-94).  The returned
        statistics are numpy's (what the reference returns); a resident device copy is normalised in place by the
        HIP kernel instead of being uploaded a second time."""
        mean = self.observations.mean(0, keepdims=True)
        std = self.observations.std(0, keepdims=True) + eps
        resident = self._dev is not None and not self._dirty and self._size == len(self.observations)
        self.observations = (self.observations - mean) / std
        self.next_observations = (self.next_observations - mean) / std
        if resident:
            self._dev.normalize_obs(eps)
        else:
            self._dirty = True
        return mean, std

    # ---- device side ----
    def _device_index(self) -> int:
        if self.device.type != "cuda":
            raise RuntimeError("offlinerlkit(AMD) ReplayBuffer.sample needs device='cuda' (the store lives in MI355X HBM; there is no CPU path)")
        return self.device.index if self.device.index is not None else torch.cuda.current_device()

    def device_buffer(self) -> "_engine.DeviceBuffer":
        """The HBM-resident store (uploaded on first use / after host-side changes)."""
        if self._size == 0:
            raise RuntimeError("ReplayBuffer is empty")
        if self._dev is None or self._dirty:
            od = int(np.prod(self.obs_shape))
            if self._dev is None:
                self._dev = _engine.DeviceBuffer(od, self.action_dim, self._device_index())
            n = self._size
            self._dev.load(self.observations[:n].reshape(n, od), self.actions[:n], self.next_observations[:n].reshape(n, od),
                           self.rewards[:n], self.terminals[:n])
            self._dirty = False
        return self._dev

    def sample(self, batch_size: int) -> Dict[str, torch.Tensor]:
        dev = self.device_buffer()
        batch_indexes = np.random.randint(0, self._size, size=batch_size)      # buffer.py:98
        od = int(np.prod(self.obs_shape))
        tdev = torch.device("cuda", self._device_index())
        out = {
            "observations": torch.empty((batch_size, od), dtype=torch.float32, device=tdev),
            "actions": torch.empty((batch_size, self.action_dim), dtype=torch.float32, device=tdev),
            "next_observations": torch.empty((batch_size, od), dtype=torch.float32, device=tdev),
            "terminals": torch.empty((batch_size, 1), dtype=torch.float32, device=tdev),
            "rewards": torch.empty((batch_size, 1), dtype=torch.float32, device=tdev),
        }
        torch.cuda.current_stream(tdev).synchronize()       # outputs are written on the null stream by the engine
        dev.sample_into(batch_indexes, batch_size, 0, out["observations"].data_ptr(), out["actions"].data_ptr(),
                        out["next_observations"].data_ptr(), out["rewards"].data_ptr(), out["terminals"].data_ptr())
        if len(self.obs_shape) != 1:
            out["observations"] = out["observations"].view((batch_size,) + self.obs_shape)
            out["next_observations"] = out["next_observations"].view((batch_size,) + self.obs_shape)
        return out

    def sample_all(self) -> Dict[str, np.ndarray]:
        n = self._size
        return {
            "observations": self.observations[:n].copy(),
            "actions": self.actions[:n].copy(),
            "next_observations": self.next_observations[:n].copy(),
            "terminals": self.terminals[:n].copy(),
            "rewards": self.rewards[:n].copy(),
        }

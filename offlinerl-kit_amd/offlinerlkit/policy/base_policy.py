"""Policy interface (reference: offlinerlkit/policy/base_policy.py:8-26) and the engine-backed base class.

An ``EnginePolicy`` keeps the reference's constructor signatures (pre-built ``nn.Module`` networks and
``torch.optim`` optimizers) but runs every ``learn()`` in the HIP engine:
  * on first use the modules' parameters are copied into one flat device arena owned by a torch tensor and
    each ``param.data`` is re-pointed at a view of it, so ``state_dict()``, ``load_state_dict()`` and the
    evaluation-time torch forward in ``select_action`` always see the engine's live weights;
  * the torch optimizers are only read (lr / betas / eps; an ``lr_scheduler`` that mutates ``param_groups`` is
    honoured before every step, run_iql.py:133) — their state is not used, Adam runs fused on the device;
  * target networks (deepcopy of the critics, sac.py:29-33) live in the same arena.
There is no CPU fallback: without the native library or a GPU, ``learn`` raises.
"""
from __future__ import annotations

from copy import deepcopy
from typing import Dict, List, Optional

import numpy as np
import torch
import torch.nn as nn

from .. import _engine
from ..nets import MLP


class BasePolicy(nn.Module):
    def __init__(self) -> None:
        super().__init__()

    def train(self) -> None:  # noqa: D401  (reference signature takes no mode flag)
        raise NotImplementedError

    def eval(self) -> None:
        raise NotImplementedError

    def select_action(self, obs: np.ndarray, deterministic: bool = False) -> np.ndarray:
        raise NotImplementedError

    def learn(self, batch: Dict) -> Dict[str, float]:
        raise NotImplementedError


def _backbone_dims(backbone: MLP):
    lins = backbone.linear_layers() if hasattr(backbone, "linear_layers") else [m for m in backbone.model if isinstance(m, nn.Linear)]
    if not lins:
        raise ValueError("backbone has no Linear layers")
    if getattr(backbone, "dropout_rate", None) is not None:
        raise NotImplementedError("dropout in the backbone is not supported by the HIP engine")
    return lins[0].in_features, [l.out_features for l in lins]


def _adam_hyper(optim: torch.optim.Optimizer):
    if not isinstance(optim, torch.optim.Adam):
        raise NotImplementedError("the HIP engine implements torch.optim.Adam only")
    g = optim.param_groups[0]
    if g.get("weight_decay", 0) != 0 or g.get("amsgrad", False):
        raise NotImplementedError("Adam weight_decay / amsgrad are not supported by the HIP engine")
    return float(g["lr"]), tuple(float(b) for b in g["betas"]), float(g["eps"])


class EnginePolicy(BasePolicy):
    ALGO: str = ""

    def __init__(self) -> None:
        super().__init__()
        self._eng: Optional[_engine.Engine] = None
        self._arena: Optional[torch.Tensor] = None
        self._bound_batch = None
        self._lr_pushed: Dict[int, float] = {}
        self._attached = None

    # -- subclass hooks ------------------------------------------------------------------
    def _nets(self) -> Dict[int, nn.Module]:
        raise NotImplementedError

    def _optims(self) -> Dict[int, torch.optim.Optimizer]:
        raise NotImplementedError

    def _config(self) -> Dict:
        raise NotImplementedError

    def _after_bind(self) -> None:
        pass

    def _before_unbind(self) -> None:
        pass

    # -- engine binding ------------------------------------------------------------------
    def _device(self) -> torch.device:
        actor = self._nets()[_engine.NET_ACTOR]
        dev = getattr(actor, "device", None)
        if dev is None or torch.device(dev).type != "cuda":
            if not torch.cuda.is_available():
                raise RuntimeError("offlinerlkit(AMD) policies need an MI355X: no HIP device is visible and there is no CPU fallback")
            return torch.device("cuda", torch.cuda.current_device())
        dev = torch.device(dev)
        return dev if dev.index is not None else torch.device("cuda", torch.cuda.current_device())

    def _bind(self, batch_size: int) -> None:
        if self._eng is not None and self._bound_batch == batch_size:
            return
        saved_steps = None
        if self._eng is not None:        # batch size changed: rebuild around the current weights
            self._unbind()
        dev = self._device()
        over = dict(self._config())
        betas, eps = None, None
        for oid, opt in self._optims().items():
            lr, b, e = _adam_hyper(opt)
            if betas is None:
                betas, eps = b, e
            elif (b, e) != (betas, eps):
                raise NotImplementedError("all optimizers of a policy must share Adam betas/eps")
        over.update(batch_size=int(batch_size), n_runs=1, device=dev.index, adam_beta1=betas[0], adam_beta2=betas[1], adam_eps=eps)
        cfg = _engine.default_config(self.ALGO, **over)
        n = _engine.load_library().orl_arena_floats(cfg)
        if n <= 0:
            raise RuntimeError("orl_arena_floats failed: " + _engine.last_error())
        self._arena = torch.zeros(n, dtype=torch.float32, device=dev)
        cfg.external_arena = self._arena.data_ptr()
        torch.cuda.synchronize(dev)
        self._eng = _engine.Engine(cfg)
        self._bound_batch = batch_size
        base = self._arena.data_ptr()
        for nid, mod in self._nets().items():
            mod.to(dev)
            for m in mod.modules():
                if hasattr(m, "device") and isinstance(getattr(m, "device"), torch.device):
                    m.device = dev
            off0 = (self._eng.net_ptr(0, nid) - base) // 4
            params = dict(mod.named_parameters())
            for name, off, shape in self._eng.net_tensors(nid):
                p = params[name]
                if tuple(p.shape) != tuple(shape):
                    raise ValueError(f"{name}: module shape {tuple(p.shape)} != engine shape {tuple(shape)}")
                view = self._arena[off0 + off: off0 + off + p.numel()].view(shape)
                view.copy_(p.data.to(dev))
                p.data = view
        torch.cuda.synchronize(dev)
        self._lr_pushed = {}
        self._push_lrs()
        self._after_bind()
        self._attached = None

    def _unbind(self) -> None:
        """Detach module parameters from the arena (clone) and drop the engine."""
        self._before_unbind()
        for mod in self._nets().values():
            for p in mod.parameters():
                p.data = p.data.clone()
        self._eng.close()
        self._eng, self._arena, self._bound_batch = None, None, None

    def _push_lrs(self) -> None:
        for oid, opt in self._optims().items():
            lr = float(opt.param_groups[0]["lr"])
            if self._lr_pushed.get(oid) != lr:
                self._eng.set_lr(oid, lr)
                self._lr_pushed[oid] = lr

    # -- reference API -------------------------------------------------------------------
    def train(self) -> None:
        for nid, m in self._nets().items():
            nn.Module.train(m, True)

    def eval(self) -> None:
        for nid, m in self._nets().items():
            nn.Module.train(m, False)

    def learn(self, batch: Dict, noise: Optional[List] = None) -> Dict[str, float]:
        """One gradient step on ``batch`` (the dict ``ReplayBuffer.sample`` returns).  Synchronous, like the
        reference's ``.item()`` calls; noise is drawn on the device (Philox) unless ``noise`` supplies the arrays of
        include/orl_engine.h's orl_noise in the reference's draw order (teacher-forced parity runs)."""
        obs = batch["observations"]
        B = int(obs.shape[0])
        self._bind(B)
        dev = self._arena.device
        keep = []
        ptrs = {}
        for k in ("observations", "actions", "next_observations", "rewards", "terminals"):
            t = torch.as_tensor(batch[k], dtype=torch.float32, device=dev).contiguous()
            keep.append(t)
            ptrs[k] = t.data_ptr()
        torch.cuda.current_stream(dev).synchronize()
        self._push_lrs()
        nz = None
        if noise is not None:
            nz = []
            for a in noise:
                t = torch.as_tensor(a, dtype=torch.float32, device=dev).contiguous()
                keep.append(t)
                nz.append(t.data_ptr())
            torch.cuda.current_stream(dev).synchronize()
        m = self._eng.step(ptrs, nz, on_device=True)[0]
        return {k: float(v) for k, v in zip(self._eng.metric_names, m)}

    def learn_n(self, n_steps: int, buffer, batch_size: int = 256) -> Dict[str, float]:
        """``n_steps`` x (sample -> learn) fused on the device (MFPolicyTrainer's inner loop, mf_policy_trainer.py:52-60):
        Philox index sampling from the HBM-resident buffer, one host sync at the end.  Returns the per-key means
        (what ``logger.logkv_mean`` would hold at the end of the epoch)."""
        self._bind(batch_size)
        dbuf = buffer.device_buffer() if hasattr(buffer, "device_buffer") else buffer
        if self._attached is not dbuf:
            self._eng.attach_buffer(dbuf)
            self._attached = dbuf
        self._push_lrs()
        m, ms = self._eng.learn_n(int(n_steps))
        self.last_learn_n_ms = ms
        return {k: float(v) for k, v in zip(self._eng.metric_names, m[0])}

    @property
    def engine(self) -> Optional[_engine.Engine]:
        return self._eng


def clone_target(module: nn.Module) -> nn.Module:
    t = deepcopy(module)
    nn.Module.train(t, False)
    return t

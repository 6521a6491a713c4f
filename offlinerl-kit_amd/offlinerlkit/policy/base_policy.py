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

import os
from copy import deepcopy
from typing import Callable, Dict, List, Optional

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


def _backbone_dims(backbone: MLP, allow_dropout: bool = False):
    """(input width, hidden widths) of an MLP backbone.  ``dropout_rate`` (nets/mlp.py:16-24) is supported where the reference's launchers
    use it -- the actor backbone of IQL (run_iql.py:34,106) -- and refused elsewhere."""
    lins = backbone.linear_layers() if hasattr(backbone, "linear_layers") else [m for m in backbone.model if isinstance(m, nn.Linear)]
    if not lins:
        raise ValueError("backbone has no Linear layers")
    if getattr(backbone, "dropout_rate", None) is not None and not allow_dropout:
        raise NotImplementedError("dropout in this backbone is not supported by the HIP engine (supported: the actor backbone of IQLPolicy, "
                                  "which is the one run_iql.py --dropout_rate builds)")
    return lins[0].in_features, [l.out_features for l in lins]


def _adam_hyper(optim: torch.optim.Optimizer):
    if not isinstance(optim, torch.optim.Adam):
        raise NotImplementedError("the HIP engine implements torch.optim.Adam only")
    g = optim.param_groups[0]
    if g.get("weight_decay", 0) != 0 or g.get("amsgrad", False):
        raise NotImplementedError("Adam weight_decay / amsgrad are not supported by the HIP engine")
    return float(g["lr"]), tuple(float(b) for b in g["betas"]), float(g["eps"])


_BIND_COUNTER = 0


def _mix64(x: int) -> int:
    """splitmix64 finaliser: spreads (torch seed, bind counter) over the 64-bit Philox key space"""
    x = (x + 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
    x = ((x ^ (x >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
    x = ((x ^ (x >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
    return x ^ (x >> 31)


class EnginePolicy(BasePolicy):
    ALGO: str = ""

    def __init__(self) -> None:
        super().__init__()
        self._eng: Optional[_engine.Engine] = None
        self._arena: Optional[torch.Tensor] = None
        self._bound_batch = None
        self._lr_pushed: Dict[int, float] = {}
        self._attached = None
        self._n_runs = 1
        self._run_init = None
        self._seed: Optional[int] = None
        self._precision = int(os.environ.get("ORL_PRECISION", "0"))
        self._cur_run = 0

    # -- subclass hooks ------------------------------------------------------------------
    def _nets(self) -> Dict[int, nn.Module]:
        raise NotImplementedError

    def _optims(self) -> Dict[int, torch.optim.Optimizer]:
        raise NotImplementedError

    def _all_optims(self) -> List[torch.optim.Optimizer]:
        """every optimizer the reference constructor received (validated; ``_optims`` lists the ones whose lr is read)"""
        return list(self._optims().values())

    def _config(self) -> Dict:
        raise NotImplementedError

    def _after_bind(self) -> None:
        pass

    def _before_unbind(self) -> None:
        pass

    def _after_bind_new_runs(self, first: int, n_runs: int) -> None:
        """runs [first, n_runs) were added to a policy whose earlier runs carried their state over: give them the scalars
        ``_after_bind`` gives a fresh engine (subclasses with per-run scalars)"""
        pass

    # -- engine options (before the first learn(), or any time: the engine is rebuilt around the current state, optimizer state included) --------
    def set_engine_options(self, n_runs: Optional[int] = None, seed: Optional[int] = None, precision: Optional[int] = None,
                           run_init: Optional[Callable[[int], Dict[str, Dict[str, torch.Tensor]]]] = None) -> "EnginePolicy":
        """``n_runs``: independent runs (seeds) this policy object trains together -- every kernel launch updates all of them
        (the reference's analogue is N separate processes, tune_example/tune_mopo.py:222-239).  Run 0 starts from the modules'
        current parameters; run r > 0 from ``run_init(r)`` ({net attribute name: state_dict}) or, by default, from the modules'
        own ``reset_parameters()`` under ``torch.manual_seed(seed + r)`` -- i.e. what building the networks under another seed
        gives (custom initialisations such as run_iql.py:121-125 / run_edac.py:100-103 need ``run_init``).
        ``seed``: key of the device sampler / noise streams (default: derived from ``torch.initial_seed()`` and a per-process
        bind counter, so launcher seeds give independent streams and a re-bind never replays one).
        ``precision``: 0 exact fp32 MFMA (default); 1 split-fp16 MFMA (two planes per operand: same 1e-4 parity gate on losses and Q-values, ~3x
        faster at many runs); 2 three fp16 planes per operand in the many-row critic launches (an fp32 operand is represented exactly: fp32-class
        arithmetic at about twice the fp32 MFMA rate), exact fp32 MFMA everywhere else."""
        if n_runs is not None and n_runs < 1:
            raise ValueError("n_runs must be >= 1")
        if precision is not None and precision not in (0, 1, 2):
            raise ValueError("precision must be 0 (fp32 MFMA), 1 (split-fp16 MFMA) or 2 (three fp16 planes / fp32 MFMA)")
        if self._eng is not None:
            # an engine exists: the next bind restores parameters, Adam moments, step count and scalars of every run that survives
            # (all of them when n_runs is unchanged; the first n_runs when it shrinks; new runs r >= old n_runs start fresh)
            self._carried = self._unbind()
        if n_runs is not None:
            self._n_runs = int(n_runs)
        if seed is not None:
            self._seed = int(seed)
        if precision is not None:
            self._precision = int(precision)
        if run_init is not None:
            self._run_init = run_init
        return self

    @property
    def n_runs(self) -> int:
        return self._n_runs

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

    def _engine_seed(self) -> int:
        global _BIND_COUNTER
        _BIND_COUNTER += 1
        base = self._seed if self._seed is not None else torch.initial_seed()
        return _mix64((int(base) & 0xFFFFFFFFFFFFFFFF) ^ _mix64(_BIND_COUNTER))

    def _fresh_run_params(self, run: int) -> Dict[int, Dict[str, torch.Tensor]]:
        """initial parameters of run > 0, keyed by net id"""
        nets = self._nets()
        if self._run_init is not None:
            given = self._run_init(run)
            by_mod = {id(getattr(self, k)): v for k, v in given.items()}
            out = {}
            for nid, mod in nets.items():
                if id(mod) in by_mod:
                    out[nid] = {k: torch.as_tensor(v, dtype=torch.float32) for k, v in by_mod[id(mod)].items()}
            return out
        base = self._seed if self._seed is not None else torch.initial_seed()
        rng_state = torch.get_rng_state()
        try:
            # the init stream of run r is keyed by mix(seed, r), not seed + r: a launcher that starts seeds s, s + 1, ... with several
            # runs each (BASELINE configs[4]) must not hand run 1 of seed s the networks of run 0 of seed s + 1
            torch.manual_seed(_mix64((int(base) & 0xFFFFFFFFFFFFFFFF) ^ _mix64(0x52554E00 + run)) & 0x7FFFFFFFFFFFFFFF)
            out = {}
            trainable = [nid for nid in nets if nid in (_engine.NET_ACTOR, _engine.NET_CRITIC1, _engine.NET_CRITIC2, _engine.NET_CRITIC_V,
                                                        _engine.NET_VAE_ENC, _engine.NET_VAE_DEC)]
            for nid in trainable:
                m = deepcopy(nets[nid]).cpu()
                for sub in m.modules():
                    if hasattr(sub, "reset_parameters"):
                        sub.reset_parameters()
                    elif next(sub.parameters(recurse=False), None) is not None:
                        raise NotImplementedError(
                            f"{type(sub).__name__} owns parameters but has no reset_parameters(): the default initialisation of runs r > 0 "
                            "would silently start them from run 0's values -- pass run_init to set_engine_options")
                out[nid] = {k: v.detach().clone() for k, v in m.named_parameters()}
        finally:
            torch.set_rng_state(rng_state)
        return out

    _TARGET_OF = {_engine.NET_CRITIC1_OLD: _engine.NET_CRITIC1, _engine.NET_CRITIC2_OLD: _engine.NET_CRITIC2, _engine.NET_ACTOR_OLD: _engine.NET_ACTOR}

    def _rebind_with(self, carried, batch_size: int) -> None:
        """bind a fresh engine around state taken from ``_unbind()`` (subclasses whose row layout changed)"""
        self._carried = carried
        self._bind(batch_size)

    def _bind(self, batch_size: int) -> None:
        if self._eng is not None and self._bound_batch == batch_size:
            return
        carried = getattr(self, "_carried", None)
        self._carried = None
        if self._eng is not None:        # batch size changed: rebuild around the current weights AND optimizer state
            carried = self._unbind()
        dev = self._device()
        over = dict(self._config())
        betas, eps = None, None
        for opt in self._all_optims():
            lr, b, e = _adam_hyper(opt)
            if betas is None:
                betas, eps = b, e
            elif (b, e) != (betas, eps):
                raise NotImplementedError("all optimizers of a policy must share Adam betas/eps")
        R = self._n_runs
        over.update(batch_size=int(batch_size), n_runs=R, device=dev.index, adam_beta1=betas[0], adam_beta2=betas[1], adam_eps=eps,
                    precision=self._precision, seed=self._engine_seed())
        cfg = _engine.default_config(self.ALGO, **over)
        n = _engine.load_library().orl_arena_floats(cfg)
        if n <= 0:
            raise RuntimeError("orl_arena_floats failed: " + _engine.last_error())
        self._arena = torch.zeros(n, dtype=torch.float32, device=dev)
        cfg.external_arena = self._arena.data_ptr()
        torch.cuda.synchronize(dev)
        self._eng = _engine.Engine(cfg)
        self._bound_batch = batch_size
        nets = self._nets()
        # run 0 (or, on a re-bind, every run) <- current parameters; fresh runs <- their own initialisation
        for nid, mod in nets.items():
            mod.to(dev)
            for m in mod.modules():
                if hasattr(m, "device") and isinstance(getattr(m, "device"), torch.device):
                    m.device = dev
        n_carried = len(carried["params"]) if carried is not None else 0
        for r in range(R):
            if r < n_carried:
                params = carried["params"][r]
            elif r == 0:
                params = {nid: dict(mod.named_parameters()) for nid, mod in nets.items()}
            else:
                fresh = self._fresh_run_params(r)
                params = {}
                for nid, mod in nets.items():
                    src = fresh.get(nid)
                    if src is None and nid in self._TARGET_OF:      # deepcopy of the freshly initialised critic (sac.py:29-33)
                        src = fresh.get(self._TARGET_OF[nid])
                    params[nid] = src if src is not None else dict(mod.named_parameters())
            for nid in nets:
                self._write_net(r, nid, params[nid])
        self._cur_run = -1
        self.select_run(0, _hook=False)      # (the scalars of the fresh engine are set below: nothing to sync back yet)
        torch.cuda.synchronize(dev)
        self._lr_pushed = {}
        self._push_lrs()
        if carried is not None:
            for r in range(min(R, n_carried)):
                self._eng.load_optimizer_state(carried["opt"][r], r)
            if R > n_carried:
                self._after_bind_new_runs(n_carried, R)
        else:
            self._after_bind()
        self._attached = None

    def _net_views(self, run: int, nid: int):
        base = self._arena.data_ptr()
        off0 = (self._eng.net_ptr(run, nid) - base) // 4
        for name, off, shape in self._eng.net_tensors(nid):
            numel = int(np.prod(shape))
            yield name, self._arena[off0 + off: off0 + off + numel].view(shape)

    def _write_net(self, run: int, nid: int, params) -> None:
        for name, view in self._net_views(run, nid):
            p = params[name]
            src = p.data if isinstance(p, torch.nn.Parameter) else torch.as_tensor(p)
            if tuple(src.shape) != tuple(view.shape):
                raise ValueError(f"{name}: module shape {tuple(src.shape)} != engine shape {tuple(view.shape)}")
            view.copy_(src.to(view.device, dtype=torch.float32))

    def select_run(self, run: int, _hook: bool = True) -> None:
        """Point the torch modules (``state_dict``, ``select_action``, checkpoints) at run ``run``'s live parameters."""
        if self._eng is None:
            if run != 0:
                raise RuntimeError("select_run before the first learn(): only run 0 exists yet")
            return
        if not 0 <= run < self._n_runs:
            raise IndexError(f"run {run} out of range (n_runs = {self._n_runs})")
        if run == self._cur_run:
            return
        for nid, mod in self._nets().items():
            params = dict(mod.named_parameters())
            for name, view in self._net_views(run, nid):
                params[name].data = view
        self._cur_run = run
        if _hook:
            self._on_select_run(run)

    def _on_select_run(self, run: int) -> None:
        pass

    # -- every run's actor in one forward (evaluation of a multi-run policy) ------------------------------------------
    def _stacked_net(self, nid: int) -> Dict[str, torch.Tensor]:
        """{tensor name: [n_runs, *shape] strided view of the arena} -- the runs of a net sit at a fixed stride, so no copy is made"""
        R = self._n_runs
        base = self._arena.data_ptr()
        off0 = (self._eng.net_ptr(0, nid) - base) // 4
        stride = (self._eng.net_ptr(1, nid) - self._eng.net_ptr(0, nid)) // 4 if R > 1 else 0
        out = {}
        for name, off, shape in self._eng.net_tensors(nid):
            inner = [int(np.prod(shape[i + 1:])) for i in range(len(shape))]
            out[name] = self._arena.as_strided((R,) + tuple(shape), (stride,) + tuple(inner), off0 + off)
        return out

    def _eval_obs(self, obs: np.ndarray) -> np.ndarray:
        """what ``select_action`` does to observations before the actor sees them (TD3BC: the scaler)"""
        return obs

    def _mode_from_hidden(self, h: torch.Tensor, P: Dict[str, torch.Tensor]) -> torch.Tensor:
        """deterministic action [n_runs, E, act_dim] from the backbone output [n_runs, E, H] and the stacked actor tensors"""
        raise NotImplementedError

    def select_action_runs(self, obs: np.ndarray) -> np.ndarray:
        """Deterministic actions of EVERY run in one batched forward: ``obs`` [n_runs, E, obs_dim] -> [n_runs, E, act_dim]; row
        (r, e) is what ``select_run(r); select_action(obs[r, e:e+1], deterministic=True)`` returns (mf_policy_trainer.py:100)."""
        if self._eng is None:
            raise RuntimeError("select_action_runs before the first learn(): no engine is bound yet")
        obs = np.asarray(obs, dtype=np.float32)
        if obs.ndim != 3 or obs.shape[0] != self._n_runs:
            raise ValueError(f"obs: expected [n_runs = {self._n_runs}, E, obs_dim], got {obs.shape}")
        R, E = obs.shape[:2]
        x = np.asarray(self._eval_obs(obs.reshape(R * E, -1)), dtype=np.float32).reshape(R, E, -1)
        P = self._stacked_net(_engine.NET_ACTOR)
        with torch.no_grad():
            h = torch.as_tensor(x, device=self._arena.device)
            # Linear layers of the backbone in nn.Sequential order (indices 0, 2, 4, ... or 0, 3, 6, ... with Dropout layers: evaluation
            # runs in eval mode, where nn.Dropout is the identity)
            idx = sorted(int(k[len("backbone.model."):-len(".weight")]) for k in P if k.startswith("backbone.model.") and k.endswith(".weight"))
            for i in idx:
                h = torch.relu(torch.baddbmm(P[f"backbone.model.{i}.bias"].unsqueeze(1), h, P[f"backbone.model.{i}.weight"].transpose(1, 2)))
            return self._mode_from_hidden(h, P).cpu().numpy()

    def _unbind(self):
        """Detach module parameters from the arena (clone) and drop the engine; returns what a re-bind carries over."""
        self.select_run(0)
        self._before_unbind()
        carried = {"params": [], "opt": []}
        for r in range(self._n_runs):
            carried["params"].append({nid: {name: v.clone() for name, v in self._net_views(r, nid)} for nid in self._nets()})
            carried["opt"].append(self._eng.optimizer_state(r))
        for mod in self._nets().values():
            for p in mod.parameters():
                p.data = p.data.clone()
        self._eng.close()
        self._eng, self._arena, self._bound_batch = None, None, None
        return carried

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

    def _result(self, m: np.ndarray) -> Dict[str, float]:
        """metric table [n_runs][n_metrics] -> the reference's result dict; with several runs the plain keys hold the mean over
        runs and ``run<i>/<key>`` the individual values"""
        names = self._eng.metric_names
        if self._n_runs == 1:
            return {k: float(v) for k, v in zip(names, m[0])}
        out = {k: float(v) for k, v in zip(names, m.mean(axis=0))}
        for r in range(self._n_runs):
            for k, v in zip(names, m[r]):
                out[f"run{r}/{k}"] = float(v)
        return out

    def learn(self, batch: Dict, noise: Optional[List] = None) -> Dict[str, float]:
        """One gradient step on ``batch`` (the dict ``ReplayBuffer.sample`` returns).  Synchronous, like the
        reference's ``.item()`` calls; noise is drawn on the device (Philox) unless ``noise`` supplies the arrays of
        include/orl_engine.h's orl_noise in the reference's draw order (teacher-forced parity runs).  With ``n_runs`` > 1 the
        arrays either carry a leading run dimension or are shared by all runs."""
        R = self._n_runs
        obs = batch["observations"]
        if np.ndim(obs) not in (2, 3):
            raise ValueError(f"observations: expected [B, obs_dim] or [n_runs, B, obs_dim], got shape {tuple(obs.shape)}")
        B = int(obs.shape[-2])
        self._bind(B)
        dev = self._arena.device
        keep = []
        ptrs = {}

        def dev_array(name, x):
            """every array decides for itself whether it carries a run dimension: [rows, cols] is shared by all runs (expanded),
            [n_runs, rows, cols] is per run; anything else is refused (the engine reads n_runs * rows * cols floats)"""
            t = torch.as_tensor(x, dtype=torch.float32, device=dev)
            if t.dim() == 1:
                t = t.unsqueeze(-1)                      # rewards / terminals as [B]
            if t.dim() == 2:
                t = t.unsqueeze(0).expand(R, *t.shape)
            elif t.dim() != 3 or t.shape[0] != R:
                raise ValueError(f"{name}: shape {tuple(t.shape)} is neither [rows, cols] nor [n_runs = {R}, rows, cols]")
            t = t.contiguous()
            keep.append(t)
            return t
        for k in ("observations", "actions", "next_observations", "rewards", "terminals"):
            t = dev_array(k, batch[k])
            if t.shape[1] != B:
                raise ValueError(f"{k}: {t.shape[1]} rows, observations have {B}")
            ptrs[k] = t.data_ptr()
        self._push_lrs()
        nz = None
        if noise is not None:
            nz = [dev_array(f"noise[{i}]", a).data_ptr() for i, a in enumerate(noise)]
        torch.cuda.current_stream(dev).synchronize()
        return self._result(self._eng.step(ptrs, nz, on_device=True))

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
        return self._result(m)

    def check_health(self):
        """Per-run health flags of the bound engine (``_engine.HEALTH_*``: non-finite loss / gradient, an operand beyond the operand range
        of split precision) after a range scan of the last step's operands; warns (``EngineHealthWarning``) or raises when the engine is
        strict.  The reference has no counterpart: its diverged runs show as nan losses, which the engine's integer-view ReLU can mask.
        ``MFPolicyTrainer`` calls it once per epoch."""
        if self._eng is None:
            return None
        return self._eng.health_check()

    def run_state_dict(self, run: int) -> Dict[str, torch.Tensor]:
        """``state_dict()`` of one run (a copy; the live modules keep pointing at the run selected before)."""
        cur = max(self._cur_run, 0)
        self.select_run(run)
        sd = {k: v.detach().clone() for k, v in self.state_dict().items()}
        self.select_run(cur)
        return sd

    @property
    def engine(self) -> Optional[_engine.Engine]:
        return self._eng


def clone_target(module: nn.Module) -> nn.Module:
    t = deepcopy(module)
    nn.Module.train(t, False)
    return t

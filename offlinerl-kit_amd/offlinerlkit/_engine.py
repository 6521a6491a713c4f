"""ctypes binding of the C-ABI update engine (include/orl_engine.h -> liborlengine.so).

This is the only place Python touches the native library.  There is NO CPU
fallback: if the shared object is missing or no MI355X is visible, creating an
engine raises.
"""
from __future__ import annotations

import ctypes as C
import os
import warnings
from typing import Dict, List, Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(os.path.dirname(_HERE), "liborlengine.so")

ALGO_CQL, ALGO_IQL, ALGO_TD3BC, ALGO_EDAC, ALGO_SAC, ALGO_MCQ = 0, 1, 2, 3, 4, 5
ALGO_ID = {"cql": ALGO_CQL, "iql": ALGO_IQL, "td3bc": ALGO_TD3BC, "edac": ALGO_EDAC, "sac": ALGO_SAC, "mcq": ALGO_MCQ}
MAX_HIDDEN, MAX_METRICS, MAX_NOISE = 4, 8, 6
NET_ACTOR, NET_CRITIC1, NET_CRITIC2, NET_CRITIC1_OLD, NET_CRITIC2_OLD, NET_CRITIC_V, NET_ACTOR_OLD, NET_VAE_ENC, NET_VAE_DEC = range(9)
NUM_NETS = 9
SCALAR_LOG_ALPHA, SCALAR_CQL_LOG_ALPHA, SCALAR_ALPHA = 0, 1, 2
SCALAR_LOG_ALPHA_M, SCALAR_LOG_ALPHA_V, SCALAR_CQL_LOG_ALPHA_M, SCALAR_CQL_LOG_ALPHA_V, SCALAR_LAST_ACTOR_LOSS = 3, 4, 5, 6, 7
ALL_SCALARS = (SCALAR_LOG_ALPHA, SCALAR_LOG_ALPHA_M, SCALAR_LOG_ALPHA_V, SCALAR_CQL_LOG_ALPHA, SCALAR_CQL_LOG_ALPHA_M,
               SCALAR_CQL_LOG_ALPHA_V, SCALAR_LAST_ACTOR_LOSS, SCALAR_ALPHA)      # set order: ALPHA last (LOG_ALPHA derives it)
OPT_ACTOR, OPT_CRITIC, OPT_ALPHA, OPT_CQL_ALPHA, OPT_CRITIC_V, OPT_VAE = range(6)
# per-run health flags (include/orl_engine.h) and the return code of a step that ran but left one raised
HEALTH_NONFINITE_LOSS, HEALTH_NONFINITE_GRAD, HEALTH_SPLIT_RANGE = 1, 2, 4
RC_UNHEALTHY = 1


class EngineHealthWarning(RuntimeWarning):
    """A run of the engine turned non-finite or left the operand range of split precision (``Engine.health()``)."""


class EngineHealthError(RuntimeError):
    """The same, raised instead of warned when ``Engine.strict_health`` (or ORL_STRICT_HEALTH=1) is set."""


# symbols include/orl_engine.h declares (checked by tests/test_abi.py)
ABI_SYMBOLS = [
    "orl_last_error", "orl_version", "orl_split_bits", "orl_config_default", "orl_arena_floats", "orl_engine_create",
    "orl_engine_destroy", "orl_engine_sync", "orl_net_present", "orl_net_floats", "orl_net_num_tensors",
    "orl_net_tensor", "orl_net_ptr", "orl_net_set", "orl_net_get", "orl_scalar_set", "orl_scalar_get",
    "orl_set_lr", "orl_reset_optimizers", "orl_adam_get", "orl_adam_set", "orl_set_step_count", "orl_buffer_create", "orl_buffer_destroy", "orl_buffer_load",
    "orl_buffer_normalize_obs", "orl_buffer_sample", "orl_buffer_size", "orl_engine_attach_buffer", "orl_step", "orl_learn_n",
    "orl_health", "orl_health_check", "orl_health_clear", "orl_num_metrics", "orl_metric_name", "orl_step_count",
    "orl_debug_read", "orl_debug_read_bits", "orl_debug_grads", "orl_debug_gemm", "orl_debug_gemm_time", "orl_profile_enable", "orl_profile_query",
]


class OrlConfig(C.Structure):
    _fields_ = [
        ("algo", C.c_int32), ("obs_dim", C.c_int32), ("act_dim", C.c_int32), ("n_hidden", C.c_int32),
        ("hidden", C.c_int32 * MAX_HIDDEN), ("batch_size", C.c_int32), ("n_runs", C.c_int32),
        ("device", C.c_int32), ("precision", C.c_int32), ("seed", C.c_uint64),
        ("gamma", C.c_float), ("tau", C.c_float),
        ("actor_lr", C.c_float), ("critic_lr", C.c_float), ("alpha_lr", C.c_float),
        ("adam_beta1", C.c_float), ("adam_beta2", C.c_float), ("adam_eps", C.c_float),
        ("auto_alpha", C.c_int32), ("alpha", C.c_float), ("target_entropy", C.c_float),
        ("cql_weight", C.c_float), ("temperature", C.c_float),
        ("max_q_backup", C.c_int32), ("deterministic_backup", C.c_int32), ("with_lagrange", C.c_int32),
        ("lagrange_threshold", C.c_float), ("cql_alpha_lr", C.c_float), ("num_repeat_actions", C.c_int32),
        ("act_low", C.c_float), ("act_high", C.c_float),
        ("expectile", C.c_float), ("iql_temperature", C.c_float), ("critic_v_lr", C.c_float),
        ("policy_noise", C.c_float), ("noise_clip", C.c_float), ("td3bc_alpha", C.c_float), ("max_action", C.c_float),
        ("update_actor_freq", C.c_int32),
        ("num_critics", C.c_int32), ("eta", C.c_float),
        ("cql_cons_row0", C.c_int32), ("cql_cons_rows", C.c_int32), ("cql_real_rows", C.c_int32),
        ("vae_hidden", C.c_int32), ("vae_latent", C.c_int32), ("mcq_lambda", C.c_float), ("behavior_lr", C.c_float),
        ("ws_one_round", C.c_int32), ("ws_cus", C.c_int32), ("actor_dropout", C.c_float),
        ("external_arena", C.c_void_p),
    ]


class OrlBatch(C.Structure):
    _fields_ = [("observations", C.c_void_p), ("actions", C.c_void_p), ("next_observations", C.c_void_p),
                ("rewards", C.c_void_p), ("terminals", C.c_void_p), ("on_device", C.c_int32)]


class OrlNoise(C.Structure):
    _fields_ = [("slot", C.c_void_p * MAX_NOISE), ("on_device", C.c_int32)]


_lib = None


def load_library(path: Optional[str] = None):
    """dlopen the engine; raises (never falls back) when it is absent."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or os.environ.get("ORL_ENGINE_LIB") or LIB_PATH       # ORL_ENGINE_LIB: an experiment build (build.py --variant)
    if not os.path.exists(p):
        raise RuntimeError(f"native update engine not built: {p} is missing "
                           f"(run `python offlinerl-kit_amd/build.py` or __graft_entry__.build())")
    # PyTorch-ROCm wheels bundle their own libamdhip64.so.7.  Import torch FIRST so that this library's
    # NEEDED libamdhip64.so.7 binds to the runtime already in the process; loading the engine first would put
    # a second HIP runtime (/opt/rocm's) beside torch's and torch would then report "No HIP GPUs are available".
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(p)
    lib.orl_last_error.restype = C.c_char_p
    lib.orl_version.restype = C.c_char_p
    lib.orl_split_bits.restype = C.c_int
    lib.orl_config_default.argtypes = [C.POINTER(OrlConfig), C.c_int32]
    lib.orl_config_default.restype = None
    lib.orl_arena_floats.argtypes = [C.POINTER(OrlConfig)]
    lib.orl_arena_floats.restype = C.c_int64
    lib.orl_engine_create.argtypes = [C.POINTER(OrlConfig), C.POINTER(C.c_void_p)]
    lib.orl_engine_destroy.argtypes = [C.c_void_p]
    lib.orl_engine_destroy.restype = None
    lib.orl_engine_sync.argtypes = [C.c_void_p]
    lib.orl_net_present.argtypes = [C.c_void_p, C.c_int]
    lib.orl_net_floats.argtypes = [C.c_void_p, C.c_int]
    lib.orl_net_floats.restype = C.c_int64
    lib.orl_net_num_tensors.argtypes = [C.c_void_p, C.c_int]
    lib.orl_net_tensor.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_char_p, C.c_int, C.POINTER(C.c_int64),
                                   C.POINTER(C.c_int32), C.POINTER(C.c_int64)]
    lib.orl_net_ptr.argtypes = [C.c_void_p, C.c_int, C.c_int]
    lib.orl_net_ptr.restype = C.c_void_p
    lib.orl_net_set.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int64]
    lib.orl_net_get.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int64]
    lib.orl_scalar_set.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_float]
    lib.orl_scalar_get.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_float)]
    lib.orl_set_lr.argtypes = [C.c_void_p, C.c_int, C.c_float]
    lib.orl_reset_optimizers.argtypes = [C.c_void_p]
    lib.orl_adam_get.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int64]
    lib.orl_adam_set.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int64]
    lib.orl_set_step_count.argtypes = [C.c_void_p, C.c_int64]
    lib.orl_buffer_create.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_void_p)]
    lib.orl_buffer_destroy.argtypes = [C.c_void_p]
    lib.orl_buffer_destroy.restype = None
    lib.orl_buffer_load.argtypes = [C.c_void_p] + [C.c_void_p] * 5 + [C.c_int64]
    lib.orl_buffer_normalize_obs.argtypes = [C.c_void_p, C.c_float, C.c_void_p, C.c_void_p]
    lib.orl_buffer_sample.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_uint64] + [C.c_void_p] * 5
    lib.orl_buffer_size.argtypes = [C.c_void_p]
    lib.orl_buffer_size.restype = C.c_int64
    lib.orl_engine_attach_buffer.argtypes = [C.c_void_p, C.c_void_p]
    lib.orl_step.argtypes = [C.c_void_p, C.POINTER(OrlBatch), C.POINTER(OrlNoise), C.c_void_p]
    lib.orl_learn_n.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.POINTER(C.c_float)]
    lib.orl_health.argtypes = [C.c_void_p, C.c_void_p]
    lib.orl_health_check.argtypes = [C.c_void_p, C.c_void_p]
    lib.orl_health_clear.argtypes = [C.c_void_p]
    lib.orl_num_metrics.argtypes = [C.c_void_p]
    lib.orl_metric_name.argtypes = [C.c_void_p, C.c_int]
    lib.orl_metric_name.restype = C.c_char_p
    lib.orl_step_count.argtypes = [C.c_void_p]
    lib.orl_step_count.restype = C.c_int64
    lib.orl_debug_read.argtypes = [C.c_void_p, C.c_int, C.c_char_p, C.c_void_p, C.c_int64]
    lib.orl_debug_read.restype = C.c_int64
    lib.orl_debug_read_bits.argtypes = [C.c_void_p, C.c_int, C.c_char_p, C.c_void_p, C.c_int64]
    lib.orl_debug_read_bits.restype = C.c_int64
    lib.orl_debug_grads.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int64]
    lib.orl_debug_gemm.argtypes = [C.c_int] * 5 + [C.c_void_p] * 5 + [C.c_int, C.c_int]
    lib.orl_debug_gemm_time.argtypes = [C.c_int] * 8 + [C.POINTER(C.c_float)]
    lib.orl_profile_enable.argtypes = [C.c_void_p, C.c_int]
    lib.orl_profile_query.argtypes = [C.c_void_p, C.c_int, C.c_char_p, C.c_int, C.POINTER(C.c_double),
                                      C.POINTER(C.c_int64), C.POINTER(C.c_double), C.POINTER(C.c_double)]
    if path is None:
        _lib = lib
    return lib


def split_bits() -> int:
    """significand bits an operand carries at precision 1 (22: fp16 hi + lo planes; 16: the bf16-plane variant build)"""
    return int(load_library().orl_split_bits())


def last_error() -> str:
    return load_library().orl_last_error().decode()


def _check(rc: int, what: str):
    if rc != 0:
        raise RuntimeError(f"{what} failed: {last_error()}")


def default_config(algo: str, **over) -> OrlConfig:
    lib = load_library()
    cfg = OrlConfig()
    lib.orl_config_default(C.byref(cfg), ALGO_ID[algo])
    apply_config(cfg, over)
    return cfg


def apply_config(cfg: OrlConfig, over: Dict) -> None:
    names = {f[0] for f in OrlConfig._fields_}
    for k, v in over.items():
        if k == "hidden":
            cfg.n_hidden = len(v)
            for i, h in enumerate(v):
                cfg.hidden[i] = int(h)
        elif k in names:
            setattr(cfg, k, v)
        else:
            raise KeyError(f"unknown engine config field {k!r}")


def _f32(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float32)


class Engine:
    """Thin RAII wrapper over ``orl_engine*``."""

    def __init__(self, cfg: OrlConfig):
        self.lib = load_library()
        self.cfg = cfg
        self._h = C.c_void_p()
        _check(self.lib.orl_engine_create(C.byref(cfg), C.byref(self._h)), "orl_engine_create")
        self.n_runs = cfg.n_runs
        self.metric_names = [self.lib.orl_metric_name(self._h, i).decode() for i in range(self.lib.orl_num_metrics(self._h))]
        # a step that ran but raised a health flag: warn once per new flag (EngineHealthWarning), or raise when strict
        self.strict_health = os.environ.get("ORL_STRICT_HEALTH", "0") == "1"
        self._health_seen = 0

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self.lib.orl_engine_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- parameters ----
    def net_present(self, net: int) -> bool:
        return bool(self.lib.orl_net_present(self._h, net))

    def net_tensors(self, net: int):
        out = []
        for i in range(self.lib.orl_net_num_tensors(self._h, net)):
            name = C.create_string_buffer(128)
            off, ndim, shape = C.c_int64(), C.c_int32(), (C.c_int64 * 4)()
            _check(self.lib.orl_net_tensor(self._h, net, i, name, 128, C.byref(off), C.byref(ndim), shape), "orl_net_tensor")
            out.append((name.value.decode(), off.value, tuple(shape[k] for k in range(ndim.value))))
        return out

    def net_floats(self, net: int) -> int:
        return self.lib.orl_net_floats(self._h, net)

    def net_ptr(self, run: int, net: int) -> int:
        return self.lib.orl_net_ptr(self._h, run, net)

    def set_net(self, run: int, net: int, params: Dict[str, np.ndarray]):
        flat = np.zeros(self.net_floats(net), dtype=np.float32)
        seen = set()
        for name, off, shape in self.net_tensors(net):
            if name not in params:
                raise KeyError(f"missing parameter {name}")
            a = _f32(params[name])
            if a.size != int(np.prod(shape)):
                raise ValueError(f"{name}: expected shape {shape}, got {a.shape}")
            flat[off:off + a.size] = a.ravel()
            seen.add(name)
        _check(self.lib.orl_net_set(self._h, run, net, flat.ctypes.data, flat.size), "orl_net_set")

    def get_net(self, run: int, net: int) -> Dict[str, np.ndarray]:
        flat = np.empty(self.net_floats(net), dtype=np.float32)
        _check(self.lib.orl_net_get(self._h, run, net, flat.ctypes.data, flat.size), "orl_net_get")
        return {name: flat[off:off + int(np.prod(shape))].reshape(shape).copy() for name, off, shape in self.net_tensors(net)}

    def set_scalar(self, run: int, which: int, v: float):
        _check(self.lib.orl_scalar_set(self._h, run, which, float(v)), "orl_scalar_set")

    def get_scalar(self, run: int, which: int) -> float:
        v = C.c_float()
        _check(self.lib.orl_scalar_get(self._h, run, which, C.byref(v)), "orl_scalar_get")
        return v.value

    def set_lr(self, opt: int, lr: float):
        _check(self.lib.orl_set_lr(self._h, opt, float(lr)), "orl_set_lr")

    def reset_optimizers(self):
        _check(self.lib.orl_reset_optimizers(self._h), "orl_reset_optimizers")

    def adam_state(self, run: int, net: int):
        """(exp_avg, exp_avg_sq) of the net's Adam optimizer, flat in state_dict order."""
        n = self.net_floats(net)
        m, v = np.empty(n, np.float32), np.empty(n, np.float32)
        _check(self.lib.orl_adam_get(self._h, run, net, m.ctypes.data, v.ctypes.data, n), "orl_adam_get")
        return m, v

    def set_adam_state(self, run: int, net: int, m, v):
        m, v = _f32(m).ravel(), _f32(v).ravel()
        _check(self.lib.orl_adam_set(self._h, run, net, m.ctypes.data, v.ctypes.data, m.size), "orl_adam_set")

    def set_step_count(self, steps: int):
        _check(self.lib.orl_set_step_count(self._h, int(steps)), "orl_set_step_count")

    def trainable_nets(self) -> List[int]:
        return [n for n in range(NUM_NETS) if self.net_present(n) and n in (NET_ACTOR, NET_CRITIC1, NET_CRITIC2, NET_CRITIC_V, NET_VAE_ENC, NET_VAE_DEC)]

    def optimizer_state(self, run: int = 0) -> Dict:
        """Everything torch.optim state_dict()s would hold for this run: per-net Adam moments, the scalar optimizers, the step count."""
        st = {"step": self.step_count(), "adam": {n: self.adam_state(run, n) for n in self.trainable_nets()}, "scalars": {}}
        for w in ALL_SCALARS:
            st["scalars"][w] = self.get_scalar(run, w)
        return st

    def load_optimizer_state(self, st: Dict, run: int = 0):
        for n, (m, v) in st["adam"].items():
            self.set_adam_state(run, n, m, v)
        for w in ALL_SCALARS:
            if w in st["scalars"]:
                self.set_scalar(run, w, st["scalars"][w])
        self.set_step_count(st["step"])

    # ---- buffer ----
    def attach_buffer(self, buf: "DeviceBuffer"):
        _check(self.lib.orl_engine_attach_buffer(self._h, buf._h if buf is not None else None), "orl_engine_attach_buffer")
        self._buf = buf   # keep alive

    # ---- hot path ----
    def step(self, batch: Optional[Dict[str, np.ndarray]], noise: Optional[List[np.ndarray]], on_device=False) -> np.ndarray:
        """batch/noise: host arrays with a leading run dimension (or raw device pointers when on_device)."""
        keep = []
        bp = None
        if batch is not None:
            b = OrlBatch()
            for k in ("observations", "actions", "next_observations", "rewards", "terminals"):
                v = batch[k]
                if on_device:
                    setattr(b, k, int(v))
                else:
                    a = _f32(v)
                    keep.append(a)
                    setattr(b, k, a.ctypes.data)
            b.on_device = 1 if on_device else 0
            bp = C.byref(b)
        npz = None
        if noise is not None:
            n = OrlNoise()
            for i, v in enumerate(noise):
                if on_device:
                    n.slot[i] = int(v)
                else:
                    a = _f32(v)
                    keep.append(a)
                    n.slot[i] = a.ctypes.data
            n.on_device = 1 if on_device else 0
            npz = C.byref(n)
        m = np.zeros((self.n_runs, MAX_METRICS), dtype=np.float32)
        self._check_step(self.lib.orl_step(self._h, bp, npz, m.ctypes.data), "orl_step")
        return m[:, :len(self.metric_names)]

    def learn_n(self, n_steps: int):
        m = np.zeros((self.n_runs, MAX_METRICS), dtype=np.float32)
        ms = C.c_float()
        self._check_step(self.lib.orl_learn_n(self._h, n_steps, m.ctypes.data, C.byref(ms)), "orl_learn_n")
        return m[:, :len(self.metric_names)], ms.value

    # ---- health (include/orl_engine.h: ORL_HEALTH_*) ----
    def _check_step(self, rc: int, what: str):
        if rc == RC_UNHEALTHY:
            self._report_health(what)
        else:
            _check(rc, what)

    def _report_health(self, what: str):
        flags = int(np.bitwise_or.reduce(self.health()))
        if self.strict_health:
            raise EngineHealthError(f"{what}: {last_error()}")
        if flags & ~self._health_seen:
            warnings.warn(f"{what}: {last_error()}", EngineHealthWarning, stacklevel=3)
        self._health_seen |= flags

    def health(self) -> np.ndarray:
        """Sticky per-run flags (HEALTH_* bits) as of the last step / learn_n / health_check."""
        f = np.zeros(self.n_runs, dtype=np.uint32)
        if self.lib.orl_health(self._h, f.ctypes.data) < 0:
            raise RuntimeError(f"orl_health failed: {last_error()}")
        return f

    def health_check(self) -> np.ndarray:
        """Scans the last step's split-precision operands (inputs, stored hidden activations, parameters) for the fp16-plane range on
        top of what the steps themselves recorded; warns / raises like a step does.  One pass over the workspaces: per epoch, not per step."""
        f = np.zeros(self.n_runs, dtype=np.uint32)
        rc = self.lib.orl_health_check(self._h, f.ctypes.data)
        if rc < 0:
            raise RuntimeError(f"orl_health_check failed: {last_error()}")
        if rc:
            self._report_health("orl_health_check")
        return f

    def health_clear(self):
        _check(self.lib.orl_health_clear(self._h), "orl_health_clear")
        self._health_seen = 0

    def step_count(self) -> int:
        return self.lib.orl_step_count(self._h)

    def debug_read(self, run: int, name: str, cap: int = 1 << 22) -> np.ndarray:
        buf = np.empty(cap, dtype=np.float32)
        n = self.lib.orl_debug_read(self._h, run, name.encode(), buf.ctypes.data, cap)
        if n < 0:
            raise RuntimeError(f"orl_debug_read({name}) failed: {last_error()}")
        return buf[:n].copy()

    def debug_read_bits(self, run: int, name: str, cap: int = 1 << 24) -> np.ndarray:
        """packed ReLU-mask words of a hidden-activation workspace (uint32, [members * rows * width/32])"""
        buf = np.empty(cap, dtype=np.uint32)
        n = self.lib.orl_debug_read_bits(self._h, run, name.encode(), buf.ctypes.data, cap)
        if n < 0:
            raise RuntimeError(f"orl_debug_read_bits({name}) failed: {last_error()}")
        return buf[:n].copy()

    def debug_grads(self, run: int, net: int) -> Dict[str, np.ndarray]:
        """Gradients of the last step for every tensor of ``net`` (reference: ``param.grad`` before ``optimizer.step()``)."""
        flat = np.empty(self.net_floats(net), dtype=np.float32)
        _check(self.lib.orl_debug_grads(self._h, run, net, flat.ctypes.data, flat.size), "orl_debug_grads")
        return {name: flat[off:off + int(np.prod(shape))].reshape(shape).copy() for name, off, shape in self.net_tensors(net)}

    def profile_enable(self, on: bool):
        self.lib.orl_profile_enable(self._h, 1 if on else 0)

    def profile_table(self):
        rows = []
        i = 0
        while True:
            name = C.create_string_buffer(128)
            tot, cnt, fl, by = C.c_double(), C.c_int64(), C.c_double(), C.c_double()
            rc = self.lib.orl_profile_query(self._h, i, name, 128, C.byref(tot), C.byref(cnt), C.byref(fl), C.byref(by))
            if rc != 0:
                break
            rows.append(dict(name=name.value.decode(), total_ms=tot.value, launches=cnt.value, flops_per_launch=fl.value,
                             bytes_per_launch=by.value))
            i += 1
        return rows


class DeviceBuffer:
    """RAII wrapper over ``orl_buffer*``: the HBM-resident SoA replay store (buffer/buffer.py)."""

    def __init__(self, obs_dim: int, act_dim: int, device: int = 0):
        self.lib = load_library()
        self.obs_dim, self.act_dim, self.device = obs_dim, act_dim, device
        self._h = C.c_void_p()
        _check(self.lib.orl_buffer_create(obs_dim, act_dim, device, C.byref(self._h)), "orl_buffer_create")

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self.lib.orl_buffer_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def load(self, obs, act, next_obs, rew, term):
        obs, act, next_obs = _f32(obs), _f32(act), _f32(next_obs)
        rew, term = _f32(rew).ravel(), _f32(term).ravel()
        n = obs.shape[0]
        assert obs.shape == (n, self.obs_dim) and next_obs.shape == (n, self.obs_dim) and act.shape == (n, self.act_dim)
        assert rew.size == n and term.size == n
        _check(self.lib.orl_buffer_load(self._h, obs.ctypes.data, act.ctypes.data, next_obs.ctypes.data,
                                        rew.ctypes.data, term.ctypes.data, n), "orl_buffer_load")

    def size(self) -> int:
        return self.lib.orl_buffer_size(self._h)

    def normalize_obs(self, eps: float = 1e-3):
        mean = np.zeros(self.obs_dim, dtype=np.float32)
        std = np.zeros(self.obs_dim, dtype=np.float32)
        _check(self.lib.orl_buffer_normalize_obs(self._h, float(eps), mean.ctypes.data, std.ctypes.data), "orl_buffer_normalize_obs")
        return mean, std

    def sample_into(self, idx, batch: int, seed: int, obs_ptr: int, act_ptr: int, nobs_ptr: int, rew_ptr: int, term_ptr: int):
        """Gather into caller-owned device arrays (raw device pointers)."""
        p = None
        if idx is not None:
            idx = np.ascontiguousarray(idx, dtype=np.int64)
            assert idx.size == batch
            p = idx.ctypes.data
        _check(self.lib.orl_buffer_sample(self._h, p, batch, seed, obs_ptr, act_ptr, nobs_ptr, rew_ptr, term_ptr), "orl_buffer_sample")


def debug_gemm(cfg: int, mode: int, A, B, v0=None, v1=None, ksplit=1, precision=0, M=None, N=None, K=None) -> np.ndarray:
    """Kernel unit-test entry (orl_debug_gemm)."""
    lib = load_library()
    A, B = _f32(A), _f32(B)
    wg = mode in (2, 4)
    out = np.zeros(M * (N + 1) if wg else M * N, dtype=np.float32)
    p0 = _f32(v0) if v0 is not None else None
    p1 = _f32(v1) if v1 is not None else None
    _check(lib.orl_debug_gemm(cfg, mode, M, N, K, A.ctypes.data, B.ctypes.data,
                              p0.ctypes.data if p0 is not None else None, p1.ctypes.data if p1 is not None else None,
                              out.ctypes.data, ksplit, precision), "orl_debug_gemm")
    return out

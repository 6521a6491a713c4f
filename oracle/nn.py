"""Shared numpy (fp32) building blocks of the oracle.  TEST INFRASTRUCTURE ONLY.

Everything is explicit: forward, hand-derived backward, PyTorch-semantics Adam
and Polyak.  No autograd.  Each helper cites the reference lines it restates
(paths relative to /root/reference).

``set_matmul_mode`` lets tests emulate the reduced-precision MFMA schemes of the
HIP engine on the CPU ("fp32" exact, "bf16x3" = hi/lo split with 3 products,
"bf16" = plain bf16 operands, fp32 accumulate).
"""
from __future__ import annotations

import math
from collections import OrderedDict
from typing import Dict, List, Sequence, Tuple

import numpy as np

f32 = np.float32
LOG_SQRT_2PI = f32(math.log(math.sqrt(2.0 * math.pi)))

_MATMUL_MODE = "fp32"


def set_matmul_mode(mode: str) -> None:
    global _MATMUL_MODE
    assert mode in ("fp32", "bf16x3", "bf16")
    _MATMUL_MODE = mode


def get_matmul_mode() -> str:
    return _MATMUL_MODE


def bf16_round(x: np.ndarray) -> np.ndarray:
    """Round-to-nearest-even fp32 -> bf16, returned widened back to fp32."""
    u = np.ascontiguousarray(x, dtype=f32).view(np.uint32)
    r = (u + np.uint32(0x7FFF) + ((u >> np.uint32(16)) & np.uint32(1))) & np.uint32(0xFFFF0000)
    return r.view(f32)


def bf16_split(x: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    hi = bf16_round(x)
    lo = bf16_round(np.asarray(x, dtype=f32) - hi)
    return hi, lo


def mm(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    """a[M,K] @ b[K,N] in the active matmul mode (fp32 accumulate always)."""
    a = np.asarray(a, dtype=f32)
    b = np.asarray(b, dtype=f32)
    if _MATMUL_MODE == "fp32":
        return a @ b
    if _MATMUL_MODE == "bf16":
        return bf16_round(a) @ bf16_round(b)
    ah, al = bf16_split(a)
    bh, bl = bf16_split(b)
    return (al @ bh + ah @ bl) + ah @ bh


# ----------------------------------------------------------------------------
# MLP backbone: [Linear, ReLU] x L            (nets/mlp.py:9-33, y = x W^T + b)
# ----------------------------------------------------------------------------

def backbone_layers(net: Dict[str, np.ndarray], prefix: str = "backbone.model.") -> Tuple[List[np.ndarray], List[np.ndarray]]:
    """Collect (W, b) of the Linear layers of an MLP backbone in order.
    nn.Sequential indices are 0,2,4,... (Linear, ReLU alternating), or 0,3,6,... when every ReLU is followed by nn.Dropout
    (nets/mlp.py:20-23)."""
    idx = sorted(int(k[len(prefix):-len(".weight")]) for k in net if k.startswith(prefix) and k.endswith(".weight"))
    return [net[f"{prefix}{i}.weight"] for i in idx], [net[f"{prefix}{i}.bias"] for i in idx]


def backbone_indices(net: Dict[str, np.ndarray], prefix: str = "backbone.model.") -> List[int]:
    return sorted(int(k[len(prefix):-len(".weight")]) for k in net if k.startswith(prefix) and k.endswith(".weight"))


def mlp_fwd(x: np.ndarray, Ws: Sequence[np.ndarray], bs: Sequence[np.ndarray], drop=None) -> List[np.ndarray]:
    """Returns [x, h1, ..., hL] with h_l = relu(h_{l-1} W_l^T + b_l).
    drop = (p, [keep mask per layer]): nn.Dropout(p) in training mode behind every ReLU (nets/mlp.py:20-23): h_l *= mask_l / (1 - p)."""
    hs = [np.asarray(x, dtype=f32)]
    for l, (W, b) in enumerate(zip(Ws, bs)):
        z = mm(hs[-1], W.T) + b
        h = np.maximum(z, f32(0))
        if drop is not None:
            h = (h * (np.asarray(drop[1][l], f32) / f32(1.0 - drop[0]))).astype(f32)      # ATen: input * (bernoulli(1 - p) / (1 - p))
        hs.append(h)
    return hs


def mlp_bwd(hs: Sequence[np.ndarray], Ws: Sequence[np.ndarray], dh: np.ndarray,
            need_dx: bool, need_dw: bool = True, drop_p=None):
    """Backward through the backbone.  dh = dLoss/dh_L.
    ReLU gradient is 1[h > 0] (threshold_backward, strict).  drop_p: the forward ran with nn.Dropout(p): hs[l] is the dropped activation, so
    1[hs > 0] = ReLU mask AND keep mask, and the gradient carries the forward's 1 / (1 - p)."""
    L = len(Ws)
    dWs: List[np.ndarray] = [None] * L
    dbs: List[np.ndarray] = [None] * L
    dx = None
    for l in reversed(range(L)):
        dz = dh * (hs[l + 1] > 0)
        if drop_p is not None:
            dz = (dz * (f32(1) / f32(1.0 - drop_p))).astype(f32)
        if need_dw:
            dWs[l] = mm(dz.T, hs[l])
            dbs[l] = dz.sum(axis=0, dtype=f32)
        if l > 0 or need_dx:
            dh = mm(dz, Ws[l])
            if l == 0:
                dx = dh
    return dWs, dbs, dx


# ----------------------------------------------------------------------------
# Critic: cat(obs, act) -> backbone -> Linear(H, 1)   (critic_module.py:17-28)
# ----------------------------------------------------------------------------

def critic_fwd(net: Dict[str, np.ndarray], obs: np.ndarray, act: np.ndarray | None = None):
    x = obs if act is None else np.concatenate([obs, act], axis=1)
    Ws, bs = backbone_layers(net)
    hs = mlp_fwd(x, Ws, bs)
    q = mm(hs[-1], net["last.weight"].T) + net["last.bias"]
    return q.astype(f32), hs


def critic_bwd(net: Dict[str, np.ndarray], hs, dq: np.ndarray, need_dx: bool, need_dw: bool = True):
    """dq: (rows,1) = dLoss/dq.  Returns (grads dict or None, dx or None)."""
    Ws, _ = backbone_layers(net)
    grads = OrderedDict()
    if need_dw:
        grads["last.weight"] = mm(dq.T, hs[-1])
        grads["last.bias"] = dq.sum(axis=0, dtype=f32)
    dh = mm(dq, net["last.weight"])
    dWs, dbs, dx = mlp_bwd(hs, Ws, dh, need_dx, need_dw)
    if need_dw:
        for l, (dW, db) in enumerate(zip(dWs, dbs)):
            grads[f"backbone.model.{2 * l}.weight"] = dW
            grads[f"backbone.model.{2 * l}.bias"] = db
    return (grads if need_dw else None), dx


# ----------------------------------------------------------------------------
# Tanh-Gaussian actor head (CQL / EDAC)
#   dist_module.py:117-127 (unbounded=True, conditioned_sigma=True), :17-42
# ----------------------------------------------------------------------------

SIGMA_MIN, SIGMA_MAX = f32(-5.0), f32(2.0)
TANH_EPS = f32(1e-6)


def tanh_gauss_fwd(net: Dict[str, np.ndarray], obs: np.ndarray, eps: np.ndarray | None):
    """actforward (sac.py:66-77).  eps=None -> deterministic mode().
    Returns (a, logp(rows,1), cache)."""
    Ws, bs = backbone_layers(net)
    hs = mlp_fwd(obs, Ws, bs)
    h = hs[-1]
    mu = mm(h, net["dist_net.mu.weight"].T) + net["dist_net.mu.bias"]
    ls_raw = mm(h, net["dist_net.sigma.weight"].T) + net["dist_net.sigma.bias"]
    ls = np.clip(ls_raw, SIGMA_MIN, SIGMA_MAX)
    sigma = np.exp(ls)
    if eps is None:
        u = mu
    else:
        u = mu + sigma * np.asarray(eps, dtype=f32)
    a = np.tanh(u)
    # Normal.log_prob(u) = -(u-mu)^2/(2 var) - log(sigma) - log(sqrt(2 pi))
    var = sigma * sigma
    lp = -((u - mu) ** 2) / (f32(2) * var) - ls - LOG_SQRT_2PI
    logp = lp.sum(axis=1, keepdims=True, dtype=f32)
    logp = logp - np.log((f32(1) - a * a) + TANH_EPS).sum(axis=1, keepdims=True, dtype=f32)
    cache = dict(hs=hs, mu=mu, ls_raw=ls_raw, sigma=sigma, u=u, a=a, eps=eps)
    return a.astype(f32), logp.astype(f32), cache


def tanh_gauss_bwd(net: Dict[str, np.ndarray], cache, da: np.ndarray, dlogp: np.ndarray):
    """Backward of (a, logp) = actforward(obs) through the rsample path.
    da (rows,A) = dL/da, dlogp (rows,1) = dL/dlogp.

    With u = mu + sigma*eps the Gaussian quadratic term is the constant -eps^2/2,
    so logp depends on (mu, ls) only through -ls and the tanh Jacobian."""
    a, sigma, eps, ls_raw = cache["a"], cache["sigma"], cache["eps"], cache["ls_raw"]
    one_m = f32(1) - a * a
    t = f32(2) * a * one_m / (one_m + TANH_EPS)          # d logp / d u
    du = da * one_m + dlogp * t
    dmu = du
    dls = du * sigma * eps - dlogp                       # -log sigma term
    gate = (ls_raw >= SIGMA_MIN) & (ls_raw <= SIGMA_MAX)  # clamp backward (inclusive)
    dls_raw = dls * gate
    h = cache["hs"][-1]
    grads = OrderedDict()
    grads["dist_net.mu.weight"] = mm(dmu.T, h)
    grads["dist_net.mu.bias"] = dmu.sum(axis=0, dtype=f32)
    grads["dist_net.sigma.weight"] = mm(dls_raw.T, h)
    grads["dist_net.sigma.bias"] = dls_raw.sum(axis=0, dtype=f32)
    dh = mm(dmu, net["dist_net.mu.weight"]) + mm(dls_raw, net["dist_net.sigma.weight"])
    Ws, _ = backbone_layers(net)
    dWs, dbs, _ = mlp_bwd(cache["hs"], Ws, dh, need_dx=False)
    for l, (dW, db) in enumerate(zip(dWs, dbs)):
        grads[f"backbone.model.{2 * l}.weight"] = dW
        grads[f"backbone.model.{2 * l}.bias"] = db
    return grads


def min2_grad(q1: np.ndarray, q2: np.ndarray, g: np.ndarray):
    """Backward of torch.min(q1, q2) (elementwise minimum): ties split evenly."""
    g1 = np.where(q1 < q2, g, np.where(q1 == q2, g * f32(0.5), f32(0))).astype(f32)
    g2 = np.where(q2 < q1, g, np.where(q1 == q2, g * f32(0.5), f32(0))).astype(f32)
    return g1, g2


# ----------------------------------------------------------------------------
# Adam (torch.optim.Adam single-tensor path, defaults) and Polyak (sac.py:60-64)
# ----------------------------------------------------------------------------

def adam_init(params: Dict[str, np.ndarray]):
    return dict(step=0, m=OrderedDict(), v=OrderedDict())


def adam_step(params: Dict[str, np.ndarray], grads: Dict[str, np.ndarray], opt: dict, lr: float,
              betas=(0.9, 0.999), eps: float = 1e-8) -> None:
    """In-place PyTorch-semantics Adam.  Parameters absent from ``grads``
    (grad is None in the reference) are skipped and get no state."""
    b1, b2 = betas
    opt["step"] += 1
    t = opt["step"]
    bc1 = 1.0 - b1 ** t
    bc2 = 1.0 - b2 ** t
    step_size = f32(lr / bc1)
    bc2_sqrt = f32(math.sqrt(bc2))
    for k, g in grads.items():
        p = params[k]
        g = np.asarray(g, dtype=f32).reshape(p.shape)
        if k not in opt["m"]:
            opt["m"][k] = np.zeros_like(p)
            opt["v"][k] = np.zeros_like(p)
        m, v = opt["m"][k], opt["v"][k]
        m += (g - m) * f32(1.0 - b1)                       # exp_avg.lerp_(grad, 1-beta1)
        v *= f32(b2)
        v += f32(1.0 - b2) * g * g                         # addcmul_
        denom = np.sqrt(v) / bc2_sqrt + f32(eps)
        p -= step_size * (m / denom)                       # addcdiv_(value=-step_size)


def polyak(old: Dict[str, np.ndarray], new: Dict[str, np.ndarray], tau: float) -> None:
    for k in old:
        old[k][...] = old[k] * f32(1.0 - tau) + new[k] * f32(tau)


def copy_net(net: Dict[str, np.ndarray]) -> "OrderedDict[str, np.ndarray]":
    return OrderedDict((k, np.array(v, dtype=f32, copy=True)) for k, v in net.items())

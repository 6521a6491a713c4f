"""Synthetic workloads of bench.py: D4RL-shaped replay buffers and engines of the four hot-path algorithms with
the reference launch scripts' hyper-parameters and initialisation schemes.  Pure numpy + the C-ABI binding: nothing here imports
``tests/`` or ``oracle/`` (the timed process must not depend on the checker).

Reference for shapes / hyper-parameters / initialisation:
  CQL   run_example/run_cql.py:26-56, 80-104     halfcheetah-medium-v2: obs 17 / act 6, 1 000 000 transitions; north_star: hidden [256, 256]
  IQL   run_example/run_iql.py:25-46, 105-133    hopper-medium-replay-v2: obs 11 / act 3, 400 000; orthogonal(sqrt 2) weights, zero biases
  TD3BC run_example/run_td3bc.py:28-113          halfcheetah-medium-v2, normalised observations
  EDAC  run_example/run_edac.py:24-31, 35-128    walker2d-medium-expert-v2: obs 17 / act 6, 2 000 000; K = 10, eta = 5, hidden [256] * 3,
        trunc-normal ensemble weights (nets/ensemble_linear.py:21-26), biases 0.1, last layer U(+-3e-3)
Engine-side defaults of every other field are the scripts' ``get_args()`` values (``orl_config_default``).
"""
from __future__ import annotations

import numpy as np

WORKLOADS = {
    "cql": dict(task="halfcheetah-medium-v2", obs=17, act=6, n=1_000_000, hidden=[256, 256],
                over=dict(num_repeat_actions=10, target_entropy=-6.0), gflop=None),
    # the reference CLI's own default depth (run_cql.py:31: hidden [256, 256, 256]); SURVEY 8(d): 15.00 GFLOP per gradient step
    "cql_h3": dict(algo="cql", task="halfcheetah-medium-v2", obs=17, act=6, n=1_000_000, hidden=[256, 256, 256],
                   over=dict(num_repeat_actions=10, target_entropy=-6.0), gflop=15.0),
    "iql": dict(task="hopper-medium-replay-v2", obs=11, act=3, n=400_000, hidden=[256, 256],
                over=dict(expectile=0.7, iql_temperature=3.0), gflop=0.559),
    "td3bc": dict(task="halfcheetah-medium-v2", obs=17, act=6, n=1_000_000, hidden=[256, 256],
                  over=dict(update_actor_freq=2, td3bc_alpha=2.5), gflop=0.415),
    "edac": dict(task="walker2d-medium-expert-v2", obs=17, act=6, n=2_000_000, hidden=[256, 256, 256],
                 over=dict(num_critics=10, eta=5.0, deterministic_backup=0, target_entropy=-6.0), gflop=6.57),
}
BATCH = 256
# (target net id, source net id) pairs: targets start as deep copies (sac.py:29-33, td3.py:30-36, iql.py:33-36)
_TARGET_OF = {3: 1, 4: 2, 6: 0}


def make_dataset(seed, n, od, ad):
    """obs, next_obs ~ N(0, 1); actions = tanh(N(0, 1)); rewards ~ N(0, 1); terminals ~ Bernoulli(0.01)"""
    rng = np.random.RandomState(seed)
    return dict(
        obs=rng.standard_normal((n, od)).astype(np.float32),
        act=np.tanh(rng.standard_normal((n, ad))).astype(np.float32),
        nobs=rng.standard_normal((n, od)).astype(np.float32),
        rew=rng.standard_normal(n).astype(np.float32),
        term=(rng.uniform(size=n) < 0.01).astype(np.float32),
    )


def _uniform(rng, shape, bound):
    return rng.uniform(-bound, bound, size=shape).astype(np.float32)


def _orthogonal(rng, shape, gain):
    rows, cols = shape
    a = rng.standard_normal((max(rows, cols), min(rows, cols)))
    q, r = np.linalg.qr(a)
    q = q * np.sign(np.diag(r))
    if rows < cols:
        q = q.T
    return (gain * q[:rows, :cols]).astype(np.float32)


def _trunc_normal(rng, shape, std):
    x = rng.standard_normal(shape)
    bad = np.abs(x * std) > 2.0                      # trunc_normal_(a = -2, b = 2) in absolute units (ensemble_linear.py:23)
    while bad.any():
        x[bad] = rng.standard_normal(int(bad.sum()))
        bad = np.abs(x * std) > 2.0
    return (x * std).astype(np.float32)


def init_net(eng, run, net, rng, algo):
    """initial parameters of one net in state_dict order, by the scheme of the algorithm's launch script"""
    tensors = eng.net_tensors(net)
    names = [t[0] for t in tensors]
    last_w = [n for n in names if n.endswith("weight")][-1]
    p = {}
    for name, _, shape in tensors:
        if len(shape) == 3:                              # EnsembleLinear: (K, in, out) weights, (K, 1, out) biases
            if name.endswith("weight"):
                p[name] = _uniform(rng, shape, 3e-3) if name == last_w else _trunc_normal(rng, shape, 1.0 / (2.0 * np.sqrt(shape[1])))
            else:
                p[name] = _uniform(rng, shape, 3e-3) if name == last_w.replace("weight", "bias") else np.full(shape, 0.1, np.float32)
        elif name.endswith("sigma_param"):
            p[name] = np.zeros(shape, np.float32)
        elif algo == "iql":                              # run_iql.py:121-125
            p[name] = _orthogonal(rng, shape, np.sqrt(2.0)) if name.endswith("weight") else np.zeros(shape, np.float32)
        else:                                            # nn.Linear default: U(+-1/sqrt(fan_in)) for weight and bias
            fan_in = shape[-1] if name.endswith("weight") else dict((n, s) for n, _, s in tensors)[name.replace("bias", "weight")][-1]
            p[name] = _uniform(rng, shape, 1.0 / np.sqrt(fan_in))
    eng.set_net(run, net, p)
    return p


def make_engine(algo, n_runs, precision, device, seed, hidden=None, ws_one_round=0, **over):
    """engine of one algorithm at its BASELINE shape, every run initialised independently by the launch script's scheme"""
    from offlinerlkit import _engine
    w = WORKLOADS[algo]
    cfg = dict(obs_dim=w["obs"], act_dim=w["act"], hidden=list(hidden or w["hidden"]), batch_size=BATCH, n_runs=n_runs, device=device,
               precision=precision, seed=1234 + 7919 * seed, ws_one_round=int(ws_one_round))
    cfg.update(w["over"])
    cfg.update(over)
    algo = w.get("algo", algo)                       # WORKLOADS key -> engine algorithm
    eng = _engine.Engine(_engine.default_config(algo, **cfg))
    for r in range(n_runs):
        rng = np.random.RandomState(1000 + seed * 4096 + r)
        src = {}
        for net in range(_engine.NUM_NETS):
            if eng.net_present(net) and net not in _TARGET_OF:
                src[net] = init_net(eng, r, net, rng, algo)
        for tgt, s in _TARGET_OF.items():
            if eng.net_present(tgt):
                eng.set_net(r, tgt, src[s])
    return eng


def workload_string(algo, n_runs, engines=1, hidden=None):
    w = WORKLOADS[algo]
    extra = {"cql": "10 repeat actions, auto-alpha", "cql_h3": "10 repeat actions, auto-alpha (the reference CLI's default depth)", "iql": "expectile 0.7, temperature 3.0", "td3bc": "policy noise 0.2, actor every 2nd step",
             "edac": "10 critics, eta 5.0, auto-alpha"}[algo]
    return (f"{w.get('algo', algo).upper()} {w['task']} shape: obs{w['obs']}/act{w['act']}, batch {BATCH}, MLP {list(hidden or w['hidden'])}, {extra}, "
            f"{w['n']} synthetic transitions, device sampling+noise, {engines} engine(s) x {n_runs} run(s)")

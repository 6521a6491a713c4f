#!/usr/bin/env python3
"""bench.py — gradient-steps/sec of the CQL policy.learn() hot path on MI355X.

Workload (BASELINE.json configs[1]): CQL, halfcheetah-medium-v2-shaped synthetic replay buffer
(N=1e6 transitions, obs 17, act 6), batch 256, critics/actor MLP [256,256], 10 repeated actions,
auto-alpha, no Lagrange.  One "step" = one engine step = one policy.learn() update for every run the
engine carries (--runs-per-gpu independent seeds batched through the same kernel launches), including
on-device index sampling, replay gather and noise generation.  value = gradient steps of all runs on all
ranks / wall time (max over ranks), inputs resident in HBM before the timed region.

Contract: python bench.py --gpus N --steps K --warmup W.  N > 1: either launched by the driver through
torch.distributed.run (RANK / LOCAL_RANK / WORLD_SIZE in the environment), or — when those are absent — this
process starts `python -m torch.distributed.run --nproc-per-node N` on itself BEFORE touching the GPU and relays
rank 0's JSON line.  One rank per GPU; independent seeds per rank (replicas only) with one RCCL
all_gather of the per-run metric means at the end, as MFPolicyTrainer would log per epoch.

The timed block of K steps (barrier + synchronize on both sides, max over ranks) is repeated (>= 5 blocks and
>= 2 s, --min-reps / --min-seconds); `value` is the median block, every block is listed in `reps`.
Side records of the same run (rank 0, N = 1 only): `value_fp32` / `fp32` (exact-fp32 MFMA, same workload, one engine x 128 runs),
`value_fp32_class` / `fp32_class` (precision 2: three fp16 planes per operand in the critic launches, fp32 MFMA elsewhere; same geometry), `target`
(both over the north_star's 50k), `by_runs` (runs per GPU 1 .. 192), `other_configs` (TD3BC, IQL = BASELINE configs[2], EDAC =
configs[3]; 128 runs each), `api` (a fused MFPolicyTrainer epoch through offlinerlkit.policy.CQLPolicy / ReplayBuffer with the split
products selected), `api_default` (the same epoch with NO engine options set: what a user of the reference's API gets untouched -- one run,
exact fp32), `config5_per_gpu` (one engine x 8 seeds on the halfcheetah and the hopper shape: what each rank of BASELINE configs[4] runs),
`roofline` (+ `traffic_source`), `cpu_baseline` (the numpy port on 1 / 16 / 64 threads, each labelled with the threads it used).
`--preset config5` (BASELINE configs[4]: 8 tasks x 8 seeds): one engine x 8 runs per GPU, rank r on D4RL task r % 8 with that task's
observation / action widths and dataset size (`config.tasks_by_rank`).  `--rccl-check` opens a world-size-1 nccl group and runs the path's collectives
on device tensors (tests/test_gpu_rccl.py).  Workload construction shared with the tools lives in bench_workloads.py.
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "offlinerl-kit_amd")):       # (the checker's paths -- tests/, oracle/ -- are added by cpu_baseline() only)
    if p not in sys.path:
        sys.path.insert(0, p)

import bench_workloads as bw

OBS, ACT, HIDDEN, BATCH, NREP = 17, 6, [256, 256], 256, 10
TARGET_STEPS_PER_S = 50_000.0       # BASELINE.json north_star: >= 50k CQL gradient-steps/s on one MI355X
# MI355X_MICROARCH.md: fp32 MFMA dense peak 157.3 TFLOP/s; bf16 / fp16 MFMA dense peak ~2500 TFLOP/s.  precision=1 spends three
# 16-bit MFMAs per fp32-equivalent product, so its ceiling for ALGORITHMIC flops is 2500/3.
# precision=2: six 16-bit MFMAs per fp32-equivalent product in the launches that carry the matrix work (three planes per operand): 2500/6.
PEAK_TFLOPS = {0: 157.3, 1: 2500.0 / 3.0, 2: 2500.0 / 6.0}


def dtype_string(precision):
    if precision == 0:
        return "f32 (v_mfma_f32_16x16x4_f32: the reference's arithmetic)"
    if precision == 2:
        return ("f32 storage / accumulate, fp32-class products: in the many-row critic launches every operand = THREE fp16 planes (hi + mid + lo = 33 "
                "significand bits: an fp32 operand is represented exactly) and the six products hi*hi, hi*mid, mid*hi, hi*lo, mid*mid, lo*hi on "
                "v_mfma_f32_16x16x32_f16 drop only terms below 2^-33 of a product; every other launch on v_mfma_f32_16x16x4_f32.  Held to the exact-fp32 "
                "bars of the parity tests (tests/test_gpu_cql.py: losses 1e-4 vs the reference fixtures, per-element parameter bars; "
                "tests/test_gpu_grads.py / test_gpu_backward_f64.py: the precision-0 constants)")
    from offlinerlkit import _engine
    if _engine.split_bits() >= 22:
        return ("f32 storage / accumulate; products on split-fp16 MFMA (every operand = fp16 hi + fp16 lo plane = 22 significand bits, power-of-two "
                "operand scales folded back into the fp32 accumulators, hi*hi + hi*lo + lo*hi on v_mfma_f32_16x16x32_f16; the reference's fp32 has 24): "
                "losses and Q-values meet the 1e-4 parity gate (tests/test_gpu_cql.py), gradients agree with the fp32 oracle to 1e-6 .. 2.5e-4 of the "
                "tensor scale and are componentwise backward-stable at 2^-20 (tests/test_gpu_grads.py, test_gpu_backward_f64.py); the exact-fp32 "
                "figure of the same run is `value_fp32` / `fp32`")
    return ("f32 storage / accumulate; products on split-bf16 MFMA (bf16-plane variant build: operands = bf16 hi + bf16 lo = 16 significand bits); "
            "the exact-fp32 figure of the same run is `value_fp32` / `fp32`")
PRESETS = {
    # BASELINE config 5: 8 seeds x 8 tasks, one task (its own synthetic buffer) and its 8 seeds per GPU, one engine
    "config5": dict(runs_per_gpu=8, engines_per_gpu=1),
}
# ... and its 8 D4RL-mujoco tasks, one per GPU (rank r trains task r % 8): observation / action widths and transition counts of the D4RL v2
# datasets (halfcheetah / walker2d: obs 17, act 6; hopper: obs 11, act 3 -- the critic input is 14 columns wide there, the actor head 6)
CONFIG5_TASKS = [
    ("halfcheetah-medium-v2", 17, 6, 1_000_000), ("hopper-medium-v2", 11, 3, 1_000_000), ("walker2d-medium-v2", 17, 6, 1_000_000),
    ("halfcheetah-medium-replay-v2", 17, 6, 202_000), ("hopper-medium-replay-v2", 11, 3, 402_000), ("walker2d-medium-replay-v2", 17, 6, 302_000),
    ("halfcheetah-medium-expert-v2", 17, 6, 2_000_000), ("hopper-medium-expert-v2", 11, 3, 2_000_000),
]


# ---------------------------------------------------------------------------------------------------------------
# N > 1 without a launcher: start one child rank per GPU.  Nothing in this function (or before it) touches the GPU.
# ---------------------------------------------------------------------------------------------------------------
def free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def child_command(n_gpus: int, port: int, argv) -> list:
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}", "--master-addr", "127.0.0.1",
            "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def launch_children(n_gpus: int, argv) -> int:
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    return subprocess.call(child_command(n_gpus, free_port(), argv), env=env)


# ---------------------------------------------------------------------------------------------------------------
# workload helpers
# ---------------------------------------------------------------------------------------------------------------
def mlp_flops_per_row(in_dim, hidden, out_dim):
    f, d = 0, in_dim
    for h in hidden:
        f += 2 * d * h
        d = h
    return f + 2 * d * out_dim


def cql_algorithmic_flops(B=BATCH, N=NREP, od=OBS, ad=ACT, hidden=HIDDEN):
    """minimal necessary work of one CQL gradient step (fwd = 2*in*out per row per layer,
    bwd = wgrad + dgrad, first-layer dgrad only where input grads are needed)."""
    a_f = mlp_flops_per_row(od, hidden, 2 * ad)
    c_f = mlp_flops_per_row(od + ad, hidden, 1)
    first_c = 2 * (od + ad) * hidden[0]
    first_a = 2 * od * hidden[0]
    rows_actor_fwd = B + B + 2 * B * N
    rows_critic = B + 3 * B * N
    fl = a_f * rows_actor_fwd
    fl += 2 * c_f * (B + rows_critic)            # both critics: actor-phase rows + CQL rows
    fl += 2 * c_f * B                            # target critics
    fl += 2 * (2 * c_f - first_c) * rows_critic  # critic bwd: wgrad everywhere + dgrad except first layer
    fl += 2 * c_f * B                            # critic dgrad for the actor loss (incl. first layer -> action grads), one selected critic per row ~ 1x
    fl += (2 * a_f - first_a) * B                # actor bwd
    return float(fl)


def make_dataset(seed, n=1_000_000, od=OBS, ad=ACT):
    return bw.make_dataset(seed, n, od, ad)


def make_cql_engines(E, R, device, precision, seed0, buf, one_round=None, od=OBS, ad=ACT):
    """E engines x R runs of the headline workload on one GPU.  With several engines per GPU each engine's weight-stationary launches stay on
    CUs / nets workgroups per net (one round): the CUs they leave idle are where the other engine's kernels run (orl_config::ws_one_round)."""
    engines = []
    for e in range(E):
        g = bw.make_engine("cql", R, precision, device, seed0 * E + e, ws_one_round=(E > 1) if one_round is None else one_round,
                           obs_dim=od, act_dim=ad, target_entropy=-float(ad))
        g.attach_buffer(buf)                              # all engines of a GPU sample the same HBM-resident dataset
        engines.append(g)
    return engines


def learn_all(engines, n):
    """every engine advances n gradient steps (all its runs); one host thread per engine, each on its own HIP stream"""
    import threading
    E = len(engines)
    res = [None] * E

    def work(i):
        res[i] = engines[i].learn_n(n)
    if E == 1:
        work(0)
    else:
        ths = [threading.Thread(target=work, args=(i,)) for i in range(E)]
        for t in ths:
            t.start()
        for t in ths:
            t.join()
    return res


def timed_rate(engines, steps, min_seconds, min_reps=1, max_reps=40):
    """[(seconds, steps)] blocks of `steps` engine steps each, host wall clock around a device-synchronising call"""
    import torch
    reps = []
    t_all = time.perf_counter()
    while len(reps) < min_reps or (time.perf_counter() - t_all < min_seconds and len(reps) < max_reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        learn_all(engines, steps)
        torch.cuda.synchronize()
        reps.append(time.perf_counter() - t0)
    return reps


# ---------------------------------------------------------------------------------------------------------------
# CPU baselines (rank 0, N = 1): the torch-CPU counterpart at up to 64 threads, 16 threads and 1 thread, and the numpy oracle
# ---------------------------------------------------------------------------------------------------------------
def _checker_paths():
    for p in (os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
        if p not in sys.path:
            sys.path.insert(0, p)


def _cql_cpu_state():
    _checker_paths()
    import synth
    from helpers import clone_state
    rng = np.random.RandomState(5)
    st = dict(actor=synth.make_tanh_actor(rng, OBS, ACT, HIDDEN), critic1=synth.make_critic(rng, OBS + ACT, HIDDEN),
              critic2=synth.make_critic(rng, OBS + ACT, HIDDEN))
    st["critic1_old"] = synth.make_critic(rng, OBS + ACT, HIDDEN)
    st["critic2_old"] = synth.make_critic(rng, OBS + ACT, HIDDEN)
    st["log_alpha"] = np.zeros(1, np.float32); st["cql_log_alpha"] = np.zeros(1, np.float32)
    return clone_state(st), rng


def cpu_baseline(seconds=24.0):
    """the build's PyTorch-CPU counterpart of CQLPolicy.learn (oracle/torch_cql.py, pinned against the
    reference fixtures in tests/test_oracle_golden.py) on this box's host cores -- min(cores available, 64) threads, 16 threads and one
    thread -- plus the numpy oracle; same synthetic workload, a bounded sample of steps each."""
    _checker_paths()
    import synth
    import torch
    from oracle import cql as ocql
    from oracle.torch_cql import TorchCQL
    ds = make_dataset(0, 100_000)
    cfg = ocql.default_cfg(OBS, ACT)
    ncpu = os.cpu_count() or 1
    try:
        avail = len(os.sched_getaffinity(0))
    except Exception:
        avail = ncpu

    def sample(rng):
        idx = np.random.randint(0, 100_000, size=BATCH)          # buffer.py:98
        batch = dict(observations=ds["obs"][idx], actions=ds["act"][idx], next_observations=ds["nobs"][idx],
                     rewards=ds["rew"][idx], terminals=ds["term"][idx])
        return batch, synth.make_cql_noise(rng, BATCH, NREP, ACT)

    def time_it(step, per):
        step(); step()
        n, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < per:
            step(); n += 1
        return n / (time.perf_counter() - t0), n

    per = seconds / 4
    out = {}
    old_threads = torch.get_num_threads()
    for label, th in (("t64", max(1, min(avail, 64))), ("t16", min(16, avail)), ("one", 1)):      # (more than 64 threads only slow a batch-256 step down)
        torch.set_num_threads(th)
        st, rng = _cql_cpu_state()
        pol = TorchCQL(st, cfg)
        rate, n = time_it(lambda: pol.learn(*sample(rng)), per)
        out[label] = dict(value=rate, threads=th, steps=n)
    torch.set_num_threads(old_threads)
    st, rng = _cql_cpu_state()
    ocql.init_opt(st)
    try:
        from threadpoolctl import threadpool_limits
        ctx = threadpool_limits(limits=min(16, avail))
    except Exception:
        ctx = None
    rate, n = time_it(lambda: ocql.learn(st, cfg, *sample(rng)), per)
    if ctx is not None and hasattr(ctx, "unregister"):
        ctx.unregister()
    best = max(("t64", "t16"), key=lambda k: out[k]["value"])
    return dict(value=out[best]["value"], unit="gradient-steps/s", cores=out[best]["threads"], kind="port",
                sample=f"{out[best]['steps']} CQL learn() steps (batch 256, ~{per:.0f} s) of the PyTorch-CPU counterpart (oracle/torch_cql.py: stock "
                       f"autograd + torch.optim.Adam) at {out[best]['threads']} torch threads; host reports {ncpu} cpus, {avail} available to the process",
                nproc=ncpu, cpus_available=avail,
                torch_up_to_64_threads=out["t64"], torch_16_threads=out["t16"], torch_1_thread=out["one"],
                numpy_oracle=dict(value=rate, blas_threads=min(16, avail), steps=n))


def pmc_traffic(tag, runs, precision):
    """(bytes per launch, source) of the dominant kernel from the COMMITTED PMC passes of this same command (rocprofv3 --pmc FETCH_SIZE /
    WRITE_SIZE in separate runs, gfx950 FETCH_SIZE x2 correction; profiles/pmc_traffic.json).  PMC counters cannot be collected from inside
    the timed process, so this is not a measurement of THIS run: it is reported (with its source) only when the committed passes were taken
    on the same kernel tag, run count and precision; otherwise null."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            t = json.load(f)
        e = t.get(tag)
        if e and e["runs_per_gpu"] == runs and e["precision"] == precision:
            return e["bytes_per_launch"], "profiles/pmc_traffic.json (%s; taken at %s)" % (e.get("source", "rocprofv3 --pmc"), e.get("commit", "an earlier commit"))
    except Exception:
        pass
    return None, None


def profile_roofline(eng, steps, precision, R, flops_step=None, value=None, dump=""):
    """live per-kernel timing with HIP events on the engine stream (eager launches, not the graph)"""
    eng.profile_enable(True)
    eng.learn_n(steps)
    table = eng.profile_table()
    eng.profile_enable(False)
    if dump:
        with open(dump, "w") as f:
            tot = sum(t["total_ms"] for t in table)
            f.write("# %d profiled steps, %d runs/engine, precision %d; eager launches timed with HIP events on the engine stream\n" % (steps, R, precision))
            f.write("%-40s %9s %10s %10s %7s %9s %9s\n" % ("tag", "launches", "us/launch", "us/step", "%", "TFLOP/s", "GB/s"))
            for t in table:
                us = t["total_ms"] / t["launches"] * 1e3
                f.write("%-40s %9.1f %10.1f %10.1f %7.1f %9.1f %9.1f\n" % (
                    t["name"], t["launches"] / steps, us, t["total_ms"] / steps * 1e3, 100 * t["total_ms"] / tot,
                    t["flops_per_launch"] / us / 1e6, t["bytes_per_launch"] / us / 1e3))
            f.write("%-40s %9s %10s %10.1f\n" % ("total", "", "", tot / steps * 1e3))
    modelled = [t for t in table if t["flops_per_launch"] > 0 or t["bytes_per_launch"] > 0]
    if not modelled:
        return None
    # the dominant launch = the tag with the most time; the roof that binds it = the one it is closer to (a launch at 0.56 of HBM and 0.17
    # of the MFMA ceiling is HBM-bound; Adam has no matrix work at all)
    top = max(modelled, key=lambda t: t["total_ms"])
    avg_ms = top["total_ms"] / top["launches"]
    ach = top["flops_per_launch"] / (avg_ms * 1e-3) / 1e12
    peak = PEAK_TFLOPS[precision]
    gbps = top["bytes_per_launch"] / (avg_ms * 1e-3) / 1e9
    traffic, traffic_source = pmc_traffic(top["name"], R, precision)
    mfma = dict(flops_per_launch=top["flops_per_launch"], achieved=ach, peak=peak, unit="TFLOP/s", frac=ach / peak)
    # the same launch against the HBM roof (algorithmic bytes: operands read once, result written once)
    hbm = dict(bytes_per_launch=top["bytes_per_launch"], achieved=gbps, peak=8000.0, unit="GB/s", frac=gbps / 8000.0)
    by_hbm = hbm["frac"] > mfma["frac"]
    roof = dict(bound="hbm" if by_hbm else "mfma", kernel=top["name"], achieved=gbps if by_hbm else ach, peak=8000.0 if by_hbm else peak,
                unit="GB/s" if by_hbm else "TFLOP/s", frac=hbm["frac"] if by_hbm else mfma["frac"],
                traffic=traffic, traffic_source=traffic_source, avg_launch_ms=avg_ms, avg_launch_from="eager launches of engine 0 alone, HIP events on its stream",
                flops_per_launch=top["flops_per_launch"], hbm=hbm, mfma=mfma,
                table=[dict(name=t["name"], ms_per_step=t["total_ms"] / steps, launches_per_step=t["launches"] / steps) for t in table[:12]])
    gemms = [t for t in table if t["flops_per_launch"] > 0]
    if gemms and by_hbm:                      # the matrix launch that takes the most time, for reference
        g = max(gemms, key=lambda t: t["total_ms"])
        g_ms = g["total_ms"] / g["launches"]
        roof["dominant_gemm"] = dict(kernel=g["name"], avg_launch_ms=g_ms, mfma_frac=g["flops_per_launch"] / (g_ms * 1e-3) / 1e12 / peak,
                                     hbm_frac=g["bytes_per_launch"] / (g_ms * 1e-3) / 1e9 / 8000.0)
    if flops_step is not None and value is not None:
        roof["step_frac_of_mlp_gemm_roofline"] = value * flops_step / (peak * 1e12)
    # the whole step against the HBM roof: algorithmic bytes of every launch (operands read once, results written once) over the eager step time
    tot_ms = sum(t["total_ms"] for t in table)
    tot_bytes = sum(t["bytes_per_launch"] * t["launches"] for t in table)
    roof["whole_step_hbm"] = dict(algorithmic_gb_per_step=tot_bytes / steps / 1e9, achieved_gbps=tot_bytes / (tot_ms * 1e-3) / 1e9, peak_gbps=8000.0,
                                  frac=tot_bytes / (tot_ms * 1e-3) / 1e9 / 8000.0,
                                  note="launches without a byte model (sampling, loss and assembly kernels) count as zero bytes: a lower bound")
    return roof


def other_config(algo, device, precision, R, seconds):
    """BASELINE configs 1 / 3 / 4 through the same engine: TD3+BC halfcheetah shape, IQL hopper-medium-replay shape, EDAC walker2d-medium-expert
    shape (bench_workloads.py: the launch scripts' hyper-parameters and initialisation, synthetic buffers of the D4RL sizes)."""
    from offlinerlkit import _engine
    w = bw.WORKLOADS[algo]
    eng = bw.make_engine(algo, R, precision, device, 11)
    ds = make_dataset(3, w["n"], w["obs"], w["act"])
    buf = _engine.DeviceBuffer(w["obs"], w["act"], device)
    buf.load(ds["obs"], ds["act"], ds["nobs"], ds["rew"], ds["term"])
    if algo == "td3bc":
        buf.normalize_obs(1e-3)                           # run_td3bc.py:71
    eng.attach_buffer(buf)
    eng.learn_n(30)
    steps = 100
    reps = timed_rate([eng], steps, seconds)
    dt = float(np.median(reps))
    roof = profile_roofline(eng, 10, precision, R)
    if roof:
        roof.pop("table", None)
    value = R * steps / dt
    out = dict(workload=bw.workload_string(algo, R), value=value, unit="gradient-steps/s", ms_per_step=dt / steps * 1e3, precision=precision,
               algorithmic_gflop_per_gradient_step=w["gflop"],
               step_frac_of_mlp_gemm_roofline=value * w["gflop"] * 1e9 / (PEAK_TFLOPS[precision] * 1e12), roofline=roof)
    eng.close(); buf.close()
    return out


def api_record(device, precision, R, steps, dataset):
    """The same workload through the reference-shaped Python API: CQLPolicy built from torch modules as run_cql.py:80-128 does,
    ``set_engine_options(n_runs, precision)``, ``ReplayBuffer.load_dataset``, one fused ``MFPolicyTrainer`` epoch (sample -> learn x steps on
    the device, one readback; evaluation excluded).  The first epoch pays binding + graph capture and is not timed."""
    import tempfile
    import torch
    from offlinerlkit.buffer import ReplayBuffer
    from offlinerlkit.modules import ActorProb, Critic, TanhDiagGaussian
    from offlinerlkit.nets import MLP
    from offlinerlkit.policy import CQLPolicy
    from offlinerlkit.policy_trainer import MFPolicyTrainer
    from offlinerlkit.utils.logger import Logger

    class Space:
        low, high, shape = np.full(ACT, -1.0, np.float32), np.full(ACT, 1.0, np.float32), (ACT,)
    dev = f"cuda:{device}"
    torch.manual_seed(0)
    actor = ActorProb(MLP(OBS, HIDDEN), TanhDiagGaussian(HIDDEN[-1], ACT, unbounded=True, conditioned_sigma=True), dev)
    c1, c2 = Critic(MLP(OBS + ACT, HIDDEN), dev), Critic(MLP(OBS + ACT, HIDDEN), dev)
    log_alpha = torch.zeros(1, requires_grad=True, device=dev)
    pol = CQLPolicy(actor, c1, c2, torch.optim.Adam(actor.parameters(), lr=1e-4), torch.optim.Adam(c1.parameters(), lr=3e-4),
                    torch.optim.Adam(c2.parameters(), lr=3e-4), action_space=Space(), tau=0.005, gamma=0.99,
                    alpha=(-float(ACT), log_alpha, torch.optim.Adam([log_alpha], lr=1e-4)), cql_weight=5.0, temperature=1.0,
                    max_q_backup=False, deterministic_backup=True, with_lagrange=False, lagrange_threshold=10.0, cql_alpha_lr=3e-4,
                    num_repeart_actions=NREP)
    if precision is not None:
        pol.set_engine_options(n_runs=R, precision=precision, seed=5)
    n = len(dataset["rew"])
    buf = ReplayBuffer(n, (OBS,), np.float32, ACT, np.float32, device=dev)
    buf.load_dataset(dict(observations=dataset["obs"], actions=dataset["act"], next_observations=dataset["nobs"],
                          rewards=dataset["rew"], terminals=dataset["term"]))
    with tempfile.TemporaryDirectory() as tmp:
        logger = Logger(tmp, {"policy_training_progress": "csv"})
        tr = MFPolicyTrainer(pol, None, buf, logger, epoch=1, step_per_epoch=steps, batch_size=BATCH, eval_episodes=0)
        tr._train_epoch(1)                               # binding, buffer upload, graph capture
        torch.cuda.synchronize()
        times = []
        for e in (2, 3, 4):
            t0 = time.perf_counter()
            tr._train_epoch(e)                           # policy.learn_n(steps) + logger.logkv of the epoch means
            torch.cuda.synchronize()
            times.append(time.perf_counter() - t0)
        logger.close()
    dt = float(np.median(times))
    finite = bool(np.isfinite([v for v in getattr(logger, "_name2val", {}).values() if isinstance(v, (int, float))]).all())
    if pol.engine is not None:
        pol.engine.close()
    return dict(what="MFPolicyTrainer fused epoch through offlinerlkit.policy.CQLPolicy / ReplayBuffer (reference constructor signatures), "
                     "%d steps per epoch, %s, one engine, evaluation excluded" % (
                         steps, "n_runs=%d, precision=%d" % (R, precision) if precision is not None else
                         "NO set_engine_options call: one policy in the product's default precision (exact fp32), what run_cql.py gets unchanged"),
                value=R * steps / dt, unit="gradient-steps/s", epoch_seconds=times, finite=finite)


def rccl_check(local_rank):
    """World-size-1 rehearsal of the ONLY collective of the path on the one GPU a builder box has: the `nccl` (= RCCL) process group is
    created exactly as the N > 1 branch of main() creates it (device_id = this rank's GPU), a few engine steps run, and the three collective
    calls of the path execute on device tensors -- all_reduce(MAX) of the block time, all_gather of the per-run metric table (here and through
    MFPolicyTrainer._gather), barrier -- before the group is destroyed.  Prints one line `RCCL {json}`.  Proves that RCCL loads, binds the
    device and moves the metric table; it is not a scaling measurement."""
    import torch
    import torch.distributed as dist
    from offlinerlkit import _engine
    from offlinerlkit.policy_trainer import MFPolicyTrainer
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", str(free_port()))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(local_rank)
    dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank), rank=0, world_size=1)
    ds = make_dataset(0, 50_000)
    buf = _engine.DeviceBuffer(OBS, ACT, local_rank)
    buf.load(ds["obs"], ds["act"], ds["nobs"], ds["rew"], ds["term"])
    engines = make_cql_engines(1, 8, local_rank, 1, 0, buf)
    t0 = time.perf_counter()
    res = learn_all(engines, 5)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    t = torch.tensor([dt, 1.0], device="cuda", dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    metrics = res[0][0]
    mine = torch.tensor(metrics, device="cuda", dtype=torch.float32)
    allm = [torch.empty_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(allm, mine)
    gathered = torch.stack(allm).cpu().numpy()

    class _Log:
        def __init__(self): self.kv = {}
        def logkv(self, k, v): self.kv[k] = v
    tr = MFPolicyTrainer(None, None, None, _Log(), fused=False)
    tr._gather({"loss/critic1": float(metrics[0, 1]), "loss/actor": float(metrics[0, 0])})
    dist.barrier()
    out = dict(backend=dist.get_backend(), world=dist.get_world_size(), all_reduce_max_ok=bool(abs(float(t[0].item()) - dt) < 1e-9),
               all_gather_shape=list(gathered.shape), all_gather_equal=bool(np.array_equal(gathered[0], metrics)),
               finite=bool(np.isfinite(gathered).all()), trainer_gathered=sorted(tr.gathered_metrics[0].keys()) if tr.gathered_metrics else None,
               trainer_logged=sorted(tr.logger.kv.keys()), nccl_version=list(torch.cuda.nccl.version()) if hasattr(torch.cuda, "nccl") else None)
    dist.destroy_process_group()
    for g in engines:
        g.close()
    buf.close()
    os.write(1, ("RCCL " + json.dumps(out) + "\n").encode())


# ---------------------------------------------------------------------------------------------------------------
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--preset", choices=sorted(PRESETS), default=None, help="config5: 8 seeds of one task per GPU, one engine (BASELINE configs[4])")
    ap.add_argument("--runs-per-gpu", type=int, default=None,
                    help="independent CQL runs (seeds) carried by ONE engine; every launch updates all of them (default 96)")
    ap.add_argument("--engines-per-gpu", type=int, default=None,
                    help="independent engines per GPU (each --runs-per-gpu runs, own HIP stream and host thread): the launch-latency-bound "
                         "256-row phases of one engine overlap the many-row launches of the other (default 2)")
    ap.add_argument("--precision", type=int, default=int(os.environ.get("ORL_PRECISION", "1")), help="0 exact fp32 MFMA, 1 split MFMA (fp16 hi + lo planes; bf16 planes in the variant build), 2 three fp16 planes in the critic launches / fp32 MFMA elsewhere (all parity-gated)")
    ap.add_argument("--min-reps", type=int, default=5)
    ap.add_argument("--min-seconds", type=float, default=2.0)
    ap.add_argument("--no-sides", action="store_true", help="skip fp32 / by_runs / other_configs side records")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-seconds", type=float, default=24.0, help="CPU work of the cpu_baseline sample (four legs share it)")
    ap.add_argument("--profile-steps", type=int, default=20)
    ap.add_argument("--profile-dump", default="", help="write the full per-launch-tag timing table (HIP events) to this file")
    ap.add_argument("--dataset-size", type=int, default=None, help="transitions of the synthetic buffer (default 1 000 000; --preset config5: the rank's D4RL task size)")
    ap.add_argument("--launch-check", action="store_true", help="print this rank's RANK / LOCAL_RANK / WORLD_SIZE as JSON and exit (no GPU use)")
    ap.add_argument("--rccl-check", action="store_true", help="world-size-1 `nccl` process group on this GPU: the path's collectives on device tensors, then exit")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_children(args.gpus, sys.argv[1:]))       # the parent has made no GPU call
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU "
                         f"(python -m torch.distributed.run --nproc-per-node {args.gpus} bench.py --gpus {args.gpus} ...) or drop WORLD_SIZE")
    if args.launch_check:
        # one write(2) per rank: the ranks share the parent's stdout, and a text-mode print may split the line and its newline
        sys.stdout.flush()
        os.write(1, ("LAUNCH " + json.dumps(dict(rank=rank, local_rank=local_rank, world=world, gpus=args.gpus)) + "\n").encode())
        return
    if args.rccl_check:
        return rccl_check(local_rank)
    preset = PRESETS.get(args.preset, {})
    R = args.runs_per_gpu if args.runs_per_gpu is not None else int(os.environ.get("ORL_RUNS_PER_GPU", preset.get("runs_per_gpu", 96)))
    E = max(1, args.engines_per_gpu if args.engines_per_gpu is not None else int(os.environ.get("ORL_ENGINES_PER_GPU", preset.get("engines_per_gpu", 2))))

    import torch
    dist = None
    backend = os.environ.get("ORL_DIST_BACKEND", "nccl")      # "gloo" only to rehearse the N>1 path on one GPU
    if os.environ.get("ORL_FORCE_DEVICE") is not None:
        local_rank = int(os.environ["ORL_FORCE_DEVICE"])
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))     # RCCL over xGMI
        else:
            dist.init_process_group(backend)
    cdev = "cuda" if backend == "nccl" else "cpu"
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback); torch.cuda.is_available() is False")

    from offlinerlkit import _engine
    # one task buffer per GPU.  --preset config5: rank r trains D4RL-mujoco task r % 8 (its own observation / action widths and size)
    task = CONFIG5_TASKS[rank % len(CONFIG5_TASKS)] if args.preset == "config5" else ("halfcheetah-medium-v2", OBS, ACT, 1_000_000)
    od, ad = task[1], task[2]
    if args.dataset_size is None:
        args.dataset_size = task[3]
    ds = make_dataset(rank, args.dataset_size, od, ad)
    buf = _engine.DeviceBuffer(od, ad, local_rank)
    buf.load(ds["obs"], ds["act"], ds["nobs"], ds["rew"], ds["term"])
    engines = make_cql_engines(E, R, local_rank, args.precision, rank, buf, od=od, ad=ad)
    eng = engines[0]

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    if args.warmup > 0:
        learn_all(engines, args.warmup)
    reps, res = [], None
    t_all = time.perf_counter()
    while True:
        barrier()
        t0 = time.perf_counter()
        res = learn_all(engines, args.steps)              # every engine synchronises its stream before returning
        barrier()
        dt = time.perf_counter() - t0
        go = len(reps) + 1 < args.min_reps or (time.perf_counter() - t_all) < args.min_seconds
        if dist is not None:
            t = torch.tensor([dt, 1.0 if go else 0.0], device=cdev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)       # block time = the slowest rank's; every rank takes the same decision
            dt, go = float(t[0].item()), bool(t[1].item() > 0)
        reps.append(dt)
        if not go or len(reps) >= 200:
            break
    ev_ms = res[0][1]
    metrics = np.concatenate([m for m, _ in res], axis=0)       # (E * R, n_metrics)
    if dist is not None:
        # end-of-epoch metric all-gather over RCCL/xGMI (the only collective of the path)
        mine = torch.tensor(metrics, device=cdev, dtype=torch.float32)
        allm = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(allm, mine)
        metrics_all = torch.stack(allm).cpu().numpy()
    else:
        metrics_all = metrics[None]
    assert np.isfinite(metrics_all).all(), "non-finite losses"
    for g in engines:                                  # (the integer-view ReLU can scrub NaNs out of the forward: look at the parameters too)
        for r in (0, R - 1):
            for net in (0, 1, 2):
                assert all(np.isfinite(v).all() for v in g.get_net(r, net).values()), "non-finite parameters after the timed steps"

    total_steps = args.steps * R * E * world
    dt_med = float(np.median(reps))
    value = total_steps / dt_med
    if rank == 0:
        flops_step = cql_algorithmic_flops(od=od, ad=ad)
        roof = None
        if args.profile_steps > 0:
            roof = profile_roofline(eng, args.profile_steps, args.precision, R, flops_step, value / world, args.profile_dump)
            if roof and roof["kernel"].startswith("critic."):
                # one weight-stationary workgroup owns a CU; with several engines per GPU a launch stays on 2 * runs workgroups (one round,
                # orl_config::ws_one_round) and the other engines' kernels run on the CUs it leaves idle: `frac` is against the whole chip
                one_round = (E > 1) if os.environ.get("ORL_WS_ONE_ROUND") is None else os.environ["ORL_WS_ONE_ROUND"] == "1"
                cus = min(256, 2 * R) if one_round else 256
                roof["cus_occupied_by_the_launch"] = cus
                roof["frac_of_occupied_cus"] = roof["frac"] * 256.0 / cus
        for g in engines:
            g.close()
        engines = []
        sides = world == 1 and not args.no_sides and (od, ad) == (OBS, ACT)
        by_runs, fp32, fp32_class, others = None, None, None, None
        api, api_default, config5 = None, None, None
        if sides:
            by_runs = []                                 # (single side engines run alone on the GPU: whole rounds of workgroups)
            for r_side in (1, 8, 32, 96):
                es = make_cql_engines(1, r_side, local_rank, args.precision, 100 + r_side, buf)
                learn_all(es, 30)
                n_side = max(50, min(2000, int(4000 / r_side)))
                rr = timed_rate(es, n_side, 0.6)
                d = float(np.median(rr))
                by_runs.append(dict(runs_per_gpu=r_side, engines_per_gpu=1, value=r_side * n_side / d, ms_per_step=d / n_side * 1e3))
                es[0].close()
            by_runs.append(dict(runs_per_gpu=R * E, engines_per_gpu=E, value=value, ms_per_step=dt_med / args.steps * 1e3))
            # the same workload in the reference's own arithmetic and in fp32-CLASS arithmetic, each in its own best geometry (one engine x 128
            # runs, whole rounds of 256 workgroups): `fp32` = exact fp32 MFMA everywhere (precision 0), `fp32_class` = three fp16 planes per
            # operand in the many-row critic launches, fp32 MFMA elsewhere (precision 2)
            def side_precision(prec, n_steps, seed, e_side=1, r_side=128):
                es = make_cql_engines(e_side, r_side, local_rank, prec, seed, buf)
                learn_all(es, 10)
                rr = timed_rate(es, n_steps, 1.2, min_reps=3)
                d = float(np.median(rr))
                rf = profile_roofline(es[0], 5, prec, r_side)
                if rf:
                    rf.pop("table", None)
                rec = dict(value=e_side * r_side * n_steps / d, unit="gradient-steps/s", ms_per_step=d / n_steps * 1e3, dtype=dtype_string(prec), precision=prec,
                           steps_per_block=n_steps, reps_s=rr, seconds_timed=float(np.sum(rr)), engines_per_gpu=e_side, runs_per_engine=r_side, roofline=rf)
                for g in es:
                    g.close()
                return rec
            if args.precision != 0:
                fp32 = side_precision(0, 20, 7)                    # exact fp32: one engine x 128 runs, whole rounds (2 x 96 measured the same within 2 %)
            if args.precision != 2:
                fp32_class = side_precision(2, 40, 9)              # precision 2 in the geometry of the fp32 record: one engine x 128 runs
                # (the headline's two engines x 96 runs measure the same within box noise: 33.5k / 32.2k against 32.7k / 33.2k in two calls)
                fp32_class["two_engines_x_96"] = {k: v for k, v in side_precision(2, 40, 8, 2, 96).items() if k in ("value", "ms_per_step")}
                fp32_class["by_runs"] = []                 # few runs per engine in fp32-class arithmetic (exact fp32: 3.7k at 1 run, 11.8k at 8)
                for r_side in (1, 8):
                    es = make_cql_engines(1, r_side, local_rank, 2, 200 + r_side, buf)
                    learn_all(es, 30)
                    n_side = max(50, min(2000, int(4000 / r_side)))
                    d = float(np.median(timed_rate(es, n_side, 0.6)))
                    fp32_class["by_runs"].append(dict(runs_per_gpu=r_side, engines_per_gpu=1, value=r_side * n_side / d, ms_per_step=d / n_side * 1e3))
                    es[0].close()
                # ... and the reference CLI's default depth [256,256,256] (run_cql.py:31) at 128 runs: all six many-row launches on three planes
                h3 = other_config("cql_h3", local_rank, 2, 128, 0.8)
                h3.pop("roofline", None)
                fp32_class["cql_h3"] = h3
            others = {a: other_config(a, local_rank, args.precision, 128, 0.8) for a in ("td3bc", "iql", "edac", "cql_h3")}
            api = api_record(local_rank, args.precision, 128, 1000, ds)
            # the drop-in number: CQLPolicy exactly as run_cql.py:80-128 builds it -- one policy, no set_engine_options (n_runs 1, the product's
            # default precision = exact fp32) -- through the same fused MFPolicyTrainer epoch
            api_default = api_record(local_rank, None, 1, 1000, ds)
            # BASELINE configs[4] per GPU (what each rank of `--gpus 8 --preset config5` runs): one engine x 8 seeds on one task buffer, at
            # both D4RL-mujoco shapes, measured one after the other on this GPU
            config5 = []
            for name, t_od, t_ad, t_n in (CONFIG5_TASKS[0], CONFIG5_TASKS[1]):
                t_ds = ds if (t_od, t_ad) == (od, ad) else make_dataset(1, min(t_n, 400_000), t_od, t_ad)
                t_buf = buf if t_ds is ds else _engine.DeviceBuffer(t_od, t_ad, local_rank)
                if t_buf is not buf:
                    t_buf.load(t_ds["obs"], t_ds["act"], t_ds["nobs"], t_ds["rew"], t_ds["term"])
                es = make_cql_engines(1, 8, local_rank, args.precision, 300, t_buf, od=t_od, ad=t_ad)
                learn_all(es, 30)
                rr = timed_rate(es, 500, 0.6)
                d = float(np.median(rr))
                config5.append(dict(task=name, obs=t_od, act=t_ad, runs_per_gpu=8, engines_per_gpu=1, value=8 * 500 / d, ms_per_step=d / 500 * 1e3,
                                    algorithmic_gflop_per_gradient_step=cql_algorithmic_flops(od=t_od, ad=t_ad) / 1e9))
                es[0].close()
                if t_buf is not buf:
                    t_buf.close()
        cpu = cpu_baseline(args.cpu_baseline_seconds) if (world == 1 and not args.no_cpu_baseline) else None      # reported baseline: rank 0 at N = 1 only
        out = {
            "metric": "gradient-steps/sec (CQL, batch=256)", "value": value, "unit": "gradient-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt_med / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": dtype_string(args.precision),
            # the same workload in the reference's own arithmetic (exact fp32 MFMA), measured in this run, next to the headline
            "value_fp32": (value if args.precision == 0 else (fp32["value"] if fp32 else None)),
            # ... and in fp32-class arithmetic (precision 2: exact fp32 operands as three fp16 planes, products down to 2^-33, fp32 accumulation)
            "value_fp32_class": (value if args.precision == 2 else (fp32_class["value"] if fp32_class else None)),
            "target": {"steps_per_s": TARGET_STEPS_PER_S, "value_over_target": value / world / TARGET_STEPS_PER_S,
                       "value_fp32_over_target": ((value / world if args.precision == 0 else fp32["value"]) / TARGET_STEPS_PER_S) if (fp32 or args.precision == 0) else None,
                       "value_fp32_class_over_target": ((value / world if args.precision == 2 else fp32_class["value"]) / TARGET_STEPS_PER_S) if (fp32_class or args.precision == 2) else None,
                       "note": "BASELINE.json north_star target, per GPU; published reference numbers: none (vs_baseline null)"},
            "data": "synthetic D4RL-shaped replay buffer (N(0,1) obs, tanh actions), random-init weights",
            "config": {"workload": ("CQL halfcheetah-medium-v2 shape: obs17/act6, batch 256, MLP [256,256], 10 repeat actions, "
                                    "auto-alpha, device sampling+noise, %d engine(s) x %d run(s) per GPU x %d GPU(s) (independent seeds; value = steps of all runs)" % (E, R, world))
                                   if args.preset != "config5" else
                                   ("CQL, 8 seeds x 8 D4RL-mujoco tasks (BASELINE configs[4]): rank r trains task r %% 8 on its own synthetic buffer "
                                    "(obs/act %s), batch 256, MLP [256,256], 10 repeat actions, auto-alpha, device sampling+noise, one engine x %d seeds per GPU x %d GPU(s)"
                                    % (", ".join("%s %d/%d" % (CONFIG5_TASKS[r % len(CONFIG5_TASKS)][0], CONFIG5_TASKS[r % len(CONFIG5_TASKS)][1], CONFIG5_TASKS[r % len(CONFIG5_TASKS)][2]) for r in range(world)), R, world)),
                       "preset": args.preset, "runs_per_gpu": R * E, "engines_per_gpu": E, "runs_per_engine": R,
                       "tasks_by_rank": [CONFIG5_TASKS[r % len(CONFIG5_TASKS)][0] for r in range(world)] if args.preset == "config5" else None,
                       "dataset_transitions": args.dataset_size, "task_buffers": world, "event_ms_per_step": ev_ms / args.steps,
                       "algorithmic_gflop_per_gradient_step": flops_step / 1e9, "rccl_world_size": world,
                       "dist_backend": backend if world > 1 else None},
            "reps": {"blocks": len(reps), "steps_per_block": args.steps, "block_seconds": reps, "value_is": "median block",
                     "value_min": total_steps / max(reps), "value_max": total_steps / min(reps), "seconds_timed": float(np.sum(reps))},
            "roofline": roof, "cpu_baseline": cpu, "fp32": fp32, "fp32_class": fp32_class, "by_runs": by_runs, "other_configs": others, "api": api,
            "api_default": api_default, "config5_per_gpu": config5,
            "single_run": by_runs[0] if by_runs else None,
            "metrics_gathered": {"shape": list(metrics_all.shape), "loss_critic1_mean_per_rank": [float(x) for x in metrics_all[:, :, 1].mean(axis=1)]},
            "final_metrics_rank0_run0": dict(zip(eng.metric_names, [float(x) for x in metrics[0]])),
        }
        print(json.dumps(out), flush=True)
    for g in engines:
        g.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

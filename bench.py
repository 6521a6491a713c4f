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
Side records of the same run (rank 0, N = 1 only): `fp32` (exact-fp32 MFMA, same workload), `by_runs` (runs per
GPU 1 .. 192), `other_configs` (BASELINE configs 3 and 4), `roofline`, `cpu_baseline`.
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
for p in (ROOT, os.path.join(ROOT, "offlinerl-kit_amd"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)

OBS, ACT, HIDDEN, BATCH, NREP = 17, 6, [256, 256], 256, 10
# MI355X_MICROARCH.md: fp32 MFMA dense peak 157.3 TFLOP/s; bf16 MFMA dense peak ~2500 TFLOP/s.  precision=1 spends three
# bf16 MFMAs per fp32-equivalent product, so its ceiling for ALGORITHMIC flops is 2500/3.
PEAK_TFLOPS = {0: 157.3, 1: 2500.0 / 3.0}
DTYPE = {0: "f32 (v_mfma_f32_16x16x4_f32: the reference's arithmetic)",
         1: "f32 storage / accumulate; products on split-bf16 MFMA (operands = bf16 hi + bf16 lo = 16 significand bits, hi*hi + hi*lo + lo*hi "
            "on v_mfma_f32_16x16x32_bf16): losses and Q-values meet the 1e-4 parity gate in this mode (tests/test_gpu_cql.py), gradients are "
            "componentwise backward-stable at 2^-17 (tests/test_gpu_grads.py); the exact-fp32 figure of the same run is in `fp32`"}
PRESETS = {
    # BASELINE config 5: 8 seeds x 8 tasks, one task (its own synthetic buffer) and its 8 seeds per GPU, one engine
    "config5": dict(runs_per_gpu=8, engines_per_gpu=1),
}


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
    rng = np.random.RandomState(seed)
    return dict(
        obs=rng.standard_normal((n, od)).astype(np.float32),
        act=np.tanh(rng.standard_normal((n, ad))).astype(np.float32),
        nobs=rng.standard_normal((n, od)).astype(np.float32),
        rew=rng.standard_normal(n).astype(np.float32),
        term=(rng.uniform(size=n) < 0.01).astype(np.float32),
    )


def init_weights(eng, run, seed):
    import synth
    rng = np.random.RandomState(1000 + seed)
    actor = synth.make_tanh_actor(rng, OBS, ACT, HIDDEN)
    c1 = synth.make_critic(rng, OBS + ACT, HIDDEN)
    c2 = synth.make_critic(rng, OBS + ACT, HIDDEN)
    eng.set_net(run, 0, actor)
    eng.set_net(run, 1, c1); eng.set_net(run, 2, c2)
    eng.set_net(run, 3, c1); eng.set_net(run, 4, c2)      # deepcopy targets (sac.py:29-33)
    return dict(actor=actor, critic1=c1, critic2=c2)


def make_cql_engines(E, R, device, precision, seed0, buf):
    from offlinerlkit import _engine
    engines = []
    for e in range(E):
        cfg = _engine.default_config("cql", obs_dim=OBS, act_dim=ACT, hidden=HIDDEN, batch_size=BATCH, n_runs=R, device=device,
                                     precision=precision, seed=1234 + 7919 * (seed0 * E + e), num_repeat_actions=NREP,
                                     target_entropy=-float(ACT))
        g = _engine.Engine(cfg)
        g.attach_buffer(buf)                              # all engines of a GPU sample the same HBM-resident dataset
        for r in range(R):
            init_weights(g, r, (seed0 * E + e) * R + r)
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
# CPU baselines (rank 0, N = 1): the torch-CPU counterpart at all cores and at 1 thread, and the numpy oracle
# ---------------------------------------------------------------------------------------------------------------
def _cql_cpu_state():
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
    reference fixtures in tests/test_oracle_golden.py) on this box's host cores -- all cores available to the process and one
    thread -- plus the numpy oracle; same synthetic workload, a bounded sample of steps each."""
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
    for label, th in (("all", max(1, min(avail, 64))), ("t16", min(16, avail)), ("one", 1)):
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
    best = max(("all", "t16"), key=lambda k: out[k]["value"])
    return dict(value=out[best]["value"], unit="gradient-steps/s", cores=out[best]["threads"], kind="port",
                sample=f"{out[best]['steps']} CQL learn() steps (batch 256, ~{per:.0f} s) of the PyTorch-CPU counterpart (oracle/torch_cql.py: stock "
                       f"autograd + torch.optim.Adam) at {out[best]['threads']} torch threads; host reports {ncpu} cpus, {avail} available to the process",
                nproc=ncpu, cpus_available=avail,
                torch_all_cores=out["all"], torch_16_threads=out["t16"], torch_1_thread=out["one"],
                numpy_oracle=dict(value=rate, blas_threads=min(16, avail), steps=n))


def pmc_traffic(tag, runs, precision):
    """HBM bytes per launch of the dominant kernel from the committed PMC passes of this same command
    (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate runs, gfx950 FETCH_SIZE x2 correction; profiles/pmc_traffic.json).
    PMC counters cannot be collected from inside the timed process, so the figure is only reported when the committed
    measurement was taken on the same kernel tag, run count and precision; otherwise null."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            t = json.load(f)
        e = t.get(tag)
        if e and e["runs_per_gpu"] == runs and e["precision"] == precision:
            return e["bytes_per_launch"]
    except Exception:
        pass
    return None


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
    gemms = [t for t in table if t["flops_per_launch"] > 0]
    if not gemms:
        return None
    top = max(gemms, key=lambda t: t["total_ms"])
    avg_ms = top["total_ms"] / top["launches"]
    ach = top["flops_per_launch"] / (avg_ms * 1e-3) / 1e12
    peak = PEAK_TFLOPS[precision]
    gbps = top["bytes_per_launch"] / (avg_ms * 1e-3) / 1e9
    roof = dict(bound="mfma", kernel=top["name"], achieved=ach, peak=peak, unit="TFLOP/s", frac=ach / peak,
                traffic=pmc_traffic(top["name"], R, precision), avg_launch_ms=avg_ms, flops_per_launch=top["flops_per_launch"],
                # the same launch against the HBM roof (algorithmic bytes: operands read once, result written once)
                hbm=dict(bytes_per_launch=top["bytes_per_launch"], achieved=gbps, peak=8000.0, unit="GB/s", frac=gbps / 8000.0),
                table=[dict(name=t["name"], ms_per_step=t["total_ms"] / steps, launches_per_step=t["launches"] / steps) for t in table[:12]])
    if flops_step is not None and value is not None:
        roof["step_frac_of_mlp_gemm_roofline"] = value * flops_step / (peak * 1e12)
    return roof


def other_config(algo, device, precision, R, seconds):
    """BASELINE configs 3 / 4 through the same engine: IQL hopper-medium-replay shape, EDAC walker2d-medium-expert shape
    (the full-size parity cases' shapes and hyper-parameters, synthetic buffers of the D4RL sizes)."""
    import synth
    import test_gpu_algos as ta
    from offlinerlkit import _engine
    case = {"iql": "iql_hopper", "edac": "edac_walker2d"}[algo]
    c = getattr(synth, f"{algo.upper()}_CASES")[case]
    n = {"iql": 400_000, "edac": 2_000_000}[algo]
    eng, mod, cfg, st, _, _ = ta.make_engine(algo, case, n_runs=R, precision=precision)
    ds = make_dataset(3, n, c["obs_dim"], c["act_dim"])
    buf = _engine.DeviceBuffer(c["obs_dim"], c["act_dim"], device)
    buf.load(ds["obs"], ds["act"], ds["nobs"], ds["rew"], ds["term"])
    eng.attach_buffer(buf)
    eng.learn_n(30)
    steps = 100
    reps = timed_rate([eng], steps, seconds)
    dt = float(np.median(reps))
    roof = profile_roofline(eng, 10, precision, R)
    if roof:
        roof.pop("table", None)
    out = dict(workload=f"{algo.upper()} {case} shape: obs{c['obs_dim']}/act{c['act_dim']}, batch {c['B']}, hidden {c['hidden']}"
                        + (f", {cfg['num_critics']} critics, eta {cfg['eta']}" if algo == "edac" else f", expectile {cfg['expectile']}")
                        + f", {n} synthetic transitions, 1 engine x {R} runs, device sampling",
               value=R * steps / dt, unit="gradient-steps/s", ms_per_step=dt / steps * 1e3, precision=precision, roofline=roof)
    eng.close(); buf.close()
    return out


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
    ap.add_argument("--precision", type=int, default=int(os.environ.get("ORL_PRECISION", "1")), help="0 exact fp32 MFMA, 1 split-bf16 MFMA (parity-gated)")
    ap.add_argument("--min-reps", type=int, default=5)
    ap.add_argument("--min-seconds", type=float, default=2.0)
    ap.add_argument("--no-sides", action="store_true", help="skip fp32 / by_runs / other_configs side records")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--profile-steps", type=int, default=20)
    ap.add_argument("--profile-dump", default="", help="write the full per-launch-tag timing table (HIP events) to this file")
    ap.add_argument("--dataset-size", type=int, default=1_000_000)
    ap.add_argument("--launch-check", action="store_true", help="print this rank's RANK / LOCAL_RANK / WORLD_SIZE as JSON and exit (no GPU use)")
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
    preset = PRESETS.get(args.preset, {})
    R = args.runs_per_gpu if args.runs_per_gpu is not None else int(os.environ.get("ORL_RUNS_PER_GPU", preset.get("runs_per_gpu", 96)))
    E = max(1, args.engines_per_gpu if args.engines_per_gpu is not None else int(os.environ.get("ORL_ENGINES_PER_GPU", preset.get("engines_per_gpu", 2))))

    if E > 1:
        # several engines per GPU: each engine's weight-stationary launches stay on CUs / nets workgroups per net (one round), the CUs
        # they leave idle are where the other engine's kernels run (csrc/ws_gemm.h: ws_blocks_per_problem)
        os.environ.setdefault("ORL_WS_ONE_ROUND", "1")
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
    ds = make_dataset(rank, args.dataset_size)             # one task buffer per GPU (config 5: task = rank)
    buf = _engine.DeviceBuffer(OBS, ACT, local_rank)
    buf.load(ds["obs"], ds["act"], ds["nobs"], ds["rew"], ds["term"])
    engines = make_cql_engines(E, R, local_rank, args.precision, rank, buf)
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

    total_steps = args.steps * R * E * world
    dt_med = float(np.median(reps))
    value = total_steps / dt_med
    if rank == 0:
        flops_step = cql_algorithmic_flops()
        roof = None
        if args.profile_steps > 0:
            roof = profile_roofline(eng, args.profile_steps, args.precision, R, flops_step, value / world, args.profile_dump)
            if roof and roof["kernel"].startswith("critic."):
                # one weight-stationary workgroup owns a CU; with several engines per GPU a launch stays on 2 * runs workgroups (one round,
                # ORL_WS_ONE_ROUND) and the other engines' kernels run on the CUs it leaves idle: `frac` is against the whole chip
                cus = min(256, 2 * R) if os.environ.get("ORL_WS_ONE_ROUND") == "1" else 256
                roof["cus_occupied_by_the_launch"] = cus
                roof["frac_of_occupied_cus"] = roof["frac"] * 256.0 / cus
        for g in engines:
            g.close()
        engines = []
        sides = world == 1 and not args.no_sides
        by_runs, fp32, others = None, None, None
        if sides:
            os.environ["ORL_WS_ONE_ROUND"] = "0"          # the side engines run alone on the GPU: whole rounds of workgroups
            by_runs = []
            for r_side in (1, 8, 32, 96):
                es = make_cql_engines(1, r_side, local_rank, args.precision, 100 + r_side, buf)
                learn_all(es, 30)
                n_side = max(50, min(2000, int(4000 / r_side)))
                rr = timed_rate(es, n_side, 0.6)
                d = float(np.median(rr))
                by_runs.append(dict(runs_per_gpu=r_side, engines_per_gpu=1, value=r_side * n_side / d, ms_per_step=d / n_side * 1e3))
                es[0].close()
            by_runs.append(dict(runs_per_gpu=R * E, engines_per_gpu=E, value=value, ms_per_step=dt_med / args.steps * 1e3))
            if args.precision != 0:
                if E > 1:
                    os.environ["ORL_WS_ONE_ROUND"] = "1"
                es = make_cql_engines(E, R, local_rank, 0, 7, buf)
                learn_all(es, 10)
                n32 = 20
                rr = timed_rate(es, n32, 1.2, min_reps=3)
                d = float(np.median(rr))
                r32 = profile_roofline(es[0], 5, 0, R)
                if r32:
                    r32.pop("table", None)
                fp32 = dict(value=R * E * n32 / d, unit="gradient-steps/s", ms_per_step=d / n32 * 1e3, dtype=DTYPE[0], steps_per_block=n32,
                            reps_s=rr, seconds_timed=float(np.sum(rr)), engines_per_gpu=E, runs_per_engine=R, roofline=r32)
                for g in es:
                    g.close()
            os.environ["ORL_WS_ONE_ROUND"] = "0"
            others = {a: other_config(a, local_rank, args.precision, 128, 0.8) for a in ("iql", "edac")}
        cpu = cpu_baseline() if (world == 1 and not args.no_cpu_baseline) else None      # reported baseline: rank 0 at N = 1 only
        out = {
            "metric": "gradient-steps/sec (CQL, batch=256)", "value": value, "unit": "gradient-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt_med / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": DTYPE[args.precision],
            "data": "synthetic D4RL-shaped replay buffer (N(0,1) obs, tanh actions), random-init weights",
            "config": {"workload": "CQL halfcheetah-medium-v2 shape: obs17/act6, batch 256, MLP [256,256], 10 repeat actions, "
                                   "auto-alpha, device sampling+noise, %d engine(s) x %d run(s) per GPU x %d GPU(s) (independent seeds; value = steps of all runs)" % (E, R, world),
                       "preset": args.preset, "runs_per_gpu": R * E, "engines_per_gpu": E, "runs_per_engine": R,
                       "dataset_transitions": args.dataset_size, "task_buffers": world, "event_ms_per_step": ev_ms / args.steps,
                       "algorithmic_gflop_per_gradient_step": flops_step / 1e9, "rccl_world_size": world,
                       "dist_backend": backend if world > 1 else None},
            "reps": {"blocks": len(reps), "steps_per_block": args.steps, "block_seconds": reps, "value_is": "median block",
                     "value_min": total_steps / max(reps), "value_max": total_steps / min(reps), "seconds_timed": float(np.sum(reps))},
            "roofline": roof, "cpu_baseline": cpu, "fp32": fp32, "by_runs": by_runs, "other_configs": others,
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

#!/usr/bin/env python3
"""bench.py — gradient-steps/sec of the CQL policy.learn() hot path on MI355X.

Workload (BASELINE.json configs[1]): CQL, halfcheetah-medium-v2-shaped synthetic replay buffer
(N=1e6 transitions, obs 17, act 6), batch 256, critics/actor MLP [256,256], 10 repeated actions,
auto-alpha, no Lagrange.  One "step" = one engine step = one policy.learn() update for every run the
engine carries (--runs-per-gpu independent seeds batched through the same kernel launches), including
on-device index sampling, replay gather and noise generation.  value = gradient steps of all runs on all
ranks / wall time (max over ranks), inputs resident in HBM before the timed region.

Contract: python bench.py --gpus N --steps K --warmup W ; for N>1 launched by torch.distributed.run, one
rank per GPU; independent seeds per rank (replicas only) with one RCCL all_gather of the
per-run metric means at the end, as MFPolicyTrainer would log per epoch.
"""
from __future__ import annotations

import argparse
import json
import os
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
DTYPE = {0: "f32 (v_mfma_f32_16x16x4_f32)", 1: "f32 via split-bf16 (3x v_mfma_f32_16x16x32_bf16, fp32 accumulate)"}


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
    H = hidden[-1]
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


def make_dataset(seed, n=1_000_000):
    rng = np.random.RandomState(seed)
    return dict(
        obs=rng.standard_normal((n, OBS)).astype(np.float32),
        act=np.tanh(rng.standard_normal((n, ACT))).astype(np.float32),
        nobs=rng.standard_normal((n, OBS)).astype(np.float32),
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


def cpu_baseline(seconds=14.0):
    """The oracle (numpy port of the reference CQL learn(), parity-pinned in tests/test_oracle_golden.py) timed on
    this box's host cores on the same synthetic workload.  BLAS thread count is swept (1, 8, 16, 32) because OpenBLAS
    on all cores of a large host is SLOWER on these 256-wide layers; the best setting is reported with its thread count."""
    import synth
    from oracle import cql as ocql
    from helpers import clone_state
    try:
        from threadpoolctl import threadpool_limits
    except Exception:
        threadpool_limits = None
    rng = np.random.RandomState(5)
    st = dict(actor=synth.make_tanh_actor(rng, OBS, ACT, HIDDEN), critic1=synth.make_critic(rng, OBS + ACT, HIDDEN),
              critic2=synth.make_critic(rng, OBS + ACT, HIDDEN))
    st["critic1_old"] = synth.make_critic(rng, OBS + ACT, HIDDEN)
    st["critic2_old"] = synth.make_critic(rng, OBS + ACT, HIDDEN)
    st["log_alpha"] = np.zeros(1, np.float32); st["cql_log_alpha"] = np.zeros(1, np.float32)
    st = clone_state(st)
    ocql.init_opt(st)
    cfg = ocql.default_cfg(OBS, ACT)
    ds = make_dataset(0, 100_000)

    def one_step():
        idx = np.random.randint(0, 100_000, size=BATCH)          # buffer.py:98
        batch = dict(observations=ds["obs"][idx], actions=ds["act"][idx], next_observations=ds["nobs"][idx],
                     rewards=ds["rew"][idx], terminals=ds["term"][idx])
        ocql.learn(st, cfg, batch, synth.make_cql_noise(rng, BATCH, NREP, ACT))

    ncpu = os.cpu_count() or 1
    cands = sorted({t for t in (1, 8, 16, 32) if t <= ncpu}) if threadpool_limits else [ncpu]
    best = None
    per = seconds / max(len(cands), 1)
    for t in cands:
        ctx = threadpool_limits(limits=t) if threadpool_limits else None
        try:
            one_step(); one_step()
            n, t0 = 0, time.perf_counter()
            while time.perf_counter() - t0 < per:
                one_step(); n += 1
            rate = n / (time.perf_counter() - t0)
        finally:
            if ctx is not None:
                ctx.unregister() if hasattr(ctx, "unregister") else None
        if best is None or rate > best[0]:
            best = (rate, t, n)
    return dict(value=best[0], unit="gradient-steps/s", cores=int(best[1]), kind="port",
                sample=f"{best[2]} CQL learn() steps (batch 256, ~{per:.0f} s) of the numpy oracle at its best BLAS thread count "
                       f"({best[1]} of {cands} tried; host has {ncpu} cpus)")


def pmc_traffic(tag, runs, precision):
    """HBM bytes per launch of the dominant kernel from the committed PMC passes of this same command
    (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate runs, gfx950 FETCH_SIZE x2 correction; profiles/pmc_traffic.json).
    PMC counters cannot be collected from inside the timed process, so the figure is only reported when the committed
    measurement was taken on the same kernel tag, run count and precision; otherwise null."""
    try:
        with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "pmc_traffic.json")) as f:
            t = json.load(f)
        e = t.get(tag)
        if e and e["runs_per_gpu"] == runs and e["precision"] == precision:
            return e["bytes_per_launch"]
    except Exception:
        pass
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--runs-per-gpu", type=int, default=int(os.environ.get("ORL_RUNS_PER_GPU", "96")),
                    help="independent CQL runs (seeds) carried by one engine / GPU; every launch updates all of them")
    ap.add_argument("--precision", type=int, default=int(os.environ.get("ORL_PRECISION", "1")), help="0 exact fp32 MFMA, 1 split-bf16 MFMA (parity-gated)")
    ap.add_argument("--no-single", action="store_true", help="skip the side measurement with ONE run per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--engines-per-gpu", type=int, default=int(os.environ.get("ORL_ENGINES_PER_GPU", "2")),
                    help="independent engines per GPU (each --runs-per-gpu runs, own HIP stream and host thread): the launch-latency-bound "
                         "256-row phases of one engine overlap the many-row launches of the other")
    ap.add_argument("--profile-steps", type=int, default=20)
    ap.add_argument("--profile-dump", default="", help="write the full per-launch-tag timing table (HIP events) to this file")
    ap.add_argument("--dataset-size", type=int, default=1_000_000)
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
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
    R = args.runs_per_gpu
    cfg = _engine.default_config("cql", obs_dim=OBS, act_dim=ACT, hidden=HIDDEN, batch_size=BATCH, n_runs=R,
                                 device=local_rank, precision=args.precision, seed=1234 + 7919 * rank,
                                 num_repeat_actions=NREP, target_entropy=-float(ACT))
    E = max(1, args.engines_per_gpu)
    ds = make_dataset(rank, args.dataset_size)
    buf = _engine.DeviceBuffer(OBS, ACT, local_rank)
    buf.load(ds["obs"], ds["act"], ds["nobs"], ds["rew"], ds["term"])
    engines = []
    for e in range(E):
        cfg_e = _engine.default_config("cql", obs_dim=OBS, act_dim=ACT, hidden=HIDDEN, batch_size=BATCH, n_runs=R,
                                       device=local_rank, precision=args.precision, seed=1234 + 7919 * (rank * E + e),
                                       num_repeat_actions=NREP, target_entropy=-float(ACT))
        g = _engine.Engine(cfg_e) if e else _engine.Engine(cfg)
        g.attach_buffer(buf)                             # all engines of a GPU sample the same HBM-resident dataset
        for r in range(R):
            init_weights(g, r, (rank * E + e) * R + r)
        engines.append(g)
    eng = engines[0]

    import threading

    def learn_all(n):
        """every engine advances n gradient steps (all its runs); one host thread per engine, each on its own HIP stream"""
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

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    if args.warmup > 0:
        learn_all(args.warmup)
    barrier()
    t0 = time.perf_counter()
    res = learn_all(args.steps)                       # every engine synchronises its stream before returning
    barrier()
    dt = time.perf_counter() - t0
    metrics, ev_ms = res[0]
    metrics = np.concatenate([m for m, _ in res], axis=0)       # (E * R, n_metrics)
    if dist is not None:
        t = torch.tensor([dt], device=cdev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        # end-of-epoch metric all-gather over RCCL/xGMI (the only collective of the path)
        mine = torch.tensor(metrics, device=cdev, dtype=torch.float32)
        allm = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(allm, mine)
        metrics_all = torch.stack(allm).cpu().numpy()
    else:
        metrics_all = metrics[None]
    assert np.isfinite(metrics_all).all(), "non-finite losses"

    total_steps = args.steps * R * E * world
    value = total_steps / dt
    out = None
    single = None
    if rank == 0 and world == 1 and not args.no_single and (R > 1 or E > 1):
        # side measurement: the same workload with ONE run on the GPU (latency-bound regime), same precision
        cfg1 = _engine.default_config("cql", obs_dim=OBS, act_dim=ACT, hidden=HIDDEN, batch_size=BATCH, n_runs=1, device=local_rank,
                                      precision=args.precision, seed=99, num_repeat_actions=NREP, target_entropy=-float(ACT))
        e1 = _engine.Engine(cfg1)
        e1.attach_buffer(buf)
        init_weights(e1, 0, 12345)
        e1.learn_n(100)
        n1 = max(200, args.steps // 2)
        t1 = time.perf_counter(); e1.learn_n(n1); d1 = time.perf_counter() - t1
        single = dict(value=n1 / d1, ms_per_step=d1 / n1 * 1e3, runs_per_gpu=1)
        e1.close()
    if rank == 0:
        flops_step = cql_algorithmic_flops()
        # live per-kernel timing with HIP events on the engine stream (eager launches, not the graph)
        roof = None
        if args.profile_steps > 0:
            eng.profile_enable(True)
            eng.learn_n(args.profile_steps)
            table = eng.profile_table()
            eng.profile_enable(False)
            if args.profile_dump:
                with open(args.profile_dump, "w") as f:
                    tot = sum(t["total_ms"] for t in table)
                    f.write("# %d profiled steps, %d runs/GPU, precision %d; eager launches timed with HIP events on the engine stream\n"
                            % (args.profile_steps, R, args.precision))
                    f.write("%-40s %9s %10s %10s %7s %9s %9s\n" % ("tag", "launches", "us/launch", "us/step", "%", "TFLOP/s", "GB/s"))
                    for t in table:
                        us = t["total_ms"] / t["launches"] * 1e3
                        f.write("%-40s %9.1f %10.1f %10.1f %7.1f %9.1f %9.1f\n" % (
                            t["name"], t["launches"] / args.profile_steps, us, t["total_ms"] / args.profile_steps * 1e3,
                            100 * t["total_ms"] / tot, t["flops_per_launch"] / us / 1e6, t["bytes_per_launch"] / us / 1e3))
                    f.write("%-40s %9s %10s %10.1f\n" % ("total", "", "", tot / args.profile_steps * 1e3))
            gemms = [t for t in table if t["flops_per_launch"] > 0]
            if gemms:
                top = max(gemms, key=lambda t: t["total_ms"])
                avg_ms = top["total_ms"] / top["launches"]
                ach = top["flops_per_launch"] / (avg_ms * 1e-3) / 1e12
                peak = PEAK_TFLOPS[args.precision]
                gbps = top["bytes_per_launch"] / (avg_ms * 1e-3) / 1e9
                roof = dict(bound="mfma", kernel=top["name"], achieved=ach, peak=peak, unit="TFLOP/s", frac=ach / peak,
                            traffic=pmc_traffic(top["name"], R, args.precision), avg_launch_ms=avg_ms,
                            flops_per_launch=top["flops_per_launch"],
                            # the same launch against the HBM roof (algorithmic bytes: operands read once, result written once)
                            hbm=dict(bytes_per_launch=top["bytes_per_launch"], achieved=gbps, peak=8000.0, unit="GB/s", frac=gbps / 8000.0),
                            step_frac_of_mlp_gemm_roofline=(value / world) * flops_step / (peak * 1e12),
                            table=[dict(name=t["name"], ms_per_step=t["total_ms"] / args.profile_steps,
                                        launches_per_step=t["launches"] / args.profile_steps) for t in table[:12]])
        cpu = cpu_baseline() if (world == 1 and not args.no_cpu_baseline) else None      # reported baseline: rank 0 at N = 1 only
        out = {
            "metric": "gradient-steps/sec (CQL, batch=256)", "value": value, "unit": "gradient-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": DTYPE[args.precision],
            "data": "synthetic D4RL-shaped replay buffer (N(0,1) obs, tanh actions), random-init weights",
            "config": {"workload": "CQL halfcheetah-medium-v2 shape: obs17/act6, batch 256, MLP [256,256], 10 repeat actions, "
                                   "auto-alpha, device sampling+noise, %d engine(s) x %d run(s) per GPU x %d GPU(s) (independent seeds)" % (E, R, world),
                       "runs_per_gpu": R * E, "engines_per_gpu": E, "runs_per_engine": R, "dataset_transitions": args.dataset_size, "event_ms_per_step": ev_ms / args.steps,
                       "algorithmic_gflop_per_gradient_step": flops_step / 1e9},
            "roofline": roof, "cpu_baseline": cpu, "single_run": single,
            "final_metrics_rank0_run0": dict(zip(eng.metric_names, [float(x) for x in metrics[0]])),
        }
        print(json.dumps(out))
    for g in engines:
        g.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

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
PEAK_TFLOPS = {0: 157.3, 1: 2500.0 / 3.0}   # fp32 MFMA dense peak; split-bf16 = 3 bf16 MFMAs per product (MI355X_MICROARCH.md)


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


def cpu_baseline(seconds=12.0):
    """The oracle (numpy port of the reference CQL learn(), parity-pinned in tests/test_oracle_golden.py) timed on
    this box's host cores on the same synthetic workload."""
    import synth
    from oracle import cql as ocql
    from helpers import clone_state
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
    n, t0 = 0, None
    while True:
        idx = np.random.randint(0, 100_000, size=BATCH)          # buffer.py:98
        batch = dict(observations=ds["obs"][idx], actions=ds["act"][idx], next_observations=ds["nobs"][idx],
                     rewards=ds["rew"][idx], terminals=ds["term"][idx])
        noise = synth.make_cql_noise(rng, BATCH, NREP, ACT)
        ocql.learn(st, cfg, batch, noise)
        n += 1
        if n == 3:
            t0 = time.perf_counter(); n0 = n       # 3 warm-up steps
        if t0 is not None and time.perf_counter() - t0 > seconds:
            break
    dt = time.perf_counter() - t0
    try:
        from threadpoolctl import threadpool_info
        threads = max([p.get("num_threads", 1) for p in threadpool_info()] + [1])
    except Exception:
        threads = os.cpu_count()
    return dict(value=(n - n0) / dt, unit="gradient-steps/s", cores=int(threads), kind="port",
                sample=f"{n - n0} CQL learn() steps of the numpy oracle (OpenBLAS, {threads} threads, host has {os.cpu_count()} cpus), batch 256, ~{seconds:.0f} s")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--runs-per-gpu", type=int, default=int(os.environ.get("ORL_RUNS_PER_GPU", "1")))
    ap.add_argument("--precision", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--profile-steps", type=int, default=20)
    ap.add_argument("--dataset-size", type=int, default=1_000_000)
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch
    dist = None
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback); torch.cuda.is_available() is False")

    from offlinerlkit import _engine
    R = args.runs_per_gpu
    cfg = _engine.default_config("cql", obs_dim=OBS, act_dim=ACT, hidden=HIDDEN, batch_size=BATCH, n_runs=R,
                                 device=local_rank, precision=args.precision, seed=1234 + 7919 * rank,
                                 num_repeat_actions=NREP, target_entropy=-float(ACT))
    eng = _engine.Engine(cfg)
    ds = make_dataset(rank, args.dataset_size)
    buf = _engine.DeviceBuffer(OBS, ACT, local_rank)
    buf.load(ds["obs"], ds["act"], ds["nobs"], ds["rew"], ds["term"])
    eng.attach_buffer(buf)
    for r in range(R):
        init_weights(eng, r, rank * R + r)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    if args.warmup > 0:
        eng.learn_n(args.warmup)
    barrier()
    t0 = time.perf_counter()
    metrics, ev_ms = eng.learn_n(args.steps)          # synchronises the engine stream before returning
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        # end-of-epoch metric all-gather over RCCL/xGMI (the only collective of the path)
        mine = torch.tensor(metrics, device="cuda", dtype=torch.float32)
        allm = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(allm, mine)
        metrics_all = torch.stack(allm).cpu().numpy()
    else:
        metrics_all = metrics[None]
    assert np.isfinite(metrics_all).all(), "non-finite losses"

    total_steps = args.steps * R * world
    value = total_steps / dt
    out = None
    if rank == 0:
        flops_step = cql_algorithmic_flops()
        # live per-kernel timing with HIP events on the engine stream (eager launches, not the graph)
        roof = None
        if args.profile_steps > 0:
            eng.profile_enable(True)
            eng.learn_n(args.profile_steps)
            table = eng.profile_table()
            eng.profile_enable(False)
            gemms = [t for t in table if t["flops_per_launch"] > 0]
            if gemms:
                top = max(gemms, key=lambda t: t["total_ms"])
                avg_ms = top["total_ms"] / top["launches"]
                ach = top["flops_per_launch"] / (avg_ms * 1e-3) / 1e12
                peak = PEAK_TFLOPS[args.precision]
                roof = dict(bound="mfma", kernel=top["name"], achieved=ach, peak=peak, unit="TFLOP/s", frac=ach / peak,
                            traffic=None, avg_launch_ms=avg_ms, flops_per_launch=top["flops_per_launch"],
                            step_frac_of_mlp_gemm_roofline=(value / world) * flops_step / (peak * 1e12),
                            table=[dict(name=t["name"], ms_per_step=t["total_ms"] / args.profile_steps,
                                        launches_per_step=t["launches"] / args.profile_steps) for t in table[:12]])
        cpu = None if args.no_cpu_baseline else cpu_baseline()
        out = {
            "metric": "gradient-steps/sec (CQL, batch=256)", "value": value, "unit": "gradient-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32" if args.precision == 0 else "f32 via 3x bf16 MFMA split",
            "data": "synthetic D4RL-shaped replay buffer (N(0,1) obs, tanh actions), random-init weights",
            "config": {"workload": "CQL halfcheetah-medium-v2 shape: obs17/act6, batch 256, MLP [256,256], 10 repeat actions, "
                                   "auto-alpha, device sampling+noise, %d run(s)/GPU x %d GPU(s) (independent seeds)" % (R, world),
                       "runs_per_gpu": R, "dataset_transitions": args.dataset_size, "event_ms_per_step": ev_ms / args.steps,
                       "algorithmic_gflop_per_gradient_step": flops_step / 1e9},
            "roofline": roof, "cpu_baseline": cpu,
            "final_metrics_rank0_run0": dict(zip(eng.metric_names, [float(x) for x in metrics[0]])),
        }
        print(json.dumps(out))
    eng.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

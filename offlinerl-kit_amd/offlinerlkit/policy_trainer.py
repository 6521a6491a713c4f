"""MFPolicyTrainer (reference: offlinerlkit/policy_trainer/mf_policy_trainer.py:17-118).

Same constructor, same logged keys, same per-epoch order (train steps -> lr_scheduler.step -> evaluate -> log ->
checkpoint) and the same return value.  Two inner loops:
  * ``fused=False``: the reference's loop verbatim — ``buffer.sample`` (numpy index stream) -> ``policy.learn``
    -> ``logger.logkv_mean`` per step, one host sync per step;
  * ``fused=True`` (default when the policy offers ``learn_n`` and the buffer is HBM-resident): the whole epoch's
    sample -> learn chain runs on the device and only the epoch means come back — the values ``logkv_mean`` would
    have accumulated (only per-epoch means are ever consumed).
Evaluation: ``eval_env`` may be ONE env (the reference's sequential loop, verbatim) or a list of envs, which are stepped in
lockstep with one batched ``select_action`` forward per step under the same episode accounting.
Several runs per policy object (``policy.n_runs`` > 1, BASELINE config 5): the epoch's training is still one ``learn_n``; every
run is evaluated, logged as ``run<i>/<key>`` (plain keys = mean over runs) and checkpointed as ``policy_run<i>.pth``.  With
``eval_env`` = one list of envs per run the runs are evaluated TOGETHER (one run-batched forward per env step,
``policy.select_action_runs``); with a single env or a flat list they are evaluated one after another (``select_run``).
Multi-GPU: independent runs, one process per GPU (replicas only).  When ``torch.distributed`` is initialised the
per-epoch metric vector of every rank is all-gathered (RCCL over xGMI on GPUs, gloo on CPU) so rank 0 can log
all runs; no other collective exists on this path.
"""
from __future__ import annotations

import os
import time
from collections import deque
from typing import Dict, List, Optional

import numpy as np
import torch


class MFPolicyTrainer:
    def __init__(self, policy, eval_env, buffer, logger, epoch: int = 1000, step_per_epoch: int = 1000, batch_size: int = 256,
                 eval_episodes: int = 10, lr_scheduler=None, fused: Optional[bool] = None, progress: bool = False) -> None:
        self.policy = policy
        self.eval_env = eval_env
        self.buffer = buffer
        self.logger = logger
        self._epoch = epoch
        self._step_per_epoch = step_per_epoch
        self._batch_size = batch_size
        self._eval_episodes = eval_episodes
        self.lr_scheduler = lr_scheduler
        if fused is None:
            fused = hasattr(policy, "learn_n") and hasattr(buffer, "device_buffer")
        self._fused = fused
        self._progress = progress
        self.gathered_metrics: List[Dict[str, np.ndarray]] = []   # per epoch: key -> value of every rank

    # ---- inner loops -----------------------------------------------------------------------
    def _train_epoch(self, e: int) -> int:
        if self._fused:
            means = self.policy.learn_n(self._step_per_epoch, self.buffer, self._batch_size)
            for k, v in means.items():
                self.logger.logkv(k, v)
            self._check_health()
            return self._step_per_epoch
        it = range(self._step_per_epoch)
        if self._progress:
            from tqdm import tqdm
            it = tqdm(it, desc=f"Epoch #{e}/{self._epoch}")
        for _ in it:
            batch = self.buffer.sample(self._batch_size)
            loss = self.policy.learn(batch)
            if self._progress:
                it.set_postfix(**loss)
            for k, v in loss.items():
                self.logger.logkv_mean(k, v)
        self._check_health()
        return self._step_per_epoch

    def _check_health(self) -> None:
        """once per epoch: the engine's range / non-finite scan (EnginePolicy.check_health warns or raises); other policies have none"""
        check = getattr(self.policy, "check_health", None)
        if callable(check):
            check()

    def _gather(self, kv: Dict[str, float]) -> None:
        """End-of-epoch metric all-gather: every rank contributes its metric vector, rank 0 logs ``rank<i>/<key>``."""
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()):
            return
        keys = sorted(kv)
        backend = dist.get_backend()
        dev = torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu")
        mine = torch.tensor([float(kv[k]) for k in keys], dtype=torch.float32, device=dev)
        out = [torch.empty_like(mine) for _ in range(dist.get_world_size())]
        dist.all_gather(out, mine)
        table = torch.stack(out).cpu().numpy()
        self.gathered_metrics.append({k: table[:, i].copy() for i, k in enumerate(keys)})
        if dist.get_rank() == 0:
            for r in range(table.shape[0]):
                for i, k in enumerate(keys):
                    self.logger.logkv(f"rank{r}/{k}", float(table[r, i]))

    # ---- reference API ---------------------------------------------------------------------
    def train(self) -> Dict[str, float]:
        start_time = time.time()
        num_timesteps = 0
        last_10_performance = deque(maxlen=10)
        n_runs = int(getattr(self.policy, "n_runs", 1))
        for e in range(1, self._epoch + 1):
            self.policy.train()
            num_timesteps += self._train_epoch(e)
            if self.lr_scheduler is not None:
                self.lr_scheduler.step()
            if n_runs == 1:
                eval_info = self._evaluate()
            else:
                grouped = self._env_groups(n_runs)
                per_run_all = self._evaluate_runs_batched(grouped) if grouped is not None else None
                per_run = []
                for r in range(n_runs):
                    if per_run_all is not None:
                        per_run.append(per_run_all[r])
                    else:
                        self.policy.select_run(r)
                        per_run.append(self._evaluate())
                    self.logger.logkv(f"run{r}/eval/episode_reward", float(np.mean(per_run[-1]["eval/episode_reward"])))
                    self.logger.logkv(f"run{r}/eval/episode_length", float(np.mean(per_run[-1]["eval/episode_length"])))
                    if hasattr(self._score_env(), "get_normalized_score"):
                        self.logger.logkv(f"run{r}/eval/normalized_episode_reward",
                                          self._score_env().get_normalized_score(float(np.mean(per_run[-1]["eval/episode_reward"]))) * 100)
                self.policy.select_run(0)
                eval_info = {k: [x for info in per_run for x in info[k]] for k in per_run[0]}      # pooled over runs
            ep_reward_mean, ep_reward_std = np.mean(eval_info["eval/episode_reward"]), np.std(eval_info["eval/episode_reward"])
            ep_length_mean, ep_length_std = np.mean(eval_info["eval/episode_length"]), np.std(eval_info["eval/episode_length"])
            epoch_kv = dict(self.logger._name2val) if hasattr(self.logger, "_name2val") else {}
            if hasattr(self._score_env(), "get_normalized_score"):
                norm_ep_rew_mean = self._score_env().get_normalized_score(ep_reward_mean) * 100
                norm_ep_rew_std = self._score_env().get_normalized_score(ep_reward_std) * 100
                last_10_performance.append(norm_ep_rew_mean)
                self.logger.logkv("eval/normalized_episode_reward", norm_ep_rew_mean)
                self.logger.logkv("eval/normalized_episode_reward_std", norm_ep_rew_std)
                epoch_kv["eval/normalized_episode_reward"] = norm_ep_rew_mean
            self.logger.logkv("eval/episode_reward", ep_reward_mean)
            self.logger.logkv("eval/episode_reward_std", ep_reward_std)
            self.logger.logkv("eval/episode_length", ep_length_mean)
            self.logger.logkv("eval/episode_length_std", ep_length_std)
            epoch_kv["eval/episode_reward"] = ep_reward_mean
            self._gather({k: v for k, v in epoch_kv.items() if isinstance(v, (int, float, np.floating))})
            self.logger.set_timestep(num_timesteps)
            self.logger.dumpkvs()
            self._checkpoint(self.logger.checkpoint_dir, n_runs)
        self.logger.log("total time: {:.2f}s".format(time.time() - start_time))
        self._checkpoint(self.logger.model_dir, n_runs)
        self.logger.close()
        return {"last_10_performance": np.mean(last_10_performance)}

    def _checkpoint(self, where: str, n_runs: int) -> None:
        torch.save(self.policy.state_dict(), os.path.join(where, "policy.pth"))          # (run 0 when the policy carries several)
        if n_runs > 1:
            for r in range(n_runs):
                torch.save(self.policy.run_state_dict(r), os.path.join(where, f"policy_run{r}.pth"))

    def _score_env(self):
        e = self.eval_env
        while isinstance(e, (list, tuple)):
            e = e[0]
        return e

    def _evaluate(self) -> Dict[str, List[float]]:
        if isinstance(self.eval_env, (list, tuple)):
            return self._evaluate_batched(list(self.eval_env))
        self.policy.eval()
        obs = self.eval_env.reset()
        done_eps: List[Dict[str, float]] = []
        ep_reward, ep_len = 0, 0
        while len(done_eps) < self._eval_episodes:
            action = self.policy.select_action(obs.reshape(1, -1), deterministic=True)
            obs, reward, terminal, _ = self.eval_env.step(action.flatten())
            ep_reward += reward
            ep_len += 1
            if terminal:
                done_eps.append({"episode_reward": ep_reward, "episode_length": ep_len})
                ep_reward, ep_len = 0, 0
                obs = self.eval_env.reset()
        return {"eval/episode_reward": [d["episode_reward"] for d in done_eps],
                "eval/episode_length": [d["episode_length"] for d in done_eps]}

    def _env_groups(self, n_runs: int):
        """``eval_env`` given as one list of envs PER RUN ([[env, ...]] * n_runs) -> those groups, else None (sequential evaluation)"""
        ev = self.eval_env
        if isinstance(ev, (list, tuple)) and len(ev) == n_runs and all(isinstance(g, (list, tuple)) and len(g) > 0 for g in ev) \
                and len({len(g) for g in ev}) == 1 and hasattr(self.policy, "select_action_runs"):
            return [list(g) for g in ev]
        return None

    def _evaluate_runs_batched(self, groups) -> List[Dict[str, List[float]]]:
        """All runs of a multi-run policy evaluated together: run r owns the envs ``groups[r]``; every step is ONE run-batched
        deterministic forward over [n_runs, E, obs_dim] (``policy.select_action_runs``) instead of n_runs x E one-row forwards and
        n_runs full rollouts one after another.  Per run the episode accounting is ``_evaluate_batched``'s."""
        self.policy.eval()
        R, E = len(groups), len(groups[0])
        obs = [[np.asarray(env.reset(), dtype=np.float32).reshape(-1) for env in g] for g in groups]
        started = [min(E, self._eval_episodes)] * R
        active = [[i < started[r] for i in range(E)] for r in range(R)]
        ep_reward = [[0.0] * E for _ in range(R)]
        ep_len = [[0] * E for _ in range(R)]
        done_eps: List[List[Dict[str, float]]] = [[] for _ in range(R)]
        while any(len(d) < self._eval_episodes for d in done_eps):
            actions = self.policy.select_action_runs(np.stack([np.stack(o) for o in obs]))
            for r in range(R):
                for i in range(E):
                    if not active[r][i]:
                        continue
                    o, reward, terminal, _ = groups[r][i].step(np.asarray(actions[r, i]).flatten())
                    obs[r][i] = np.asarray(o, dtype=np.float32).reshape(-1)
                    ep_reward[r][i] += reward
                    ep_len[r][i] += 1
                    if terminal:
                        done_eps[r].append({"episode_reward": ep_reward[r][i], "episode_length": ep_len[r][i]})
                        ep_reward[r][i], ep_len[r][i] = 0.0, 0
                        if started[r] < self._eval_episodes:
                            started[r] += 1
                            obs[r][i] = np.asarray(groups[r][i].reset(), dtype=np.float32).reshape(-1)
                        else:
                            active[r][i] = False
        return [{"eval/episode_reward": [d["episode_reward"] for d in de], "eval/episode_length": [d["episode_length"] for d in de]}
                for de in done_eps]

    def _evaluate_batched(self, envs) -> Dict[str, List[float]]:
        """E envs in lockstep: one [E, obs_dim] deterministic forward per step instead of E one-row forwards.  Episode accounting
        as in the reference loop: an env that finishes an episode is reset and keeps running while episodes are still owed; the
        first ``eval_episodes`` episodes to START are the ones reported, in order of completion."""
        self.policy.eval()
        E = len(envs)
        obs = [env.reset() for env in envs]
        started = min(E, self._eval_episodes)
        active = [i < started for i in range(E)]
        ep_reward, ep_len = [0.0] * E, [0] * E
        done_eps: List[Dict[str, float]] = []
        while len(done_eps) < self._eval_episodes:
            idx = [i for i in range(E) if active[i]]
            batch = np.stack([np.asarray(obs[i], dtype=np.float32).reshape(-1) for i in idx])
            actions = self.policy.select_action(batch, deterministic=True)
            for j, i in enumerate(idx):
                o, reward, terminal, _ = envs[i].step(np.asarray(actions[j]).flatten())
                obs[i] = o
                ep_reward[i] += reward
                ep_len[i] += 1
                if terminal:
                    done_eps.append({"episode_reward": ep_reward[i], "episode_length": ep_len[i]})
                    ep_reward[i], ep_len[i] = 0.0, 0
                    if started < self._eval_episodes:
                        started += 1
                        obs[i] = envs[i].reset()
                    else:
                        active[i] = False
        return {"eval/episode_reward": [d["episode_reward"] for d in done_eps],
                "eval/episode_length": [d["episode_length"] for d in done_eps]}

"""EmbodiedRunner: the sync actor-learner loop (mirror of rlinf/runners/embodied_runner.py:52-66,163,187,478-563).

    for step in range(max_steps):
        set_global_step on actor / rollout
        update_rollout_weights            (actor.sync_model_to_rollout -> rollout.sync_model_from_actor)
        env.interact || rollout.generate || actor.recv_rollout_trajectories
        actor.compute_advantages_and_returns
        actor.run_training
        metrics

Embodied training speed is "total environment steps / iteration time" (rlinf_system.rst:107)."""

from __future__ import annotations

import time

import torch


class EmbodiedRunner:
    def __init__(self, cfg, actor, rollout, env, reward=None):
        self.cfg, self.actor, self.rollout, self.env, self.reward = cfg, actor, rollout, env, reward
        self.global_step = 0
        r = cfg.runner
        self.max_steps = r.get("max_epochs", 1) if r.get("max_steps", -1) in (-1, None) else r.max_steps
        self.weight_sync_interval = r.get("weight_sync_interval", 1)
        self.metrics_history: list = []

    def init_workers(self, share_weights: bool = True, env_tensors=None):
        self.actor.init_worker().wait()
        model = self.actor.worker.model if share_weights else None
        self.rollout.init_worker(model).wait()
        self.env.init_worker(env_tensors).wait()
        self.env.worker.connect(self.rollout.worker)

    def update_rollout_weights(self):
        weights = self.actor.sync_model_to_rollout().wait()[0]
        self.rollout.sync_model_from_actor(weights).wait()

    def run_step(self, eps=None) -> dict:
        dev = self.actor.worker.device
        t0 = time.perf_counter()
        self.actor.set_global_step(self.global_step)
        self.rollout.set_global_step(self.global_step)
        if self.global_step % self.weight_sync_interval == 0:
            self.update_rollout_weights()
        env_h = self.env.interact(eps)
        self.rollout.generate()
        trajs = self.env.send_rollout_trajectories(self.actor.worker._world_size).wait()[0]
        self.actor.recv_rollout_trajectories(trajs).wait()
        env_h.wait()
        rollout_metrics = self.actor.compute_advantages_and_returns().wait()[0]
        train_metrics = self.actor.run_training().wait()[0]
        if dev.type == "cuda":
            torch.cuda.synchronize(dev)
        dt = time.perf_counter() - t0
        tr = self.cfg.env.train
        env_steps = tr.total_num_envs * tr.max_steps_per_rollout_epoch * tr.get("rollout_epoch", 1)
        metrics = {f"rollout/{k}": v for k, v in rollout_metrics.items()}
        metrics.update({f"train/{k}": v for k, v in train_metrics.items()})
        metrics.update({"time/step": dt, "perf/env_steps_per_sec": env_steps / dt})
        self.metrics_history.append(metrics)
        self.global_step += 1
        return metrics

    def run(self, eps_fn=None):
        for _ in range(self.global_step, self.max_steps):
            self.run_step(None if eps_fn is None else eps_fn(self.global_step))
        return self.metrics_history

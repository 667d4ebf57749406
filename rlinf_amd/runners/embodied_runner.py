"""EmbodiedRunner: the sync actor-learner loop (mirror of rlinf/runners/embodied_runner.py:52-66 constructor, :163-185
init_workers + resume, :187-206 update_rollout_weights / evaluate, :308-329 _maybe_eval_and_checkpoint, :478-563 run,
:565-642 run_pipeline, :644-653 _save_checkpoint, :655-660 set_max_steps).

    for step in range(max_steps):
        set_global_step on actor / rollout
        update_rollout_weights            (rollout.sync_model_from_actor || actor.sync_model_to_rollout)
        env.interact || rollout.generate || actor.recv_rollout_trajectories
        actor.compute_advantages_and_returns
        actor.run_training  (|| env.prefetch_train_bootstrap with runner.overlap_env_bootstrap)
        evaluate every val_check_interval, checkpoint every save_interval
        metrics

The reference's constructor and method signatures are kept (channels included: rlinf_amd.scheduler.Channel is the in-process
FIFO), so ``examples/embodiment/train_embodied_agent.py`` of the reference drives this class unchanged.  ``run_step(eps)`` is
one iteration of that loop, exposed for the parity tests and bench.py (``eps`` injects the N(0,1) draws).
Embodied training speed is "total environment steps / iteration time" (rlinf_system.rst:107)."""

from __future__ import annotations

import os
import time

import torch

from ..scheduler import Channel
from ..utils.runner_utils import check_progress


class EmbodiedRunner:
    def __init__(self, cfg, actor, rollout, env, reward=None, critic=None):
        self.cfg, self.actor, self.rollout, self.env, self.reward, self.critic = cfg, actor, rollout, env, reward, critic
        r = cfg.runner
        self.weight_sync_interval = r.get("weight_sync_interval", 1)
        self.overlap_env_bootstrap = bool(r.get("overlap_env_bootstrap", False))
        self.env_channel = Channel.create("Env")
        self.rollout_channel = Channel.create("Rollout")
        self.actor_channel = Channel.create("Actor")
        self.reward_channel = Channel.create("Reward") if reward is not None else None
        self.consumed_samples = 0
        self.global_step = 0
        self.set_max_steps()
        self.metrics_history: list = []
        self.eval_history: list = []

    def set_max_steps(self):
        r = self.cfg.runner
        self.num_steps_per_epoch = 1
        self.max_steps = self.num_steps_per_epoch * r.get("max_epochs", 1)
        max_steps = r.get("max_steps", -1)
        if max_steps is not None and max_steps >= 0:
            self.max_steps = min(self.max_steps, max_steps)

    @property
    def epoch(self):
        return self.global_step // self.num_steps_per_epoch

    # ---- set-up -------------------------------------------------------------------------------------------------------
    def init_workers(self, share_weights: bool = True, env_tensors=None):
        """rollout, env, then the actor ("create worker in order to decrease the maximum memory usage", :163-170), then resume
        from ``runner.resume_dir`` (:172-185).  ``share_weights`` / ``env_tensors`` are this package's own extras: alias the
        learner's policy object in the collocated rollout worker (no copy on weight sync), inject pre-generated env tensors."""
        if not share_weights:
            self.cfg.rollout.share_actor_weights = False
        rollout_handle = self.rollout.init_worker()
        env_handle = self.env.init_worker(env_tensors)
        if self.reward is not None:
            self.reward.init_worker().wait()
        rollout_handle.wait()
        env_handle.wait()
        self.actor.init_worker().wait()
        self.env.worker.connect(self.rollout.worker)
        resume_dir = self.cfg.runner.get("resume_dir", None)
        if resume_dir is None:
            return
        actor_checkpoint_path = os.path.join(resume_dir, "actor")
        assert os.path.exists(actor_checkpoint_path), f"resume_dir {actor_checkpoint_path} does not exist."
        self.actor.load_checkpoint(actor_checkpoint_path).wait()
        self.global_step = int(resume_dir.split("global_step_")[-1])

    def close(self):
        """Tear the runner down: its workers leave the process-local peer registry (a later runner must not adopt them) and
        the actor's xGMI communicator is destroyed (its exported buffers are unmapped by the peers' own close)."""
        from ..workers.common import clear_peers
        clear_peers(self.actor, self.rollout, self.env)
        w = getattr(self.actor, "worker", self.actor)
        xg = getattr(w, "_xgmi", None)
        if xg is not None:
            if getattr(w, "device", None) is not None and w.device.type == "cuda":
                torch.cuda.synchronize(w.device)
            w._graph = None
            xg.close()
            w._xgmi = None

    def update_rollout_weights(self):
        rollout_handle = self.rollout.sync_model_from_actor()
        actor_handle = self.actor.sync_model_to_rollout()
        actor_handle.wait()
        rollout_handle.wait()

    def evaluate(self) -> dict:
        from ..utils.metric_utils import compute_evaluate_metrics
        env_handle = self.env.evaluate(input_channel=self.env_channel, rollout_channel=self.rollout_channel)
        rollout_handle = self.rollout.evaluate(input_channel=self.rollout_channel, output_channel=self.env_channel)
        env_results = env_handle.wait()
        rollout_handle.wait()
        return compute_evaluate_metrics([r for r in env_results if r is not None])

    def _save_checkpoint(self):
        log = self.cfg.runner.get("logger", None) or {}
        base_output_dir = os.path.join(log.get("log_path", "logs"), log.get("experiment_name", "default"),
                                       f"checkpoints/global_step_{self.global_step}")
        actor_save_path = os.path.join(base_output_dir, "actor")
        os.makedirs(actor_save_path, exist_ok=True)
        self.actor.save_checkpoint(actor_save_path, self.global_step).wait()
        return base_output_dir

    def _maybe_eval_and_checkpoint(self, step: int) -> dict:
        r = self.cfg.runner
        run_val, save_model, _ = check_progress(self.global_step, self.max_steps, r.get("val_check_interval", -1),
                                                r.get("save_interval", -1), 1.0, run_time_exceeded=False)
        eval_metrics = {}
        if run_val:
            self.update_rollout_weights()
            eval_metrics = {f"eval/{k}": v for k, v in self.evaluate().items()}
            self.eval_history.append((step, eval_metrics))
        if save_model:
            self._save_checkpoint()
        return eval_metrics

    # ---- one iteration ------------------------------------------------------------------------------------------------
    def run_step(self, eps=None) -> dict:
        dev = self.actor.worker.device
        step = self.global_step
        t0 = time.perf_counter()
        self.actor.set_global_step(self.global_step).wait()
        self.rollout.set_global_step(self.global_step).wait()
        if step % self.weight_sync_interval == 0:
            self.update_rollout_weights()
        env_handle = self.env.interact(input_channel=self.env_channel, rollout_channel=self.rollout_channel,
                                       reward_channel=self.reward_channel, actor_channel=self.actor_channel, eps=eps)
        rollout_handle = self.rollout.generate(input_channel=self.rollout_channel, output_channel=self.env_channel)
        self.actor.recv_rollout_trajectories(input_channel=self.actor_channel).wait()
        rollout_handle.wait()
        env_handle.wait()
        rollout_metrics = self.actor.compute_advantages_and_returns().wait()[0]
        actor_training_handle = self.actor.run_training()
        env_bootstrap_handle = None
        if self.overlap_env_bootstrap and step + 1 < self.max_steps:
            env_bootstrap_handle = self.env.prefetch_train_bootstrap(rollout_channel=self.rollout_channel)
        train_metrics = actor_training_handle.wait()[0]
        if env_bootstrap_handle is not None:
            env_bootstrap_handle.wait()
        if not rollout_metrics and hasattr(self.actor.worker, "pop_rollout_metrics"):
            rollout_metrics = self.actor.worker.pop_rollout_metrics()  # pipeline learner: produced inside run_training
        if dev is not None and dev.type == "cuda":
            torch.cuda.synchronize(dev)
        dt = time.perf_counter() - t0
        self.global_step += 1
        eval_metrics = self._maybe_eval_and_checkpoint(step)
        tr = self.cfg.env.train
        env_steps = tr.total_num_envs * tr.max_steps_per_rollout_epoch * tr.get("rollout_epoch", 1)
        metrics = {f"rollout/{k}": v for k, v in rollout_metrics.items()}
        metrics.update({f"train/{k}": v for k, v in train_metrics.items()})
        metrics.update(eval_metrics)
        metrics.update({"time/step": dt, "perf/env_steps_per_sec": env_steps / dt})
        self.metrics_history.append(metrics)
        return metrics

    def run(self, eps_fn=None):
        if self.cfg.runner.get("use_training_pipeline", False):
            return self.run_pipeline(eps_fn)
        for _ in range(self.global_step, self.max_steps):
            self.run_step(None if eps_fn is None else eps_fn(self.global_step))
        return self.metrics_history

    def run_pipeline(self, eps_fn=None):
        """runner.use_training_pipeline (:565-642): the learner consumes the rollout while it is produced.  The data path
        (statistics normalisation, per-stage shuffles, global-batch composition) lives in the actor worker; one iteration is
        the same call sequence."""
        for _ in range(self.global_step, self.max_steps):
            self.run_step(None if eps_fn is None else eps_fn(self.global_step))
        return self.metrics_history

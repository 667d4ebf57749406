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
from ..utils.pending import resolve
from ..utils.runner_utils import check_progress


class PendingStep:
    """One queued iteration: ``result()`` waits for ITS metric copies (not for the device) and returns the step's metric dict."""

    def __init__(self, finish):
        self._finish, self._value = finish, None

    def result(self) -> dict:
        if self._value is None:
            self._value = self._finish()
            self._finish = None
        return self._value


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
        self._last_landed = None  # host time at which the previous deferred step's metrics landed
        # split placement (utils/placement.py): a group that lives on other ranks is an absent handle here
        self._has = {name: getattr(h, "worker", None) is not None for name, h in (("actor", actor), ("rollout", rollout), ("env", env))}
        self.placement = next((getattr(getattr(h, "worker", None), "placement", None) for h in (actor, rollout, env)
                               if getattr(h, "worker", None) is not None), None)
        self.split = self.placement is not None and bool(getattr(self.placement, "split", False))
        self._landing = None  # the learner rank's persistent landing buffer for shipped trajectories

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
        if self._has["env"]:
            self.env.worker.connect(self.rollout.worker)
        if self.split:
            assert not self.cfg.runner.get("use_training_pipeline", False), \
                "runner.use_training_pipeline with a split placement is not served (per-epoch hand-off over the trajectory route)"
        resume_dir = self.cfg.runner.get("resume_dir", None)
        if resume_dir is None or not self._has["actor"]:
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

    def _ship_trajectories(self):
        """Split placement: the finished trajectory buffer of rollout rank i -> learner rank i (the reference's actor channel,
        env_worker.py:1463-1467 -> embodied_fsdp_actor_worker.py:186-207, is a Ray channel between actor processes).  The fields
        travel in a fixed order into a persistent landing buffer of the same shape; the learner then finds the usual trajectory
        views on its channel."""
        from ..data.embodied_types import TrajectoryBuffer
        from ..scheduler.dist import recv_tensors, send_tensors
        fields = ("states", "actions", "prev_logprobs", "versions", "prev_values", "rewards", "dones", "terminations", "truncations")
        if self._has["env"]:
            w = self.env.worker
            while not self.actor_channel.empty():
                self.actor_channel.get()  # (the views stay on this rank: the buffer itself is what travels)
            send_tensors([getattr(w.buffer, f) for f in fields], self.placement.peer_of("env", "actor"))
        if self._has["actor"]:
            a, cfg = self.actor.worker, self.cfg
            m, tr = cfg.actor.model, cfg.env.train
            if self._landing is None:
                stage_num = int(cfg.rollout.get("pipeline_stage_num", 1))
                envs = tr.total_num_envs // a._world_size // stage_num * stage_num * int(tr.get("rollout_epoch", 1))
                self._landing = TrajectoryBuffer(tr.max_steps_per_rollout_epoch // int(m.num_action_chunks), envs, m.obs_dim,
                                                 m.action_dim, m.num_action_chunks, device=a.device,
                                                 max_episode_length=int(tr.get("max_episode_steps", 0)))
                self._landing_split = (a._world_size * stage_num, stage_num)
            recv_tensors([getattr(self._landing, f) for f in fields], self.placement.peer_of("actor", "env"))
            from ..scheduler import compute_split_num
            send_num, stage_num = self._landing_split
            split = compute_split_num(a._world_size, send_num)
            for traj in self._landing.to_splited_trajectories(split * stage_num):
                self.actor_channel.put(traj)

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
    def run_step(self, eps=None, defer: bool = False):
        """One iteration with the reference's call sequence.  ``defer`` (this package's own): queue the whole iteration on the
        device and return a ``PendingStep`` instead of its metric dict -- the numbers are copied to pinned host memory behind the
        kernels that produce them and read when ``.result()`` is called.  ``run()`` calls it one iteration late, so the host
        never waits with an empty device queue (no idle between the rollout, the update phase and the next rollout)."""
        dev = (self.actor.worker if self._has["actor"] else self.rollout.worker).device
        step = self.global_step
        t0 = time.perf_counter()
        worker = getattr(self.actor, "worker", self.actor)
        if hasattr(worker, "defer_host_reads"):
            worker.defer_host_reads = bool(defer)
        self.actor.set_global_step(self.global_step).wait()
        self.rollout.set_global_step(self.global_step).wait()
        if step % self.weight_sync_interval == 0:
            self.update_rollout_weights()
        env_handle = self.env.interact(input_channel=self.env_channel, rollout_channel=self.rollout_channel,
                                       reward_channel=self.reward_channel, actor_channel=self.actor_channel, eps=eps)
        rollout_handle = self.rollout.generate(input_channel=self.rollout_channel, output_channel=self.env_channel)
        if self.split:
            self._ship_trajectories()
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
        if not rollout_metrics and self._has["actor"] and hasattr(self.actor.worker, "pop_rollout_metrics"):
            rollout_metrics = self.actor.worker.pop_rollout_metrics()  # pipeline learner: produced inside run_training
        self.global_step += 1
        tr = self.cfg.env.train
        env_steps = tr.total_num_envs * tr.max_steps_per_rollout_epoch * tr.get("rollout_epoch", 1)

        def finish() -> dict:
            rm, tm = resolve(rollout_metrics) or {}, resolve(train_metrics) or {}  # waits for this step's copies only
            # (a rollout-only rank of a split placement has neither)
            now = time.perf_counter()
            # deferred: the iteration time is the time between two consecutive steps landing (the steady-state rate)
            dt = (now - self._last_landed) if (defer and self._last_landed is not None) else now - t0
            self._last_landed = now
            metrics = {f"rollout/{k}": v for k, v in rm.items()}
            metrics.update({f"train/{k}": v for k, v in tm.items()})
            metrics.update(eval_metrics)
            metrics.update({"time/step": dt, "perf/env_steps_per_sec": env_steps / dt})
            self.metrics_history.append(metrics)
            return metrics

        if not defer:
            if dev is not None and dev.type == "cuda":
                torch.cuda.synchronize(dev)
            eval_metrics = self._maybe_eval_and_checkpoint(step)
            return finish()
        eval_metrics = self._maybe_eval_and_checkpoint(step)  # (evaluation / checkpoints read the device: they wait by themselves)
        return PendingStep(finish)

    def run(self, eps_fn=None):
        """``runner.defer_metrics`` (default true): iterations are queued one ahead of the metrics being read (see run_step)."""
        if self.cfg.runner.get("use_training_pipeline", False):
            return self.run_pipeline(eps_fn)
        return self._run_ahead(eps_fn)

    def iter_steps(self, eps_fn=None):
        """The loop as a generator: yields every iteration's metric dict, in order.  With ``runner.defer_metrics`` (default true)
        iteration i + 1 is queued on the device before iteration i's dict is yielded: the numbers are read one iteration late and
        the device never waits for the host (measured at the benchmark shape, one process, alternating: 9.27-9.33 ms per
        iteration reading each step's numbers first, 8.89-8.92 ms this way; profiles/r03_run_ahead_loop_ab_same_process.txt)."""
        defer = bool(self.cfg.runner.get("defer_metrics", True))
        pending = None
        for _ in range(self.global_step, self.max_steps):
            step = self.run_step(None if eps_fn is None else eps_fn(self.global_step), defer=defer)
            if not defer:
                yield step
                continue
            if pending is not None:
                yield pending.result()
            pending = step
        if pending is not None:
            yield pending.result()

    def _run_ahead(self, eps_fn=None):
        for _ in self.iter_steps(eps_fn):
            pass
        return self.metrics_history

    def run_pipeline(self, eps_fn=None):
        """runner.use_training_pipeline (:565-642): the learner consumes the rollout while it is produced.  The data path
        (statistics normalisation, per-stage shuffles, global-batch composition) lives in the actor worker; one iteration is
        the same call sequence."""
        return self._run_ahead(eps_fn)

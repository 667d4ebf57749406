"""EnvWorker: steps the (synthetic) simulator, adds bootstrap rewards and fills the trajectory buffer
(mirror of rlinf/workers/env/env_worker.py: shard arithmetic :137-140, compute_bootstrap_rewards :718-758,
_run_interact_once :1058-1306, send_rollout_trajectories :1026).

Row alignment: row t of the buffer holds the policy outputs for observation t and the env
outputs (reward, dones) of the env step taken WITH action t, i.e. rewards[t] pairs with values[t], values[t+1]
and dones[t+1]; row 0 of dones is the all-False bootstrap row; one extra value row closes the buffer.
"""

from __future__ import annotations

import torch

from ... import ops
from ...data import TrajectoryBuffer
from ...envs.synthetic_env import SyntheticManiSkillEnv, generate_tensors
from ...scheduler.placement import compute_split_num, env_shard
from ..common import Worker, peer


class _null:
    def __enter__(self):
        return None

    def __exit__(self, *a):
        return False


class EnvWorker(Worker):
    ROLE = "env"

    def __init__(self, cfg, ctx=None):
        super().__init__(cfg, ctx)
        self.train_cfg = cfg.env.train
        # rollout.pipeline_stage_num (env_worker.py:137-140): the reference splits a rank's envs into stages so that one
        # stage's simulator step overlaps another stage's policy forward.  Rows are independent and a rank's stages are
        # contiguous env blocks (env_shard), so here the stages of a step ride in ONE policy launch over all of them; only
        # the trajectory hand-off keeps the per-stage split the learner counts on (send_num = world * stage_num).
        self.stage_num = int(cfg.rollout.get("pipeline_stage_num", 1))
        self.stage_envs = self.train_cfg.total_num_envs // self._world_size // self.stage_num
        self.num_envs = self.stage_envs * self.stage_num
        self.n_train_chunk_steps = (self.train_cfg.max_steps_per_rollout_epoch // cfg.actor.model.num_action_chunks)
        self.rollout_epoch = self.train_cfg.get("rollout_epoch", 1)
        self.gamma = float(cfg.algorithm.get("gamma", 1))
        self.bootstrap_type = cfg.algorithm.get("bootstrap_type", "standard")
        self.auto_reset = bool(self.train_cfg.get("auto_reset", False))
        # rollout.enable_cuda_graph is the reference's switch (huggingface_worker.py / mlp_policy.py:344-440 capture
        # _generate_actions); here the WHOLE T-step loop is one hipGraph, because every launch works on fixed rows.
        self.use_graph = bool(cfg.rollout.get("enable_cuda_graph", False))
        self.env = None
        self.buffer = None
        self.rollout = None
        self._graph = None
        self._eps = None
        self._prefetched_train_bootstrap = None
        self._eval_env = None
        # runner.use_training_pipeline with rollout_epoch > 1 (embodied_runner.py:565-642, env_worker.py:1074,1324-1330): the
        # reference hands every finished rollout epoch to the learner at once and rolls the next one out while the learner
        # trains on it.  Here: one trajectory buffer per epoch, the rollout loop on its own HIP stream with an event per epoch;
        # the learner's stream waits on epoch e's event only (runner.pipeline_overlap: true; the default keeps everything on one
        # stream).  Round 5 measured what the second stream buys on one MI355X (profiles/r05_pipeline_overlap_*.txt): nothing.
        # The kernel trace shows 0.0 ms with kernels of both queues in flight -- every launch of either chain wants whole CUs (the
        # fused step holds all 512 VGPRs per SIMD on every CU, the rollout step 41 KB of LDS and ~180 VGPRs per wave), so the
        # hardware runs the two chains one kernel at a time; E = 4 overlapped costs 9.01 ms per iteration, on one stream 9.01 ms
        # (synchronous mode 8.66 ms), and a rollout stream restricted to 16 / 32 / 64 CUs 13.6-14.5 ms.  Hence: off by default.
        self.pipeline_epochs = bool(cfg.runner.get("use_training_pipeline", False)) and self.rollout_epoch > 1
        self.overlap = self.pipeline_epochs and bool(cfg.runner.get("pipeline_overlap", False))
        self.buffers: list = []
        self.epoch_events: list = []
        self._rollout_stream = None
        self._epoch_graphs: dict = {}

    def init_worker(self, env_tensors: dict | None = None):
        m = self.cfg.actor.model
        begin, _ = env_shard(self.train_cfg.total_num_envs, self._world_size, self.stage_num, self._rank, 0)
        _, end = env_shard(self.train_cfg.total_num_envs, self._world_size, self.stage_num, self._rank, self.stage_num - 1)
        if env_tensors is None:
            # the synthetic horizon is counted in ENV steps: num_action_chunks of them per chunk step
            env_tensors = generate_tensors(int(self.train_cfg.get("seed", 0)),
                                           self.n_train_chunk_steps * self.rollout_epoch * int(m.num_action_chunks),
                                           self.train_cfg.total_num_envs, m.obs_dim,
                                           int(self.train_cfg.get("max_episode_steps", 50)),
                                           mode=self.train_cfg.get("synthetic_done_mode", "periodic"))
        self.env = SyntheticManiSkillEnv(env_tensors, self.device, m.num_action_chunks, self.auto_reset, slice(begin, end))
        mel = int(self.train_cfg.get("max_episode_steps", 0))
        if self.pipeline_epochs:  # one contiguous buffer per rollout epoch: each is a complete batch for the learner on its own
            self.buffers = [TrajectoryBuffer(self.n_train_chunk_steps, self.num_envs, m.obs_dim, m.action_dim, m.num_action_chunks,
                                             device=self.device, max_episode_length=mel) for _ in range(self.rollout_epoch)]
            self.buffer = self.buffers[0]
            if self.device.type == "cuda":
                self.epoch_events = [torch.cuda.Event() for _ in range(self.rollout_epoch)]
                if self.overlap:
                    # RLX_ROLLOUT_CUS / runner.rollout_cus: the rollout stream dispatches to that many CUs only (utils/streams.py)
                    import os
                    from ...utils.streams import rollout_stream
                    n_cus = int(os.environ.get("RLX_ROLLOUT_CUS", "0") or 0) or int(self.cfg.runner.get("rollout_cus", 0) or 0)
                    self._rollout_stream, self._rollout_stream_owner = rollout_stream(self.device, n_cus or None)
        else:
            # rollout epochs are laid out side by side on the batch axis: the learner's fold (a8) costs nothing
            self.buffer = TrajectoryBuffer(self.n_train_chunk_steps, self.num_envs * self.rollout_epoch, m.obs_dim, m.action_dim,
                                           m.num_action_chunks, device=self.device, max_episode_length=mel)
        self._bootstrap_v = torch.zeros(self.num_envs, self.buffer.V, device=self.device)

    def connect(self, rollout):
        """In-process stand-in for the env<->rollout channels (env_worker.py:964-984,1120-1129)."""
        self.rollout = rollout

    def compute_bootstrap_rewards(self, rewards, dones, truncations, bootstrap_values):
        """r[:, -1] += gamma * V(final_obs) where the env finished (env_worker.py:718-758); in place on the buffer row."""
        if rewards is None or not self.auto_reset:
            return rewards
        flags = dones if self.bootstrap_type == "always" else truncations
        return ops.bootstrap_rewards_(rewards, flags, bootstrap_values, self.gamma)

    def interact(self, input_channel=None, rollout_channel=None, reward_channel=None, actor_channel=None,
                 eps: torch.Tensor | None = None, mode: str = "train"):
        """``rollout_epoch`` epochs of T chunk steps + a closing value row each.  ``eps`` [rollout_epoch * T, B, A] injects
        the N(0,1) draws (drawn on the device otherwise).

        The reference's signature (env_worker.py:1058: four channels): observations and actions travel over
        ``rollout_channel`` / ``input_channel`` there, the finished trajectories over ``actor_channel``.  In-process the
        policy is called directly; with an ``actor_channel`` the trajectory views are put on it for
        ``actor.recv_rollout_trajectories(input_channel=...)``.  This package's own callers pass ``eps`` as the first
        positional argument (a tensor, or None)."""
        if isinstance(input_channel, torch.Tensor):  # interact(eps): the package's own call style
            eps, input_channel = input_channel, None
        if self.rollout is None:
            self.rollout = peer("rollout", self.cfg)
        self._prefetched_train_bootstrap = None  # consumed: the first observation batch is resident (env.reset(0))
        out = self._interact(eps, mode)
        if actor_channel is not None:
            for traj in self.send_rollout_trajectories():
                actor_channel.put(traj)
        return out

    def prefetch_train_bootstrap(self, rollout_channel=None) -> None:
        """"Prepare and send the first env batch for the next training rollout" (env_worker.py:999-1009, behind
        runner.overlap_env_bootstrap): overlaps the simulator reset with the learner's update.  The synthetic env's
        observations are resident tensors, so there is nothing to prepare; the call keeps the reference's contract (one
        outstanding prefetch, consumed by interact())."""
        if self._prefetched_train_bootstrap is not None:
            raise RuntimeError("A prefetched train bootstrap already exists. "
                               "Call interact() to consume it before prefetching again.")
        self._prefetched_train_bootstrap = True

    def _interact(self, eps: torch.Tensor | None = None, mode: str = "train"):
        if self._eps is None:
            self._eps = torch.empty(self.n_train_chunk_steps * self.rollout_epoch, self.num_envs, self.buffer.A,
                                    device=self.device)
        if eps is None:
            self._eps.normal_()
        else:
            self._eps.copy_(eps)
        # PolicyOutput.versions = full_like(prev_logprobs, version) (huggingface_worker.py:579-610): the weights do not change
        # inside a rollout, so one fill per rollout -- outside the captured step loop, the value changes every iteration
        if self.pipeline_epochs:
            return self._interact_pipeline_epochs(self._eps, mode)
        self.buffer.versions.fill_(float(self.rollout.version))
        if not (self.use_graph and mode == "train" and self.device.type == "cuda"):
            return self._interact_eager(self._eps, mode)
        if self._graph is None:
            self._interact_eager(self._eps, mode)  # real run; also warms every allocation
            torch.cuda.synchronize(self.device)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                self._interact_eager(self._eps, mode)
            self._graph = g
            return None
        self._graph.replay()
        return None

    def _interact_pipeline_epochs(self, eps: torch.Tensor, mode: str = "train"):
        """Epoch after epoch into its own buffer, on the rollout stream when overlapping; event e marks "epoch e is complete"
        (all its rows incl. the bootstrap folds and the closing value row).  With rollout.enable_cuda_graph every epoch is one
        hipGraph (captured on the stream it replays on)."""
        cuda = self.device.type == "cuda"
        main = torch.cuda.current_stream(self.device) if cuda else None
        rs = self._rollout_stream if (self.overlap and cuda) else main
        if cuda and rs is not main:
            rs.wait_stream(main)  # weight sync, noise copy, the previous iteration's reads of these buffers
        ctx = torch.cuda.stream(rs) if (cuda and rs is not main) else _null()
        with ctx:
            for e, buf in enumerate(self.buffers):
                buf.versions.fill_(float(self.rollout.version))
                if self.use_graph and mode == "train" and cuda:
                    g = self._epoch_graphs.get(e)
                    if g is None:
                        self._run_epoch(e, eps, mode)  # real run; also warms every allocation
                        torch.cuda.synchronize(self.device)
                        g = torch.cuda.CUDAGraph()
                        with torch.cuda.graph(g, stream=rs if rs is not main else None):
                            self._run_epoch(e, eps, mode)
                        self._epoch_graphs[e] = g
                    else:
                        g.replay()
                else:
                    self._run_epoch(e, eps, mode)
                if cuda:
                    self.epoch_events[e].record(rs)
        return None

    def _run_epoch(self, epoch: int, eps: torch.Tensor, mode: str = "train"):
        """One rollout epoch of the pipeline layout into buffers[epoch] (same launches as _interact_eager's inner loop)."""
        buf, env, ro = self.buffers[epoch], self.env, self.rollout
        buf.reset()
        T = self.n_train_chunk_steps
        cols = slice(None)
        obs, _ = env.reset(epoch * T)
        for t in range(T):
            chunk_actions = ro.predict(obs, out=buf.policy_rows(t, cols), eps=None if eps is None else eps[epoch * T + t],
                                       mode=mode, states_copy=buf.states[t, cols])
            obs, rewards, term, trunc, infos = env.chunk_step(chunk_actions)
            r_row, d_row, te_row, tr_row = buf.env_rows(t, cols)
            if self.auto_reset:
                ro.queue_bootstrap(infos["final_obs"], r_row, None, self.gamma, env=(rewards, term, trunc),
                                   rows=(d_row, te_row, tr_row), flag_is_truncation=self.bootstrap_type != "always")
            else:
                ops.store_env_rows_(rewards, term, trunc, r_row, d_row, te_row, tr_row)
        ro.get_bootstrap_values(obs, out=buf.prev_values[T, cols])

    def _interact_eager(self, eps: torch.Tensor, mode: str = "train"):
        """ONE launch per chunk step with auto_reset: the fused policy launch also carries the value job that stores the
        previous step's env outputs and folds its bootstrap value into the reward row.  Without auto_reset (no bootstrap)
        a second, tiny launch stores the env outputs."""
        buf, env, ro = self.buffer, self.env, self.rollout
        buf.reset()
        T = self.n_train_chunk_steps
        with self.timer("env/interact"):
            for epoch in range(self.rollout_epoch):  # env_worker.py:1074
                cols = slice(epoch * self.num_envs, (epoch + 1) * self.num_envs)
                obs, _ = env.reset(epoch * T)  # bootstrap_step (:898-952): dones row 0 of this block stays all-False
                for t in range(T):
                    # forward_inputs.states of step t (return_obs=True) are written by the policy launch itself
                    chunk_actions = ro.predict(obs, out=buf.policy_rows(t, cols),
                                               eps=None if eps is None else eps[epoch * T + t], mode=mode,
                                               states_copy=buf.states[t, cols])
                    obs, rewards, term, trunc, infos = env.chunk_step(chunk_actions)
                    r_row, d_row, te_row, tr_row = buf.env_rows(t, cols)
                    if self.auto_reset:  # value of the true terminal observation enters through the reward (A.2); the job
                        # that computes it (riding in the next policy launch) also stores this step's env outputs
                        ro.queue_bootstrap(infos["final_obs"], r_row, None, self.gamma, env=(rewards, term, trunc),
                                           rows=(d_row, te_row, tr_row), flag_is_truncation=self.bootstrap_type != "always")
                    else:
                        ops.store_env_rows_(rewards, term, trunc, r_row, d_row, te_row, tr_row)
                ro.get_bootstrap_values(obs, out=buf.prev_values[T, cols])  # closing row of the epoch: values only
        return None

    def evaluate(self, input_channel=None, rollout_channel=None) -> dict:
        """EnvWorker.evaluate (env_worker.py:1374-1461): ``eval_rollout_epoch`` epochs of ``n_eval_chunk_steps`` on the eval
        envs with the policy acting in eval mode (its mean, mlp_policy.py:230-236); the per-episode records of the envs that
        finished (ManiSkill's ``final_info["episode"]``: return / episode_len / reward, maniskill_env.py:300-325) are
        returned as tensors for compute_evaluate_metrics.  Nothing is written to the training buffer."""
        ev = self.cfg.env.get("eval", None) or self.train_cfg
        m = self.cfg.actor.model
        if self.rollout is None:
            self.rollout = peer("rollout", self.cfg)
        steps = int(ev.get("max_steps_per_rollout_epoch", self.train_cfg.max_steps_per_rollout_epoch)) // m.num_action_chunks
        epochs = int(ev.get("rollout_epoch", 1))
        total = int(ev.get("total_num_envs", self.train_cfg.total_num_envs))
        auto_reset = bool(ev.get("auto_reset", True))
        if self._eval_env is None:
            begin, end = (total // self._world_size) * self._rank, (total // self._world_size) * (self._rank + 1)
            tensors = generate_tensors(int(ev.get("seed", 0)) + 1_000_003, steps * epochs * int(m.num_action_chunks), total, m.obs_dim,
                                       int(ev.get("max_episode_steps", self.train_cfg.get("max_episode_steps", 50))),
                                       mode=ev.get("synthetic_done_mode", "periodic"))
            self._eval_env = SyntheticManiSkillEnv(tensors, self.device, m.num_action_chunks, auto_reset, slice(begin, end))
        env, ro = self._eval_env, self.rollout
        n = env.num_envs
        ret = torch.zeros(n, device=self.device)
        length = torch.zeros(n, device=self.device)
        prev_done = torch.zeros(n, dtype=torch.bool, device=self.device)
        records: dict = {"return": [], "episode_len": [], "reward": []}
        for epoch in range(epochs):
            if not auto_reset or epoch == 0:
                obs, _ = env.reset(epoch * steps)
                prev_done.zero_()
            for _ in range(steps):
                actions = ro.predict(obs, eps=None, mode="eval")
                obs, rewards, term, trunc, _ = env.chunk_step(actions)
                ret += rewards.sum(dim=1)
                length += rewards.shape[1]
                done = (term | trunc).any(dim=1)
                newly = done if auto_reset else done & ~prev_done
                prev_done |= done
                if bool(newly.any()):
                    records["return"].append(ret[newly].cpu())
                    records["episode_len"].append(length[newly].cpu())
                    records["reward"].append((ret[newly] / length[newly]).cpu())
                    if auto_reset:
                        ret[newly], length[newly] = 0.0, 0.0
        ro.flush_bootstrap()
        return {k: torch.cat(v, dim=0).contiguous() for k, v in records.items() if v}

    def send_rollout_trajectories(self, actor_world_size: int | None = None) -> list:
        """to_splited_trajectories(actor_split_num) (env_worker.py:1026,1463-1467): views, no copies."""
        split = compute_split_num(actor_world_size or self._world_size, self._world_size * self.stage_num)
        if self.pipeline_epochs:  # epoch-major: the learner takes `split * stage_num` trajectories per rollout epoch
            return [t for buf in self.buffers for t in buf.to_splited_trajectories(split * self.stage_num)]
        return self.buffer.to_splited_trajectories(split * self.stage_num)

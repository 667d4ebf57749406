"""EmbodiedFSDPActor: the PPO/GRPO learner (mirror of rlinf/workers/actor/embodied_fsdp_actor_worker.py:
recv_rollout_trajectories :186-207, _process_received_rollout_batch :209-284, compute_advantages_and_returns
:286-321, run_training :483-589, train_micro_batch :591-700) in its NO_SHARD data-parallel form
(rlinf/hybrid_engines/fsdp: optimizer_step fsdp_model_manager.py:429-463, two-group AdamW :501-590).

One optimizer step is a fixed chain of HIP launches on preallocated buffers -- no autograd graph, no
``.item()``: [forward + loss + metrics + backward-data, one launch] -> [split-K weight gradients + head gradients +
metric row, one launch] -> [RCCL all-reduce of one flat buffer] -> clip + AdamW (two launches).  Metrics of every
step land in one device matrix read back once per ``run_training``.  Because the chain is static it can be captured in a
hipGraph (``actor.enable_hip_graph``) and replayed: the step counter lives on the device for that reason.
"""

from __future__ import annotations

import numpy as np
import torch

from ... import ops
from ..._lib import PPO_OUT_FLOATS, PPO_OUT_NAMES
from ...algorithms import calculate_adv_and_returns
from ...algorithms.losses import EV_PREFIX, _ACTOR_KEYS, _CRITIC_KEYS, _EV_MAP, explained_variance_from_stats
from ...data import convert_trajectories_to_batch
from ...models import get_model
from ...scheduler.dist import all_reduce_flat_, all_reduce_scalars
from ...scheduler.placement import compute_split_num, minibatch_plan
from ...utils.lr_scheduler import LearnerLRScheduler
from ...utils.pending import PendingMetrics
from ..common import Worker

CRITIC_EXPLAINED_VARIANCE_KEY = "critic/explained_variance"


def _host_randperm(n: int, generator: torch.Generator) -> torch.Tensor:
    """torch.randperm on the host with the caller's generator -- the reference's own call (nested_dict_process.py:272-285,
    env_worker.py:1519-1537), so the order is ITS order -- on ONE thread: the permutation itself is a serial Fisher-Yates walk of
    the mt19937 stream (randperm_cpu), but torch first fills the arange through at::parallel_for, and above its 32768-element
    grain that wakes the whole intra-op pool: measured 57-65 ms per call for 131072 elements against 0.9 ms single-threaded,
    identical result (tests/test_end_to_end.py::test_single_thread_randperm_is_torchs_randperm)."""
    old = torch.get_num_threads()
    if old == 1:
        return torch.randperm(n, generator=generator)
    torch.set_num_threads(1)
    try:
        return torch.randperm(n, generator=generator)
    finally:
        torch.set_num_threads(old)


class EmbodiedFSDPActor(Worker):
    ROLE = "actor"

    def __init__(self, cfg, ctx=None):
        super().__init__(cfg, ctx)
        self.model = None
        self.rollout_batch: dict = {}
        self.version = 0
        self.optimizer_steps = 0
        a = cfg.actor
        # cfg.weight_syncer (embodied_fsdp_actor_worker.py:80-83,97): bucket or patch; None when the configuration names none (a
        # collocated rollout worker that aliases this learner's policy object needs no transport at all)
        from ..weight_link import weight_syncer_config
        from ...hybrid_engines.weight_syncer import WeightSyncer
        ws_cfg = weight_syncer_config(cfg)
        self.weight_syncer = WeightSyncer.create(ws_cfg) if ws_cfg is not None else None
        self._is_weight_sender = self._rank == 0
        self.gradient_accumulation = a.global_batch_size // a.micro_batch_size // self._world_size  # :91-95
        self.critic_warmup_steps = int(a.optim.get("critic_warmup_steps", 0))
        self.enable_hip_graph = bool(a.get("enable_hip_graph", False))
        # rlx_ppo_step (fused forward + loss + backward); False = the stage-by-stage entry points
        self.fused_step = bool(a.get("fused_step", True))
        # let the optimizer kernel scatter the new weights into the fragment-tile image (instead of one re-pack launch per step)
        self.optimizer_writes_tiles = bool(a.get("optimizer_writes_tiles", True))
        self._graph = None
        self._graph_key = None
        self._lr_log: list = []
        # runner.use_training_pipeline (embodied_runner.py:479, env_worker.py:1469-1572, fsdp_actor_worker_pipeline.py): the
        # env side normalises advantages from (count, sum, sumsq) statistics and hands the learner per-stage shuffled
        # micro-batches.  With rollout and learner sharing the resident buffer nothing streams; what remains is the data
        # path: that normalisation, the per-stage shuffles with the rank's stateful generator, fixed global batches.
        self.use_training_pipeline = bool(cfg.runner.get("use_training_pipeline", False))
        self.pipeline_epochs = 1
        self.rollout_batches: list = []
        self._deferred_rollout_metrics: dict = {}
        # True (set by the runner's run-ahead loop): metric dicts come back as PendingMetrics, read one iteration late
        self.defer_host_reads = False
        if self.use_training_pipeline:
            assert cfg.algorithm.adv_type == "gae", ("algorithm.adv_type only supports 'gae' now"
                                                     "when runner.use_training_pipeline is True.")  # config.py:992-996
            # rollout_epoch > 1: every epoch is its own batch (prepare_pipeline_batch, :1512; sent per epoch, :1324-1330) with
            # its own statistics normalisation and stage shuffles; the learner trains on epoch e while epoch e + 1 rolls out
            self.pipeline_epochs = int(cfg.env.train.get("rollout_epoch", 1))
            # _init_pipeline_params (:355-361): seed = actor.seed + actor_rank + env_rank * actor_world_size, and with one
            # env rank feeding the learner rank of the same process both ranks are this rank
            self._pipe_gen = torch.Generator().manual_seed(int(a.get("seed", 1234)) + self._rank + self._rank * self._world_size)
            self._pipe_shuffle = bool(cfg.algorithm.get("shuffle_rollout", True))
        self._perm_prefetch: dict = {}
        self._perm_prefetch_state = None

    # ---- set-up ---------------------------------------------------------------------------------------------
    def init_worker(self):
        a = self.cfg.actor
        seed = int(a.get("seed", 1234))
        torch.manual_seed(seed)  # every rank builds identical initial weights
        self.model = get_model(a.model).to(self.device)
        n = self.model.n_params
        dev = self.device
        self.exp_avg = torch.zeros(n, device=dev)
        self.exp_avg_sq = torch.zeros(n, device=dev)
        self.step_state = torch.zeros(2, dtype=torch.int32, device=dev)
        self.opt_stats = torch.zeros(2, device=dev)
        self.adamw_ws = torch.empty(ops._lib.load().rlx_adamw_workspace_bytes(n), dtype=torch.uint8, device=dev)
        # slab sum + norm + clip + AdamW as ONE launch (None: two).  It needs all its workgroups resident together, so not when
        # two ranks of the job share a GPU: their launches could each hold half the device and wait for the other half
        from ...scheduler import ranks_share_a_device
        self.adamw_sync = None if ranks_share_a_device(self.ctx) else ops.adamw_sync_words(n, dev)
        self.grad_flat = torch.zeros(n, device=dev)
        self._ws = {}
        # gradient all-reduce transport (world_size > 1): hand-written xGMI peer reads, validated against torch.distributed at
        # start-up on every rank; RCCL when that is unavailable.  Either way the update phase is graph-captured.
        self._xgmi = None
        if self._exchange and dev is not None and dev.type == "cuda" and self.ctx.group is None:
            # (a split placement's learner group exchanges over RCCL: the xGMI communicator is set up over the whole job's ranks)
            from ...scheduler import xgmi
            self._xgmi = xgmi.build(self.ctx, n)
        self.grad_allreduce_backend = "none" if not self._exchange else ("xgmi" if self._xgmi is not None else "rccl")
        self._build_lr_scheduler()

    def _build_lr_scheduler(self):
        """build_lr_scheduler (fsdp_model_manager.py:464-498): a fresh schedule over (lr, value_lr)."""
        o = self.cfg.actor.optim
        self.lr_scheduler = LearnerLRScheduler(o, [o.lr, o.value_lr])
        self._apply_lrs()

    def _apply_lrs(self):
        """The schedule's current learning rates -> the AdamW ranges; launch plans and graphs that captured the old values
        are dropped."""
        self._lrs = self.lr_scheduler.get_last_lr()
        self.groups = self.model.group_ranges(self._lrs[0], self._lrs[1], train_value_head=self.cfg.algorithm.loss_type != "actor")
        for key in ("prepared_key", "aplan_key", "agraph_key"):  # (the async learner's plan and graph: async_ppo_fsdp_worker.py)
            self._ws.pop(key, None)
        self._graph = None

    def _one_launch_expired(self, grad_norm: float):
        """ops.check_adamw_sync's verdict on an iteration's gradient norm: when the one-launch optimizer step's exchange expired
        (a foreign process or stream held part of the GPU), continue on the two-launch form -- the sync words are retired (kept
        alive: launches already queued still point at them and skip as a whole), prepared launches and graphs that captured their
        pointer are dropped."""
        if ops.check_adamw_sync(self.adamw_sync, grad_norm):
            self._retired_adamw_sync, self.adamw_sync = self.adamw_sync, None
            for key in ("prepared_key", "aplan_key", "agraph_key"):
                self._ws.pop(key, None)
            self._graph = None

    def _step_lr_scheduler(self):
        """lr_scheduler.step() once per run_training (embodied_fsdp_actor_worker.py:568)."""
        self.lr_scheduler.step()
        if self.lr_scheduler.get_last_lr() != self._lrs:
            self._apply_lrs()

    def set_global_step(self, global_step: int):
        self.version = global_step

    def sync_model_to_rollout(self):
        """Weights for the rollout workers (embodied_fsdp_actor_worker.py:131-178).  Split placement: the reference's sequence --
        ``init_sender`` once (the receiver's metadata in, an init sync out when the patch syncer asks for one), then
        ``sync(state_dict, send, version)`` -- over the placement's weight-sync group.  Collocated: the rollout worker of this
        process either aliases this learner's policy object (nothing to send) or drives both halves of the sync itself over an
        in-process link (MultiStepRolloutWorker.sync_model_from_actor); returns the flat buffer for this package's own callers."""
        placement = getattr(self, "placement", None)
        if placement is not None and placement.split:
            from ..weight_link import GroupLink
            assert self.weight_syncer is not None, "weight_syncer config must be provided for a split placement"
            if getattr(self, "_weight_link", None) is None:
                self._weight_link = GroupLink(placement, self.device)
            self.serve_weight_sync(self._weight_link)
        return self.model.flat.data

    def serve_weight_sync(self, link) -> None:
        """The sender half (:131-178) over ``link`` (workers/weight_link.py): the state dict is the policy's reference-named views,
        ``param_names_need_sync`` all of them (collect_param_names_need_sync: trainable parameters and persistent buffers)."""
        state_dict = self.model.state_dict()
        syncer = self.weight_syncer
        if not syncer.sender_initialized():
            syncer.init_sender(state_dict=state_dict, send=link.actor_send, recv=link.actor_recv,
                               param_names_need_sync=list(state_dict.keys()), is_sender=self._is_weight_sender)
        syncer.sync(state_dict, link.actor_send, version=self.version)

    def state_dict(self):
        return self.model.reference_state_dict()

    # ---- checkpoints ------------------------------------------------------------------------------------------
    def save_checkpoint(self, save_path: str, step: int = 0) -> None:
        """FSDPModelManager.save_checkpoint (fsdp_model_manager.py:360-389 -> strategy/base.py:184-268), "local_shard" form:
        every rank writes its training state (NO_SHARD: the full model, the optimizer moments and step counter, the LR
        schedule, the RNG streams) to ``local_shard_checkpoint/checkpoint_rank_{r}.pt``; rank 0 also writes
        ``model_state_dict/full_weights.pt`` under the reference's parameter names -- the file the reference's own
        MLPPolicy.load_state_dict accepts."""
        import os
        shard_dir = os.path.join(save_path, "local_shard_checkpoint")
        os.makedirs(shard_dir, exist_ok=True)
        state = {
            "model": {k: v.detach().cpu().clone() for k, v in self.model.state_dict().items()},
            "optimizer": {"exp_avg": self.exp_avg.cpu(), "exp_avg_sq": self.exp_avg_sq.cpu(), "step_state": self.step_state.cpu(),
                          "optimizer_steps": self.optimizer_steps, "critic_warmup_steps": self.critic_warmup_steps},
            "lr_scheduler": self.lr_scheduler.state_dict(),
            "rng": {"torch": torch.get_rng_state(),
                    "cuda": torch.cuda.get_rng_state(self.device) if self.device is not None and self.device.type == "cuda" else None,
                    "pipeline_shuffle": ((self._perm_prefetch_state if self._perm_prefetch else self._pipe_gen.get_state())
                                         if self.use_training_pipeline else None)},
            "version": self.version, "step": int(step), "world_size": self._world_size, "n_params": self.model.n_params,
        }
        torch.save(state, os.path.join(shard_dir, f"checkpoint_rank_{self._rank}.pt"))
        if self._rank == 0:
            sd_dir = os.path.join(save_path, "model_state_dict")
            os.makedirs(sd_dir, exist_ok=True)
            torch.save(state["model"], os.path.join(sd_dir, "full_weights.pt"))
        if self._world_size > 1:
            import torch.distributed as dist
            dist.barrier(group=self.ctx.group)

    def load_checkpoint(self, load_path: str) -> None:
        """FSDPModelManager.load_checkpoint (:342-358): restores what save_checkpoint wrote, IN PLACE -- every device buffer
        keeps its address, so prepared launch plans and captured hipGraphs stay valid.  A directory that only holds
        ``model_state_dict/full_weights.pt`` (e.g. written by the reference) loads the weights alone."""
        import os
        shard = os.path.join(load_path, "local_shard_checkpoint", f"checkpoint_rank_{self._rank}.pt")
        if not os.path.exists(shard):
            full = os.path.join(load_path, "model_state_dict", "full_weights.pt")
            assert os.path.exists(full), f"no checkpoint under {load_path}"
            self.model.load_state_dict(torch.load(full, map_location="cpu", weights_only=False))
            return
        state = torch.load(shard, map_location="cpu", weights_only=False)
        assert state["n_params"] == self.model.n_params, "checkpoint holds a different model"
        self.model.load_state_dict(state["model"])
        o = state["optimizer"]
        self.exp_avg.copy_(o["exp_avg"]), self.exp_avg_sq.copy_(o["exp_avg_sq"]), self.step_state.copy_(o["step_state"])
        self.optimizer_steps, self.critic_warmup_steps = int(o["optimizer_steps"]), int(o["critic_warmup_steps"])
        self.lr_scheduler.load_state_dict(state["lr_scheduler"])
        self._apply_lrs()
        torch.set_rng_state(state["rng"]["torch"])
        if state["rng"]["cuda"] is not None and self.device is not None and self.device.type == "cuda":
            torch.cuda.set_rng_state(state["rng"]["cuda"], self.device)
        if self.use_training_pipeline and state["rng"]["pipeline_shuffle"] is not None:
            self._pipe_gen.set_state(state["rng"]["pipeline_shuffle"])
            self._perm_prefetch.clear()  # orders drawn ahead belong to the run that was replaced
        self.version = int(state["version"])

    # ---- trajectories -----------------------------------------------------------------------------------------
    def recv_rollout_trajectories(self, input_channel=None):
        """``input_channel``: the reference's channel (:186-207: ``split_num`` gets) or, for this package's own callers, the
        list of trajectory views itself."""
        send_num = self._world_size * self.cfg.rollout.get("pipeline_stage_num", 1)
        split_num = compute_split_num(send_num, self._world_size)
        E = self.pipeline_epochs if self.pipeline_epochs > 1 else 1
        trajectories = (input_channel if isinstance(input_channel, (list, tuple))
                        else [input_channel.get() for _ in range(split_num * E)])
        assert len(trajectories) == split_num * E, f"expected {split_num * E} trajectories, got {len(trajectories)}"
        if E > 1:  # epoch-major: one batch per rollout epoch, each trained as soon as its epoch is complete
            self.rollout_batches = [self._process_received_rollout_batch(convert_trajectories_to_batch(
                list(trajectories[e * split_num:(e + 1) * split_num]))) for e in range(E)]
            self.rollout_batch = self.rollout_batches[0]
            return
        self.rollout_batch = self._process_received_rollout_batch(convert_trajectories_to_batch(trajectories))

    def _process_received_rollout_batch(self, batch: dict) -> dict:
        tr = self.cfg.env.train
        # rollout_epoch > 1: the env worker already wrote epoch e into batch columns [e*B, (e+1)*B), i.e. the layout
        # process_nested_dict_for_adv folds to (:208-216); nothing to move
        if not tr.get("auto_reset", False) and not tr.get("ignore_terminations", False):  # :219-233
            dones = batch["dones"].contiguous()
            loss_mask, mask_sum = ops.done_prefix_mask(dones)
            loss_mask_sum = mask_sum.view(1, -1, 1).expand_as(loss_mask)
            if self.cfg.algorithm.reward_type == "chunk_level":
                loss_mask = loss_mask.any(dim=-1, keepdim=True)
                loss_mask_sum = loss_mask_sum[..., -1:]
            batch["loss_mask"], batch["loss_mask_sum"] = loss_mask, loss_mask_sum
        alg = self.cfg.algorithm
        if alg.get("filter_rewards", False):  # :235-281: drop prompt groups whose mean reward is out of bounds
            mask = batch.get("loss_mask")
            rewards = batch["rewards"].contiguous()
            if mask is not None and mask.shape != rewards.shape:
                mask = mask.expand_as(rewards).contiguous()
            batch["loss_mask"] = ops.reward_filter_mask(rewards, mask, int(alg.group_size), float(alg.rewards_lower_bound),
                                                        float(alg.rewards_upper_bound))
        return batch

    def compute_advantages_and_returns(self) -> dict:
        if self.pipeline_epochs > 1:
            # per-epoch work happens inside run_training (the reference's pipeline learner does everything there and returns
            # {"rollout_metrics", "training_metrics"}, fsdp_actor_worker_pipeline.py:84-196): enqueueing the advantage pass of
            # epoch e + 1 here would put its wait for that epoch's rollout IN FRONT of epoch e's training on the learner stream
            return {}
        return self._advantages_for(self.rollout_batch)

    def pop_rollout_metrics(self) -> dict:
        out, self._deferred_rollout_metrics = self._deferred_rollout_metrics, {}
        return out

    def _advantages_for(self, b: dict, metrics: bool = True):
        alg = self.cfg.algorithm
        with self.timer("actor/compute_adv"):
            out = calculate_adv_and_returns(
                task_type=self.cfg.runner.task_type, adv_type=alg.adv_type, rewards=b["rewards"], dones=b["dones"],
                values=b.get("prev_values"), gamma=alg.get("gamma", 1), gae_lambda=alg.get("gae_lambda", 1),
                group_size=alg.get("group_size", 8), reward_type=alg.reward_type, loss_mask=b.get("loss_mask"),
                loss_mask_sum=b.get("loss_mask_sum"),
                **({"normalize_advantages": False} if self.use_training_pipeline else {}))
            if self.use_training_pipeline and alg.get("normalize_advantages", True):
                adv = out["advantages"].contiguous()
                mask = b.get("loss_mask")
                if mask is not None and mask.shape != adv.shape:
                    mask = mask.expand_as(adv).contiguous()
                stats = ops.masked_stats(adv, mask)  # this rank's stage batches together; one env rank feeds one learner
                out["advantages"] = ops.normalize_from_stats(adv, stats)
            b.update(out)
            return self._rollout_metrics(b) if metrics else None

    def _rollout_metrics(self, b: dict) -> dict:
        """compute_rollout_metrics (rlinf/utils/metric_utils.py:422-506): masked mean / min / max of rewards, advantages,
        returns -- ONE pass over the three arrays on the device (rlx_rollout_metrics), reduced over ranks in one SUM call + one MAX
        call, read back in one copy.  No boolean-index gathers."""
        mask = b.get("loss_mask")
        names = [k for k in ("rewards", "advantages", "returns") if b.get(k) is not None]
        if self.device is None or self.device.type != "cuda":
            raise ops.RlxError("rollout metrics run on the accelerator: the trajectory buffer lives there (no CPU fallback)")
        key = ("rollout_metrics", len(names))
        if key not in self._ws:
            self._ws[key] = (torch.empty((len(names), 4), dtype=torch.float64, device=self.device),
                             torch.empty(ops._lib.load().rlx_rollout_metrics_workspace_bytes(), dtype=torch.uint8, device=self.device))
        out, ws = self._ws[key]
        ops.rollout_metrics([b[k] for k in names], mask, out=out, workspace=ws)
        return self._metrics_from_reductions(names, out)

    def _metrics_from_reductions(self, names: list, red: torch.Tensor) -> dict:
        """red [len(names), 4] f64 = (sum, count, -min, max) of this rank's selection -> the reference's metric dict: one SUM and
        one MAX all-reduce over ranks (metric_utils.py:451-454 does two per metric), one read-back."""
        s, m = all_reduce_scalars(red[:, :2].reshape(-1).clone(), red[:, 2:].reshape(-1).clone(), self.ctx)
        ns = s.numel()

        def finish(host: list) -> dict:
            s, m = host[:ns], host[ns:]
            res = {}
            for i, key in enumerate(names):
                cnt = s[2 * i + 1]
                if cnt > 0:
                    mean, vmax, vmin = s[2 * i] / cnt, m[2 * i + 1], -m[2 * i]
                else:  # nothing selected on any rank: all three are NaN (metric_utils.py:458-460)
                    mean = vmax = vmin = float("nan")
                if key == "rewards":
                    res["rewards"] = mean
                else:
                    res[f"{key}_mean"], res[f"{key}_max"], res[f"{key}_min"] = mean, vmax, vmin
            return res

        if self.defer_host_reads:  # the runner reads one iteration late (utils/pending.py)
            return PendingMetrics(torch.cat([s, m]), finish)
        return finish(s.tolist() + m.tolist())

    # ---- update -------------------------------------------------------------------------------------------------
    def _pipeline_perm_host(self, T: int, B: int) -> np.ndarray:
        stages = int(self.cfg.rollout.get("pipeline_stage_num", 1))
        n = B // stages
        parts = []
        for st in range(stages):
            # index arithmetic in numpy: torch's element-wise CPU ops fan out over every core of the host above 32768 elements
            # (measured on the 256-core box: 30 ms per iteration for three int64 ops on 131072 elements)
            local = (_host_randperm(T * n, self._pipe_gen) if self._pipe_shuffle else torch.arange(T * n)).numpy()
            parts.append((local // n) * B + st * n + (local % n))
        return np.concatenate(parts)

    def _pipeline_perm(self, T: int, B: int, epoch: int) -> torch.Tensor:
        """pack_pipeline_micro_batches (env_worker.py:1519-1537): every stage's [T, B/stages] block flattened and shuffled on
        its own, stage after stage, with a generator that is seeded ONCE -- a new order every call, written into the same
        device buffer so that prepared launches / captured graphs keep reading the right rows.  The host side of it (a serial
        Fisher-Yates walk, ~1 ms per 131072 rows) is normally already done: _prefetch_pipeline_perms drew this iteration's
        orders while the GPU was still busy with the previous iteration."""
        N = T * B
        host = self._perm_prefetch.pop((T, B, epoch), None)
        if host is None:
            if self._perm_prefetch:  # a shape change: orders drawn ahead for another shape are void -> rewind the generator
                self._pipe_gen.set_state(self._perm_prefetch_state)
                self._perm_prefetch.clear()
            host = torch.from_numpy(self._pipeline_perm_host(T, B))
        pkey = ("perm", N, epoch)
        if pkey not in self._ws:
            self._ws[pkey] = torch.empty(N, dtype=torch.int64, device=self.device)
        if self.device.type != "cuda":
            self._ws[pkey].copy_(host)
            return self._ws[pkey]
        # Upload without stalling the host (a copy from pageable memory waits for everything queued before it -- that alone kept
        # the host in lock-step with the device in pipeline mode): a small ring of pinned staging buffers per order, each reused
        # only after the upload that last read it has finished (two iterations ago: the event is long set).
        skey = ("perm_stage", N, epoch)
        ring = self._ws.setdefault(skey, {"bufs": [], "events": [], "next": 0})
        if len(ring["bufs"]) < 3:
            ring["bufs"].append(torch.empty(N, dtype=torch.int64, pin_memory=True))
            ring["events"].append(torch.cuda.Event())
            i = len(ring["bufs"]) - 1
        else:
            i = ring["next"] % 3
            ring["events"][i].synchronize()
        ring["next"] = i + 1
        np.copyto(ring["bufs"][i].numpy(), host.numpy())  # (torch's CPU copy_ wakes the whole intra-op pool above 32768 elements)
        self._ws[pkey].copy_(ring["bufs"][i], non_blocking=True)
        ring["events"][i].record(torch.cuda.current_stream(self.device))
        return self._ws[pkey]

    def _prefetch_pipeline_perms(self, T: int, B: int, n_epochs: int) -> None:
        """Draw the NEXT iteration's shuffles now -- the GPU work of this iteration is enqueued and the host would otherwise sit
        in the metric read-back -- in the order the next iteration would draw them (epoch 0 .. E - 1, stages inside): the
        generator stream is consumed exactly as without the prefetch.  A checkpoint written in between saves the generator's
        state from BEFORE the draw (``_perm_prefetch_state``), and loading one drops the prefetched orders."""
        if not (self.use_training_pipeline and self._pipe_shuffle) or self._perm_prefetch:
            return
        self._perm_prefetch_state = self._pipe_gen.get_state()
        for e in range(n_epochs):
            self._perm_prefetch[(T, B, e)] = torch.from_numpy(self._pipeline_perm_host(T, B))

    def _flatten_and_shuffle(self, b: dict | None = None, epoch: int = 0, n_epochs: int = 1, perm_ready: bool = False):
        """process_nested_dict_for_train (rlinf/utils/nested_dict_process.py:272-285): one randperm per
        run_training with Generator(seed = actor.seed + rank) (:511-513); one gather launch for every field.
        ``epoch`` / ``n_epochs`` (pipeline mode, rollout_epoch > 1): batch ``b`` is rollout epoch ``epoch`` and lands in rows
        [epoch * N, (epoch + 1) * N) of shuffled buffers that hold all epochs -> (views of that slice, N, the whole buffers)."""
        b = self.rollout_batch if b is None else b
        T, B = b["prev_logprobs"].shape[:2]
        N = T * B
        pkey = ("perm", N, epoch)
        if self.use_training_pipeline:
            if not perm_ready:
                self._pipeline_perm(T, B, epoch)
        elif pkey not in self._ws:  # the reference re-seeds the generator on every call: the permutation never changes
            g = torch.Generator()
            g.manual_seed(int(self.cfg.actor.seed) + self._rank)
            self._ws[pkey] = _host_randperm(N, g).to(self.device)
        perm = self._ws[pkey]
        names = ["states", "action", "prev_logprobs", "advantages", "prev_values"]
        src = [b["forward_inputs"]["states"], b["forward_inputs"]["action"], b["prev_logprobs"], b["advantages"],
               b["prev_values"][:-1]]
        if b.get("returns") is not None:
            names.append("returns")
            src.append(b["returns"])
        if b.get("loss_mask") is not None:
            names.append("loss_mask")
            src.append(b["loss_mask"])
        if b.get("loss_mask_sum") is not None:
            names.append("loss_mask_sum")
            src.append(b["loss_mask_sum"].contiguous())
        flat = [t.reshape(N, *t.shape[2:]).contiguous() for t in src]
        # the field set can change between calls with the same N (a loss mask or returns appearing): the cached output
        # buffers are keyed by every field's name, row shape and dtype, never zipped against a different list
        key = ("shuf", N, n_epochs, tuple((n, tuple(t.shape[1:]), t.dtype) for n, t in zip(names, flat)))
        if key not in self._ws:
            self._ws[key] = [torch.empty((N * n_epochs, *t.shape[1:]), dtype=t.dtype, device=t.device) for t in flat]
        big = self._ws[key]
        outs = ops.gather_rows(flat, perm, [t[epoch * N:(epoch + 1) * N] for t in big])
        if n_epochs == 1:
            return dict(zip(names, outs)), N
        return dict(zip(names, outs)), N, dict(zip(names, big))

    def _minibatch_workspace(self, mb: int):
        key = ("mb", mb, self.fused_step)
        if key not in self._ws:
            lay, dev = self.model.layout, self.device
            if self.fused_step:
                self._ws[key] = dict(slabs=ops.ppo_step_slabs(lay, mb, self.model.compute_dtype == torch.bfloat16),
                                     step_ws=torch.empty(ops.ppo_step_workspace_bytes(lay, mb), dtype=torch.uint8, device=dev))
                return self._ws[key]
            f = lambda *s: torch.empty(s, dtype=torch.float32, device=dev)  # noqa: E731
            slabs = ops.mlp_bwd_slabs(mb)
            lib = ops._lib.load()
            self._ws[key] = dict(
                logprob=f(mb, lay.act_dim), entropy=f(mb, lay.act_dim), value=f(mb, lay.val_dim), mean=f(mb, lay.act_dim),
                acts=f(2, 3, mb, 256), g_lp=f(mb * lay.act_dim), g_v=f(mb * lay.val_dim), d_lp=f(mb, lay.act_dim),
                d_v=f(mb, lay.val_dim), slabs=slabs,
                loss_ws=torch.empty(lib.rlx_ppo_loss_workspace_bytes(mb), dtype=torch.uint8, device=dev),
                bwd_ws=torch.empty(lib.rlx_mlp_bwd_workspace_bytes(ops.byref(lay), mb), dtype=torch.uint8, device=dev))
        return self._ws[key]

    def _loss_params(self, critic_warmup: bool):
        alg, m = self.cfg.algorithm, self.cfg.actor.model
        has_critic = alg.loss_type == "actor_critic"
        return ops.make_ppo_params(
            logprob_type=alg.logprob_type, action_dim=m.action_dim, chunks=m.num_action_chunks,
            clip_ratio_low=alg.clip_ratio_low, clip_ratio_high=alg.clip_ratio_high, value_clip=alg.get("value_clip"),
            huber_delta=alg.get("huber_delta"), max_episode_steps=self.cfg.env.train.get("max_episode_steps"),
            critic_warmup=critic_warmup, has_critic=has_critic, reward_type=alg.get("reward_type", "action_level"))

    def train_micro_batch(self, mbatch: dict, ws: dict, grads: torch.Tensor, out_row: torch.Tensor, grad_out: torch.Tensor,
                          lp: "ops.PpoLossParams"):
        """forward -> loss -> backward for one micro-batch; parameter gradients land in ``grads`` [slabs, n]."""
        m = self.model
        lay = m.layout
        mb = mbatch["states"].shape[0]
        if self.fused_step:
            bf16 = m.compute_dtype == torch.bfloat16
            ops.ppo_step(m.flat.data, lay, lp, mbatch, grads, out_row, ws["step_ws"], grad_out=self._grad_out_host,
                         tiles=m.tiles() if self.optimizer_writes_tiles else None, bf16=bf16)
            return
        ops.mlp_train_fwd(m.flat.data, m.packed(), lay, mbatch["states"], mbatch["action"], acts=ws["acts"],
                          out=(ws["logprob"], ws["entropy"], ws["value"], ws["mean"]))
        n_adv = mb * lay.act_dim // lp.raw_per_adv  # advantage elements: [mb, C] (action/token level) or [mb] (chunk level)
        has_critic = bool(lp.has_critic)
        ops.ppo_loss_fwd_raw(lp, n_adv, ws["logprob"], mbatch["prev_logprobs"], mbatch["advantages"],
                             ws["value"] if has_critic else None, mbatch["prev_values"] if has_critic else None,
                             mbatch.get("returns") if has_critic else None,
                             None if mbatch.get("loss_mask") is None else mbatch["loss_mask"].view(torch.uint8),
                             mbatch.get("loss_mask_sum"), ws["g_lp"], ws["g_v"] if has_critic else None, out_row,
                             ws["loss_ws"])
        ops.ppo_loss_bwd_raw(lp, n_adv, ws["g_lp"], ws["g_v"] if has_critic else None, out_row, grad_out, ws["d_lp"],
                             ws["d_v"] if has_critic else None)
        if not has_critic:
            ws["d_v"].zero_()
        ops.mlp_train_bwd(m.flat.data, m.packed(), lay, mbatch["states"], mbatch["action"], ws["mean"], ws["acts"],
                          ws["d_lp"], None, ws["d_v"], grads=grads, workspace=ws["bwd_ws"])

    def _entropy_bonus(self, mbatch: dict, grads: torch.Tensor, out_row: torch.Tensor, critic_warmup: bool = False,
                       actor_scale: torch.Tensor | None = None):
        """loss -= entropy_bonus * masked_mean(entropy) (:679-690); a no-op at the default entropy_bonus = 0.
        ``actor_scale``: see ops.gaussian_entropy_bonus_ (the decoupled fused step's sum-form actor gradients)."""
        alg = self.cfg.algorithm
        bonus = float(alg.get("entropy_bonus", 0) or 0)
        if bonus <= 0 or critic_warmup:
            return
        has_mask = mbatch.get("loss_mask") is not None
        etype = alg.get("entropy_type", "action_level")
        if has_mask and etype == "chunk_level":
            # reshape_entropy (utils.py:406-407) leaves a [bsz] vector that masked_mean (utils.py:323-330) multiplies with the
            # [bsz, C] mask.  C = 1: the product broadcasts to [bsz, bsz], so (sum_i e_i)(sum_j m_j) / sum_j m_j = the SUM of the
            # row entropies -- bsz times the mean, whatever the mask holds (zero for an all-False mask); reproduced as written.
            # C > 1: torch refuses to broadcast [bsz] against [bsz, C]; so do we, with its message.
            mb, C = mbatch["loss_mask"].shape[0], int(mbatch["loss_mask"].numel() // mbatch["loss_mask"].shape[0])
            if C != 1:
                raise RuntimeError(f"The size of tensor a ({mb}) must match the size of tensor b ({C}) at non-singleton dimension 1")
            ops.gaussian_entropy_bonus_(self.model.flat.data, self.model.layout, grads[0], out_row, bonus, self._grad_out_host,
                                        True, float(mb), actor_scale=actor_scale)
            return
        per_elem = alg.get("entropy_type", "action_level") == "token_level" and not has_mask
        ops.gaussian_entropy_bonus_(self.model.flat.data, self.model.layout, grads[0], out_row, bonus, self._grad_out_host,
                                    has_mask, 1.0 / self.model.layout.act_dim if per_elem else 1.0, actor_scale=actor_scale)

    def optimizer_step(self, grads: torch.Tensor, stats: torch.Tensor | None = None, critic_warmup: bool = False):
        """clip_grad_norm_ + AdamW (+ data-parallel mean of the gradient) -> stats (norm, applied) on device.  The
        optimizer kernel also refreshes the fragment-tile weight image the next forward (and the rollout) streams.
        Critic warm-up (fsdp_model_manager.py:523-531, :451-459): the optimizer holds the value head only (everything else
        has requires_grad False), and after step ``critic_warmup_steps`` the reference builds a NEW optimizer over all
        parameters -- every moment and every step count starts again from zero, the value head's included."""
        o = self.cfg.actor.optim
        groups = self.groups
        if critic_warmup:
            vh = [(self.model.offsets[n], self.model.offsets[n] + self.model.view(n).numel())
                  for n in self.model.shapes if "value_head" in n]
            groups = [(max(b, vb), min(e, ve), lr) for (b, e, lr) in self.groups for (vb, ve) in vh if max(b, vb) < min(e, ve)]
        tiles = self.model.tiles() if (self.fused_step and self.optimizer_writes_tiles) else None
        kw = dict(betas=(o.adam_beta1, o.adam_beta2), eps=o.adam_eps, weight_decay=o.weight_decay, max_grad_norm=o.clip_grad,
                  stats=self.opt_stats if stats is None else stats, step_state=self.step_state, workspace=self.adamw_ws,
                  tile_layout=self.model.layout if tiles is not None else None, tiles=tiles, sync=self.adamw_sync)
        if self._xgmi is not None:  # stage + peer-read reduce + clip + AdamW: three launches, no host round trip
            ops.PreparedAdamw(self.model.flat.data, grads, self.exp_avg, self.exp_avg_sq, groups, grad_scale=1.0 / self._world_size,
                              xgmi=self._xgmi, grad_flat=self.grad_flat, **kw)(torch.cuda.current_stream(self.device).cuda_stream)
        elif self._exchange:
            ops.sum_slabs(grads, out=self.grad_flat)
            all_reduce_flat_(self.grad_flat, self.ctx)  # RCCL, one flat 1.15 MB buffer (C1)
            ops.clip_adamw_step_(self.model.flat.data, self.grad_flat, self.exp_avg, self.exp_avg_sq, groups, 0,
                                 grad_scale=1.0 / self._world_size, **kw)
        else:
            ops.clip_adamw_step_(self.model.flat.data, grads, self.exp_avg, self.exp_avg_sq, groups, 0, grad_scale=1.0, **kw)
        self.model.mark_updated(tiles_fresh=tiles is not None)
        self.optimizer_steps += 1
        # lr_list of FSDPModelManager.optimizer_step (:451-461): while the critic warms up the optimizer holds ONE group and
        # reports 0.0 for it (-> actor/lr = 0.0, no critic/lr entry for this step), the scheduled rates afterwards
        self._lr_log.append((0.0, None) if self.critic_warmup_steps > 0 else (self._lrs[0], self._lrs[1]))
        if self.critic_warmup_steps > 0 and self.optimizer_steps >= self.critic_warmup_steps:
            self.exp_avg.zero_(), self.exp_avg_sq.zero_(), self.step_state.zero_()  # build_optimizer(model) anew (:453-455)
            self.critic_warmup_steps = 0
            self._build_lr_scheduler()  # ... and a new scheduler over it (:456-459)

    def _run_update(self, flat: dict, N: int, metrics_dev: torch.Tensor, norms_dev: torch.Tensor):
        a, alg = self.cfg.actor, self.cfg.algorithm
        n_mb, per_rank, accum = minibatch_plan(N, a.global_batch_size, a.micro_batch_size, self._world_size)
        assert accum == self.gradient_accumulation
        micro = a.micro_batch_size
        ws = self._minibatch_workspace(micro)
        key = ("grads", micro, accum)
        if key not in self._ws:
            self._ws[key] = torch.zeros((ws["slabs"] * accum, self.model.n_params), dtype=torch.float32, device=self.device)
            self._ws["grad_out"] = torch.full((1,), 1.0 / accum, dtype=torch.float32, device=self.device)
        grads, grad_out = self._ws[key], self._ws["grad_out"]
        self._grad_out_host = 1.0 / accum
        if self.fused_step and self.critic_warmup_steps == 0:
            return self._run_update_prepared(flat, N, metrics_dev, norms_dev, grads, ws, n_mb, per_rank, accum, micro)
        step = 0
        for _ in range(alg.get("update_epoch", 1)):
            for i in range(n_mb):
                lp = self._loss_params(self.optimizer_steps < self.critic_warmup_steps)
                for j in range(accum):
                    lo = i * per_rank + j * micro
                    mbatch = {k: v[lo:lo + micro] for k, v in flat.items()}
                    self.train_micro_batch(mbatch, ws, grads[j * ws["slabs"]:(j + 1) * ws["slabs"]],
                                           metrics_dev[step * accum + j], grad_out, lp)
                    self._entropy_bonus(mbatch, grads[j * ws["slabs"]:(j + 1) * ws["slabs"]], metrics_dev[step * accum + j],
                                        bool(lp.critic_warmup))
                self.optimizer_step(grads, stats=norms_dev[step],  # (norm, applied) straight into this step's row
                                    critic_warmup=bool(lp.critic_warmup))
                step += 1
        return step

    def _run_update_prepared(self, flat, N, metrics_dev, norms_dev, grads, ws, n_mb, per_rank, accum, micro, only_build=False):
        """The same loop with every launch marshalled once: all buffers are persistent, so an optimizer step is a handful of
        ctypes calls (+ one torch.distributed all-reduce when world_size > 1) -- the eager path of multi-GPU runs is
        otherwise bound by Python argument marshalling, not by the GPU."""
        m, o = self.model, self.cfg.actor.optim
        epochs = self.cfg.algorithm.get("update_epoch", 1)
        pkey = ("prepared", N, micro, accum, epochs, tuple(t.data_ptr() for t in flat.values()), metrics_dev.data_ptr())
        if self._ws.get("prepared_key") != pkey:
            bf16 = m.compute_dtype == torch.bfloat16
            tiles = m.tiles() if self.optimizer_writes_tiles else None
            lp = self._loss_params(False)
            multi = self._exchange
            xg = self._xgmi
            plan, step = [], 0
            for _ in range(epochs):
                for i in range(n_mb):
                    micro_calls = []
                    for j in range(accum):
                        lo = i * per_rank + j * micro
                        mbatch = {k: v[lo:lo + micro] for k, v in flat.items()}
                        micro_calls.append(ops.PreparedPpoStep(
                            m.flat.data, m.layout, lp, mbatch, grads[j * ws["slabs"]:(j + 1) * ws["slabs"]],
                            metrics_dev[step * accum + j], ws["step_ws"], grad_out=self._grad_out_host, tiles=tiles, bf16=bf16))
                        if float(self.cfg.algorithm.get("entropy_bonus", 0) or 0) > 0:
                            micro_calls.append(lambda _stream, mb=mbatch, g=grads[j * ws["slabs"]:(j + 1) * ws["slabs"]],
                                               row=metrics_dev[step * accum + j]: self._entropy_bonus(mb, g, row))
                    adam = ops.PreparedAdamw(
                        m.flat.data, self.grad_flat if (multi and xg is None) else grads, self.exp_avg, self.exp_avg_sq, self.groups,
                        betas=(o.adam_beta1, o.adam_beta2), eps=o.adam_eps, weight_decay=o.weight_decay,
                        max_grad_norm=o.clip_grad, grad_scale=1.0 / self._world_size if multi else 1.0, stats=norms_dev[step],
                        step_state=self.step_state, workspace=self.adamw_ws, tile_layout=m.layout if tiles is not None else None,
                        tiles=tiles, xgmi=xg, grad_flat=self.grad_flat if xg is not None else None, sync=self.adamw_sync)
                    plan.append((micro_calls, adam))
                    step += 1
            self._ws["prepared_key"], self._ws["prepared_plan"] = pkey, plan
        plan = self._ws["prepared_plan"]
        if only_build:
            return plan
        return self._exec_plan(plan, grads)

    def _exec_plan(self, plan: list, grads: torch.Tensor, lo: int = 0, hi: int | None = None) -> int:
        """Steps [lo, hi) of a prepared plan on the current stream."""
        m = self.model
        steps = plan[lo:hi]
        tiles_fresh = self.optimizer_writes_tiles
        if tiles_fresh:
            m.tiles()  # make sure the image is current before the first forward (no-op when the optimizer kept it fresh)
        stream = torch.cuda.current_stream(self.device).cuda_stream
        with torch.cuda.device(self.device):
            for micro_calls, adam in steps:
                for call in micro_calls:
                    call(stream)
                if self._exchange and self._xgmi is None:
                    ops.sum_slabs(grads, out=self.grad_flat, deferred=adam.deferred)
                    all_reduce_flat_(self.grad_flat, self.ctx)  # RCCL, one flat 1.15 MB buffer (C1); capturable in a hipGraph
                adam(stream)
        m.mark_updated(tiles_fresh=tiles_fresh)
        self.optimizer_steps += len(steps)
        self._lr_log.extend([(self._lrs[0], self._lrs[1])] * len(steps))
        return len(steps)

    def _pipeline_schedule(self, E: int, n_e: int, passes: int) -> list:
        """Order in which the learner trains the (pass, global batch) pairs of one iteration in pipeline mode with E rollout epochs
        of n_e global batches each: [(k, g)], pass k in [0, passes), global batch g in [0, E * n_e) -- g // n_e is its epoch.
        This base class: every epoch's first pass as the epochs complete, then every stored global batch again, pass by pass,
        oldest first.  PipelineEmbodiedFSDPActor derives the order from the reference's own queue logic instead."""
        return [(0, g) for g in range(E * n_e)] + [(k, g) for k in range(1, passes) for g in range(E * n_e)]

    def _run_pipeline_epochs(self) -> dict:
        """runner.use_training_pipeline with rollout_epoch E > 1 (fsdp_actor_worker_pipeline.py:84-160 over the per-epoch sends of
        env_worker.py:1324-1330).  For every epoch, when the schedule first touches it and as soon as ITS rollout is complete (the
        env worker's event; the rollout of the next epoch keeps running on its own stream, with its own frozen weights):
        un-normalised GAE, statistics normalisation over the epoch's batch, per-stage stateful shuffles; then the global batches
        in the order _pipeline_schedule gives -- by default every epoch's first pass on arrival and the stored global batches
        again, pass by pass, oldest first (select_global_batch).  The reference's interleaving of first and later passes depends
        on arrival timing; the default is the schedule of a rollout that delivers each epoch just as the previous one's first
        pass ends -- the one its overlap is built for."""
        a, alg = self.cfg.actor, self.cfg.algorithm
        from ..common import peer
        E = self.pipeline_epochs
        env = peer("env", self.cfg)
        events = env.epoch_events if (env is not None and env.epoch_events) else None
        stream = torch.cuda.current_stream(self.device)
        T0, B0 = self.rollout_batches[0]["prev_logprobs"].shape[:2]
        for e in range(E):  # the shuffles do not depend on the data: staged before the learner stream starts waiting on rollouts
            self._pipeline_perm(T0, B0, e)
        reds, bigs, names_box = [None] * E, [None], [None]

        def prepare(e: int):
            b = self.rollout_batches[e]
            if events is not None:
                stream.wait_event(events[e])
            self._advantages_for(b, metrics=False)
            names_box[0] = [k for k in ("rewards", "advantages", "returns") if b.get(k) is not None]
            reds[e] = ops.rollout_metrics([b[k] for k in names_box[0]], b.get("loss_mask"))
            _, N_e, big = self._flatten_and_shuffle(b, epoch=e, n_epochs=E, perm_ready=True)
            bigs[0] = big
            return N_e

        N_e = prepare(0)
        N = N_e * E
        n_mb, per_rank, accum = minibatch_plan(N, a.global_batch_size, a.micro_batch_size, self._world_size)
        assert N_e % per_rank == 0, f"a rollout epoch ({N_e} rows) must hold whole global batches ({per_rank} rows per rank)"
        n_e, epochs = N_e // per_rank, alg.get("update_epoch", 1)
        key = ("metrics", n_mb * epochs, accum)
        if key not in self._ws:
            self._ws[key] = (torch.zeros(n_mb * epochs * accum, PPO_OUT_FLOATS, device=self.device),
                             torch.zeros(n_mb * epochs, 2, device=self.device))
        metrics_dev, norms_dev = self._ws[key]
        micro = a.micro_batch_size
        ws = self._minibatch_workspace(micro)
        gkey = ("grads", micro, accum)
        if gkey not in self._ws:
            self._ws[gkey] = torch.zeros((ws["slabs"] * accum, self.model.n_params), dtype=torch.float32, device=self.device)
            self._ws["grad_out"] = torch.full((1,), 1.0 / accum, dtype=torch.float32, device=self.device)
        grads = self._ws[gkey]
        self._grad_out_host = 1.0 / accum
        assert self.fused_step and self.critic_warmup_steps == 0, "the pipeline learner runs the fused prepared step"
        # the plan is laid out pass-major: entry k * n_mb + g trains global batch g for the (k + 1)-th time
        plan = self._run_update_prepared(bigs[0], N, metrics_dev, norms_dev, grads, ws, n_mb, per_rank, accum, micro, only_build=True)
        order = self._pipeline_schedule(E, n_e, epochs)
        assert sorted(order) == [(k, g) for k in range(epochs) for g in range(n_mb)], "every (pass, global batch) exactly once"
        prepared, seen, run = {0}, set(), []

        def flush():
            if run:
                self._exec_plan(plan, grads, run[0], run[-1] + 1)
                run.clear()

        for k, g in order:
            e = g // n_e
            assert k == 0 or (k - 1, g) in seen, "a global batch is revisited only after its previous pass"
            if e not in prepared:  # first touch of a later epoch: its advantage pass and shuffle go in front of its first batch
                assert k == 0
                flush()
                prepare(e)
                prepared.add(e)
            idx = k * n_mb + g
            if run and idx != run[-1] + 1:
                flush()
            run.append(idx)
            seen.add((k, g))
        flush()
        self._prefetch_pipeline_perms(T0, B0, E)  # host work for the NEXT iteration, behind this iteration's enqueued launches
        names = names_box[0]
        red = torch.stack(reds)  # [E, k, 4]: sums add, (-min, max) take the maximum
        self._deferred_rollout_metrics = self._metrics_from_reductions(
            names, torch.cat([red[:, :, :2].sum(dim=0), red[:, :, 2:].amax(dim=0)], dim=1))
        out = self._collect_metrics(metrics_dev, norms_dev, accum)
        if self._xgmi is not None and not self.defer_host_reads:
            self._xgmi.check_status()
        self._step_lr_scheduler()
        return out

    def run_training(self, input_channel=None) -> dict:
        a, alg = self.cfg.actor, self.cfg.algorithm
        if self.pipeline_epochs > 1:
            with self.timer("run_training"):
                self._lr_log = []
                return self._run_pipeline_epochs()
        with self.timer("run_training"):
            flat, N = self._flatten_and_shuffle()
            n_mb, _, accum = minibatch_plan(N, a.global_batch_size, a.micro_batch_size, self._world_size)
            n_steps = n_mb * alg.get("update_epoch", 1)
            key = ("metrics", n_steps, accum)
            if key not in self._ws:
                self._ws[key] = (torch.zeros(n_steps * accum, PPO_OUT_FLOATS, device=self.device),
                                 torch.zeros(n_steps, 2, device=self.device))
            metrics_dev, norms_dev = self._ws[key]
            self._lr_log = []  # (actor lr, critic lr or None) per optimizer step of this call
            if self.enable_hip_graph and self.critic_warmup_steps == 0 and self.lr_scheduler.is_static and self._capturable():
                self._replay_or_capture(flat, N, metrics_dev, norms_dev, n_steps)
            else:
                self._run_update(flat, N, metrics_dev, norms_dev)
            if self.use_training_pipeline:
                Tb, Bb = self.rollout_batch["prev_logprobs"].shape[:2]
                self._prefetch_pipeline_perms(Tb, Bb, 1)  # host work for the NEXT iteration, behind the enqueued update phase
            out = self._collect_metrics(metrics_dev, norms_dev, accum)
            if self._xgmi is not None and not self.defer_host_reads:
                self._xgmi.check_status()  # a peer that never published its gradient: raise instead of training on garbage
            self._step_lr_scheduler()
            return out

    def _capturable(self) -> bool:
        """The update phase can live in a hipGraph when its launches are all stream operations: always on one GPU; at
        world_size > 1 with the xGMI all-reduce (pure kernels) or with RCCL (stream-ordered, capturable) -- not with a
        host-staged backend such as gloo."""
        if not self._exchange or self._xgmi is not None:
            return True
        import torch.distributed as dist
        return dist.get_backend() == "nccl"

    def _replay_or_capture(self, flat, N, metrics_dev, norms_dev, n_steps):
        """hipGraph of the whole update phase (all epochs x minibatches): every buffer is persistent and the step
        counter lives on the device, so the captured launch chain can be replayed as is."""
        gkey = (N, n_steps, tuple(sorted(flat)), tuple(t.data_ptr() for t in flat.values()))
        if self._graph is None or self._graph_key != gkey:
            self._run_update(flat, N, metrics_dev, norms_dev)  # real run; also warms every workspace
            torch.cuda.synchronize(self.device)
            steps_before = self.optimizer_steps
            g = torch.cuda.CUDAGraph()
            ok = 1
            try:
                with torch.cuda.graph(g):
                    self._run_update(flat, N, metrics_dev, norms_dev)
            except Exception as e:  # noqa: BLE001 -- e.g. an RCCL build that cannot be stream-captured
                if not self._exchange:
                    raise
                ok = 0
                print(f"[rlinf_amd] rank {self._rank}: capturing the update phase failed ({type(e).__name__}: {e}); running it eagerly",
                      flush=True)
            self.optimizer_steps = steps_before  # capture records, it does not execute
            if self._exchange:  # every rank replays, or none does
                import torch.distributed as dist
                verdict = torch.tensor([ok], dtype=torch.int32, device=self.device)
                dist.all_reduce(verdict, op=dist.ReduceOp.MIN, group=self.ctx.group)
                ok = int(verdict.item())
            if not ok:
                self.enable_hip_graph = False
                return
            self._graph, self._graph_key = g, gkey
            return
        self._graph.replay()
        # the raw-pointer kernels do not bump flat._version: tell the model which derived images are stale now (the tile
        # image is kept fresh by the optimizer kernel only with fused_step + optimizer_writes_tiles)
        self.model.mark_updated(tiles_fresh=self.fused_step and self.optimizer_writes_tiles)
        self.optimizer_steps += n_steps

    def _collect_metrics(self, metrics_dev, norms_dev, accum) -> dict:
        """Mean over micro-batches, then AVG over ranks (embodied_fsdp_actor_worker.py:576-587); EV statistics are
        summed (metric_utils.py:293-319).  One D2H copy for everything."""
        has_critic = self.cfg.algorithm.loss_type == "actor_critic"
        m = metrics_dev.mean(dim=0)
        ev = metrics_dev[:, PPO_OUT_NAMES["ev/count"]:PPO_OUT_NAMES["ev/errors_sq_sum"] + 1].sum(dim=0)
        gn = norms_dev[:, 0].mean()
        vec = torch.cat([m, ev, gn.view(1)])
        if self._world_size > 1:
            avg = torch.cat([m, gn.view(1)])
            all_reduce_flat_(avg, self.ctx, average=True)
            all_reduce_flat_(ev, self.ctx)
            vec = torch.cat([avg[:-1], ev, avg[-1:]])
        # averaged per optimizer step like every other entry of the reference's metric lists (append_to_dict + np.mean);
        # a replayed graph ran every step at the current rates and logs nothing per step
        log = list(self._lr_log) or [(self._lrs[0], self._lrs[1])]
        xgmi = self._xgmi
        snap = xgmi is not None and self.defer_host_reads
        if snap:  # the exchange's status word rides along with THIS step's numbers (a blocking read here would wait for the whole queue,
            #       and one iteration late it would report -- and clear -- the NEXT step's time-out)
            vec = torch.cat([vec, xgmi.status_snapshot(torch.empty(1, dtype=torch.float32, device=vec.device))])

        def finish(host: list) -> dict:
            if snap:
                timed_out, host = host[-1] != 0.0, host[:-1]
                if timed_out:
                    xgmi.check_status()  # raises (and clears the word): a peer never published its gradient during this step
            out = {k: host[PPO_OUT_NAMES[k]] for k in _ACTOR_KEYS}
            if has_critic:
                out.update({k: host[PPO_OUT_NAMES[k]] for k in _CRITIC_KEYS})
                stats = {name: host[PPO_OUT_FLOATS + i] for i, name in enumerate(_EV_MAP.values())}
                out[CRITIC_EXPLAINED_VARIANCE_KEY] = explained_variance_from_stats(stats)
            out["actor/total_loss"] = host[PPO_OUT_NAMES["loss"]] / max(accum, 1)
            out["actor/entropy_loss"] = host[PPO_OUT_NAMES["actor/entropy_loss"]]
            out["actor/grad_norm"] = host[-1]
            self._one_launch_expired(host[-1])
            out["actor/lr"] = float(np.mean([a for a, _ in log]))
            critic = [c for _, c in log if c is not None]
            if critic:
                out["critic/lr"] = float(np.mean(critic))
            return out

        if self.defer_host_reads:  # the runner reads one iteration late (utils/pending.py)
            return PendingMetrics(vec, finish)
        return finish(vec.tolist())

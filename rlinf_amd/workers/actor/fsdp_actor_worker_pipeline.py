"""PipelineEmbodiedFSDPActor (rlinf/workers/actor/fsdp_actor_worker_pipeline.py:35-232): the learner of
``runner.use_training_pipeline`` -- it trains micro-batches while the env side is still producing them.

Mirrored here, with the reference's names: ``GlobalBatchState`` (:32-35), ``compute_micro_batches`` (:211-232),
``try_recv_micro_batch`` / ``recv_micro_batch`` (:56-76), ``select_global_batch`` (:78-84); the scheduling policy of
``run_training`` (:86-196) is restated over counters in ``training_schedule`` -- newly arrived micro-batches first; a complete global batch gets its optimizer step and is stored with
train_count 1; with nothing pending the learner takes the stored global batch with the LOWEST train count (oldest first) and
trains it again; with nothing stored either it blocks on the channel; done when every global batch has been trained
``update_epoch`` times.

What differs is what a "micro-batch on the channel" is.  Rollout and learner share one resident trajectory buffer per rollout
epoch: nothing is packed, sent or unpacked, a micro-batch is a row range of the epoch's shuffled buffer, and it "arrives" when the
rollout stream has recorded that epoch's completion event (the learner's stream waits for that event on the device; the host
never blocks).  ``MicroBatchFeed`` is that channel: it hands out micro-batch descriptors in the order the env side packs them
(epoch by epoch, stage shuffles inside, env_worker.py:1519-1590) and models WHEN the host sees them as available -- by default a
rollout that delivers every epoch just as the learner runs out of new micro-batches (``lag`` = 0: try_recv never comes back
empty), or ``lag`` stored-batch revisits later (the situation of a slow simulator, where first and later passes interleave).  The
queue logic turns the feed into the order of (pass, global batch) pairs that EmbodiedFSDPActor._run_pipeline_epochs executes as
prepared launches; tests/test_pipeline_schedule.py runs the REFERENCE's own run_training loop (compiled from its source) against a
scripted channel and checks that both produce the same training order, arrival pattern by arrival pattern."""

from __future__ import annotations

from collections import deque
from dataclasses import dataclass, field

from .embodied_fsdp_actor_worker import EmbodiedFSDPActor


@dataclass
class GlobalBatchState:
    micro_batches: list
    train_count: int = 0


@dataclass
class MicroBatchFeed:
    """The learner's side of the pipeline channel: ``total`` micro-batch descriptors (epoch, index in epoch), epoch-major.
    ``lag``: how many times ``get_nowait`` reports "nothing yet" in front of each NEW epoch's first micro-batch (0: a rollout that
    keeps up with the learner).  ``get`` blocks in the reference; here it means "wait for the epoch's event on the device"."""
    per_epoch: int
    epochs: int
    lag: int = 0
    _next: int = 0
    _misses: int = 0
    log: list = field(default_factory=list)

    def _ready(self) -> bool:
        if self._next >= self.per_epoch * self.epochs:
            return False
        if self._next % self.per_epoch != 0 or self._next == 0:
            return True  # the rest of an epoch that has started arriving is there (the env side sends an epoch at once)
        return self._misses >= self.lag

    def get_nowait(self):
        if not self._ready():
            self._misses += 1
            return None
        return self.get()

    def get(self):
        assert self._next < self.per_epoch * self.epochs, "the feed is exhausted"
        mb = (self._next // self.per_epoch, self._next % self.per_epoch)
        self._next += 1
        if self._next % self.per_epoch == 0:
            self._misses = 0
        self.log.append(mb)
        return mb


class PipelineEmbodiedFSDPActor(EmbodiedFSDPActor):
    def __init__(self, cfg, ctx=None):
        assert bool(cfg.runner.get("use_training_pipeline", False)), "PipelineEmbodiedFSDPActor needs runner.use_training_pipeline"
        super().__init__(cfg, ctx)
        tr = cfg.env.train
        self.update_epoch = int(cfg.algorithm.get("update_epoch", 1))
        self.micro_batches_per_step = self.compute_micro_batches(
            total_num_envs=tr.total_num_envs, actor_world_size=self._world_size, micro_batch_size=cfg.actor.micro_batch_size,
            rollout_epoch=tr.get("rollout_epoch", 1),
            n_train_chunk_steps=tr.max_steps_per_rollout_epoch // cfg.actor.model.num_action_chunks)
        assert self.micro_batches_per_step % self.gradient_accumulation == 0, (
            f"micro_batches_per_step ({self.micro_batches_per_step}) must be divisible by "
            f"gradient_accumulation ({self.gradient_accumulation}).")
        self.global_batches_per_step = self.micro_batches_per_step // self.gradient_accumulation
        # runner.pipeline_arrival_lag: see MicroBatchFeed (this package's knob; the reference's arrivals are whatever the
        # simulator's speed makes them)
        self.arrival_lag = int(cfg.runner.get("pipeline_arrival_lag", 0))
        self.last_schedule: list = []

    def compute_micro_batches(self, total_num_envs: int, actor_world_size: int, micro_batch_size: int, rollout_epoch: int,
                              n_train_chunk_steps: int) -> int:
        """How many pipeline-local micro-batches each actor rank receives per iteration (:211-232)."""
        total_rollout_samples = total_num_envs * rollout_epoch * n_train_chunk_steps
        per_rank_micro_batch_samples = actor_world_size * micro_batch_size
        assert total_rollout_samples % per_rank_micro_batch_samples == 0, (
            f"Total flattened rollout samples ({total_rollout_samples}) must be divisible by actor_world_size * micro_batch_size "
            f"({per_rank_micro_batch_samples}).")
        return total_rollout_samples // per_rank_micro_batch_samples

    # ---- the channel -------------------------------------------------------------------------------------------------
    def try_recv_micro_batch(self, input_channel):
        return input_channel.get_nowait()

    def recv_micro_batch(self, input_channel):
        return input_channel.get()

    def select_global_batch(self, global_batches):
        for epoch in range(1, self.update_epoch):
            if global_batches[epoch]:
                return global_batches[epoch].popleft()
        return None

    # ---- the training order, derived from the arrival pattern -----------------------------------------------------------------
    def training_schedule(self, input_channel) -> list:
        """-> [(pass k, global batch g)] in training order -- the order the reference's run_training (:86-165) trains in, derived from
        counters instead of queues of micro-batch objects (the arithmetic runs afterwards, as prepared launches in exactly this
        order; tests/test_pipeline_schedule.py holds it to the reference's own loop, arrival pattern by arrival pattern).

        The policy: (1) take every micro-batch that has ALREADY arrived; (2) global batch g -- arrivals g A .. (g + 1) A - 1 for
        gradient-accumulation A -- gets its first pass as soon as its last micro-batch is there; (3) with no first pass to give,
        revisit the stored global batch with the fewest passes so far, oldest first; (4) with nothing stored either, wait for one
        more arrival; done when every global batch has had ``update_epoch`` passes."""
        accum, n_micro, n_global, passes = self.gradient_accumulation, self.micro_batches_per_step, self.global_batches_per_step, self.update_epoch
        order: list = []
        arrived = 0                                          # micro-batches taken off the channel
        first_passes = 0                                     # global batches 0 .. first_passes - 1 have had pass 0
        had = [deque() for _ in range(passes + 1)]           # had[c]: global batches with c passes behind them, oldest first
        while len(had[passes]) < n_global:
            while arrived < n_micro and self.try_recv_micro_batch(input_channel) is not None:
                arrived += 1
            while (first_passes + 1) * accum <= arrived:     # complete global batches, in arrival order
                order.append((0, first_passes))
                had[1].append(first_passes)
                first_passes += 1
            fewest = next((c for c in range(1, passes) if had[c]), None)
            if fewest is not None:
                g = had[fewest].popleft()
                order.append((fewest, g))
                had[fewest + 1].append(g)
            elif arrived < n_micro and len(had[passes]) < n_global:
                self.recv_micro_batch(input_channel)         # nothing to revisit: block for the next micro-batch
                arrived += 1
        return order

    def _pipeline_schedule(self, E: int, n_e: int, passes: int) -> list:
        assert passes == self.update_epoch and E * n_e == self.global_batches_per_step, (E, n_e, passes)
        feed = MicroBatchFeed(per_epoch=n_e * self.gradient_accumulation, epochs=E, lag=self.arrival_lag)
        self.last_schedule = self.training_schedule(feed)
        return self.last_schedule

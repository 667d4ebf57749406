"""The ``send`` / ``recv`` callables the weight syncers are driven with, for the two placements of this build.

The reference hands its syncers closures over ``Worker.broadcast`` / ``Worker.send`` / ``Worker.recv``
(rlinf/workers/actor/embodied_fsdp_actor_worker.py:142-164: the learner's rank 0 broadcasts every payload to all rollout ranks and
receives the receiver's metadata from rollout rank 0; rlinf/workers/rollout/hf/huggingface_worker.py:633-655: the rollout ranks
receive that broadcast, rollout rank 0 sends the metadata to EVERY learner rank).  The same four closures here:

* ``GroupLink`` -- split placement: torch.distributed over the placement's weight-sync group (actor rank 0 + all rollout ranks:
  RCCL over xGMI for device payloads, one flat byte buffer per bucket / one broadcast per patch field, scheduler/dist.py) and
  point-to-point pickled objects for the metadata.
* ``InProcessLink`` -- learner and rollout worker in ONE process, the rollout keeping its own copy of the weights
  (``rollout.share_actor_weights: false``): two FIFO queues.  The receiver half runs in the calling thread, the sender half in a
  helper thread for the duration of the sync (the syncers' init hand-shake blocks on both sides, like the reference's two
  concurrently running actors).
"""

from __future__ import annotations

import queue
import threading
from typing import Any, Callable


class InProcessLink:
    TIMEOUT_S = 120.0

    def __init__(self):
        self._to_rollout: queue.Queue = queue.Queue()
        self._to_actor: queue.Queue = queue.Queue()

    def _get(self, q: queue.Queue, who: str):
        try:
            item = q.get(timeout=self.TIMEOUT_S)
        except queue.Empty:
            raise RuntimeError(f"in-process weight sync: {who} waited {self.TIMEOUT_S:.0f} s for a payload that never came") from None
        if isinstance(item, _Failure):
            raise RuntimeError(f"in-process weight sync: the other half failed: {item.error!r}") from item.error
        return item

    # the learner's closures
    def actor_send(self, data: Any) -> None:
        self._to_rollout.put(data)

    def actor_recv(self) -> Any:
        return self._get(self._to_actor, "the learner")

    # the rollout worker's closures
    def rollout_send(self, data: Any) -> None:
        self._to_actor.put(data)

    def rollout_recv(self) -> Any:
        return self._get(self._to_rollout, "the rollout worker")

    def run(self, sender_half: Callable[[], None], receiver_half: Callable[[], Any]):
        """``sender_half`` on a helper thread, ``receiver_half`` here; a failure on either side reaches the other as an exception
        instead of a hang."""
        failure: list = []

        def target():
            try:
                sender_half()
            except BaseException as e:  # noqa: BLE001 -- handed to the receiver, re-raised below
                failure.append(e)
                self._to_rollout.put(_Failure(e))

        t = threading.Thread(target=target, name="rlx-weight-sync-sender", daemon=True)
        t.start()
        try:
            out = receiver_half()
        except BaseException as e:
            self._to_actor.put(_Failure(e))
            t.join(self.TIMEOUT_S)
            if failure:
                raise failure[0] from e
            raise
        t.join(self.TIMEOUT_S)
        if failure:
            raise failure[0]
        return out


class _Failure:
    def __init__(self, error: BaseException):
        self.error = error


class GroupLink:
    """``placement``: a split HybridComponentPlacement (utils/placement.py); built on every rank that takes part."""

    def __init__(self, placement, device=None):
        self.placement, self.device = placement, device
        self.sync_ctx = placement.sync_ctx          # actor rank 0 is index 0 of this group, the rollout ranks follow
        self.actor_ranks = placement.ranks("actor")
        self.rollout_ranks = placement.ranks("rollout")
        self.me = placement.world_ctx.rank

    # the learner's closures (embodied_fsdp_actor_worker.py:142-164)
    def actor_send(self, data: Any) -> None:
        if self.me != self.actor_ranks[0]:
            return  # (only the weight sender broadcasts; the other learner ranks are not in the group)
        from ..scheduler.dist import broadcast_payload
        broadcast_payload(data, self.sync_ctx, src=0, device=self.device)

    def actor_recv(self) -> Any:
        from ..scheduler.dist import recv_object
        return recv_object(self.rollout_ranks[0])

    # the rollout worker's closures (huggingface_worker.py:633-655)
    def rollout_recv(self) -> Any:
        from ..scheduler.dist import broadcast_payload
        return broadcast_payload(None, self.sync_ctx, src=0, device=self.device)

    def rollout_send(self, data: Any) -> None:
        if self.me != self.rollout_ranks[0]:
            return
        from ..scheduler.dist import send_object
        for dst in self.actor_ranks:
            send_object(data, dst)


def weight_syncer_config(cfg):
    """``cfg.weight_syncer`` (hydra default ``weight_syncer/patch_syncer@weight_syncer`` of the shipped configurations; read by both
    workers: embodied_fsdp_actor_worker.py:80-82, huggingface_worker.py:120-125), else ``cfg.rollout.weight_syncer``; None when the
    configuration names none."""
    get = cfg.get if hasattr(cfg, "get") else (lambda k, d=None: getattr(cfg, k, d))
    node = get("weight_syncer", None)
    if node is None:
        ro = get("rollout", None)
        node = ro.get("weight_syncer", None) if ro is not None and hasattr(ro, "get") else None
    return node

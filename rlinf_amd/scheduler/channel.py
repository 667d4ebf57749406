"""Channel: the in-process stand-in for RLinf's distributed FIFO (rlinf/scheduler/channel/channel.py: ``Channel.create(name)``,
``put(item, key=...)``, ``get(key=...)``, ``qsize``).  The reference moves trajectories, observations and actions between Ray
actors through these; here the three workers of a rank live in one process and share device memory, so a channel only carries
object references (views of the resident trajectory buffer) between calls that keep the reference's signatures
(``env.interact(input_channel=..., rollout_channel=..., actor_channel=...)``,
``actor.recv_rollout_trajectories(input_channel=...)``, embodied_runner.py:478-563).  Control plane: nothing here is on the
measured path."""

from __future__ import annotations

from collections import defaultdict, deque
from typing import Any


class Channel:
    DEFAULT_KEY = "default_queue"

    def __init__(self, name: str, maxsize: int = 0):
        self.name, self.maxsize = name, maxsize
        self._queues: dict = defaultdict(deque)

    @classmethod
    def create(cls, name: str, maxsize: int = 0, **_ignored) -> "Channel":
        return cls(name, maxsize)

    def put(self, item: Any, weight: int = 0, key: Any = None, async_op: bool = False):
        self._queues[self.DEFAULT_KEY if key is None else key].append(item)
        return None

    def get(self, key: Any = None, async_op: bool = False):
        q = self._queues[self.DEFAULT_KEY if key is None else key]
        if not q:
            raise RuntimeError(f"channel {self.name!r} is empty (key={key!r}): in-process channels never block -- the producer "
                               "call must come first")
        return q.popleft()

    def qsize(self, key: Any = None) -> int:
        return len(self._queues[self.DEFAULT_KEY if key is None else key])

    def empty(self, key: Any = None) -> bool:
        return self.qsize(key) == 0

"""In-process worker plumbing that stands in for RLinf's Ray worker groups on this path.

The reference launches every worker class as a Ray actor group and drives it through RPC handles
(``Worker.create_group(cfg).launch(...)``, ``group.method(...).wait()``,
rlinf/scheduler/worker/worker_group.py:144-197,406-557).  Here there is one process per GPU (torchrun-style
ranks) and the three workers of a rank live in that process, so a "group" is the local instance and a handle
resolves immediately; the method names, call order and return conventions are kept so the runner reads like
the reference's.
"""

from __future__ import annotations

import time
from contextlib import contextmanager
from typing import Any


class Handle:
    """Result of a (synchronous) worker call; mirrors WorkerGroupFuncResult.wait()/consume_durations()."""

    def __init__(self, value: Any, duration: float = 0.0):
        self._value, self._duration = value, duration

    def wait(self):
        return [self._value]  # one entry per rank in the group: this rank's

    def consume_durations(self):
        return {"": self._duration}


class Worker:
    """Base class: rank bookkeeping + the ``timer`` metrics of rlinf/scheduler/worker/worker.py:1322-1376."""

    ROLE = None  # "actor" / "rollout" / "env": the key under which the process-local peers find each other

    def __init__(self, cfg, ctx=None):
        from ..scheduler import DistContext

        self.cfg = cfg
        self.ctx = ctx or DistContext()
        self._rank, self._world_size = self.ctx.rank, self.ctx.world_size
        # True at world_size > 1, and at world_size 1 under RLX_FORCE_EXCHANGE (scheduler/dist.py): the gradient exchange runs
        self._exchange = bool(getattr(self.ctx, "exchanges_gradients", self._world_size > 1))
        self.device = self.ctx.device
        self._timer_metrics: dict = {}

    @classmethod
    def create_group(cls, cfg, ctx=None):
        return _Group(cls, cfg, ctx)

    @contextmanager
    def timer(self, tag: str):
        t0 = time.perf_counter()
        yield
        self._timer_metrics[tag] = self._timer_metrics.get(tag, 0.0) + time.perf_counter() - t0

    def pop_execution_times(self) -> dict:
        out, self._timer_metrics = self._timer_metrics, {}
        return out


# The workers of THIS process by role ("actor" / "rollout" / "env"): what the reference reaches through named Ray groups and
# channels (``self.send_to(group_name=cfg.rollout.group_name, ...)``) is a direct object reference here.
_PEERS: dict = {}


def peer(role: str, cfg=None):
    """The worker of ``role`` launched in this process, or None.  ``cfg``: the asking worker's own configuration -- a worker that
    was launched with a DIFFERENT configuration (an earlier runner of the same process, a tool that built one worker on its own)
    is not a peer: its model / buffers must never be adopted silently."""
    w = _PEERS.get(role)
    if w is not None and cfg is not None and not (w.cfg is cfg or w.cfg == cfg):
        return None
    return w


def clear_peers(*workers) -> None:
    """Forget the given workers (all of them without arguments): a runner that is torn down must not leave its workers behind
    for the next one to adopt."""
    if not workers:
        _PEERS.clear()
        return
    for role in [r for r, w in _PEERS.items() if any(w is x or w is getattr(x, "worker", None) for x in workers)]:
        del _PEERS[role]


def _role_of(cls) -> str:
    return getattr(cls, "ROLE", None) or cls.__name__


class _Group:
    """``Worker.create_group(cfg).launch(cluster, name=..., placement_strategy=...)`` -> the local worker, whose
    methods return Handles.  The rank context comes from ``create_group(cfg, ctx)`` (this package's own callers), from the
    ``cluster`` (the reference's call, rlinf_amd.scheduler.Cluster) or from the torchrun environment, in that order."""

    def __init__(self, cls, cfg, ctx):
        self._cls, self._cfg, self._ctx = cls, cfg, ctx

    def launch(self, cluster=None, name: str = "", placement_strategy=None):
        ctx = self._ctx
        placement = getattr(placement_strategy, "placement", None)
        if placement is not None and getattr(placement, "split", False):
            # split placement (utils/placement.py): the component runs on ITS ranks only, in the component's own rank context
            if not placement_strategy.present:
                return _AbsentProxy(_role_of(self._cls), name)
            ctx = placement_strategy.ctx
        if ctx is None and cluster is not None:
            ctx = getattr(cluster, "ctx", None)
        if ctx is None and placement_strategy is not None:
            ctx = getattr(placement_strategy, "ctx", None)
        if ctx is None:
            from ..scheduler import init_distributed
            ctx = init_distributed()
        worker = self._cls(self._cfg, ctx)
        worker.group_name = name
        worker.placement = placement  # None: launched without a placement (collocated, this package's own callers)
        _PEERS[_role_of(self._cls)] = worker
        return _HandleProxy(worker)


class _HandleProxy:
    def __init__(self, worker):
        object.__setattr__(self, "worker", worker)

    def __getattr__(self, item):
        attr = getattr(self.worker, item)
        if not callable(attr):
            return attr

        def call(*a, **kw):
            t0 = time.perf_counter()
            a = tuple(x.worker if isinstance(x, _HandleProxy) else x for x in a)
            kw = {k: (v.worker if isinstance(v, _HandleProxy) else v) for k, v in kw.items()}
            out = attr(*a, **kw)
            return Handle(out, time.perf_counter() - t0)

        return call


class _AbsentProxy:
    """The handle of a worker group none of whose workers runs in THIS process (split placement): the reference's runner calls
    every group from its one driver process and the call reaches the ranks that host the group; here every rank runs the runner,
    and a call on a group that is elsewhere is simply not this rank's to execute."""

    worker = None

    def __init__(self, role: str, name: str):
        object.__setattr__(self, "role", role)
        object.__setattr__(self, "group_name", name)

    def __getattr__(self, item):
        if item.startswith("__"):
            raise AttributeError(item)
        return lambda *a, **kw: Handle(None)

    def __bool__(self):
        return False

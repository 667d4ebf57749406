"""HybridComponentPlacement (rlinf/utils/placement.py:86-): maps components (actor / rollout / env / reward) to hardware ranks
from ``cluster.component_placement``.  This build runs every component in every rank of the torchrun-style world (the shipped
collocated ``actor,env,rollout: all`` pattern), so a strategy is just the rank context; the class keeps the entry point's
calls (``HybridComponentPlacement(cfg, cluster).get_strategy("actor")``, ``get_world_size``) working unchanged."""

from __future__ import annotations


class _CollocatedStrategy:
    def __init__(self, component: str, ctx):
        self.component, self.ctx = component, ctx


class HybridComponentPlacement:
    def __init__(self, cfg, cluster):
        self._cfg, self._cluster = cfg, cluster

    def get_strategy(self, component: str) -> _CollocatedStrategy:
        return _CollocatedStrategy(component, self._cluster.ctx)

    def get_world_size(self, component: str) -> int:
        return self._cluster.ctx.world_size

"""HybridComponentPlacement (rlinf/utils/placement.py:86-97 over ComponentPlacement): maps the components of a job (actor /
rollout / env) to the ranks of the torchrun-style world from ``cluster.component_placement``.

Two placements are served:

* **collocated** -- every component on every rank (the shipped ``env,rollout,actor: 0`` / ``all`` pattern, scaled to N GPUs by the
  launcher): actor, rollout and env of a rank live in one process and share device memory.  Any placement whose actor and rollout
  rank sets are EQUAL is run this way.
* **split** -- actor ranks and rollout ranks DISJOINT (``actor: 0-3`` / ``env,rollout: 4-7``; BASELINE.json configs[3]: "rollout /
  learner split across 8 x MI355X over xGMI"): the learner ranks form their own process group (gradient exchange, metric
  reductions), rollout rank i generates on its env shard and ships the finished trajectory buffer to learner rank i, and the
  weights travel the reference's way -- ``cfg.weight_syncer`` (bucket or patch) from actor rank 0, broadcast to every rollout rank
  (embodied_fsdp_actor_worker.py:131-178, huggingface_worker.py:629-675).  ``env`` must sit with ``rollout`` (the policy is called
  in process; the reference's env <-> rollout channel layer is control plane this build does not have) and the two sides must have
  the same number of ranks (the trajectory route is 1 : 1).

The reference's entry point (``HybridComponentPlacement(cfg, cluster).get_strategy("actor")``, ``get_world_size``) works unchanged;
a strategy here carries the component's rank context, or says that the component does not run in this process."""

from __future__ import annotations

from typing import Optional


def parse_rank_spec(spec, world_size: int) -> list:
    """``all`` | ``3`` | ``0-3`` | ``0-1,4,6-7`` -> sorted rank list (the reference's hardware-rank syntax)."""
    if isinstance(spec, int):
        return [int(spec)]
    text = str(spec).strip().lower()
    if text == "all":
        return list(range(world_size))
    ranks = set()
    for part in text.split(","):
        part = part.strip()
        if not part:
            continue
        if "-" in part:
            lo, hi = part.split("-", 1)
            lo, hi = int(lo), int(hi)
            if hi < lo:
                raise ValueError(f"component_placement: empty range {part!r}")
            ranks.update(range(lo, hi + 1))
        else:
            ranks.add(int(part))
    if not ranks:
        raise ValueError(f"component_placement: no ranks in {spec!r}")
    return sorted(ranks)


def parse_component_placement(placement_cfg, world_size: int) -> dict:
    """{"env,rollout,actor": "all"} / {"actor": "0-3", "env,rollout": "4-7"} -> {component: [ranks]}."""
    out: dict = {}
    items = placement_cfg.items() if hasattr(placement_cfg, "items") else []
    for key, spec in items:
        if hasattr(spec, "get"):  # the reference's long form {placement: "0-3", ...}: only the ranks matter here
            spec = spec.get("placement", "all")
        ranks = parse_rank_spec(spec, world_size)
        for comp in str(key).split(","):
            comp = comp.strip()
            if comp:
                if comp in out:
                    raise ValueError(f"component_placement names {comp!r} twice")
                out[comp] = ranks
    return out


class ComponentStrategy:
    """What ``launch(..., placement_strategy=...)`` needs: does the component run in this process, and with which rank context."""

    def __init__(self, component: str, ctx, present: bool = True, placement: Optional["HybridComponentPlacement"] = None):
        self.component, self.ctx, self.present, self.placement = component, ctx, present, placement


class HybridComponentPlacement:
    COMPONENTS = ("actor", "rollout", "env")

    def __init__(self, cfg, cluster):
        self._cfg, self._cluster = cfg, cluster
        self.world_ctx = cluster.ctx
        world = self.world_ctx.world_size
        placement_cfg = None
        cl = cfg.get("cluster", None) if hasattr(cfg, "get") else None
        if cl is not None:
            placement_cfg = cl.get("component_placement", None)
        parsed = parse_component_placement(placement_cfg, world) if placement_cfg is not None else {}
        self._ranks = {c: parsed.get(c) for c in self.COMPONENTS}
        a, r, e = (self._ranks[c] for c in self.COMPONENTS)
        self.split = a is not None and r is not None and not (set(a) & set(r)) and world > 1
        self._ctxs: dict = {}
        self.sync_ctx = None
        if not self.split:
            if a is not None and r is not None and set(a) != set(r) and world > 1:
                raise NotImplementedError(f"component_placement: actor ranks {a} and rollout ranks {r} overlap without being equal; "
                                          "collocated (equal sets) and split (disjoint sets) placements are served")
            self._ranks = {c: list(range(world)) for c in self.COMPONENTS}  # collocated: every component on every rank
            return
        if e is None or set(e) != set(r):
            raise NotImplementedError(f"component_placement: env ranks {e} must equal rollout ranks {r} -- the policy is called in "
                                      "process (the reference's env <-> rollout channel layer is not part of this build)")
        if len(a) != len(r):
            raise NotImplementedError(f"component_placement: {len(a)} actor ranks against {len(r)} rollout ranks; the trajectory "
                                      "route of the split placement is 1 : 1")
        bad = [x for x in a + r if not 0 <= x < world]
        if bad:
            raise ValueError(f"component_placement names ranks {bad} outside the job's world of {world}")
        self._build_split_contexts()

    # ---- split placement: process groups and per-component contexts ------------------------------------------------------
    def _build_split_contexts(self):
        """Collective: every rank of the job creates the same groups in the same order (torch.distributed.new_group's rule)."""
        import torch.distributed as dist

        from ..scheduler.dist import DistContext

        w = self.world_ctx
        assert dist.is_initialized(), "a split placement needs an initialised process group (torchrun-style environment)"
        a, r = self._ranks["actor"], self._ranks["rollout"]
        g_actor = dist.new_group(a)
        g_rollout = dist.new_group(r)
        sync_ranks = [a[0]] + list(r)
        g_sync = dist.new_group(sync_ranks)

        def ctx_of(ranks, group):
            if w.rank not in ranks:
                return None
            c = DistContext(ranks.index(w.rank), w.local_rank, len(ranks), w.device, group=group, global_ranks=list(ranks))
            return c

        self._ctxs = {"actor": ctx_of(a, g_actor), "rollout": ctx_of(r, g_rollout), "env": ctx_of(r, g_rollout)}
        # the weight-sync group: actor rank 0 (index 0) + every rollout rank, the broadcast domain of the reference's send_func
        self.sync_ctx = ctx_of(sync_ranks, g_sync)

    # ---- the reference's accessors --------------------------------------------------------------------------------------------
    def get_strategy(self, component: str) -> ComponentStrategy:
        if not self.split:
            return ComponentStrategy(component, self.world_ctx, True, self)
        ctx = self._ctxs.get(component)
        return ComponentStrategy(component, ctx, ctx is not None, self)

    def get_world_size(self, component: str) -> int:
        return len(self._ranks[component])

    def ranks(self, component: str) -> list:
        return list(self._ranks[component])

    def has(self, component: str, global_rank: Optional[int] = None) -> bool:
        return (self.world_ctx.rank if global_rank is None else global_rank) in self._ranks[component]

    def peer_of(self, component_from: str, component_to: str) -> int:
        """Global rank of the ``component_to`` rank paired with THIS rank of ``component_from`` (1 : 1 route)."""
        i = self._ranks[component_from].index(self.world_ctx.rank)
        return self._ranks[component_to][i]

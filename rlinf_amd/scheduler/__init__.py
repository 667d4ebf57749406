"""Minibatch / placement arithmetic and the one-process-per-GPU launcher that replaces RLinf's Ray scheduler
for this path (rlinf/scheduler/: only the rank/env-shard arithmetic and the routing maps are in scope)."""

from .channel import Channel  # noqa: F401
from .dist import DistContext, all_reduce_flat_, all_reduce_scalars, init_distributed, ranks_share_a_device  # noqa: F401
from .placement import compute_split_num, env_shard  # noqa: F401
from .routing import CommMapper, RouteEntry, RoutePlan, build_recv_plan, build_send_plan  # noqa: F401


class Cluster:
    """``Cluster(cluster_cfg=..., distributed_log_dir=...)`` (rlinf/scheduler/cluster/cluster.py): the reference starts / joins
    a Ray cluster here.  One process per GPU under torchrun has nothing to start: this records the rank context the worker
    groups are launched into (control plane, out of scope beyond the call signature)."""

    def __init__(self, cluster_cfg=None, distributed_log_dir=None, ctx=None):
        self.cfg, self.distributed_log_dir = cluster_cfg, distributed_log_dir
        self.ctx = ctx if ctx is not None else init_distributed()
        self.num_nodes = int(cluster_cfg.get("num_nodes", 1)) if cluster_cfg is not None else 1

"""Minibatch / placement arithmetic and the one-process-per-GPU launcher that replaces RLinf's Ray scheduler
for this path (rlinf/scheduler/: only the rank/env-shard arithmetic and the routing maps are in scope)."""

from .dist import DistContext, all_reduce_flat_, all_reduce_scalars, init_distributed  # noqa: F401
from .placement import compute_split_num, env_shard  # noqa: F401
from .routing import CommMapper, RouteEntry, RoutePlan, build_recv_plan, build_send_plan  # noqa: F401

"""Batch-range -> rank routing maps (mirror of rlinf/scheduler/worker/routing.py:132-196 CommMapper and the
send/recv plans pinned by the reference's tests/unit_tests/test_comm_mapper.py:40-120).  Pure integer logic:
the global batch is cut into equal contiguous ranges per source rank and per destination rank; a source range is
sent to every destination whose range it overlaps."""

from __future__ import annotations

from dataclasses import dataclass, field


class CommMapper:
    @staticmethod
    def _check(batch_size: int, src_world_size: int, dst_world_size: int):
        assert batch_size % src_world_size == 0, (
            f"batch_size ({batch_size}) must be divisible by src_world_size ({src_world_size}).")
        assert batch_size % dst_world_size == 0, (
            f"batch_size ({batch_size}) must be divisible by dst_world_size ({dst_world_size}).")

    @staticmethod
    def get_dst_ranks(batch_size: int, src_world_size: int, dst_world_size: int, src_rank: int) -> list[tuple[int, int]]:
        CommMapper._check(batch_size, src_world_size, dst_world_size)
        assert 0 <= src_rank < src_world_size, f"src_rank ({src_rank}) must be in [0, {src_world_size})."
        per_src, per_dst = batch_size // src_world_size, batch_size // dst_world_size
        lo, hi = src_rank * per_src, (src_rank + 1) * per_src
        out = []
        while lo < hi:
            dst = lo // per_dst
            take = min((dst + 1) * per_dst, hi) - lo
            out.append((dst, take))
            lo += take
        return out

    @staticmethod
    def get_src_ranks(batch_size: int, src_world_size: int, dst_world_size: int, dst_rank: int) -> list[tuple[int, int]]:
        CommMapper._check(batch_size, src_world_size, dst_world_size)
        assert 0 <= dst_rank < dst_world_size, f"dst_rank ({dst_rank}) must be in [0, {dst_world_size})."
        out = []
        for src in range(src_world_size):
            for dst, size in CommMapper.get_dst_ranks(batch_size, src_world_size, dst_world_size, src):
                if dst == dst_rank:
                    out.append((src, size))
        assert sum(s for _, s in out) == batch_size // dst_world_size
        return out


@dataclass(frozen=True)
class RouteEntry:
    peer_rank: int
    batch_size: int
    offset: int  # offset of this shard inside the LOCAL batch of the planning rank


@dataclass(frozen=True)
class RoutePlan:
    src_group_name: str
    dst_group_name: str
    tag: str
    entries: list = field(default_factory=list)


def build_send_plan(src_group_name, dst_group_name, src_rank, src_world_size, dst_world_size, tag, batch_size) -> RoutePlan:
    entries, off = [], 0
    for dst, size in CommMapper.get_dst_ranks(batch_size, src_world_size, dst_world_size, src_rank):
        entries.append(RouteEntry(dst, size, off))
        off += size
    return RoutePlan(src_group_name, dst_group_name, tag, entries)


def build_recv_plan(src_group_name, dst_group_name, dst_rank, src_world_size, dst_world_size, tag, batch_size) -> RoutePlan:
    entries, off = [], 0
    for src, size in CommMapper.get_src_ranks(batch_size, src_world_size, dst_world_size, dst_rank):
        entries.append(RouteEntry(src, size, off))
        off += size
    return RoutePlan(src_group_name, dst_group_name, tag, entries)

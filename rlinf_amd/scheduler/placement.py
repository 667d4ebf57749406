"""Env-shard and trajectory-split arithmetic (rlinf/workers/env/env_worker.py:137-140,1463-1467,
rlinf/utils/metric_utils.py:228-229).  Pure integer logic."""

from __future__ import annotations

import math


def compute_split_num(num: int, split_num: int) -> int:
    """lcm(num, split_num) // split_num (metric_utils.py:228-229)."""
    return math.lcm(num, split_num) // split_num


def env_shard(total_num_envs: int, env_world_size: int, stage_num: int, rank: int, stage: int = 0) -> tuple[int, int]:
    """[begin, end) of the envs owned by (rank, stage): total // world // stages each (env_worker.py:137-140).
    Shards are contiguous in rank-major, stage-minor order, so GRPO groups (consecutive envs) stay intact when
    num_envs_per_stage % group_size == 0 (rlinf/config.py:1109-1117)."""
    assert total_num_envs % env_world_size == 0 and (total_num_envs // env_world_size) % stage_num == 0
    per = total_num_envs // env_world_size // stage_num
    begin = (rank * stage_num + stage) * per
    return begin, begin + per


def minibatch_plan(rollout_size: int, global_batch_size: int, micro_batch_size: int, world_size: int):
    """(num_minibatches, per_rank_batch, grad_accum) as EmbodiedFSDPActor.run_training derives them
    (rlinf/workers/actor/embodied_fsdp_actor_worker.py:91-95,518-549)."""
    per_rank = global_batch_size // world_size
    assert rollout_size % per_rank == 0, f"{rollout_size} is not divisible by {per_rank}"
    assert per_rank % micro_batch_size == 0, f"train_global_batch_size={per_rank}, {micro_batch_size}"
    return rollout_size // per_rank, per_rank, global_batch_size // micro_batch_size // world_size

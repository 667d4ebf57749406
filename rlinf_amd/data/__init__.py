"""Trajectory buffer and per-step payload types (mirror of rlinf/data/schema/embodied_types.py and
embodied_trajectory_builder.py:37-311), device-resident."""

from .embodied_types import (  # noqa: F401
    ChunkStepResult,
    EnvOutput,
    PolicyOutput,
    Trajectory,
    TrajectoryBuffer,
    convert_trajectories_to_batch,
)

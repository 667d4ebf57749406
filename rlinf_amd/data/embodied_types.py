"""Device-resident trajectory buffer.

The reference appends per-step CPU tensors to Python lists, ``torch.stack``s them at the end and forces every
payload to the CPU on construction (rlinf/data/schema/embodied_types.py:56-96,297-311,358-376;
embodied_trajectory_builder.py:72-93,176-230), ~46 MB of copies per iteration at 1024 x 128.  Here the buffer is
preallocated once in HBM as struct-of-arrays, time-major ``[T(+1), B, ...]`` -- the layout the scan kernels
index directly -- and the rollout kernel writes its outputs straight into row ``t``; nothing is stacked, nothing
leaves the device.  Field names and row counts are the reference's:
    actions / prev_logprobs / versions / forward_inputs.* / rewards : T rows
    dones / terminations / truncations / prev_values              : T+1 rows
"""

from __future__ import annotations

from dataclasses import dataclass, field
from typing import Any, Optional

import torch


@dataclass
class EnvOutput:
    obs: dict
    final_obs: Optional[dict] = None
    dones: Optional[torch.Tensor] = None         # [B, C] bool
    terminations: Optional[torch.Tensor] = None
    truncations: Optional[torch.Tensor] = None
    rewards: Optional[torch.Tensor] = None       # [B, C] f32


@dataclass
class PolicyOutput:
    actions: Optional[torch.Tensor] = None       # [B, C, A]
    prev_logprobs: Optional[torch.Tensor] = None  # [B, C*A]
    prev_values: Optional[torch.Tensor] = None   # [B, val]
    forward_inputs: dict = field(default_factory=dict)
    versions: Optional[torch.Tensor] = None
    bootstrap_values: Optional[torch.Tensor] = None  # [B, 1]


@dataclass
class ChunkStepResult:
    actions: Optional[torch.Tensor] = None
    prev_logprobs: Optional[torch.Tensor] = None
    prev_values: Optional[torch.Tensor] = None
    forward_inputs: dict = field(default_factory=dict)
    versions: Optional[torch.Tensor] = None
    dones: Optional[torch.Tensor] = None
    terminations: Optional[torch.Tensor] = None
    truncations: Optional[torch.Tensor] = None
    rewards: Optional[torch.Tensor] = None


@dataclass
class Trajectory:
    """A view of (a batch slice of) the buffer in the reference's field layout."""
    max_episode_length: int = 0
    actions: Optional[torch.Tensor] = None
    rewards: Optional[torch.Tensor] = None
    terminations: Optional[torch.Tensor] = None
    truncations: Optional[torch.Tensor] = None
    dones: Optional[torch.Tensor] = None
    prev_logprobs: Optional[torch.Tensor] = None
    prev_values: Optional[torch.Tensor] = None
    versions: Optional[torch.Tensor] = None
    forward_inputs: dict = field(default_factory=dict)
    # (buffer, begin, end) when this is a column range of a resident TrajectoryBuffer: adjacent ranges of one buffer are
    # merged back into a single view by convert_trajectories_to_batch instead of being concatenated (copied)
    origin: Optional[tuple] = field(default=None, repr=False, compare=False)

    _TENSOR_FIELDS = ("actions", "rewards", "terminations", "truncations", "dones", "prev_logprobs", "prev_values",
                      "versions")


class TrajectoryBuffer:
    def __init__(self, num_steps: int, batch: int, obs_dim: int, action_dim: int, num_action_chunks: int = 1,
                 value_dim: Optional[int] = None, device: Any = "cuda", max_episode_length: int = 0):
        self.T, self.B, self.C = int(num_steps), int(batch), int(num_action_chunks)
        self.A = self.C * int(action_dim)
        self.D = int(obs_dim)
        self.V = self.C if value_dim is None else int(value_dim)
        self.max_episode_length = max_episode_length
        dev = torch.device(device)
        T, B = self.T, self.B
        z = lambda *s, dt=torch.float32: torch.zeros(s, dtype=dt, device=dev)  # noqa: E731
        self.states = z(T, B, self.D)
        self.actions = z(T, B, self.A)
        self.prev_logprobs = z(T, B, self.A)
        self.versions = z(T, B, self.A)
        self.prev_values = z(T + 1, B, self.V)
        self.rewards = z(T, B, self.C)
        self.dones = z(T + 1, B, self.C, dt=torch.bool)
        self.terminations = z(T + 1, B, self.C, dt=torch.bool)
        self.truncations = z(T + 1, B, self.C, dt=torch.bool)
        self.reset()

    def reset(self):
        self.t = 0          # rows of actions / logprobs / states filled so far
        self.t_env = 0      # rows of dones (after the bootstrap row 0) / rewards filled so far
        self.dones[0] = False
        self.terminations[0] = False
        self.truncations[0] = False

    # ---- zero-copy write targets -------------------------------------------------------------------------
    def policy_rows(self, t: int, cols: slice = slice(None)):
        """(action, logprob, value) rows the rollout kernel writes step t into.  ``cols``: a block of batch columns --
        rollout epoch e owns columns [e*B_env, (e+1)*B_env), which IS process_nested_dict_for_adv's folded layout
        [n, rollout_epoch * bsz, ...] (nested_dict_process.py:251-269), written in place instead of reshaped into."""
        return self.actions[t, cols], self.prev_logprobs[t, cols], self.prev_values[t, cols]

    def env_rows(self, t: int, cols: slice = slice(None)):
        """(rewards[t], dones[t+1], terminations[t+1], truncations[t+1]) written after env step t."""
        return (self.rewards[t, cols], self.dones[t + 1, cols], self.terminations[t + 1, cols],
                self.truncations[t + 1, cols])

    # ---- reference-style append (copies; kept for API compatibility) ----------------------------------------
    def append_step_result(self, result: ChunkStepResult):
        """embodied_trajectory_builder.py:72-93: a row carries the policy fields of step t and the env fields of the
        PREVIOUS env step; the bootstrap row has no reward; the last row has only env fields + prev_values."""
        if result.dones is not None:
            row = self.t_env  # dones rows run 0..T
            self.dones[row].copy_(result.dones.reshape(self.B, self.C))
            if result.terminations is not None:
                self.terminations[row].copy_(result.terminations.reshape(self.B, self.C))
            if result.truncations is not None:
                self.truncations[row].copy_(result.truncations.reshape(self.B, self.C))
            if result.rewards is not None:
                self.rewards[row - 1].copy_(result.rewards.reshape(self.B, self.C))
            self.t_env += 1
        if result.prev_values is not None:
            self.prev_values[self.t if result.actions is not None else self.T].copy_(result.prev_values.reshape(self.B, self.V))
        if result.actions is not None:
            t = self.t
            self.actions[t].copy_(result.actions.reshape(self.B, self.A))
            if result.prev_logprobs is not None:
                self.prev_logprobs[t].copy_(result.prev_logprobs.reshape(self.B, self.A))
            if result.versions is not None:
                self.versions[t].copy_(result.versions.reshape(self.B, self.A))
            if result.forward_inputs and "states" in result.forward_inputs:
                self.states[t].copy_(result.forward_inputs["states"])
            self.t += 1

    # ---- views in the reference's layout ------------------------------------------------------------------
    def to_trajectory(self, begin: int = 0, end: Optional[int] = None) -> Trajectory:
        end = self.B if end is None else end
        sl = slice(begin, end)
        return Trajectory(
            origin=(self, begin, end),
            max_episode_length=self.max_episode_length, actions=self.actions[:, sl], rewards=self.rewards[:, sl],
            terminations=self.terminations[:, sl], truncations=self.truncations[:, sl], dones=self.dones[:, sl],
            prev_logprobs=self.prev_logprobs[:, sl], prev_values=self.prev_values[:, sl], versions=self.versions[:, sl],
            forward_inputs={"states": self.states[:, sl], "action": self.actions[:, sl], "model_action": self.actions[:, sl]})

    def to_splited_trajectories(self, split_size: int) -> list:
        """torch.chunk on the batch dim (embodied_trajectory_builder.py:232-281), as views."""
        assert self.B % split_size == 0
        per = self.B // split_size
        return [self.to_trajectory(i * per, (i + 1) * per) for i in range(split_size)]


def convert_trajectories_to_batch(trajectories: list) -> dict:
    """Trajectory list -> ``[T, B, ...]`` batch dict, concatenating on the batch dim (embodied_types.py:500-559).
    A single trajectory is passed through as views (no copy)."""
    if not trajectories:
        return {}
    if len(trajectories) > 1 and all(t.origin is not None for t in trajectories):
        buf, begin, end = trajectories[0].origin
        contiguous = True
        for t in trajectories[1:]:
            contiguous &= t.origin[0] is buf and t.origin[1] == end
            end = t.origin[2]
        if contiguous:  # e.g. the per-stage / per-actor splits of one rank's buffer: still views, nothing moves
            trajectories = [buf.to_trajectory(begin, end)]
    cat = (lambda ts: ts[0]) if len(trajectories) == 1 else (lambda ts: torch.cat(ts, dim=1))
    batch: dict = {}
    if trajectories[0].forward_inputs:
        batch["forward_inputs"] = {k: cat([t.forward_inputs[k] for t in trajectories]) for k in trajectories[0].forward_inputs}
    for name in Trajectory._TENSOR_FIELDS:
        vals = [getattr(t, name) for t in trajectories if getattr(t, name) is not None]
        if vals:
            batch[name] = cat(vals)
    return batch

"""Synthetic ManiSkill-shaped vectorised env (no simulator in this image; BASELINE.json asks for "synthetic
ManiSkill-shaped buffers").  Shapes and step semantics follow the ManiSkill wrapper
(rlinf/envs/maniskill/maniskill_env.py:327-391): ``chunk_step(actions [B,C,A])`` returns obs, rewards [B,C],
terminations / truncations [B,C] and, with auto_reset, ``final_obs`` = the true terminal observation of the envs
that just finished.  Transitions are pre-generated tensors indexed by the step counter, so a step costs nothing
and the loop measures the actor-learner path only.
"""

from __future__ import annotations

from typing import Optional

import torch


def generate_tensors(seed: int, num_steps: int, num_envs: int, obs_dim: int = 42, max_episode_steps: int = 50,
                     mode: str = "periodic", p_done: float = 0.02) -> dict:
    """obs [T+1,B,D] ~ N(0,1), final_obs [T,B,D] ~ N(0,1), rewards [T,B] ~ U(0,1) (PickCube's dense reward is bounded),
    dones [T+1,B]: row 0 False; 'periodic' = truncation every max_episode_steps steps, else Bernoulli(p_done)."""
    g = torch.Generator().manual_seed(seed)
    T, B = num_steps, num_envs
    obs = torch.randn(T + 1, B, obs_dim, generator=g)
    final_obs = torch.randn(T, B, obs_dim, generator=g)
    rewards = torch.rand(T, B, generator=g)
    if mode == "periodic":
        dones = torch.zeros(T + 1, B, dtype=torch.bool)
        for t in range(1, T + 1):
            if t % max_episode_steps == 0:
                dones[t] = True
    else:
        dones = torch.rand(T + 1, B, generator=g) < p_done
        dones[0] = False
    return dict(obs=obs, final_obs=final_obs, rewards=rewards, dones=dones)


class SyntheticManiSkillEnv:
    """``tensors`` are indexed by ENV step; a chunk step consumes ``num_action_chunks`` of them (maniskill_env.py:327-372): the
    rewards of its C sub-steps side by side, terminations / truncations raised in the LAST column for every env that finished
    anywhere inside the chunk, the observation after the chunk, and -- with auto_reset -- the true observation the chunk ended
    on as ``final_obs``.  The per-chunk views are laid out once at construction, so a chunk step is pure indexing."""

    def __init__(self, tensors: dict, device, num_action_chunks: int = 1, auto_reset: bool = True,
                 env_slice: Optional[slice] = None, done_is_truncation: bool = True):
        sl = env_slice if env_slice is not None else slice(None)
        C = self.C = int(num_action_chunks)
        obs, final_obs = tensors["obs"][:, sl], tensors["final_obs"][:, sl]
        rewards, dones = tensors["rewards"][:, sl], tensors["dones"][:, sl]
        n_env_steps = rewards.shape[0]
        assert n_env_steps % C == 0, "the pre-generated horizon must hold whole action chunks"
        Tc, B = n_env_steps // C, rewards.shape[1]
        self.obs = obs[::C].to(device).contiguous()                         # [Tc + 1, B, D]: before chunk t / after chunk t - 1
        self.final_obs = final_obs[C - 1::C].to(device).contiguous()       # [Tc, B, D]
        self.rewards = rewards.reshape(Tc, C, B).transpose(1, 2).to(device).contiguous()     # [Tc, B, C]
        flags = torch.zeros(Tc, B, C, dtype=torch.bool)
        flags[:, :, -1] = dones[1:].reshape(Tc, C, B).any(dim=1)            # past_dones -> last column (:359-371)
        self.flags = flags.to(device).contiguous()
        self.auto_reset = auto_reset
        self.done_is_truncation = done_is_truncation
        self.num_steps = Tc
        self.num_envs = B
        self.t = 0
        self._false = torch.zeros(B, C, dtype=torch.bool, device=device)

    def reset(self, start_step: int = 0):
        """``start_step`` (in CHUNK steps): where in the pre-generated horizon this episode batch begins (rollout epoch e of T
        chunk steps starts at e * T, so that every epoch sees fresh transitions)."""
        self.t = int(start_step)
        return {"states": self.obs[self.t]}, {}

    def chunk_step(self, chunk_actions: torch.Tensor):
        """-> (obs dict, rewards [B,C], terminations [B,C], truncations [B,C], infos)."""
        t = self.t
        assert t < self.num_steps, "rollout longer than the pre-generated horizon"
        done = self.flags[t]
        trunc, term = (done, self._false) if self.done_is_truncation else (self._false, done)
        infos = {"final_obs": {"states": self.final_obs[t]}} if self.auto_reset else {}
        self.t += 1
        return {"states": self.obs[t + 1]}, self.rewards[t], term, trunc, infos

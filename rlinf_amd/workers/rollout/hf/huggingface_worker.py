"""MultiStepRolloutWorker: batched obs-preprocess -> policy forward -> Gaussian sample, one fused HIP launch per
step (mirror of rlinf/workers/rollout/hf/huggingface_worker.py: predict :469-554, _build_policy_output :579-627,
get_bootstrap_values :612-627, sync_model_from_actor :629-675, generate_one_epoch :677-800)."""

from __future__ import annotations

import torch

from .... import ops
from ....models import get_model
from ...common import Worker


class MultiStepRolloutWorker(Worker):
    def __init__(self, cfg, ctx=None):
        super().__init__(cfg, ctx)
        self.hf_model = None
        self.version = 0
        self._shares_actor_weights = False

    def init_worker(self, model=None):
        """``model``: when rollout and learner are collocated (component_placement ``env,rollout,actor: 0``) the
        rollout worker can alias the learner's policy object -- weight sync becomes a no-op."""
        if model is not None:
            self.hf_model, self._shares_actor_weights = model, True
        else:
            self.hf_model = get_model(self.cfg.actor.model).to(self.device)

    def set_global_step(self, step: int):
        self.version = step

    def sync_model_from_actor(self, flat_params: torch.Tensor | None = None):
        """Apply the learner's weights (huggingface_worker.py:629-675).  Flat-buffer copy, or nothing when aliased."""
        if self._shares_actor_weights or flat_params is None:
            self.hf_model.mark_updated()
            return
        with torch.no_grad():
            self.hf_model.flat.data.copy_(flat_params)
        self.hf_model.mark_updated()

    def predict(self, env_obs: dict, out=None, eps=None, mode: str = "train"):
        """-> chunk_actions [B, C, A]; action / logprob / value rows land in ``out`` (the trajectory buffer)."""
        m = self.hf_model
        states = env_obs["states"]
        if mode == "train" and eps is None:
            eps = torch.randn((states.shape[0], m.layout.act_dim), dtype=torch.float32, device=states.device)
        action, _, _ = ops.mlp_rollout(m.flat.data, m.packed(), m.layout, states, eps if mode == "train" else None, out=out)
        return action.view(-1, m.num_action_chunks, m.action_dim)

    def get_bootstrap_values(self, final_obs: dict, out=None) -> torch.Tensor:
        m = self.hf_model
        return ops.mlp_value(m.flat.data, m.packed(), m.layout, final_obs["states"], out=out)[:, :1]

    def generate(self, *args, **kwargs):
        """The reference runs this concurrently with EnvWorker.interact over channels; in-process the env worker
        calls predict() directly, so there is nothing left to do here."""
        return None

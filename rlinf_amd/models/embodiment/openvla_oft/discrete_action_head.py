"""Everything OpenVLA-OFT does between the language model's logits and the env action / training log-probs
(rlinf/models/embodiment/openvla_oft/official/openvla_oft_action_model.py:258-287 ``_compute_logprobs_and_entropy``,
:363-414 the sampling branch of ``_discrete_prediction``), on token_ops.hip.

Rollout: ONE launch reads the ``n_action_bins`` window of each action-token row in place out of the model's
``[B, seq, V]`` logits and produces token, log-prob and normalised action (the reference: slice copy, div, top-k warper,
softmax, multinomial, a D2H copy of the ids, numpy clip/lookup, cross_entropy).  Training: the same window feeds
``compute_logprobs_from_logits`` / ``compute_entropy_from_logits`` (rlinf_amd.utils.utils), again without a copy.
"""

from __future__ import annotations

from typing import Optional

import torch

from .... import token_ops
from ....utils.utils import compute_logprobs_and_entropy_from_logits, compute_logprobs_from_logits


class DiscreteActionHead:
    def __init__(self, n_action_bins: int = 256, pad_to_multiple_of: int = 64, action_dim: int = 7,
                 num_action_chunks: int = 8, bin_centers: Optional[torch.Tensor] = None):
        self.n_action_bins, self.pad_to_multiple_of = int(n_action_bins), int(pad_to_multiple_of)
        self.action_dim, self.num_action_chunks = int(action_dim), int(num_action_chunks)
        if bin_centers is None:  # the reference's bins: n_action_bins edges on [-1, 1], centers between them
            edges = torch.linspace(-1.0, 1.0, self.n_action_bins)
            bin_centers = (edges[:-1] + edges[1:]) / 2.0
        self.bin_centers = bin_centers.float()

    def action_logits(self, response_logits: torch.Tensor) -> torch.Tensor:
        """[B, A, V] -> the [B, A, n_action_bins] window (a view), :363-368."""
        hi = response_logits.shape[-1] - self.pad_to_multiple_of
        return response_logits[..., hi - self.n_action_bins:hi]

    @torch.no_grad()
    def predict(self, response_logits: torch.Tensor, do_sample: bool = True, temperature: float = 1.0, top_k: int = -1,
                generator: Optional[torch.Generator] = None, noise: Optional[torch.Tensor] = None):
        """-> (normalized_actions [B*chunks, action_dim], action_tokens [B, A] i64, logprobs [B, A] f32).
        ``noise`` (Exp(1), the window's dtype and shape) may be injected for reproducibility; otherwise it is drawn
        with ``generator`` the way torch.multinomial draws it."""
        window = self.action_logits(response_logits)
        if do_sample:
            if not temperature > 0:
                raise AssertionError("temperature must be positive")
            if noise is None:
                noise = torch.empty(window.shape, dtype=window.dtype, device=window.device).exponential_(1, generator=generator)
        else:
            noise = None
        centers = self.bin_centers.to(window.device)
        tokens, logprobs, actions = token_ops.categorical_sample(window, noise, temperature=temperature, top_k=top_k,
                                                                 bin_centers=centers)
        return actions.reshape(-1, self.action_dim), tokens, logprobs

    def logprobs_and_entropy(self, response_logits: torch.Tensor, action_tokens: torch.Tensor, compute_entropy: bool = False,
                             temperature: float = 1.0):
        """Training-time recomputation (:258-287), differentiable w.r.t. the logits."""
        window = self.action_logits(response_logits)
        if compute_entropy:
            lp, ent = compute_logprobs_and_entropy_from_logits(window, action_tokens, temperature=temperature)
            return {"logprobs": lp, "entropy": ent}
        return {"logprobs": compute_logprobs_from_logits(window, action_tokens, temperature=temperature)}

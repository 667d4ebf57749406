"""The token tier of the reasoning learner: the part of ``FSDPActor`` (rlinf/workers/actor/fsdp_actor_worker.py)
that starts at the model's logits and ends at ``loss.backward()`` -- lines 476-505 (logits -> log-probs / entropy)
and 694-781 (micro-batch loss) -- on the kernels of token_ops.hip.

The transformer itself, FSDP wrapping, sequence packing and weight sync are outside this path (SURVEY.md 8:
model backends are out of scope); ``TokenLearnerStep`` is what a maintainer drops into ``training_step`` in
place of ``forward_batch``'s tail + the loss block:

    step = TokenLearnerStep.from_cfg(cfg)
    logits = model(**inputs).logits                          # [bsz, S, V], bf16 under amp or fp32
    loss, metrics = step(logits, m_batch, gradient_accumulation)
    loss.backward()                                          # d_logits is written by ONE kernel

Per micro-batch the reference runs: div_ (read+write logits), a reshape copy of the response slice, log_softmax
(+exp, *, where, sum for the entropy), cross_entropy, ~40 elementwise [bsz, seq] kernels and ~12 .item()-free
reductions for the loss, and autograd's mirror image of all of it.  Here: one read of the logits forward, one
read + one write backward, three small [bsz, seq] launches each way.
"""

from __future__ import annotations

from dataclasses import dataclass
from typing import Mapping, Optional

import torch

from ... import ops
from ...algorithms.registry import calculate_adv_and_returns, policy_loss
from ...utils.utils import compute_logprobs_and_entropy_from_logits, compute_logprobs_from_logits


def _get(node, key, default=None):
    if node is None:
        return default
    if isinstance(node, Mapping):
        return node.get(key, default)
    return getattr(node, key, default)


@dataclass
class TokenLearnerStep:
    response_len: int
    loss_agg: str = "token-mean"
    loss_type: str = "actor"
    task_type: str = "reasoning"
    clip_ratio_low: float = 0.2
    clip_ratio_high: float = 0.2
    clip_ratio_c: Optional[float] = 3.0
    clip_log_ratio_min: Optional[float] = None
    clip_log_ratio_max: Optional[float] = None
    temperature: float = 1.0
    calculate_entropy: bool = False
    entropy_bonus: float = 0.0
    kl_beta: float = 0.0
    kl_penalty_type: str = "low_var_kl"
    importance_sampling_fix: bool = False
    importance_sampling_clip: Optional[float] = None
    logprob_op_type: str = "torch"   # "torch" rounds like the reference's default, "flash_attn" keeps fp32
    inplace_grad: bool = False       # let backward overwrite the logits buffer with d_logits
    # advantage stage (fsdp_actor_worker.py:900-907,941-978)
    adv_type: str = "grpo"
    group_size: int = 1
    reinpp_kl_beta: float = 0.0
    use_reinpp_baseline: bool = False
    normalize_advantages: bool = False

    @classmethod
    def from_cfg(cfg_cls, cfg) -> "TokenLearnerStep":
        """Field names of the reference's yaml (fsdp_actor_worker.py:116-160,710-722)."""
        algo, actor, data = _get(cfg, "algorithm"), _get(cfg, "actor"), _get(cfg, "data")
        eps = _get(algo, "ratio_clip_eps", 0.2)
        low, high = _get(algo, "clip_ratio_low"), _get(algo, "clip_ratio_high")
        enc = _get(_get(actor, "model"), "encoder_seq_length")
        return cfg_cls(
            response_len=int(enc - _get(data, "max_prompt_length")),
            loss_agg=_get(algo, "loss_agg_func", "token-mean"), loss_type=_get(algo, "loss_type", "actor"),
            task_type=_get(_get(cfg, "runner"), "task_type", "reasoning"),
            clip_ratio_low=eps if low is None else low, clip_ratio_high=eps if high is None else high,
            clip_ratio_c=_get(algo, "clip_ratio_c", 3.0), clip_log_ratio_min=_get(algo, "clip_log_ratio_min"),
            clip_log_ratio_max=_get(algo, "clip_log_ratio_max"),
            temperature=float(_get(_get(algo, "sampling_params"), "temperature", 1.0)),
            calculate_entropy=bool(_get(algo, "calculate_entropy", False)),
            entropy_bonus=float(_get(algo, "entropy_bonus", 0.0)), kl_beta=float(_get(algo, "kl_beta", 0.0)),
            kl_penalty_type=_get(algo, "kl_penalty_type", "low_var_kl"),
            importance_sampling_fix=bool(_get(algo, "importance_sampling_fix", False)),
            importance_sampling_clip=_get(algo, "importance_sampling_clip"),
            logprob_op_type=_get(algo, "logprob_op_type", "torch"),
            adv_type=_get(algo, "adv_type", "grpo"), group_size=int(_get(algo, "group_size", 1)),
            reinpp_kl_beta=float(_get(algo, "reinpp_kl_beta", 0.0)),
            use_reinpp_baseline=bool(_get(algo, "use_reinpp_baseline", False)),
            normalize_advantages=bool(_get(algo, "normalize_advantages", False)))

    # fsdp_actor_worker.py:941-978
    def compute_advantages_and_returns(self, batch: dict) -> dict:
        """Fill ``batch["advantages"]`` ([bsz, response_len]) unless the rollout already did: GRPO, or Reinforce++ with its
        per-token KL reward from (recomputed | rollout) log-probs against the reference policy's."""
        if batch.get("advantages") is None:
            mask = batch["response_mask"][:, -self.response_len:]
            logprob = batch.get("recomputed_logprobs")
            if logprob is None:
                logprob = batch.get("rollout_logprobs")
            advantages, _ = calculate_adv_and_returns(
                task_type=self.task_type, adv_type=self.adv_type, rewards=batch["rewards"], loss_mask=mask,
                group_size=self.group_size, kl_beta=self.reinpp_kl_beta, kl_penalty_type=self.kl_penalty_type, logprob=logprob,
                ref_logprob=batch.get("ref_logprobs"), use_reinpp_baseline=self.use_reinpp_baseline)
            batch["advantages"] = advantages
        return batch

    # fsdp_actor_worker.py:900-907
    def normalize_batch_advantages(self, batch: dict, ctx=None) -> dict:
        """``algorithm.normalize_advantages``: masked_normalization over the data-parallel group -- (count, sum, sumsq) of the
        masked advantages in one f64[3] all-reduce (the reference issues three), then one elementwise pass."""
        if self.normalize_advantages:
            from ...scheduler import all_reduce_flat_
            adv = batch["advantages"]
            mask = batch["response_mask"][:, -self.response_len:].contiguous()
            stats = ops.masked_stats(adv, mask)
            if ctx is not None:
                all_reduce_flat_(stats, ctx)
            batch["advantages"] = ops.masked_normalize(adv, mask, stats)
        return batch

    # fsdp_actor_worker.py:476-505 (the fixed-length branch)
    def logprobs_and_entropy(self, logits: torch.Tensor, input_ids: torch.Tensor):
        resp = self.response_len
        window = logits[:, -resp - 1:-1, :]  # a strided view; the kernels address it in place
        responses = input_ids[:, -resp:]
        if self.calculate_entropy:
            return compute_logprobs_and_entropy_from_logits(window, responses, temperature=self.temperature,
                                                            op_type=self.logprob_op_type,
                                                            inplace_grad=self.inplace_grad)
        return compute_logprobs_from_logits(window, responses, self.logprob_op_type, temperature=self.temperature,
                                            inplace_grad=self.inplace_grad), None

    # fsdp_actor_worker.py:694-781
    def __call__(self, logits: torch.Tensor, m_batch: Mapping, gradient_accumulation: int = 1):
        logprobs, entropy = self.logprobs_and_entropy(logits, m_batch["input_ids"])
        old_logprobs = m_batch.get("recomputed_logprobs")
        if old_logprobs is None:
            old_logprobs = m_batch["rollout_logprobs"]
        advantages = m_batch["advantages"]
        ref_logprobs = m_batch.get("ref_logprobs")
        loss_mask = m_batch["response_mask"][:, -self.response_len:]
        if self.importance_sampling_fix:
            if "rollout_logprobs" not in m_batch or "recomputed_logprobs" not in m_batch:
                raise ValueError("importance_sampling_fix requires both rollout_logprobs and recomputed_logprobs")
            advantages = advantages * torch.clamp(
                (m_batch["recomputed_logprobs"] - m_batch["rollout_logprobs"]).exp(), max=self.importance_sampling_clip)
        use_kl = self.kl_beta > 0 and ref_logprobs is not None
        bonus = self.entropy_bonus if (self.entropy_bonus > 0 and self.calculate_entropy) else 0.0
        loss, metrics = policy_loss(
            task_type=self.task_type, loss_type=self.loss_type, loss_agg=self.loss_agg, logprobs=logprobs,
            old_logprobs=old_logprobs, advantages=advantages, clip_ratio_c=self.clip_ratio_c,
            clip_ratio_low=self.clip_ratio_low, clip_ratio_high=self.clip_ratio_high, loss_mask=loss_mask,
            clip_log_ratio_min=self.clip_log_ratio_min, clip_log_ratio_max=self.clip_log_ratio_max,
            fast_path_zero_loss_mask=True,
            # the two terms the reference adds right after policy_loss(), fused into the same kernel:
            entropy=entropy, entropy_bonus=bonus, ref_logprobs=ref_logprobs if use_kl else None,
            kl_beta=self.kl_beta if use_kl else 0.0, kl_penalty_type=self.kl_penalty_type)
        if not self.calculate_entropy:
            metrics.update({"actor/final_loss": loss.detach(), "actor/entropy_loss": 0.0})
        if not use_kl:
            metrics.update({"actor/kl_loss": 0.0})
        return loss / gradient_accumulation, metrics

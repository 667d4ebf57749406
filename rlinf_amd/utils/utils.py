"""Mirror of rlinf/utils/utils.py:307-381,454-512 -- the loss aggregations and the logits -> log-prob /
entropy entry points of the reasoning learner.

``compute_logprobs_from_logits`` / ``compute_entropy_from_logits`` run the HIP kernels of token_ops.hip
(autograd included) and raise without a HIP device.  The aggregation helpers are the small tensor
expressions the reference passes around as ``loss_agg_func``; they carry an ``rlx_agg`` tag so that
``policy_loss(task_type="reasoning", loss_agg_func=...)`` maps them onto the fused token-loss kernel instead
of calling them.  They remain callable (plain tensor ops on the caller's device) for user-registered losses
that receive them as an argument, exactly like the reference's.
"""

from __future__ import annotations

from typing import Callable, Literal, Optional

import torch

from .. import token_ops


def _reduce_mean(values, axis):
    return values.mean() if axis is None else values.mean(dim=axis)


def _reduce_sum(values, axis):
    return values.sum() if axis is None else values.sum(dim=axis)


def masked_mean(values: torch.Tensor, mask: Optional[torch.Tensor], axis=None):
    """utils.py:323-330: mean over the mask; the plain sum (zero) when the mask is all False."""
    if mask is None:
        return _reduce_mean(values, axis)
    if (~mask).all():
        return _reduce_sum(values * mask, axis)
    return _reduce_sum(values * mask, axis) / _reduce_sum(mask, axis)


def masked_sum(values: torch.Tensor, mask: torch.Tensor, axis=None):
    return _reduce_sum(values * mask, axis)


def seq_mean_token_sum(values: torch.Tensor, mask: torch.Tensor, dim: int = -1):
    return torch.mean(torch.sum(values * mask, dim=-1))


def seq_mean_token_mean(values: torch.Tensor, mask: torch.Tensor, dim: int = -1):
    return torch.mean(torch.sum(values * mask, dim=-1) / torch.sum(mask, dim=-1))


def masked_mean_ratio(values: torch.Tensor, mask: torch.Tensor, loss_mask_ratio: torch.Tensor):
    return (values / loss_mask_ratio * mask).mean()


masked_mean.rlx_agg = "token-mean"
seq_mean_token_sum.rlx_agg = "seq-mean-token-sum"
seq_mean_token_mean.rlx_agg = "seq-mean-token-mean"


def get_loss_agg_func(loss_agg: str) -> Callable:
    """utils.py:359-381."""
    if loss_agg == "seq-mean-token-sum":
        return seq_mean_token_sum
    if loss_agg == "seq-mean-token-mean":
        return seq_mean_token_mean
    if loss_agg == "token-mean":
        return masked_mean
    raise ValueError(f"Unsupported loss aggregation method: {loss_agg}")


def compute_logprobs_from_logits(logits: torch.Tensor, target: torch.Tensor,
                                 op_type: Literal["torch", "flash_attn", "liger_kernel"] = "torch", *,
                                 temperature: float = 1.0, inplace_grad: bool = False) -> torch.Tensor:
    """logits [B, seq, V], target [B, seq] -> fp32 logprobs [B, seq] (utils.py:454-492).

    ``op_type="torch"`` keeps the reference's rounding (the result is rounded to the logits' dtype before the
    final ``.float()``); ``"flash_attn"`` is the reference's fp32 variant, i.e. no rounding.  Both run the same
    one-pass HIP kernel; ``"liger_kernel"`` is accepted as an alias of ``"torch"``.  ``temperature`` fuses the
    learner's ``logits.div_(temperature)`` (fsdp_actor_worker.py:478)."""
    if op_type not in ("torch", "flash_attn", "liger_kernel"):
        raise AssertionError(f"Unsupported op_type: {op_type} for logprobs computation. Supported types are "
                             "'torch', 'flash_attn', 'liger_kernel'.")
    logprobs, _ = token_ops.token_logprobs(logits, target, temperature=temperature, with_entropy=False,
                                           round_outputs=(op_type != "flash_attn"), inplace_grad=inplace_grad)
    return logprobs


def compute_entropy_from_logits(logits: torch.Tensor, dim: int = -1, *, temperature: float = 1.0) -> torch.Tensor:
    """H = -sum p log p over the vocabulary (utils.py:495-512), computed in fp32."""
    if dim not in (-1, logits.dim() - 1):
        raise NotImplementedError("compute_entropy_from_logits: only the vocabulary (last) dim is supported")
    labels = torch.zeros(logits.shape[:-1], dtype=torch.int64, device=logits.device)
    _, entropy = token_ops.token_logprobs(logits, labels, temperature=temperature, with_entropy=True)
    return entropy


def compute_logprobs_and_entropy_from_logits(logits: torch.Tensor, target: torch.Tensor, *, temperature: float = 1.0,
                                             op_type: str = "torch", inplace_grad: bool = False):
    """Both outputs of FSDPActor.forward_batch (fsdp_actor_worker.py:476-505) from ONE pass over the logits."""
    return token_ops.token_logprobs(logits, target, temperature=temperature, with_entropy=True,
                                    round_outputs=(op_type != "flash_attn"), inplace_grad=inplace_grad)

"""Built-in advantage estimators behind the registry names "gae" and "grpo"
(mirror of rlinf/algorithms/advantages.py:24-121), served by gae_scan.hip / grpo_adv.hip.

Two entry shapes exist for each name:
  * the reference's callee contract -- flattened ``[T, B]`` tensors, ``fn(**kwargs) -> (advantages,
    returns_or_None)`` -- so that a maintainer can re-register these functions inside the real RLinf
    (rlinf_amd/ext.py) and everything upstream keeps working;
  * a native ``[n_chunk, B, C]`` path used by ``calculate_adv_and_returns`` here, which skips the
    transpose/reshape copies entirely.
"""

from __future__ import annotations

from typing import Optional

import torch

from .. import ops, token_ops
from . import utils as _u
from .registry import _mark_native_adv, register_advantage


def _tb(t: Optional[torch.Tensor], dev) -> Optional[torch.Tensor]:
    return None if t is None else _u.stage(t, dev).contiguous().unsqueeze(-1)


@register_advantage("gae")
def compute_gae_advantages_and_returns(rewards: torch.Tensor, gamma: float = 1.0, gae_lambda: float = 1.0,
                                       values: Optional[torch.Tensor] = None, normalize_advantages: bool = True,
                                       normalize_returns: bool = False, loss_mask: Optional[torch.Tensor] = None,
                                       dones: Optional[torch.Tensor] = None, **kwargs):
    """rewards [T,B], values [T+1,B] (None = critic-free), dones [T+1,B] bool -> (advantages, returns) [T,B]."""
    dev = _u.compute_device(rewards, values, dones, loss_mask)
    adv, ret = ops.gae_scan(_tb(rewards.float(), dev), _tb(values, dev), _tb(dones, dev), _tb(loss_mask, dev),
                            gamma, gae_lambda, normalize_advantages, normalize_returns,
                            variant=int(kwargs.get("rlx_variant", 0)))
    return _u.unstage(adv[..., 0], rewards), _u.unstage(ret[..., 0], rewards)


@register_advantage("grpo")
def compute_grpo_advantages(rewards: torch.Tensor, loss_mask: torch.Tensor, group_size: int, **kwargs):
    """rewards = per-env scores (any shape with B elements, the reference passes [num_groups, group_size]);
    loss_mask [T,B] -> (advantages [T,B], None).  advantages.py:89-121."""
    dev = _u.compute_device(rewards, loss_mask)
    scores = _u.stage(rewards, dev).float().reshape(-1)
    mask = _u.stage(loss_mask, dev).contiguous().unsqueeze(-1)
    adv = ops.grpo_from_scores(scores, mask, int(group_size))
    return _u.unstage(adv[..., 0], loss_mask), None


# ---- native [n_chunk, B, C] paths -------------------------------------------------------------------------
def _chunk_level(rewards, dones, loss_mask, loss_mask_sum):
    rewards = rewards.sum(dim=-1, keepdim=True)
    dones = dones.max(dim=-1, keepdim=True)[0]
    if loss_mask is not None:
        loss_mask = loss_mask.max(dim=-1, keepdim=True)[0]
    if loss_mask_sum is not None:
        loss_mask_sum = loss_mask_sum.max(dim=-1, keepdim=True)[0]
    return rewards, dones, loss_mask, loss_mask_sum


def _native_gae(*, rewards, dones, values=None, loss_mask=None, loss_mask_sum=None, gamma=1.0, gae_lambda=1.0,
                reward_type="action_level", normalize_advantages=True, normalize_returns=False, **kwargs):
    dev = _u.compute_device(rewards, dones, values, loss_mask)
    r, d, m = _u.stage(rewards, dev), _u.stage(dones, dev), _u.stage(loss_mask, dev)
    if reward_type == "chunk_level":  # reductions over the chunk dim happen on device views (utils.py:80-89)
        r, d, m, _ = _chunk_level(r, d, m, None)
    v = _u.stage(values, dev)
    adv, ret = ops.gae_scan(r.float(), v, d, m, gamma, gae_lambda, normalize_advantages, normalize_returns,
                            variant=int(kwargs.get("rlx_variant", 0)))
    return {"advantages": _u.unstage(adv, rewards), "returns": _u.unstage(ret, rewards)}


def _native_grpo(*, rewards, dones, loss_mask=None, group_size=8, reward_type="action_level", **kwargs):
    if loss_mask is None:
        raise TypeError("compute_grpo_advantages() missing required argument: 'loss_mask'")
    dev = _u.compute_device(rewards, dones, loss_mask)
    r, d, m = _u.stage(rewards, dev), _u.stage(dones, dev), _u.stage(loss_mask, dev)
    if reward_type == "chunk_level":
        r, d, m, _ = _chunk_level(r, d, m, None)
    adv, _ = ops.grpo_group_adv(r.float(), d, m, int(group_size))
    return {"advantages": _u.unstage(adv, rewards)}


def _native_grpo_reasoning(*, rewards, loss_mask, group_size, **kwargs):
    """Reasoning GRPO: rewards [bsz], loss_mask [bsz, seq] -> (advantages [bsz, seq], None); replaces
    preprocess_reasoning_advantages_inputs + compute_grpo_advantages + postprocess (utils.py:177-277)."""
    if rewards.ndim != 1:
        raise AssertionError(f"Unsupported reward shape {rewards.shape}")
    dev = _u.compute_device(rewards, loss_mask)
    adv = token_ops.grpo_seq_adv(_u.stage(rewards, dev).float(), _u.stage(loss_mask, dev), int(group_size))
    return _u.unstage(adv, loss_mask), None


def _native_gae_reasoning(*, rewards, loss_mask, values=None, gamma=1.0, gae_lambda=1.0, normalize_advantages=True,
                          normalize_returns=False, **kwargs):
    """Reasoning GAE on [bsz, seq] tensors: (advantages, returns) -- the scan runs along the contiguous axis, nothing is
    transposed or padded (utils.py:177-277 + advantages.py:24-86)."""
    if rewards.ndim != 1:
        raise AssertionError(f"Unsupported reward shape {rewards.shape}")
    if values is None:  # critic-free GAE forces gamma = lambda = 1 and delta = r: the generic route handles it
        return None
    if values.ndim != 2:
        raise AssertionError(f"Unsupported values shape {values.shape}")
    dev = _u.compute_device(rewards, loss_mask, values)
    adv, ret = token_ops.gae_seq(_u.stage(values, dev).float(), _u.stage(rewards, dev).float(), gamma, gae_lambda)
    mask = _u.stage(loss_mask, dev)
    if normalize_advantages:
        ops.masked_standardize_(adv, mask)
    if normalize_returns:
        ops.masked_standardize_(ret, mask)
    return _u.unstage(adv, values), _u.unstage(ret, values)


def _native_reinpp_reasoning(*, rewards, loss_mask, group_size=None, use_reinpp_baseline=False, kl_beta=0.0, logprob=None,
                             ref_logprob=None, kl_penalty_type="", **kwargs):
    """Reinforce++ on [bsz, seq] tensors -> (advantages, None): replaces preprocess_reasoning_advantages_inputs +
    compute_reinpp_advantages (advantages.py:300-364) + postprocess, reward placement as the reference computes it."""
    if rewards.ndim != 1:
        raise AssertionError(f"Unsupported reward shape {rewards.shape}")
    if use_reinpp_baseline:
        # advantages.py:329-332,343: the group-centred rewards come back 1-D and scatter_ then rejects them against the
        # 2-D index -- the reference cannot run this option; fail the same way rather than guess what was meant
        raise IndexError("Dimension out of range (expected to be in range of [-1, 0], but got 1) -- use_reinpp_baseline "
                         "fails at the reward scatter in the reference (rlinf/algorithms/advantages.py:343)")
    dev = _u.compute_device(rewards, loss_mask, logprob, ref_logprob)
    kl = float(kl_beta) > 0
    adv = token_ops.reinpp_seq_adv(_u.stage(rewards, dev).float(), _u.stage(loss_mask, dev),
                                   _u.stage(logprob, dev).float() if kl else None,
                                   _u.stage(ref_logprob, dev).float() if kl else None, float(kl_beta), kl_penalty_type if kl else None)
    return _u.unstage(adv, loss_mask), None


@register_advantage("reinpp")
def compute_reinpp_advantages(rewards: torch.Tensor, loss_mask: torch.Tensor, group_size: int, use_reinpp_baseline: bool = False,
                              kl_beta: float = 0.0, logprob=None, ref_logprob=None, kl_penalty_type: str = "", **kwargs):
    """The reference's own signature (advantages.py:300-364): rewards [1, B], loss_mask / logprob / ref_logprob [L, B] ->
    (advantages [L, B], None).  The kernel works on sequence-major rows, so this entry transposes on the way in and out;
    ``calculate_adv_and_returns`` skips it and hands the [bsz, seq] tensors over untouched."""
    t = lambda x: None if x is None else x.transpose(0, 1)  # noqa: E731
    adv, _ = _native_reinpp_reasoning(rewards=rewards.reshape(-1), loss_mask=t(loss_mask), group_size=group_size,
                                      use_reinpp_baseline=use_reinpp_baseline, kl_beta=kl_beta, logprob=t(logprob),
                                      ref_logprob=t(ref_logprob), kl_penalty_type=kl_penalty_type)
    return adv.transpose(0, 1), None


_mark_native_adv("gae", _native_gae)
_mark_native_adv("grpo", _native_grpo)
_mark_native_adv("gae", _native_gae_reasoning, task_type="reasoning")
_mark_native_adv("grpo", _native_grpo_reasoning, task_type="reasoning")
_mark_native_adv("reinpp", _native_reinpp_reasoning, task_type="reasoning")

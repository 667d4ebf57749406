"""Host-side helpers of the algorithm layer (mirror of rlinf/algorithms/utils.py, hot-path subset).

Shape bookkeeping only (views, no arithmetic) plus the device-staging rule of the drop-in boundary:
the reference keeps rollout buffers on the CPU, so registered callees may be handed CPU tensors.  They
are copied to the current HIP device, computed there, and results return on the caller's device.  With
no HIP device this raises -- there is no CPU implementation in this package.
"""

from __future__ import annotations

from typing import Optional

import torch

from .._lib import RlxError


def compute_device(*tensors) -> torch.device:
    for t in tensors:
        if isinstance(t, torch.Tensor) and t.is_cuda:
            return t.device
    if not torch.cuda.is_available():
        raise RlxError("rlinf_amd needs a HIP device: inputs are CPU tensors and no GPU is visible "
                       "(there is no CPU fallback by design)")
    return torch.device("cuda", torch.cuda.current_device())


def stage(t: Optional[torch.Tensor], device: torch.device) -> Optional[torch.Tensor]:
    if t is None:
        return None
    return t if t.device == device else t.to(device, non_blocking=True)


def unstage(t: Optional[torch.Tensor], like: torch.Tensor) -> Optional[torch.Tensor]:
    if t is None:
        return None
    return t if t.device == like.device else t.to(like.device)


# ---- [n_chunk, B, C] <-> [T, B] views for user-registered callees (utils.py:67-131,155-174) ------------
def preprocess_embodied_advantages_inputs(rewards, dones, values=None, loss_mask=None, loss_mask_sum=None, **kwargs):
    if kwargs["reward_type"] == "chunk_level":
        rewards = rewards.sum(dim=-1, keepdim=True)
        dones = dones.max(dim=-1, keepdim=True)[0]
        if loss_mask is not None:
            loss_mask = loss_mask.max(dim=-1, keepdim=True)[0]
        if loss_mask_sum is not None:
            loss_mask_sum = loss_mask_sum.max(dim=-1, keepdim=True)[0]
    num_chunk, bsz, chunk_size = rewards.shape
    n_steps = num_chunk * chunk_size
    kwargs.update(num_chunk=num_chunk, batch_size=bsz, chunk_size=chunk_size, n_steps=n_steps)
    rewards = rewards.transpose(1, 2).reshape(n_steps, bsz)
    if loss_mask is not None:
        loss_mask = loss_mask.transpose(1, 2).reshape(n_steps, bsz)
    dones = dones.transpose(1, 2).reshape((num_chunk + 1) * chunk_size, bsz)[-(n_steps + 1):]
    if kwargs["adv_type"] == "gae":
        values = values.transpose(1, 2).reshape((num_chunk + 1) * chunk_size, bsz)[: n_steps + 1]
    kwargs.update(rewards=rewards, dones=dones, values=values, loss_mask=loss_mask, loss_mask_sum=loss_mask_sum)
    return kwargs


def calculate_scores(rewards, dones, **kwargs):
    """Per-env first-episode return on the flattened [T,B] inputs (utils.py:134-152), via the HIP scan."""
    from .. import ops

    dev = compute_device(rewards, dones)
    r = stage(rewards, dev).float().contiguous().unsqueeze(-1)
    d = stage(dones, dev).contiguous().unsqueeze(-1)
    scores = ops.episode_scores(r, d).reshape(-1, kwargs["group_size"])
    kwargs.update(rewards=unstage(scores, rewards), dones=dones)
    return kwargs


def postprocess_embodied_advantages_outputs(advantages, num_chunk, chunk_size, returns=None, **kwargs):
    res = {"advantages": advantages.reshape(num_chunk, chunk_size, -1).transpose(1, 2)}
    if returns is not None:
        res["returns"] = returns.reshape(num_chunk, chunk_size, -1).transpose(1, 2)
    return res


def preprocess_reasoning_advantages_inputs(rewards, loss_mask, values=None, logprob=None, ref_logprob=None, **kwargs):
    """Reasoning -> time-major shaping for registered advantage callees (utils.py:177-262).  Views only, except
    the reward row the GAE branch has to materialise; the built-in "grpo" never comes here (its kernel reads the
    [bsz, seq] layout directly)."""
    bsz, seq_len = loss_mask.shape
    if rewards.ndim != 1:
        raise AssertionError(f"Unsupported reward shape {rewards.shape}")
    adv_type = kwargs["adv_type"]
    if adv_type == "gae":
        last_token = rewards.new_zeros((seq_len, bsz))
        last_token[-1] = rewards  # the scalar reward sits on the final token
        shaped = last_token
    elif adv_type == "grpo":
        shaped = rewards.reshape(-1, kwargs["group_size"]).contiguous()
    elif adv_type == "grpo_dynamic":
        shaped = rewards.reshape(-1, kwargs["num_sequence"]).transpose(0, 1).contiguous()
    elif adv_type == "reinpp":
        shaped = rewards.unsqueeze(0)
    elif adv_type == "raw":
        shaped = rewards
    else:
        raise AssertionError(f"Unsupported adv_type {adv_type}")
    kwargs["rewards"] = shaped
    if values is not None:
        if values.ndim != 2:
            raise AssertionError(f"Unsupported values shape {values.shape}")
        vt = values.transpose(0, 1)
        kwargs["values"] = torch.cat([vt, vt.new_zeros((1, bsz))], dim=0)  # zero bootstrap row
    if logprob is not None:
        kwargs["logprob"] = logprob.transpose(0, 1)
    if ref_logprob is not None:
        kwargs["ref_logprob"] = ref_logprob.transpose(0, 1)
    dones = torch.zeros(seq_len + 1, bsz, dtype=torch.bool, device=rewards.device)
    dones[-1] = True  # one episode per sequence, ending after its last token
    kwargs["dones"] = dones
    kwargs["loss_mask"] = loss_mask.transpose(0, 1)
    return kwargs


def postprocess_reasoning_advantages_outputs(advantages, returns=None):
    """Back to [bsz, seq], contiguous (utils.py:265-277)."""
    advantages = advantages.transpose(0, 1).contiguous()
    if returns is not None:
        returns = returns.transpose(0, 1).contiguous()
    return advantages, returns


def kl_penalty(logprob: torch.Tensor, ref_logprob: torch.Tensor, kl_penalty: str) -> torch.Tensor:
    """Per-token KL estimators (utils.py:26-64) for callers outside the fused token loss, which has them built in."""
    diff = logprob - ref_logprob
    if kl_penalty in ("kl", "k1"):
        return diff
    if kl_penalty == "abs":
        return diff.abs()
    if kl_penalty in ("mse", "k2"):
        return 0.5 * diff.square()
    if kl_penalty in ("low_var_kl", "k3"):
        neg = torch.clamp(ref_logprob - logprob, min=-20, max=20)
        return torch.clamp((torch.exp(neg) - neg - 1).contiguous(), min=-10, max=10)
    raise NotImplementedError


# ---- loss-input shaping for user-registered loss callees (utils.py:280-376) ------------------------------
def expand_to_target_dim(tensor, target_shape):
    if tensor is None:
        return None
    if tensor.shape != target_shape:
        while tensor.dim() < len(target_shape):
            tensor = tensor.unsqueeze(-1)
    return tensor


def preprocess_loss_inputs(logprobs, old_logprobs, advantages, logprob_type=None, single_action_dim=None,
                           loss_mask=None, loss_mask_sum=None, values=None, prev_values=None, returns=None,
                           reward_type=None, versions=None, **kwargs):
    if reward_type == "chunk_level":
        advantages = advantages.flatten()
        loss_mask = None if loss_mask is None else loss_mask.flatten()
        loss_mask_sum = None if loss_mask_sum is None else loss_mask_sum.flatten()
        values = None if values is None else values.flatten()
        prev_values = None if prev_values is None else prev_values.flatten()
        returns = None if returns is None else returns.flatten()
    bsz = logprobs.shape[0]
    if logprob_type == "token_level":
        logprobs = logprobs.reshape(bsz, -1, single_action_dim)
        old_logprobs = old_logprobs.reshape(bsz, -1, single_action_dim)
        advantages = advantages.unsqueeze(-1)
        loss_mask = None if loss_mask is None else loss_mask.unsqueeze(-1)
        loss_mask_sum = None if loss_mask_sum is None else loss_mask_sum.unsqueeze(-1)
    elif logprob_type == "action_level":
        logprobs = logprobs.reshape(bsz, -1, single_action_dim).sum(dim=-1)
        old_logprobs = old_logprobs.reshape(bsz, -1, single_action_dim).sum(dim=-1)
    elif logprob_type == "chunk_level":
        logprobs = logprobs.reshape(bsz, -1, single_action_dim).sum(dim=[1, 2])
        old_logprobs = old_logprobs.reshape(bsz, -1, single_action_dim).sum(dim=[1, 2])
    shape = logprobs.shape
    kwargs.update(logprobs=logprobs, old_logprobs=old_logprobs, versions=expand_to_target_dim(versions, shape),
                  advantages=expand_to_target_dim(advantages, shape), loss_mask=expand_to_target_dim(loss_mask, shape),
                  loss_mask_sum=expand_to_target_dim(loss_mask_sum, shape), values=expand_to_target_dim(values, shape),
                  prev_values=expand_to_target_dim(prev_values, shape), returns=expand_to_target_dim(returns, shape))
    return kwargs


def postprocess_loss_metric(metrics_data: dict) -> dict:
    for k, v in metrics_data.items():
        if isinstance(v, torch.Tensor):
            metrics_data[k] = v.detach().item()
    return metrics_data


def safe_normalize(array: torch.Tensor, loss_mask: Optional[torch.Tensor]):
    """utils.py:397-404 on the HIP path (returns a new tensor like the reference)."""
    from .. import ops

    dev = compute_device(array, loss_mask)
    x = stage(array, dev).float().contiguous().clone()
    ops.masked_standardize_(x, stage(loss_mask, dev))
    return unstage(x, array)

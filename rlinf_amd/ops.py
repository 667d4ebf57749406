"""Tensor-level launchers over the C ABI (include/rlx.h).  torch is plumbing only: device memory,
the current HIP stream and dtype/shape checks.  Every function requires HIP tensors and raises
otherwise -- there is no eager/CPU path.
"""

from __future__ import annotations

from ctypes import byref
from typing import Optional

import ctypes
import os

import torch

from . import _lib
from ._lib import (AdamwGroup, AdamwParams, GaeParams, GatherField, MlpLayout, PpoLossParams, PpoStepArgs, RlxError,
                   RolloutStep)


def _stream_ptr(device: torch.device) -> int:
    return torch.cuda.current_stream(device).cuda_stream


def _dev(*tensors) -> torch.device:
    dev = None
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise RlxError("rlinf_amd ops need HIP (cuda) tensors; got a CPU tensor and there is no CPU fallback")
        if dev is None:
            dev = t.device
        elif t.device != dev:
            raise RlxError(f"tensors on different devices: {dev} vs {t.device}")
    if dev is None:
        raise RlxError("no tensor argument")
    return dev


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _as_u8(t: Optional[torch.Tensor]) -> Optional[torch.Tensor]:
    if t is None:
        return None
    if t.dtype == torch.bool:
        return t.contiguous().view(torch.uint8)
    if t.dtype == torch.uint8:
        return t.contiguous()
    raise RlxError(f"expected a bool/uint8 tensor, got {t.dtype}")


def _as_f32(t: Optional[torch.Tensor], name: str) -> Optional[torch.Tensor]:
    if t is None:
        return None
    if t.dtype != torch.float32:
        raise RlxError(f"{name} must be float32 (got {t.dtype})")
    return t.contiguous()


def _nbc(t: torch.Tensor, name: str):
    if t.dim() != 3:
        raise RlxError(f"{name} must be [n_chunk, batch, chunk]; got {tuple(t.shape)}")
    return t.shape


def done_prefix_mask(dones: torch.Tensor):
    """dones bool [n+1, B, C] -> (loss_mask bool [n, B, C], mask_sum int64 [B]).  a9."""
    lib = _lib.load()
    dev = _dev(dones)
    n1, B, C = _nbc(dones, "dones")
    if n1 < 1:
        raise RlxError("dones needs at least one row")
    d8 = _as_u8(dones)
    mask = torch.empty((n1 - 1, B, C), dtype=torch.bool, device=dev)
    cnt = torch.empty((B,), dtype=torch.int64, device=dev)
    with torch.cuda.device(dev):
        _lib.check(lib.rlx_done_prefix_mask(d8.data_ptr(), mask.data_ptr(), cnt.data_ptr(), n1 - 1, B, C,
                                            _stream_ptr(dev)), "rlx_done_prefix_mask")
    return mask, cnt


def gae_scan(rewards: torch.Tensor, values: Optional[torch.Tensor], dones: torch.Tensor,
             loss_mask: Optional[torch.Tensor] = None, gamma: float = 1.0, gae_lambda: float = 1.0,
             normalize_advantages: bool = True, normalize_returns: bool = False, norm_eps: float = 1e-5,
             variant: int = 0, out: Optional[tuple] = None):
    """rewards [n,B,C] f32, values [n+1,B,C] f32|None, dones [n+1,B,C] bool -> (adv, ret) [n,B,C].  a10-a12."""
    lib = _lib.load()
    dev = _dev(rewards, values, dones, loss_mask)
    n, B, C = _nbc(rewards, "rewards")
    r = _as_f32(rewards, "rewards")
    v = _as_f32(values, "values")
    d8 = _as_u8(dones)
    m8 = _as_u8(loss_mask)
    if tuple(d8.shape) != (n + 1, B, C):
        raise RlxError(f"dones must be {(n + 1, B, C)}, got {tuple(d8.shape)}")
    if v is not None and tuple(v.shape) != (n + 1, B, C):
        raise RlxError(f"values must be {(n + 1, B, C)}, got {tuple(v.shape)}")
    if m8 is not None and tuple(m8.shape) != (n, B, C):
        raise RlxError(f"loss_mask must be {(n, B, C)}, got {tuple(m8.shape)}")
    if v is None:  # critic-free: the reference forces gamma = lambda = 1 (advantages.py:61-64)
        gamma, gae_lambda = 1.0, 1.0
    if out is not None:
        adv, ret = out
        if adv.shape != r.shape or ret.shape != r.shape or not (adv.is_contiguous() and ret.is_contiguous()):
            raise RlxError("gae_scan: out tensors must be contiguous and shaped like rewards")
    else:
        adv = torch.empty_like(r)
        ret = torch.empty_like(r)
    ws_bytes = max(8, lib.rlx_gae_workspace_bytes(n, B, C))
    ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=dev)
    p = GaeParams(float(gamma), float(float(gamma) * float(gae_lambda)), int(bool(normalize_advantages)),
                  int(bool(normalize_returns)), float(norm_eps), int(variant))
    with torch.cuda.device(dev):
        _lib.check(lib.rlx_gae_scan(r.data_ptr(), _ptr(v), d8.data_ptr(), _ptr(m8), adv.data_ptr(), ret.data_ptr(),
                                    ws.data_ptr(), ws_bytes, n, B, C, byref(p), _stream_ptr(dev)), "rlx_gae_scan")
    return adv, ret


def masked_standardize_(x: torch.Tensor, mask: Optional[torch.Tensor] = None, eps: float = 1e-5) -> torch.Tensor:
    """In place x <- (x - mean(x[mask])) / (std(x[mask]) + eps).  a12."""
    lib = _lib.load()
    dev = _dev(x, mask)
    if x.dtype != torch.float32 or not x.is_contiguous():
        raise RlxError("masked_standardize_ needs a contiguous float32 tensor")
    m8 = _as_u8(mask)
    if m8 is not None and m8.numel() != x.numel():
        raise RlxError("mask must have as many elements as x")
    ws_bytes = lib.rlx_standardize_workspace_bytes(x.numel())
    ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        _lib.check(lib.rlx_masked_standardize(x.data_ptr(), _ptr(m8), x.numel(), float(eps), ws.data_ptr(), ws_bytes,
                                              _stream_ptr(dev)), "rlx_masked_standardize")
    return x


def masked_stats(x: torch.Tensor, mask: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None,
                 accumulate: bool = False) -> torch.Tensor:
    """(count, sum, sum of squares) of x[mask] as a device float64[3] (rlinf/utils/distributed.py:942-954);
    ``accumulate`` adds into ``out`` (several batches of one rank before the cross-rank sum)."""
    lib = _lib.load()
    dev = _dev(x, mask)
    xf = _as_f32(x, "x")
    m8 = _as_u8(mask)
    if m8 is not None and m8.numel() != xf.numel():
        raise AssertionError((tuple(mask.shape), tuple(x.shape)))
    if out is None:
        out = torch.zeros(3, dtype=torch.float64, device=dev)
    ws_bytes = lib.rlx_masked_stats_workspace_bytes(xf.numel())
    ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        _lib.check(lib.rlx_masked_stats(xf.data_ptr(), _ptr(m8), xf.numel(), out.data_ptr(), int(bool(accumulate)),
                                        ws.data_ptr(), ws_bytes, _stream_ptr(dev)), "rlx_masked_stats")
    return out


def normalize_from_stats(x: torch.Tensor, stats: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """(x - mean) * rsqrt(var + 1e-5) from (count, sum, sumsq), f64 arithmetic -> f32 (rlinf/utils/distributed.py:957-965)."""
    dev = _dev(x, stats)
    xf = _as_f32(x, "x")
    if stats.dtype != torch.float64 or stats.numel() != 3:
        raise RlxError("stats must be float64[3]")
    if out is None:
        out = torch.empty_like(xf)
    with torch.cuda.device(dev):
        _lib.check(_lib.load().rlx_normalize_from_stats(xf.data_ptr(), stats.contiguous().data_ptr(), out.data_ptr(),
                                                        xf.numel(), _stream_ptr(dev)), "rlx_normalize_from_stats")
    return out.view(x.shape)


def masked_normalize(x: torch.Tensor, mask: Optional[torch.Tensor], stats: torch.Tensor, eps: float = 1e-5,
                     out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """masked_normalization's elementwise part (rlinf/utils/distributed.py:917-937) from (count, sum, sumsq):
    ((mask ? x : 0) - mean) / (sqrt(var_biased) + eps), f64 arithmetic -> f32."""
    dev = _dev(x, mask, stats)
    xf = _as_f32(x, "x")
    m8 = _as_u8(mask)
    if m8 is not None and m8.numel() != xf.numel():
        raise AssertionError((tuple(mask.shape), tuple(x.shape)))
    if stats.dtype != torch.float64 or stats.numel() != 3:
        raise RlxError("stats must be float64[3]")
    if out is None:
        out = torch.empty_like(xf)
    with torch.cuda.device(dev):
        _lib.check(_lib.load().rlx_masked_normalize(xf.data_ptr(), _ptr(m8), stats.contiguous().data_ptr(), float(eps),
                                                    out.data_ptr(), xf.numel(), _stream_ptr(dev)), "rlx_masked_normalize")
    return out.view(x.shape)


def grpo_group_adv(rewards: torch.Tensor, dones: torch.Tensor, loss_mask: torch.Tensor, group_size: int,
                   eps: float = 1e-6):
    """rewards [n,B,C], dones [n+1,B,C], loss_mask [n,B,C] -> (advantages [n,B,C], scores [B]).  a13."""
    lib = _lib.load()
    dev = _dev(rewards, dones, loss_mask)
    n, B, C = _nbc(rewards, "rewards")
    r = _as_f32(rewards, "rewards")
    d8 = _as_u8(dones)
    m8 = _as_u8(loss_mask)
    if tuple(d8.shape) != (n + 1, B, C) or tuple(m8.shape) != (n, B, C):
        raise RlxError("dones must be [n+1,B,C] and loss_mask [n,B,C]")
    if group_size < 1 or B % group_size != 0:
        raise RlxError(f"batch {B} not divisible by group_size {group_size}")
    adv = torch.empty_like(r)
    scores = torch.empty((B,), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        _lib.check(lib.rlx_grpo_group_adv(r.data_ptr(), d8.data_ptr(), m8.data_ptr(), scores.data_ptr(), adv.data_ptr(),
                                          n, B, C, int(group_size), float(eps), _stream_ptr(dev)), "rlx_grpo_group_adv")
    return adv, scores


def grpo_from_scores(scores: torch.Tensor, loss_mask: torch.Tensor, group_size: int, eps: float = 1e-6) -> torch.Tensor:
    """scores [B] f32, loss_mask [n,B,C] -> advantages [n,B,C] (advantages.py:107-121)."""
    lib = _lib.load()
    dev = _dev(scores, loss_mask)
    n, B, C = _nbc(loss_mask, "loss_mask")
    sc = _as_f32(scores.reshape(-1), "scores")
    if sc.numel() != B or group_size < 1 or B % group_size != 0:
        raise RlxError(f"scores must have {B} elements and batch must divide by group_size={group_size}")
    m8 = _as_u8(loss_mask)
    adv = torch.empty((n, B, C), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        _lib.check(lib.rlx_grpo_from_scores(sc.data_ptr(), m8.data_ptr(), adv.data_ptr(), n, B, C, int(group_size),
                                            float(eps), _stream_ptr(dev)), "rlx_grpo_from_scores")
    return adv


def gaussian_entropy_bonus_(params: torch.Tensor, layout: MlpLayout, grads_slab0: torch.Tensor, out_row: torch.Tensor,
                            entropy_bonus: float, grad_scale: float, has_mask: bool, elem_scale: float = 1.0,
                            actor_scale: Optional[torch.Tensor] = None):
    """a22 for the Gaussian MLP policy: fold -entropy_bonus * masked_mean(entropy) into the loss row and the logstd
    gradient (see include/rlx.h).  ``grads_slab0`` is the first gradient slab [n_params].  ``actor_scale`` (one device float):
    the slab is in sum form behind a decoupled ppo_step and will be multiplied by it later -- the bonus is pre-divided."""
    dev = _dev(params, grads_slab0, out_row)
    off, A = int(layout.off_logstd), int(layout.act_dim)
    logstd = params[off:off + A]
    g = grads_slab0[off:off + A]
    if actor_scale is not None:
        with torch.cuda.device(dev):
            _lib.check(_lib.load().rlx_gaussian_entropy_bonus_deferred(
                logstd.data_ptr(), A, g.data_ptr(), out_row.data_ptr(), float(entropy_bonus), float(grad_scale), int(bool(has_mask)),
                float(elem_scale), actor_scale.data_ptr(), _stream_ptr(dev)), "rlx_gaussian_entropy_bonus_deferred")
        return
    with torch.cuda.device(dev):
        _lib.check(_lib.load().rlx_gaussian_entropy_bonus(logstd.data_ptr(), A, g.data_ptr(), out_row.data_ptr(),
                                                          float(entropy_bonus), float(grad_scale), int(bool(has_mask)),
                                                          float(elem_scale), _stream_ptr(dev)),
                   "rlx_gaussian_entropy_bonus")


def rollout_metrics(arrays: list, loss_mask: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None,
                    workspace: Optional[torch.Tensor] = None) -> torch.Tensor:
    """a15: masked (sum, count, -min, max) of up to three f32 arrays in one pass -> f64 [len(arrays), 4] on the device
    (compute_rollout_metrics, metric_utils.py:422-506).  ``loss_mask`` broadcasts over trailing dimensions like the reference's
    ``v[mask.expand_as(v)]``: every array's element count must be a multiple of the mask's."""
    lib = _lib.load()
    dev = _dev(*arrays, loss_mask)
    xs = [_as_f32(a, "metric array") for a in arrays]
    k = len(xs)
    m8 = _as_u8(loss_mask)
    ptrs = (ctypes.c_void_p * k)(*[x.data_ptr() for x in xs])
    sizes = (ctypes.c_int64 * k)(*[x.numel() for x in xs])
    out = torch.empty((k, 4), dtype=torch.float64, device=dev) if out is None else out
    wsb = lib.rlx_rollout_metrics_workspace_bytes()
    ws = workspace if workspace is not None and workspace.numel() >= wsb else torch.empty(wsb, dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        _lib.check(lib.rlx_rollout_metrics(ptrs, sizes, k, _ptr(m8), 0 if m8 is None else m8.numel(), out.data_ptr(), ws.data_ptr(),
                                           ws.numel(), _stream_ptr(dev)), "rlx_rollout_metrics")
    return out


def reward_filter_mask(rewards: torch.Tensor, loss_mask: Optional[torch.Tensor], group_size: int, lower: float,
                       upper: float) -> torch.Tensor:
    """rewards [n, B, C] f32 (+ loss_mask [n, B, C] bool) -> the filtered loss mask (bool): [n, B, C] with a mask,
    [n, B, 1] without (embodied_fsdp_actor_worker.py:235-281)."""
    dev = _dev(rewards, loss_mask)
    n, B, C = _nbc(rewards, "rewards")
    r = _as_f32(rewards, "rewards")
    m = _as_u8(loss_mask)
    if m is not None and m.shape != r.shape:
        raise RlxError("loss_mask must have the shape of rewards")
    if group_size < 1 or B % group_size != 0:
        raise AssertionError(f"batch {B} not divisible by group_size {group_size}")
    out = torch.empty((n, B, C if m is not None else 1), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        _lib.check(_lib.load().rlx_reward_filter_mask(r.data_ptr(), _ptr(m), out.data_ptr(), n, B, C, int(group_size),
                                                      float(lower), float(upper), _stream_ptr(dev)),
                   "rlx_reward_filter_mask")
    return out.view(torch.bool)


def episode_scores(rewards: torch.Tensor, dones: torch.Tensor) -> torch.Tensor:
    """rewards [n,B,C], dones [n+1,B,C] -> per-env first-episode return [B] (utils.py:134-152)."""
    lib = _lib.load()
    dev = _dev(rewards, dones)
    n, B, C = _nbc(rewards, "rewards")
    r = _as_f32(rewards, "rewards")
    d8 = _as_u8(dones)
    if tuple(d8.shape) != (n + 1, B, C):
        raise RlxError("dones must be [n+1,B,C]")
    scores = torch.empty((B,), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        _lib.check(lib.rlx_episode_scores(r.data_ptr(), d8.data_ptr(), scores.data_ptr(), n, B, C, _stream_ptr(dev)),
                   "rlx_episode_scores")
    return scores


# --------------------------------------------------------------------------------------------
# a18-a21  fused PPO loss (forward + backward) as an autograd node
# --------------------------------------------------------------------------------------------
_LEVELS = ("action_level", "token_level", "chunk_level")


def _loss_geometry(logprobs: torch.Tensor, logprob_type: str, action_dim: int, reward_type: Optional[str] = None):
    """(n_adv, raw_per_adv, sub_per_adv) for a [bsz, C*A] log-prob tensor (algorithms/utils.py:325-356).  With
    reward_type='chunk_level' the advantage / mask / value tensors are per env-step ([bsz], utils.py:296-308) while
    token-level log-probs keep one loss element per action dimension of every chunk: bsz x (C*A) x (C*A)."""
    bsz = logprobs.shape[0]
    per_row = logprobs.numel() // max(bsz, 1)
    if logprob_type not in _LEVELS:
        raise RlxError(f"logprob_type must be one of {_LEVELS}, got {logprob_type!r}")
    if per_row % action_dim != 0:
        raise RlxError(f"logprobs row of {per_row} elements is not a multiple of action_dim={action_dim}")
    chunks = per_row // action_dim
    if reward_type == "chunk_level" and logprob_type == "token_level":
        return bsz, per_row, per_row
    if reward_type == "chunk_level" and logprob_type == "action_level" and chunks > 1:
        # ratio [bsz, C] against [bsz, 1] advantages / mask: one advantage element per env step, C ratios under it, each the
        # sum of action_dim raw log-probs; the ratio metrics keep the un-broadcast mask count (PpoLossParams.metric_unbroadcast)
        return bsz, per_row, chunks
    if logprob_type == "action_level":
        return bsz * chunks, action_dim, 1
    if logprob_type == "token_level":
        return bsz * chunks, action_dim, action_dim
    return bsz, per_row, 1


class _PpoLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logprobs, values, old_logprobs, advantages, prev_values, returns, loss_mask, loss_mask_sum,
                params: PpoLossParams, n_adv: int):
        lib = _lib.load()
        dev = logprobs.device
        lp = logprobs.contiguous()
        g_lp = torch.empty((n_adv * params.sub_per_adv,), dtype=torch.float32, device=dev)
        g_v = torch.empty((n_adv,), dtype=torch.float32, device=dev) if params.has_critic else None
        out = torch.empty((_lib.PPO_OUT_FLOATS,), dtype=torch.float32, device=dev)
        ws_bytes = lib.rlx_ppo_loss_workspace_bytes(n_adv)
        ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=dev)
        with torch.cuda.device(dev):
            _lib.check(lib.rlx_ppo_loss_fwd(lp.data_ptr(), old_logprobs.data_ptr(), advantages.data_ptr(), _ptr(values),
                                            _ptr(prev_values), _ptr(returns), _ptr(loss_mask), _ptr(loss_mask_sum), n_adv,
                                            byref(params), g_lp.data_ptr(), _ptr(g_v), out.data_ptr(), ws.data_ptr(),
                                            ws_bytes, _stream_ptr(dev)), "rlx_ppo_loss_fwd")
        ctx.save_for_backward(g_lp, g_v if g_v is not None else out, out)
        ctx.has_critic = bool(params.has_critic)
        ctx.geom = (n_adv, params.raw_per_adv, params.sub_per_adv)
        ctx.lp_shape = logprobs.shape
        ctx.v_shape = None if values is None else values.shape
        ctx.mark_non_differentiable(out)
        return out[0].clone(), out

    @staticmethod
    def backward(ctx, grad_loss, _grad_out):
        lib = _lib.load()
        g_lp, g_v, out = ctx.saved_tensors
        n_adv, raw, sub = ctx.geom
        dev = g_lp.device
        go = grad_loss.to(dtype=torch.float32).contiguous()
        d_lp = torch.empty((n_adv * raw,), dtype=torch.float32, device=dev)
        d_v = torch.empty((n_adv,), dtype=torch.float32, device=dev) if ctx.has_critic else None
        with torch.cuda.device(dev):
            _lib.check(lib.rlx_ppo_loss_bwd(g_lp.data_ptr(), g_v.data_ptr() if ctx.has_critic else None, out.data_ptr(),
                                            go.data_ptr(), d_lp.data_ptr(), _ptr(d_v), n_adv, raw, sub,
                                            _stream_ptr(dev)), "rlx_ppo_loss_bwd")
        return (d_lp.view(ctx.lp_shape), None if d_v is None else d_v.view(ctx.v_shape), None, None, None, None, None,
                None, None, None)


def ppo_loss(logprobs: torch.Tensor, old_logprobs: torch.Tensor, advantages: torch.Tensor, *,
             logprob_type: str = "action_level", action_dim: int = 8, clip_ratio_low: float, clip_ratio_high: float,
             values: Optional[torch.Tensor] = None, prev_values: Optional[torch.Tensor] = None,
             returns: Optional[torch.Tensor] = None, value_clip: Optional[float] = None,
             huber_delta: Optional[float] = None, loss_mask: Optional[torch.Tensor] = None,
             loss_mask_sum: Optional[torch.Tensor] = None, max_episode_steps: Optional[int] = None,
             clip_ratio_c: Optional[float] = None, clip_log_ratio_min: Optional[float] = None,
             clip_log_ratio_max: Optional[float] = None, critic_warmup: bool = False, has_critic: bool = True,
             decoupled: Optional[dict] = None, reward_type: Optional[str] = None):
    """Fused actor(+critic) PPO loss.  Inputs are the RAW per-dimension tensors the reference hands to
    policy_loss (before preprocess_loss_inputs); returns (loss 0-dim tensor with grad, out f32[20] on device,
    see _lib.PPO_OUT_NAMES).  losses.py asserts float32 inputs (:232-240); so do we."""
    dev = _dev(logprobs, old_logprobs, advantages, values, prev_values, returns, loss_mask, loss_mask_sum)
    for name, t in (("logprobs", logprobs), ("old_logprobs", old_logprobs), ("advantages", advantages)):
        if t.dtype != torch.float32:
            raise RlxError(f"{name} must be float32 to keep numerical stability")
    n_adv, raw, sub = _loss_geometry(logprobs, logprob_type, action_dim, reward_type)
    unbroadcast = reward_type == "chunk_level" and logprob_type == "action_level" and sub > 1
    if old_logprobs.numel() != logprobs.numel():
        raise RlxError("old_logprobs must have the shape of logprobs")
    if advantages.numel() != n_adv:
        raise RlxError(f"advantages has {advantages.numel()} elements, expected {n_adv} for {logprob_type}")
    if has_critic:
        if values is None or prev_values is None or returns is None or value_clip is None or huber_delta is None:
            raise RlxError("actor_critic loss needs values, prev_values, returns, value_clip and huber_delta")
        for name, t in (("values", values), ("prev_values", prev_values), ("returns", returns)):
            if t.numel() != n_adv:
                raise RlxError(f"{name} has {t.numel()} elements, expected {n_adv}")
            if t.dtype != torch.float32:
                raise RlxError(f"{name} must be float32")
    m8 = _as_u8(loss_mask)
    if m8 is not None and m8.numel() != n_adv:
        raise RlxError(f"loss_mask has {m8.numel()} elements, expected {n_adv}")
    msum = None
    if loss_mask_sum is not None:
        if loss_mask_sum.numel() != n_adv:
            msum = loss_mask_sum.expand(advantages.shape if loss_mask_sum.dim() == advantages.dim() else loss_mask_sum.shape)
        else:
            msum = loss_mask_sum
        msum = msum.to(torch.int64).contiguous()
        if msum.numel() != n_adv:
            raise RlxError(f"loss_mask_sum has {msum.numel()} elements, expected {n_adv}")
    if clip_ratio_c is not None and not clip_ratio_c > 1.0:
        raise AssertionError("clip_ratio_c must be greater than 1.0")
    p = PpoLossParams()
    p.ratio_lo, p.ratio_hi = float(1.0 - clip_ratio_low), float(1.0 + clip_ratio_high)
    p.clip_ratio_c = float(clip_ratio_c) if clip_ratio_c is not None else 0.0
    p.use_dual_clip = int(clip_ratio_c is not None)
    p.clip_log_ratio_min = float(clip_log_ratio_min) if clip_log_ratio_min is not None else 0.0
    p.clip_log_ratio_max = float(clip_log_ratio_max) if clip_log_ratio_max is not None else 0.0
    p.use_clip_log_ratio_min = int(clip_log_ratio_min is not None)
    p.use_clip_log_ratio_max = int(clip_log_ratio_max is not None)
    p.value_clip = float(value_clip) if has_critic else 0.0
    p.huber_delta = float(huber_delta) if has_critic else 0.0
    p.has_critic = int(has_critic)
    p.critic_warmup = int(bool(critic_warmup))
    p.max_episode_steps = int(max_episode_steps) if max_episode_steps else 0
    p.raw_per_adv, p.sub_per_adv, p.metric_unbroadcast = raw, sub, int(unbroadcast)
    v = values.contiguous() if has_critic else None
    if decoupled is not None:
        dp, prox, versions = _decoupled_params(p, logprobs, **decoupled)
        return _DecoupledLossFn.apply(logprobs, v, old_logprobs.contiguous(), advantages.contiguous(),
                                      prev_values.contiguous() if has_critic else None,
                                      returns.contiguous() if has_critic else None, m8, msum, prox, versions, dp, n_adv)
    return _PpoLossFn.apply(logprobs, v, old_logprobs.contiguous(), advantages.contiguous(),
                            prev_values.contiguous() if has_critic else None,
                            returns.contiguous() if has_critic else None, m8, msum, p, n_adv)


def _decoupled_params(p: PpoLossParams, logprobs, proximal_logprobs=None, versions=None, current_version=None,
                      behave_weight_threshold=None):
    """rlx_decoupled_loss_params + the two optional raw tensors (losses.py:68-89: which proximal policy applies)."""
    dp = _lib.DecoupledLossParams()
    dp.ppo = p
    for name, t in (("proximal_logprobs", proximal_logprobs), ("versions", versions)):
        if t is not None and t.numel() != logprobs.numel():
            raise RlxError(f"{name} must have the shape of logprobs")
    if proximal_logprobs is not None:
        if proximal_logprobs.dtype != torch.float32:
            raise AssertionError("proximal_logprobs must be float32 to keep numerical stability")
        dp.proximal_mode = _lib.PROX_GIVEN
    elif versions is None or current_version is None:
        dp.proximal_mode = _lib.PROX_IS_OLD
    else:
        dp.proximal_mode = _lib.PROX_FROM_VERSIONS
    dp.current_version = float(current_version) if current_version is not None else 0.0
    dp.use_behave_threshold = int(behave_weight_threshold is not None)
    dp.behave_weight_threshold = float(behave_weight_threshold) if behave_weight_threshold is not None else 0.0
    prox = proximal_logprobs.contiguous() if proximal_logprobs is not None else None
    ver = versions.float().contiguous() if (versions is not None and current_version is not None) else None
    return dp, prox, ver


class _DecoupledLossFn(torch.autograd.Function):
    """rlx_decoupled_loss_fwd + the shared rlx_ppo_loss_bwd."""

    @staticmethod
    def forward(ctx, logprobs, values, old_logprobs, advantages, prev_values, returns, loss_mask, loss_mask_sum,
                proximal_logprobs, versions, dp, n_adv: int):
        lib = _lib.load()
        dev = logprobs.device
        params = dp.ppo
        lp = logprobs.contiguous()
        g_lp = torch.empty((n_adv * params.sub_per_adv,), dtype=torch.float32, device=dev)
        g_v = torch.empty((n_adv,), dtype=torch.float32, device=dev) if params.has_critic else None
        out = torch.empty((_lib.PPO_OUT_FLOATS,), dtype=torch.float32, device=dev)
        ws_bytes = lib.rlx_decoupled_loss_workspace_bytes(n_adv)
        ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=dev)
        with torch.cuda.device(dev):
            _lib.check(lib.rlx_decoupled_loss_fwd(
                lp.data_ptr(), old_logprobs.data_ptr(), _ptr(proximal_logprobs), _ptr(versions), advantages.data_ptr(),
                _ptr(values), _ptr(prev_values), _ptr(returns), _ptr(loss_mask), _ptr(loss_mask_sum), n_adv, byref(dp),
                g_lp.data_ptr(), _ptr(g_v), out.data_ptr(), ws.data_ptr(), ws_bytes, _stream_ptr(dev)),
                "rlx_decoupled_loss_fwd")
        ctx.save_for_backward(g_lp, g_v if g_v is not None else out, out)
        ctx.has_critic = bool(params.has_critic)
        ctx.geom = (n_adv, params.raw_per_adv, params.sub_per_adv)
        ctx.lp_shape = logprobs.shape
        ctx.v_shape = None if values is None else values.shape
        ctx.mark_non_differentiable(out)
        return out[0].clone(), out

    @staticmethod
    def backward(ctx, grad_loss, _grad_out):
        d_lp, d_v = _PpoLossFn.backward(ctx, grad_loss, _grad_out)[:2]
        return (d_lp, d_v) + (None,) * 10


def make_ppo_params(*, logprob_type="action_level", action_dim=8, chunks=1, clip_ratio_low, clip_ratio_high,
                    value_clip=None, huber_delta=None, max_episode_steps=None, clip_ratio_c=None,
                    clip_log_ratio_min=None, clip_log_ratio_max=None, critic_warmup=False, has_critic=True,
                    reward_type=None) -> PpoLossParams:
    p = PpoLossParams()
    p.ratio_lo, p.ratio_hi = float(1.0 - clip_ratio_low), float(1.0 + clip_ratio_high)
    p.clip_ratio_c = float(clip_ratio_c) if clip_ratio_c is not None else 0.0
    p.use_dual_clip = int(clip_ratio_c is not None)
    p.clip_log_ratio_min = float(clip_log_ratio_min) if clip_log_ratio_min is not None else 0.0
    p.clip_log_ratio_max = float(clip_log_ratio_max) if clip_log_ratio_max is not None else 0.0
    p.use_clip_log_ratio_min, p.use_clip_log_ratio_max = int(clip_log_ratio_min is not None), int(clip_log_ratio_max is not None)
    p.value_clip = float(value_clip) if has_critic else 0.0
    p.huber_delta = float(huber_delta) if has_critic else 0.0
    p.has_critic, p.critic_warmup = int(has_critic), int(bool(critic_warmup))
    p.max_episode_steps = int(max_episode_steps) if max_episode_steps else 0
    if reward_type == "chunk_level" and logprob_type == "token_level":  # one advantage per env step, a ratio per raw log-prob
        p.raw_per_adv, p.sub_per_adv = action_dim * chunks, action_dim * chunks
    elif reward_type == "chunk_level" and logprob_type == "action_level" and chunks > 1:  # ... a ratio per action chunk
        p.raw_per_adv, p.sub_per_adv, p.metric_unbroadcast = action_dim * chunks, chunks, 1
    elif logprob_type == "action_level":
        p.raw_per_adv, p.sub_per_adv = action_dim, 1
    elif logprob_type == "token_level":
        p.raw_per_adv, p.sub_per_adv = action_dim, action_dim
    elif logprob_type == "chunk_level":
        p.raw_per_adv, p.sub_per_adv = action_dim * chunks, 1
    else:
        raise RlxError(f"logprob_type must be one of {_LEVELS}")
    return p


def ppo_loss_fwd_raw(params: PpoLossParams, n_adv: int, logprobs, old_logprobs, advantages, values, prev_values, returns,
                     loss_mask, loss_mask_sum, g_logp, g_value, out, workspace):
    """Allocation-free launcher (all tensors caller-owned) used by the learner's fused/graph-captured step."""
    lib = _lib.load()
    dev = logprobs.device
    with torch.cuda.device(dev):
        _lib.check(lib.rlx_ppo_loss_fwd(logprobs.data_ptr(), old_logprobs.data_ptr(), advantages.data_ptr(), _ptr(values),
                                        _ptr(prev_values), _ptr(returns), _ptr(loss_mask), _ptr(loss_mask_sum), n_adv,
                                        byref(params), g_logp.data_ptr(), _ptr(g_value), out.data_ptr(),
                                        workspace.data_ptr(), workspace.numel(), _stream_ptr(dev)), "rlx_ppo_loss_fwd")


def ppo_loss_bwd_raw(params: PpoLossParams, n_adv: int, g_logp, g_value, out, grad_out, d_logprobs, d_values):
    lib = _lib.load()
    dev = g_logp.device
    with torch.cuda.device(dev):
        _lib.check(lib.rlx_ppo_loss_bwd(g_logp.data_ptr(), _ptr(g_value), out.data_ptr(), _ptr(grad_out),
                                        d_logprobs.data_ptr(), _ptr(d_values), n_adv, params.raw_per_adv, params.sub_per_adv,
                                        _stream_ptr(dev)), "rlx_ppo_loss_bwd")


# --------------------------------------------------------------------------------------------
# a16  shuffle gather
# --------------------------------------------------------------------------------------------
def gather_rows(fields: list, index: torch.Tensor, outs: Optional[list] = None) -> list:
    """fields: list of [N, ...] contiguous HIP tensors; returns [t[index] for t in fields] (one launch per 16)."""
    lib = _lib.load()
    if not fields:
        return []
    dev = _dev(index, *fields)
    if index.dtype != torch.int64:
        raise RlxError("index must be int64")
    idx = index.contiguous()
    n = idx.numel()
    for t in fields:
        if not t.is_contiguous():
            raise RlxError("gather_rows needs contiguous fields")
    if outs is None:
        outs = [torch.empty((n, *t.shape[1:]), dtype=t.dtype, device=dev) for t in fields]
    with torch.cuda.device(dev):
        for lo in range(0, len(fields), _lib.GATHER_MAX_FIELDS):
            chunk = list(zip(fields, outs))[lo:lo + _lib.GATHER_MAX_FIELDS]
            arr = (GatherField * len(chunk))()
            k = 0
            for src, dst in chunk:
                row_bytes = src.element_size() * (src.numel() // max(src.shape[0], 1))
                if row_bytes == 0:
                    continue
                arr[k].src, arr[k].dst, arr[k].row_bytes = src.data_ptr(), dst.data_ptr(), row_bytes
                k += 1
            if k and n:
                _lib.check(lib.rlx_gather_rows(arr, k, idx.data_ptr(), n, _stream_ptr(dev)), "rlx_gather_rows")
    return outs


# --------------------------------------------------------------------------------------------
# a23  clip + AdamW on flat buffers
# --------------------------------------------------------------------------------------------
def actor_param_ranges(layout: MlpLayout) -> list:
    """Element ranges of the flat buffer that belong to the ACTOR network (actor_logstd; backbone.* + actor_mean.*): the
    gradients a decoupled ppo_step leaves in sum form."""
    A = int(layout.act_dim)
    lo = int(layout.off_w[1][0])
    hi = max(int(layout.off_w[1][3]) + A * int(layout.hidden), (int(layout.off_b[1][3]) + A) if int(layout.off_b[1][3]) >= 0 else 0)
    return [(int(layout.off_logstd), int(layout.off_logstd) + A), (lo, hi)]


def deferred_actor_scale(layout: MlpLayout, rows: torch.Tensor, groups: int) -> dict:
    """``deferred`` argument of the AdamW wrappers behind decoupled ppo_steps: ``rows`` = the [groups, PPO_OUT_FLOATS] metric rows
    of the optimizer step's micro-batches (contiguous); their out[RLX_PPO_ACTOR_GRAD_SCALE] scale the actor ranges."""
    assert rows.dim() == 2 and rows.shape[0] == groups and rows.stride(1) == 1
    return dict(scale=rows[0, _lib.PPO_ACTOR_GRAD_SCALE:], stride=int(rows.stride(0)), groups=int(groups),
                ranges=actor_param_ranges(layout), keep=rows)


def adamw_sync_words(n: int, device) -> Optional[torch.Tensor]:
    """Zeroed ``rlx_adamw_params.sync_words`` buffer for a set of ``n`` parameters: with it the optimizer step is ONE launch
    (slab sum, norm, clip and AdamW around a device-side exchange of the norm partials) where the plan allows it.  One buffer per
    parameter set, reused by every step on it.  ``RLX_ADAMW_ONE_LAUNCH=0`` -> None (the two-launch form)."""
    if os.environ.get("RLX_ADAMW_ONE_LAUNCH", "1") == "0":
        return None
    return torch.zeros((int(_lib.load().rlx_adamw_sync_words(int(n))),), dtype=torch.int64, device=device)


def check_adamw_sync(sync: Optional[torch.Tensor], grad_norm: float) -> bool:
    """Call with a step's gradient norm once it is on the host.  A NON-FINITE norm is either a real one (the step was skipped, like
    the reference's) or the one-launch optimizer step reporting that one of its workgroups never became resident within its 2 s
    bound (word 1 of the sync buffer, sticky) -- another process or stream holds part of the GPU.  Steps since then were skipped
    as a whole (the kernel reads the word first thing), so the parameters are those of the last complete step.  Returns True in
    that case, after a warning: the caller drops ``sync`` (and whatever captured its pointer) and continues on two launches."""
    import math
    if sync is None or math.isfinite(grad_norm):
        return False
    if int(sync[1].item()) == 0:
        return False
    import warnings
    warnings.warn("the one-launch optimizer step timed out waiting for its own workgroups (another process or stream holds part of "
                  "this GPU); the optimizer steps since then were skipped. Continuing with the two-launch form "
                  "(RLX_ADAMW_ONE_LAUNCH=0 selects it from the start).", RuntimeWarning, stacklevel=2)
    return True


def _set_deferred(p: AdamwParams, deferred: Optional[dict]):
    if deferred is None:
        return
    p.deferred_scale = deferred["scale"].data_ptr()
    p.deferred_stride, p.deferred_groups = int(deferred["stride"]), int(deferred["groups"])
    for k, (b, e) in enumerate(deferred["ranges"]):
        p.deferred_range[k][0], p.deferred_range[k][1] = int(b), int(e)


def clip_adamw_step_(params: torch.Tensor, grads: torch.Tensor, exp_avg: torch.Tensor, exp_avg_sq: torch.Tensor,
                     groups: list, step: int, *, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 0.01,
                     max_grad_norm: float = 0.5, grad_scale: float = 1.0, stats: Optional[torch.Tensor] = None,
                     step_state: Optional[torch.Tensor] = None, workspace: Optional[torch.Tensor] = None,
                     tile_layout: Optional[MlpLayout] = None, tiles: Optional[torch.Tensor] = None,
                     deferred: Optional[dict] = None, sync: Optional[torch.Tensor] = None):
    """In-place clip_grad_norm_ + AdamW over flat f32 buffers.  groups = [(begin, end, lr), ...].
    sync: ``adamw_sync_words(n, device)`` -> the one-launch form.
    grads may be [n] or [slabs, n] (split-K slabs, summed first).  Returns stats f32[2] = (norm, applied).
    step_state: device int32[2] keeping the step count on the device (then ``step`` is ignored).
    deferred: see ``deferred_actor_scale`` (scales read on the device while the slabs are summed)."""
    lib = _lib.load()
    dev = _dev(params, grads, exp_avg, exp_avg_sq)
    n = params.numel()
    for t in (params, grads, exp_avg, exp_avg_sq):
        if t.dtype != torch.float32 or not t.is_contiguous():
            raise RlxError("clip_adamw_step_ needs contiguous float32 buffers")
    if grads.numel() % max(n, 1) != 0 or exp_avg.numel() != n or exp_avg_sq.numel() != n:
        raise RlxError("buffer sizes do not match")
    slabs = grads.numel() // max(n, 1)
    if len(groups) > _lib.ADAMW_MAX_GROUPS:
        raise RlxError("too many parameter groups")
    p = AdamwParams()
    p.beta1, p.beta2, p.eps, p.weight_decay = float(betas[0]), float(betas[1]), float(eps), float(weight_decay)
    p.max_grad_norm = float(max_grad_norm) if max_grad_norm else 0.0
    p.step, p.n_groups, p.grad_partials, p.grad_scale = int(step), len(groups), int(slabs), float(grad_scale)
    for k, (b, e, lr) in enumerate(groups):
        p.groups[k] = AdamwGroup(int(b), int(e), float(lr))
    if tile_layout is not None and tiles is not None:  # keep the fragment-tile weight image in step with the parameters
        p.tile_layout, p.tiles = ctypes_pointer(tile_layout), tiles.data_ptr()
        p.tiles_bf16 = int(tiles.dtype == torch.bfloat16)
    _set_deferred(p, deferred)
    if sync is not None:
        p.sync_words = sync.data_ptr()
    if stats is None:
        stats = torch.empty((2,), dtype=torch.float32, device=dev)
    ws_bytes = lib.rlx_adamw_workspace_bytes(n)
    ws = workspace if workspace is not None and workspace.numel() >= ws_bytes else torch.empty((ws_bytes,), dtype=torch.uint8, device=dev)
    if step_state is not None and (step_state.dtype != torch.int32 or step_state.numel() < 2):
        raise RlxError("step_state must be int32[2]")
    with torch.cuda.device(dev):
        _lib.check(lib.rlx_clip_adamw_step(params.data_ptr(), grads.data_ptr(), exp_avg.data_ptr(), exp_avg_sq.data_ptr(), n,
                                           byref(p), stats.data_ptr(), _ptr(step_state), ws.data_ptr(), ws.numel(),
                                           _stream_ptr(dev)), "rlx_clip_adamw_step")
    return stats


def bootstrap_rewards_(rewards: torch.Tensor, flags: torch.Tensor, bootstrap_values: torch.Tensor, gamma: float):
    """rewards [B,C] f32 in place: r[:, -1] += gamma * V[:, 0] where flags[:, -1]  (env_worker.py:744-758)."""
    lib = _lib.load()
    dev = _dev(rewards, flags, bootstrap_values)
    if rewards.dim() != 2 or not rewards.is_contiguous() or rewards.dtype != torch.float32:
        raise RlxError("rewards must be a contiguous float32 [B, C] tensor")
    f8 = _as_u8(flags)
    v = _as_f32(bootstrap_values, "bootstrap_values")
    B, C = rewards.shape
    if tuple(f8.shape) != (B, C) or v.shape[0] != B:
        raise RlxError("flags must be [B,C] and bootstrap_values [B, >=1]")
    with torch.cuda.device(dev):
        _lib.check(lib.rlx_bootstrap_rewards(rewards.data_ptr(), f8.data_ptr(), v.data_ptr(), B, C,
                                             v.numel() // max(B, 1), float(gamma), _stream_ptr(dev)), "rlx_bootstrap_rewards")
    return rewards


def store_env_rows_(rewards, terminations, truncations, reward_row, done_row, termination_row, truncation_row):
    """One env step's [B, C] outputs -> buffer rows; dones = terminations | truncations.  One launch."""
    lib = _lib.load()
    dev = _dev(rewards, terminations, truncations, reward_row, done_row, termination_row, truncation_row)
    r = _as_f32(rewards, "rewards")
    te, tr = _as_u8(terminations), _as_u8(truncations)
    n = r.numel()
    for t in (te, tr, reward_row, done_row, termination_row, truncation_row):
        if t.numel() != n or not t.is_contiguous():
            raise RlxError("store_env_rows_: all tensors must be contiguous with rewards' number of elements")
    with torch.cuda.device(dev):
        _lib.check(lib.rlx_store_env_rows(r.data_ptr(), te.data_ptr(), tr.data_ptr(), reward_row.data_ptr(),
                                          done_row.data_ptr(), termination_row.data_ptr(), truncation_row.data_ptr(), n,
                                          _stream_ptr(dev)), "rlx_store_env_rows")


# --------------------------------------------------------------------------------------------
# a1-a5, a17  MLP policy kernels (flat parameter buffer + layout descriptor)
# --------------------------------------------------------------------------------------------
def mlp_pack(params: torch.Tensor, layout: MlpLayout, packed: Optional[torch.Tensor] = None) -> torch.Tensor:
    lib = _lib.load()
    dev = _dev(params)
    nbytes = lib.rlx_mlp_packed_bytes(byref(layout))
    if packed is None:
        packed = torch.empty((nbytes // 4,), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        _lib.check(lib.rlx_mlp_pack(params.data_ptr(), byref(layout), packed.data_ptr(), _stream_ptr(dev)), "rlx_mlp_pack")
    return packed


def mlp_rollout(params: torch.Tensor, packed: torch.Tensor, layout: MlpLayout, states: torch.Tensor,
                eps: Optional[torch.Tensor], out: Optional[tuple] = None):
    """states [M,D], eps [M,act]|None -> (action [M,act], logprob [M,act], value [M,val])."""
    lib = _lib.load()
    dev = _dev(params, packed, states, eps)
    st = _as_f32(states, "states")
    M = st.shape[0]
    if st.dim() != 2 or st.shape[1] != layout.obs_dim:
        raise RlxError(f"states must be [M, {layout.obs_dim}], got {tuple(st.shape)}")
    e = _as_f32(eps, "eps")
    if e is not None and tuple(e.shape) != (M, layout.act_dim):
        raise RlxError(f"eps must be [{M}, {layout.act_dim}]")
    if out is not None:  # write straight into caller-owned rows (e.g. the trajectory buffer)
        action, logprob, value = out
        for t, w in ((action, layout.act_dim), (logprob, layout.act_dim), (value, layout.val_dim)):
            if t.dtype != torch.float32 or not t.is_contiguous() or t.numel() != M * w:
                raise RlxError("mlp_rollout: out tensors must be contiguous float32 of the right size")
    else:
        action = torch.empty((M, layout.act_dim), dtype=torch.float32, device=dev)
        logprob = torch.empty_like(action)
        value = torch.empty((M, layout.val_dim), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        _lib.check(lib.rlx_mlp_rollout(params.data_ptr(), packed.data_ptr(), byref(layout), st.data_ptr(), _ptr(e), M,
                                       action.data_ptr(), logprob.data_ptr(), value.data_ptr(), _stream_ptr(dev)),
                   "rlx_mlp_rollout")
    return action, logprob, value


def mlp_value(params, packed, layout: MlpLayout, states, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Value head only: states [M,D] -> value [M,val]."""
    lib = _lib.load()
    dev = _dev(params, packed, states)
    st = _as_f32(states, "states")
    M = st.shape[0]
    value = out if out is not None else torch.empty((M, layout.val_dim), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        _lib.check(lib.rlx_mlp_value(params.data_ptr(), packed.data_ptr(), byref(layout), st.data_ptr(), M, value.data_ptr(),
                                     _stream_ptr(dev)), "rlx_mlp_value")
    return value


def sum_slabs(grads: torch.Tensor, out: Optional[torch.Tensor] = None, deferred: Optional[dict] = None) -> torch.Tensor:
    """[slabs, n] -> [n].  ``deferred`` (deferred_actor_scale): the slabs are micro-batch groups behind decoupled ppo_steps."""
    lib = _lib.load()
    dev = _dev(grads)
    slabs, n = grads.shape
    if out is None:
        out = torch.empty((n,), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        if deferred is not None:
            p = AdamwParams()
            _set_deferred(p, deferred)
            _lib.check(lib.rlx_sum_slabs_deferred(grads.data_ptr(), n, slabs, out.data_ptr(), byref(p), _stream_ptr(dev)),
                       "rlx_sum_slabs_deferred")
        else:
            _lib.check(lib.rlx_sum_slabs(grads.data_ptr(), n, slabs, out.data_ptr(), _stream_ptr(dev)), "rlx_sum_slabs")
    return out


def mlp_train_fwd(params, packed, layout: MlpLayout, states, action, acts: Optional[torch.Tensor] = None,
                  out: Optional[tuple] = None):
    """-> (logprob, entropy [M,act], value [M,val], mean [M,act], acts [2,3,M,256])."""
    lib = _lib.load()
    dev = _dev(params, packed, states, action)
    st, ac = _as_f32(states, "states"), _as_f32(action, "action")
    M = st.shape[0]
    if tuple(ac.shape) != (M, layout.act_dim):
        raise RlxError(f"action must be [{M}, {layout.act_dim}], got {tuple(ac.shape)}")
    if out is not None:
        logprob, entropy, value, mean = out
    else:
        logprob = torch.empty((M, layout.act_dim), dtype=torch.float32, device=dev)
        entropy = torch.empty_like(logprob)
        mean = torch.empty_like(logprob)
        value = torch.empty((M, layout.val_dim), dtype=torch.float32, device=dev)
    if acts is None or acts.numel() != 6 * M * 256:
        acts = torch.empty((2, 3, M, 256), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        _lib.check(lib.rlx_mlp_train_fwd(params.data_ptr(), packed.data_ptr(), byref(layout), st.data_ptr(), ac.data_ptr(), M,
                                         logprob.data_ptr(), entropy.data_ptr(), value.data_ptr(), mean.data_ptr(),
                                         acts.data_ptr(), _stream_ptr(dev)), "rlx_mlp_train_fwd")
    return logprob, entropy, value, mean, acts


def mlp_bwd_slabs(m: int) -> int:
    return _lib.load().rlx_mlp_bwd_slabs(int(m))


def mlp_train_bwd(params, packed, layout: MlpLayout, states, action, mean, acts, d_logprob, d_entropy, d_value,
                  grads: Optional[torch.Tensor] = None, workspace: Optional[torch.Tensor] = None):
    """-> grads [slabs, n_params] (split-K slabs; every element written)."""
    lib = _lib.load()
    dev = _dev(params, packed, states, action, mean, acts, d_logprob, d_entropy, d_value)
    M = states.shape[0]
    slabs = lib.rlx_mlp_bwd_slabs(M) if grads is None else grads.shape[0]
    if grads is None:
        grads = torch.empty((slabs, layout.n_params), dtype=torch.float32, device=dev)
    ws_bytes = lib.rlx_mlp_bwd_workspace_bytes(byref(layout), M)
    if workspace is None or workspace.numel() < ws_bytes:
        workspace = torch.empty((ws_bytes,), dtype=torch.uint8, device=dev)
    dl, dv = _as_f32(d_logprob, "d_logprob"), _as_f32(d_value, "d_value")
    de = _as_f32(d_entropy, "d_entropy")
    with torch.cuda.device(dev):
        _lib.check(lib.rlx_mlp_train_bwd(params.data_ptr(), packed.data_ptr(), byref(layout), states.data_ptr(),
                                         action.data_ptr(), mean.data_ptr(), acts.data_ptr(), dl.data_ptr(), _ptr(de),
                                         dv.data_ptr(), M, grads.data_ptr(), slabs, workspace.data_ptr(),
                                         workspace.numel(), _stream_ptr(dev)), "rlx_mlp_train_bwd")
    return grads


# --------------------------------------------------------------------------------------------
# fused hot launches (ppo_step.hip)
# --------------------------------------------------------------------------------------------
def mlp_pack_tiles(params: torch.Tensor, layout: MlpLayout, tiles: Optional[torch.Tensor] = None, bf16: bool = False) -> torch.Tensor:
    """Fragment-tile weight image the fused launches stream (rebuild after every change of ``params``); f32 or bf16."""
    lib = _lib.load()
    dev = _dev(params)
    dt = torch.bfloat16 if bf16 else torch.float32
    n = lib.rlx_mlp_tiles_bytes_for(byref(layout), 1 if bf16 else 0) // (2 if bf16 else 4)  # the f32 image is opaque (include/rlx.h)
    if tiles is None or tiles.dtype != dt or tiles.numel() != n:
        tiles = torch.empty((n,), dtype=dt, device=dev)
    fn = lib.rlx_mlp_pack_tiles_bf16 if bf16 else lib.rlx_mlp_pack_tiles
    with torch.cuda.device(dev):
        _lib.check(fn(params.data_ptr(), byref(layout), tiles.data_ptr(), _stream_ptr(dev)), "rlx_mlp_pack_tiles")
    return tiles


def mlp_rollout_step(params: torch.Tensor, tiles: torch.Tensor, layout: MlpLayout, states: Optional[torch.Tensor],
                     eps: Optional[torch.Tensor], out: Optional[tuple] = None, states_copy: Optional[torch.Tensor] = None,
                     value_jobs: tuple = ()):
    """ONE launch: policy job on ``states`` (-> action, logprob, value rows in ``out``) plus up to two value-only jobs
    ``dict(states=[m,D], values=[m,val]|None, rewards=[m,C]|None, flags=[m,C] bool, gamma=float)``: values <- V(states),
    rewards[:, -1] += gamma * V(states)[:, 0] where flags[:, -1].  With ``env=(rewards, terminations, truncations)`` and
    ``rows=(done_row, termination_row, truncation_row)`` the job also stores the env step's outputs into the buffer rows
    (``rewards`` is then the destination row; the flag is done, or truncation with ``flag_is_truncation``)."""
    lib = _lib.load()
    dev = _dev(params, states, eps, *[j["states"] for j in value_jobs])
    st = RolloutStep()
    st.params, st.tiles, st.layout = params.data_ptr(), tiles.data_ptr(), ctypes_pointer(layout)
    st.bf16 = int(tiles.dtype == torch.bfloat16)  # the tile image's element type selects the MFMA operand precision
    action = logprob = value = None
    if states is not None:
        s_ = _as_f32(states, "states")
        M = s_.shape[0]
        if s_.dim() != 2 or s_.shape[1] != layout.obs_dim:
            raise RlxError(f"states must be [M, {layout.obs_dim}], got {tuple(s_.shape)}")
        e = _as_f32(eps, "eps")
        if e is not None and tuple(e.shape) != (M, layout.act_dim):
            raise RlxError(f"eps must be [{M}, {layout.act_dim}]")
        if out is not None:
            action, logprob, value = out
            for t, w in ((action, layout.act_dim), (logprob, layout.act_dim), (value, layout.val_dim)):
                if t.dtype != torch.float32 or not t.is_contiguous() or t.numel() != M * w:
                    raise RlxError("mlp_rollout_step: out tensors must be contiguous float32 of the right size")
        else:
            action = torch.empty((M, layout.act_dim), dtype=torch.float32, device=dev)
            logprob = torch.empty_like(action)
            value = torch.empty((M, layout.val_dim), dtype=torch.float32, device=dev)
        if states_copy is not None and (states_copy.dtype != torch.float32 or not states_copy.is_contiguous()
                                        or states_copy.numel() != s_.numel()):
            raise RlxError("mlp_rollout_step: states_copy must be a contiguous float32 tensor shaped like states")
        st.states, st.eps, st.m = s_.data_ptr(), _ptr(e), M
        st.action, st.logprob, st.value, st.states_copy = action.data_ptr(), logprob.data_ptr(), value.data_ptr(), _ptr(states_copy)
    if len(value_jobs) > 2:
        raise RlxError("at most two value jobs per launch")
    keep = []
    for k, j in enumerate(value_jobs):
        js = _as_f32(j["states"], "value job states")
        m = js.shape[0]
        vals, rew, flags = j.get("values"), j.get("rewards"), j.get("flags")
        if vals is not None and (vals.dtype != torch.float32 or not vals.is_contiguous() or vals.numel() != m * layout.val_dim):
            raise RlxError("value job: values must be contiguous float32 [m, val_dim]")
        chunk = 1
        env = j.get("env")
        if rew is not None:
            if rew.dim() != 2 or rew.dtype != torch.float32 or not rew.is_contiguous() or rew.shape[0] != m:
                raise RlxError("value job: rewards must be a contiguous float32 [m, C] tensor")
            chunk = rew.shape[1]
            if env is None:
                flags = _as_u8(flags)
                if flags is None or tuple(flags.shape) != tuple(rew.shape):
                    raise RlxError("value job: flags must be a bool tensor shaped like rewards")
        keep += [js, flags]
        vj = st.value_jobs[k]
        vj.states, vj.m, vj.values, vj.rewards, vj.flags = js.data_ptr(), m, _ptr(vals), _ptr(rew), _ptr(flags)
        vj.chunk, vj.gamma = chunk, float(j.get("gamma", 1.0))
        if env is not None:
            if rew is None:
                raise RlxError("value job: env rows need the destination reward row")
            er, ete, etr = _as_f32(env[0], "env rewards"), _as_u8(env[1]), _as_u8(env[2])
            rows = [t if t.dtype == torch.uint8 else t.view(torch.uint8) for t in j["rows"]]
            for t in (er, ete, etr, *rows):
                if t.numel() != rew.numel() or not t.is_contiguous():
                    raise RlxError("value job: env tensors and rows must be contiguous with the reward row's number of elements")
            keep += [er, ete, etr, *rows]
            vj.env_rewards, vj.env_terminations, vj.env_truncations = er.data_ptr(), ete.data_ptr(), etr.data_ptr()
            vj.done_row, vj.termination_row, vj.truncation_row = (t.data_ptr() for t in rows)
            vj.flag_is_truncation = int(bool(j.get("flag_is_truncation", False)))
    st.n_value_jobs = len(value_jobs)
    with torch.cuda.device(dev):
        _lib.check(lib.rlx_mlp_rollout_step(byref(st), _stream_ptr(dev)), "rlx_mlp_rollout_step")
    return action, logprob, value


def ctypes_pointer(layout: MlpLayout):
    import ctypes
    return ctypes.pointer(layout)


def ppo_step_slabs(layout: MlpLayout, m: int, bf16: bool = False) -> int:
    """Split-K slabs `grads` must hold for a minibatch of m rows at the given operand precision."""
    return _lib.load().rlx_ppo_step_slabs_for(byref(layout), int(m), 1 if bf16 else 0)


def ppo_step_workspace_bytes(layout: MlpLayout, m: int) -> int:
    return _lib.load().rlx_ppo_step_workspace_bytes(byref(layout), int(m))


def _ppo_step_args(params, layout, loss, mbatch, grads, out_row, workspace, grad_out, tiles, bf16, decoupled) -> PpoStepArgs:
    a = PpoStepArgs()
    a.params, a.layout, a.loss = params.data_ptr(), ctypes_pointer(layout), ctypes_pointer(loss)
    st = mbatch["states"]
    a.states, a.action = st.data_ptr(), mbatch["action"].data_ptr()
    a.old_logprobs, a.advantages = mbatch["prev_logprobs"].data_ptr(), mbatch["advantages"].data_ptr()
    has_critic = bool(loss.has_critic)
    a.prev_values = mbatch["prev_values"].data_ptr() if has_critic else None
    a.returns = mbatch["returns"].data_ptr() if has_critic else None
    lm = mbatch.get("loss_mask")
    a.loss_mask = None if lm is None else lm.data_ptr()
    a.loss_mask_sum = _ptr(mbatch.get("loss_mask_sum"))
    a.m, a.grad_out = st.shape[0], float(grad_out)
    a.grads, a.slabs, a.out = grads.data_ptr(), grads.shape[0], out_row.data_ptr()
    a.workspace, a.workspace_bytes = workspace.data_ptr(), workspace.numel()
    a.tiles, a.bf16 = _ptr(tiles), int(bool(bf16))
    if tiles is not None and (tiles.dtype == torch.bfloat16) != bool(bf16):
        raise RlxError("ppo_step: the tile image's dtype does not match the requested precision")
    if decoupled is not None:  # dict(params=DecoupledLossParams, proximal_logprobs, versions, current_version_dev): see decoupled_step_args
        a.decoupled = ctypes.addressof(decoupled["params"])
        a.proximal_logprobs, a.versions = _ptr(decoupled.get("proximal_logprobs")), _ptr(decoupled.get("versions"))
        a.current_version_dev = _ptr(decoupled.get("current_version_dev"))
    return a


def decoupled_step_args(loss: PpoLossParams, mbatch: dict, *, current_version, behave_weight_threshold,
                        current_version_dev: Optional[torch.Tensor] = None) -> dict:
    """``decoupled`` argument of ppo_step / PreparedPpoStep: which proximal policy applies (losses.py:68-89) from what the
    micro-batch carries -- ``proximal_logprobs`` (given), else ``versions`` (+ a current version: interpolated), else the
    behaviour policy itself.  ``current_version_dev`` (one device float) is read when the launch executes."""
    dp, prox, ver = _decoupled_params(loss, mbatch["prev_logprobs"], mbatch.get("proximal_logprobs"), mbatch.get("versions"),
                                      current_version, behave_weight_threshold)
    for name, t, src in (("proximal_logprobs", prox, mbatch.get("proximal_logprobs")), ("versions", ver, mbatch.get("versions"))):
        if t is not None and t.data_ptr() != src.data_ptr():
            raise RlxError(f"decoupled ppo_step needs contiguous float32 {name}")
    return dict(params=dp, proximal_logprobs=prox, versions=ver, current_version_dev=current_version_dev)


def ppo_step(params: torch.Tensor, layout: MlpLayout, loss: PpoLossParams, mbatch: dict, grads: torch.Tensor,
             out_row: torch.Tensor, workspace: torch.Tensor, grad_out: float = 1.0, tiles: Optional[torch.Tensor] = None,
             bf16: bool = False, decoupled: Optional[dict] = None):
    """forward + loss + backward of one micro-batch (two launches).  ``mbatch``: states, action, prev_logprobs,
    advantages [, prev_values, returns, loss_mask, loss_mask_sum] as flattened minibatch views; ``grads`` [slabs, n].
    ``decoupled`` (decoupled_step_args): the asynchronous-PPO actor loss; the actor network's gradients then leave in sum
    form (include/rlx.h) -- pass ``deferred_actor_scale`` of the row(s) to the AdamW step."""
    dev = params.device
    a = _ppo_step_args(params, layout, loss, mbatch, grads, out_row, workspace, grad_out, tiles, bf16, decoupled)
    with torch.cuda.device(dev):
        _lib.check(_lib.load().rlx_ppo_step(byref(a), _stream_ptr(dev)), "rlx_ppo_step")


# --------------------------------------------------------------------------------------------
# prepared launches: marshal once, replay with one ctypes call (the update loop issues ~500 launches per iteration;
# building the argument structs in Python every time costs more host time than the kernels take on the GPU)
# --------------------------------------------------------------------------------------------
class PreparedPpoStep:
    """rlx_ppo_step with every pointer fixed (persistent minibatch views and workspaces)."""

    def __init__(self, params, layout, loss, mbatch, grads, out_row, workspace, grad_out=1.0, tiles=None, bf16=False,
                 decoupled=None):
        self._lib = _lib.load()
        self._keep = (params, layout, loss, dict(mbatch), grads, out_row, workspace, tiles, decoupled)
        self._args = _ppo_step_args(params, layout, loss, mbatch, grads, out_row, workspace, grad_out, tiles, bf16, decoupled)
        self._ref = byref(self._args)

    def __call__(self, stream: int):
        rc = self._lib.rlx_ppo_step(self._ref, stream)
        if rc:
            _lib.check(rc, "rlx_ppo_step")


class PreparedAdamw:
    """rlx_clip_adamw_step with every pointer fixed; ``stats`` is this step's (norm, applied) row."""

    def __init__(self, params, grads, exp_avg, exp_avg_sq, groups, *, betas, eps, weight_decay, max_grad_norm, grad_scale,
                 stats, step_state, workspace, tile_layout=None, tiles=None, xgmi=None, grad_flat=None, deferred=None,
                 deferred_in_caller=False, sync=None):
        """``xgmi`` (scheduler.xgmi.XgmiAllReduce) + ``grad_flat`` [n]: rlx_xgmi_clip_adamw_step instead -- ``grads`` are this
        rank's slabs, the reduced gradient (scaled by grad_scale = 1 / world_size) lands in grad_flat, then clip + AdamW."""
        self._lib = _lib.load()
        n = params.numel()
        p = AdamwParams()
        p.beta1, p.beta2, p.eps, p.weight_decay = float(betas[0]), float(betas[1]), float(eps), float(weight_decay)
        p.max_grad_norm = float(max_grad_norm) if max_grad_norm else 0.0
        p.step, p.n_groups, p.grad_partials, p.grad_scale = 0, len(groups), grads.numel() // n, float(grad_scale)
        for k, (b, e, lr) in enumerate(groups):
            p.groups[k] = AdamwGroup(int(b), int(e), float(lr))
        if tile_layout is not None and tiles is not None:
            p.tile_layout, p.tiles, p.tiles_bf16 = ctypes_pointer(tile_layout), tiles.data_ptr(), int(tiles.dtype == torch.bfloat16)
        _set_deferred(p, deferred)
        self.deferred = deferred  # (a data-parallel caller that collapses the slabs itself -- RCCL path -- applies it there)
        if deferred is not None and deferred_in_caller:
            p.deferred_scale = None  # the caller sums (and scales: sum_slabs(deferred=...)) the slabs in front of its all-reduce
        # one GPU: slab sum + norm + clip + AdamW as one launch.  Over xGMI: the exchange INSIDE that launch, when the communicator's
        # start-up validation has passed it (XgmiAllReduce.one_launch); otherwise the exchange's launch chain with its hand-shakes
        if sync is not None and (xgmi is None or getattr(xgmi, "one_launch", False)):
            p.sync_words = sync.data_ptr()
        self._keep = (params, grads, exp_avg, exp_avg_sq, stats, step_state, workspace, tile_layout, tiles, p, xgmi, grad_flat, deferred, sync)
        if xgmi is not None:
            self._fn, self._name = self._lib.rlx_xgmi_clip_adamw_step, "rlx_xgmi_clip_adamw_step"
            self._argv = (xgmi.handle, params.data_ptr(), grads.data_ptr(), grad_flat.data_ptr(), exp_avg.data_ptr(),
                          exp_avg_sq.data_ptr(), n, byref(p), stats.data_ptr(), step_state.data_ptr(), workspace.data_ptr(),
                          workspace.numel())
        else:
            self._fn, self._name = self._lib.rlx_clip_adamw_step, "rlx_clip_adamw_step"
            self._argv = (params.data_ptr(), grads.data_ptr(), exp_avg.data_ptr(), exp_avg_sq.data_ptr(), n, byref(p),
                          stats.data_ptr(), step_state.data_ptr(), workspace.data_ptr(), workspace.numel())

    def __call__(self, stream: int):
        rc = self._fn(*self._argv, stream)
        if rc:
            _lib.check(rc, self._name)

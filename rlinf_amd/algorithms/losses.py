"""Built-in policy losses behind the registry names "actor_critic" and "actor"
(mirror of rlinf/algorithms/losses.py:170-380,397-425,513-535), served by ppo_loss.hip.

``policy_loss(**kwargs)`` in the reference runs preprocess_loss_inputs, the loss, and then ``.item()``
on every metric (15 device syncs per micro-batch).  Here the embodied entry takes the RAW kwargs the
learner builds (rlinf/workers/actor/embodied_fsdp_actor_worker.py:640-662), the kernel fuses the
shaping, and metrics come back as ONE lazily-read device vector: ``LossMetrics`` behaves like the
reference's dict (same keys, float values) but performs a single D2H copy, on first read.
"""

from __future__ import annotations

from collections.abc import Mapping
from typing import Optional

import torch

from .. import ops, token_ops
from .._lib import DPPO_OUT_NAMES, PPO_OUT_NAMES, TOK_OUT_NAMES
from . import utils as _u
from .registry import _mark_native_loss, register_policy_loss

# hidden explained-variance keys, named as in rlinf/utils/metric_utils.py (popped by the learner)
EV_PREFIX = "__sum__/_critic_explained_variance/"
_EV_MAP = {"ev/count": EV_PREFIX + "count", "ev/returns_sum": EV_PREFIX + "returns_sum",
           "ev/returns_sq_sum": EV_PREFIX + "returns_sq_sum", "ev/errors_sum": EV_PREFIX + "errors_sum",
           "ev/errors_sq_sum": EV_PREFIX + "errors_sq_sum"}
_ACTOR_KEYS = ("actor/policy_loss", "actor/policy_loss_abs", "actor/ratio", "actor/ratio_abs", "actor/clipped_ratio",
               "actor/dual_cliped_ratio", "actor/approx_kl", "actor/clip_fraction")
_CRITIC_KEYS = ("critic/value_loss", "critic/value_clip_ratio")


class LossMetrics(Mapping):
    """Read-only metrics dict over the kernel's f32[20] output; one D2H copy on first access."""

    def __init__(self, out: torch.Tensor, has_critic: bool, actor_keys=_ACTOR_KEYS, names=PPO_OUT_NAMES):
        self.device_vector = out
        self._keys = list(actor_keys) + (list(_CRITIC_KEYS) + list(_EV_MAP.values()) if has_critic else [])
        self._names = names
        self._host = None
        self._extra = {}

    def _load(self):
        if self._host is None:
            self._host = self.device_vector.tolist()
        return self._host

    def __getitem__(self, key):
        if key in self._extra:
            return self._extra[key]
        if key not in self._keys:
            raise KeyError(key)
        host = self._load()
        inv = {v: k for k, v in _EV_MAP.items()}
        return host[self._names[inv.get(key, key)]]

    def __setitem__(self, key, value):  # the learner adds actor/entropy_loss, actor/total_loss
        self._extra[key] = value

    def pop(self, key, default=None):
        if key in self._extra:
            return self._extra.pop(key)
        if key in self._keys:
            val = self[key]
            self._keys.remove(key)
            return val
        return default

    def __iter__(self):
        yield from self._keys
        yield from self._extra

    def __len__(self):
        return len(self._keys) + len(self._extra)


def _fused(kw: dict, has_critic: bool):
    logprobs = kw["logprobs"]
    dev = _u.compute_device(logprobs)
    st = lambda name: _u.stage(kw.get(name), dev)  # noqa: E731
    logprob_type = kw.get("logprob_type") or "action_level"
    action_dim = kw.get("single_action_dim")
    if action_dim is None:
        raise TypeError("policy_loss(task_type='embodied') needs single_action_dim")
    loss, out = ops.ppo_loss(
        _u.stage(logprobs, dev), st("old_logprobs"), st("advantages"), logprob_type=logprob_type,
        reward_type=kw.get("reward_type"),
        action_dim=int(action_dim), clip_ratio_low=kw["clip_ratio_low"], clip_ratio_high=kw["clip_ratio_high"],
        values=st("values") if has_critic else None, prev_values=st("prev_values") if has_critic else None,
        returns=st("returns") if has_critic else None, value_clip=kw.get("value_clip"),
        huber_delta=kw.get("huber_delta"), loss_mask=st("loss_mask"), loss_mask_sum=st("loss_mask_sum"),
        max_episode_steps=kw.get("max_episode_steps"), clip_ratio_c=kw.get("clip_ratio_c"),
        clip_log_ratio_min=kw.get("clip_log_ratio_min"), clip_log_ratio_max=kw.get("clip_log_ratio_max"),
        critic_warmup=bool(kw.get("critic_warmup", False)), has_critic=has_critic)
    return loss, LossMetrics(out, has_critic)


# ---- reasoning (token) tier ---------------------------------------------------------------------------------
_TOKEN_KEYS = _ACTOR_KEYS
# the reference's zero-loss-mask fast path returns a different key set (losses.py:206-219)
_TOKEN_KEYS_ZERO = ("actor/token_num", "actor/policy_loss", "actor/policy_loss_mbs_mean", "actor/policy_loss_abs",
                    "actor/ratio", "actor/clipped_ratio", "actor/dual_cliped_ratio", "actor/approx_kl",
                    "actor/clip_fraction")
_TOKEN_FUSED_KEYS = ("actor/final_loss", "actor/entropy_loss", "actor/kl_loss")


class TokenLossMetrics(Mapping):
    """Metrics of the fused token loss over the kernel's f32[16] output; one D2H copy on first read.  The key set
    follows the reference: the fast-path keys when the first sequence's mask was empty, the usual eight otherwise,
    plus final/entropy/kl when those terms were fused in."""

    def __init__(self, out: torch.Tensor, fused_terms: bool):
        self.device_vector = out
        self._fused = fused_terms
        self._host = None
        self._extra = {}

    def _load(self):
        if self._host is None:
            self._host = self.device_vector.tolist()
        return self._host

    def _kernel_keys(self):
        on = self._load()[TOK_OUT_NAMES["policy_on"]] != 0.0
        return (_TOKEN_KEYS if on else _TOKEN_KEYS_ZERO) + (_TOKEN_FUSED_KEYS if self._fused else ())

    def __getitem__(self, key):
        if key in self._extra:
            return self._extra[key]
        if key not in self._kernel_keys():
            raise KeyError(key)
        host = self._load()
        if key == "actor/final_loss":
            return host[TOK_OUT_NAMES["loss"]]
        if host[TOK_OUT_NAMES["policy_on"]] == 0.0 and key not in _TOKEN_FUSED_KEYS:
            return 0.0
        return host[TOK_OUT_NAMES[key]]

    def __setitem__(self, key, value):
        self._extra[key] = value

    def update(self, other):  # the learner adds actor/final_loss etc. (fsdp_actor_worker.py:773-779)
        self._extra.update(other)

    def __iter__(self):
        for k in self._kernel_keys():
            if k not in self._extra:
                yield k
        yield from self._extra

    def __len__(self):
        return len(set(self._kernel_keys()) | set(self._extra))


def _token_fused(kw: dict):
    """compute_ppo_actor_loss on [bsz, seq] token tensors with a caller-chosen aggregation (losses.py:170-312), plus --
    when the caller hands them over -- the entropy bonus and KL penalty the reasoning learner adds right after
    (fsdp_actor_worker.py:750-766)."""
    agg = kw.get("loss_agg_func")
    # our own aggregation helpers carry a tag; the reference's (rlinf.utils.utils) are recognised by name, which is how
    # the RLINF_EXT_MODULE route sees them
    by_name = {"masked_mean": "token-mean", "seq_mean_token_sum": "seq-mean-token-sum",
               "seq_mean_token_mean": "seq-mean-token-mean"}
    agg_name = (kw.get("loss_agg") or getattr(agg, "rlx_agg", None) or by_name.get(getattr(agg, "__name__", None))
                or ("token-mean" if agg is None else None))
    if agg_name is None:
        raise NotImplementedError("policy_loss(task_type='reasoning'): loss_agg_func must come from "
                                  "rlinf_amd.utils.utils.get_loss_agg_func (or pass loss_agg='token-mean'|...)")
    if kw.get("max_episode_steps") is not None and kw.get("loss_mask_sum") is not None:
        raise NotImplementedError("masked_mean_ratio is an embodied aggregation; not used by the token path")
    logprobs = kw["logprobs"]
    dev = _u.compute_device(logprobs)
    st = lambda name: _u.stage(kw.get(name), dev)  # noqa: E731
    entropy, ref = st("entropy"), st("ref_logprobs")
    kl_beta = float(kw.get("kl_beta") or 0.0)
    fused_terms = entropy is not None or (ref is not None and kl_beta > 0)
    params = token_ops.make_token_loss_params(
        loss_agg=agg_name, clip_ratio_low=kw["clip_ratio_low"], clip_ratio_high=kw["clip_ratio_high"],
        clip_ratio_c=kw.get("clip_ratio_c"), clip_log_ratio_min=kw.get("clip_log_ratio_min"),
        clip_log_ratio_max=kw.get("clip_log_ratio_max"), critic_warmup=bool(kw.get("critic_warmup", False)),
        fast_path_zero_loss_mask=bool(kw.get("fast_path_zero_loss_mask", False)),
        kl_penalty_type=kw.get("kl_penalty_type", "low_var_kl") if ref is not None and kl_beta > 0 else None,
        kl_beta=kl_beta if ref is not None else 0.0, use_entropy=entropy is not None,
        entropy_bonus=float(kw.get("entropy_bonus") or 0.0))
    lp = _u.stage(logprobs, dev)
    shape2d = lp.shape if lp.dim() == 2 else (1, lp.numel())
    as2d = lambda t: None if t is None else t.reshape(shape2d)  # noqa: E731
    loss, out = token_ops.token_loss(as2d(lp), as2d(st("old_logprobs")), as2d(st("advantages")), as2d(st("loss_mask")),
                                     params, entropy=as2d(entropy), ref_logprobs=as2d(ref))
    return loss, TokenLossMetrics(out, fused_terms)


_DECOUPLED_KEYS = ("actor/policy_loss", "actor/proximal_ratio", "actor/clipped_proximal_ratio", "actor/clip_fraction",
                   "actor/dual_clip_fraction", "actor/behav_clip_fraction", "actor/proximal_approx_kl",
                   "actor/behav_approx_kl")


class DecoupledLossMetrics(LossMetrics):
    """Adds the two version metrics the reference reports only when versions and loss_mask share a shape and some
    element is unmasked (losses.py:156-165)."""

    def __init__(self, out, has_critic, current_version, versions_match_mask: bool):
        super().__init__(out, has_critic, actor_keys=_DECOUPLED_KEYS, names=DPPO_OUT_NAMES)
        self._current_version = current_version
        self._versions_match = versions_match_mask
        self._versions_resolved = False

    def _resolve(self):
        if not self._versions_resolved:
            self._versions_resolved = True
            if self._versions_match and self._load()[DPPO_OUT_NAMES["mask_count"]] > 0:
                self._keys += ["actor/average_version", "actor/current_version"]

    def __getitem__(self, key):
        self._resolve()
        if key == "actor/current_version" and key in self._keys:
            return float(self._current_version)
        return super().__getitem__(key)

    def __iter__(self):
        self._resolve()
        return super().__iter__()

    def __len__(self):
        self._resolve()
        return super().__len__()


@register_policy_loss("decoupled_actor_critic")
def compute_decoupled_ppo_actor_critic_loss(**kwargs) -> tuple[torch.Tensor, Mapping]:
    """Decoupled (async) PPO: clip against the proximal policy, importance-weight by exp(prox - behaviour)
    (losses.py:27-167,383-393) on RAW per-dimension inputs; the proximal policy is ``proximal_logprobs`` when given,
    else interpolated from ``versions`` / ``current_version``, else the behaviour policy."""
    kw = kwargs
    logprobs = kw["logprobs"]
    dev = _u.compute_device(logprobs)
    st = lambda name: _u.stage(kw.get(name), dev)  # noqa: E731
    logprob_type = kw.get("logprob_type") or "action_level"
    action_dim = kw.get("single_action_dim")
    if action_dim is None:
        raise TypeError("policy_loss(loss_type='decoupled_actor_critic') needs single_action_dim")
    versions, current_version = st("versions"), kw.get("current_version")
    loss, out = ops.ppo_loss(
        _u.stage(logprobs, dev), st("old_logprobs"), st("advantages"), logprob_type=logprob_type,
        reward_type=kw.get("reward_type"),
        action_dim=int(action_dim), clip_ratio_low=kw["clip_ratio_low"], clip_ratio_high=kw["clip_ratio_high"],
        values=st("values"), prev_values=st("prev_values"), returns=st("returns"), value_clip=kw.get("value_clip"),
        huber_delta=kw.get("huber_delta"), loss_mask=st("loss_mask"), loss_mask_sum=st("loss_mask_sum"),
        max_episode_steps=kw.get("max_episode_steps"), clip_ratio_c=kw.get("clip_ratio_c"),
        critic_warmup=bool(kw.get("critic_warmup", False)), has_critic=True,
        decoupled=dict(proximal_logprobs=st("proximal_logprobs"), versions=versions, current_version=current_version,
                       behave_weight_threshold=kw.get("behave_weight_threshold")))
    # versions.shape == loss_mask.shape after the reference's shaping <=> one loss element per advantage element
    match = (versions is not None and current_version is not None
             and (logprob_type != "token_level" or kw.get("loss_mask") is None))
    return loss, DecoupledLossMetrics(out, True, current_version, match)


@register_policy_loss("actor_critic")
def compute_ppo_actor_critic_loss(**kwargs) -> tuple[torch.Tensor, Mapping]:
    """PPO clipped surrogate + clipped-Huber value loss (losses.py:397-425) on RAW per-dimension inputs."""
    return _fused(kwargs, has_critic=True)


@register_policy_loss("actor")
def compute_grpo_actor_loss_fn(**kwargs) -> tuple[torch.Tensor, Mapping]:
    """Actor-only clipped surrogate used by GRPO (losses.py:513-535): the embodied per-dimension kernel, or the
    token kernel for reasoning batches."""
    if kwargs.get("task_type", "embodied") != "embodied" or kwargs.get("single_action_dim") is None:
        return _token_fused(kwargs)
    return _fused(kwargs, has_critic=False)


_mark_native_loss("actor_critic", compute_ppo_actor_critic_loss)
_mark_native_loss("actor", compute_grpo_actor_loss_fn)
_mark_native_loss("decoupled_actor_critic", compute_decoupled_ppo_actor_critic_loss)


def explained_variance_from_stats(stats: Mapping) -> float:
    """metric_utils.py:261-290 on summed sufficient statistics (host scalars)."""
    n = float(stats[EV_PREFIX + "count"])
    if n < 2:
        return float("nan")
    f32 = lambda x: torch.tensor(float(x), dtype=torch.float32)  # noqa: E731
    rs, rss = f32(stats[EV_PREFIX + "returns_sum"]), f32(stats[EV_PREFIX + "returns_sq_sum"])
    es, ess = f32(stats[EV_PREFIX + "errors_sum"]), f32(stats[EV_PREFIX + "errors_sq_sum"])
    rc = rss - rs * rs / n
    if torch.isnan(rc) or rc == 0:
        return float("nan")
    ec = ess - es * es / n
    if torch.isnan(ec):
        return float("nan")
    return float(1 - ec / rc)

"""Plugin boundary: same names, argument meaning and error behaviour as rlinf/algorithms/registry.py:30-124.

``register_advantage`` / ``register_policy_loss`` keep the reference's "last registration wins,
names are lower-cased" rule, ``get_*`` raise ValueError for unknown names, and the two unified
entries accept the reference's keyword dictionaries.  The built-in names ("gae", "grpo",
"actor_critic", "actor") are served by hand-written HIP kernels: for them the embodied
``[n_chunk, B, C]`` buffers go to the kernels as they are (the kernels index the time-chunk layout
directly), so the reference's transpose/reshape pre- and post-processing disappears.  A user-registered
function still gets the reference's flattened ``[T, B]`` views.
"""

from __future__ import annotations

from functools import wraps
from typing import Callable, Optional

import torch

from . import utils as _u

ADV_REGISTRY: dict[str, Callable] = {}
LOSS_REGISTRY: dict[str, Callable] = {}
# names whose registered callee is one of ours -> the native [n,B,C] fast path may be used
_NATIVE_ADV: dict[str, Callable] = {}
_NATIVE_ADV_REASONING: dict[str, Callable] = {}
_NATIVE_LOSS: dict[str, Callable] = {}


def register_advantage(name: str):
    """Decorator storing an advantage/returns function under ``name.lower()`` (registry.py:33-44)."""

    def decorator(fn):
        @wraps(fn)
        def wrapper(*args, **kwargs):
            return fn(*args, **kwargs)

        ADV_REGISTRY[name.lower()] = wrapper
        _NATIVE_ADV.pop(name.lower(), None)
        _NATIVE_ADV_REASONING.pop(name.lower(), None)
        return wrapper

    return decorator


def get_adv_and_returns(name: str) -> Callable:
    if name.lower() not in ADV_REGISTRY:
        raise ValueError(f"Advantage '{name}' not registered. Available: {list(ADV_REGISTRY.keys())}")
    return ADV_REGISTRY[name.lower()]


def register_policy_loss(name: str):
    def decorator(fn):
        @wraps(fn)
        def wrapper(*args, **kwargs):
            return fn(*args, **kwargs)

        LOSS_REGISTRY[name.lower()] = wrapper
        _NATIVE_LOSS.pop(name.lower(), None)
        return wrapper

    return decorator


def get_policy_loss(name: str):
    if name not in LOSS_REGISTRY:  # the reference does not lower-case on lookup (registry.py:71-74)
        raise ValueError(f"Loss {name} not registered")
    return LOSS_REGISTRY[name]


def _mark_native_adv(name: str, native: Callable, task_type: str = "embodied"):
    (_NATIVE_ADV if task_type == "embodied" else _NATIVE_ADV_REASONING)[name] = native


def _mark_native_loss(name: str, native: Callable):
    _NATIVE_LOSS[name] = native


def policy_loss(**kwargs) -> tuple[torch.Tensor, dict]:
    """Unified actor loss entry (registry.py:77-92).  Requires ``loss_type`` and ``task_type``."""
    loss_type = kwargs["loss_type"]
    loss_fn = get_policy_loss(loss_type)
    task_type = kwargs["task_type"]
    native = _NATIVE_LOSS.get(loss_type)
    if task_type == "embodied" and native is not None:
        return native(**kwargs)  # raw per-dimension inputs: preprocess_loss_inputs is fused into the kernel
    if task_type == "embodied":
        kwargs = _u.preprocess_loss_inputs(**kwargs)
    loss, metrics_data = loss_fn(**kwargs)
    if task_type == "embodied":
        metrics_data = _u.postprocess_loss_metric(metrics_data)
    return loss, metrics_data


def calculate_adv_and_returns(**kwargs):
    """Unified advantage + return entry (registry.py:95-124): embodied -> dict, reasoning -> tuple."""
    adv_type = kwargs["adv_type"]
    fn = get_adv_and_returns(adv_type)
    task_type = kwargs["task_type"]
    if task_type == "embodied":
        native = _NATIVE_ADV.get(adv_type.lower())
        if native is not None:
            return native(**kwargs)
        if adv_type == "opd":
            advantages, returns = fn(**kwargs)
            res = {"advantages": advantages}
            if returns is not None:
                res["returns"] = returns
            return res
        kwargs = _u.preprocess_embodied_advantages_inputs(**kwargs)
        if adv_type not in ("gae", "grpo_video"):
            kwargs = _u.calculate_scores(**kwargs)
        advantages, returns = fn(**kwargs)
        return _u.postprocess_embodied_advantages_outputs(advantages=advantages, returns=returns, **kwargs)
    native = _NATIVE_ADV_REASONING.get(adv_type.lower())
    if native is not None:
        out = native(**kwargs)  # reads the [bsz, seq] layout directly: no transposes, no copies
        if out is not None:
            return out
    kwargs = _u.preprocess_reasoning_advantages_inputs(**kwargs)
    advantages, returns = fn(**kwargs)
    return _u.postprocess_reasoning_advantages_outputs(advantages, returns)

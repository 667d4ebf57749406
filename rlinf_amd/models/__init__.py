"""Model registry boundary (mirror of rlinf/models/__init__.py:31-53,337-355): register_model / get_model.

Only the MLP policy of the hot path is built in; other model types are out of scope for this tier.
"""

from __future__ import annotations

from typing import Callable, Optional

import torch

ModelBuilder = Callable[[object, Optional[object]], object]
_MODEL_REGISTRY: dict[str, ModelBuilder] = {}


def torch_dtype_from_precision(precision) -> Optional[torch.dtype]:
    """rlinf/config.py:171: '32'/'fp32' -> float32, 'bf16' -> bfloat16, '16'/'fp16' -> float16."""
    p = str(precision).lower()
    if p in ("32", "fp32", "float32"):
        return torch.float32
    if p in ("bf16", "bfloat16", "bf16-mixed"):
        return torch.bfloat16
    if p in ("16", "fp16", "float16", "16-mixed"):
        return torch.float16
    if p in ("none", "null"):
        return None
    raise ValueError(f"Unsupported precision: {precision}")


def register_model(model_type: str, model_builder: ModelBuilder, category: str = "embodied", force: bool = False):
    if not model_type:
        raise ValueError("model_type must be a non-empty string.")
    if not callable(model_builder):
        raise TypeError("model_builder must be callable.")
    if not force and model_type in _MODEL_REGISTRY:
        raise ValueError(f"Model type `{model_type}` is already registered. Set force=True to override it.")
    _MODEL_REGISTRY[model_type] = model_builder


def _cfg_get(cfg, key, default=None):
    if hasattr(cfg, "get"):
        return cfg.get(key, default)
    return getattr(cfg, key, default)


def _build_mlp_policy(cfg, torch_dtype):
    from .embodiment.mlp_policy import MLPPolicy

    return MLPPolicy(
        obs_dim=int(_cfg_get(cfg, "obs_dim")), action_dim=int(_cfg_get(cfg, "action_dim")),
        num_action_chunks=int(_cfg_get(cfg, "num_action_chunks", 1)),
        add_value_head=bool(_cfg_get(cfg, "add_value_head", True)), add_q_head=bool(_cfg_get(cfg, "add_q_head", False)),
        compute_dtype=torch.bfloat16 if torch_dtype == torch.bfloat16 else torch.float32)


register_model("mlp_policy", _build_mlp_policy, force=True)


def get_model(cfg):
    builder = _MODEL_REGISTRY.get(str(_cfg_get(cfg, "model_type")))
    if builder is None:
        return None
    model = builder(cfg, torch_dtype_from_precision(_cfg_get(cfg, "precision", "32")))
    if torch.cuda.is_available() and _cfg_get(cfg, "load_to_device", True) and hasattr(model, "to"):
        model = model.to(torch.device("cuda", torch.cuda.current_device()))
    return model

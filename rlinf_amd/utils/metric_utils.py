"""compute_evaluate_metrics (rlinf/utils/metric_utils.py:372-419): the per-episode records every env process returned from
``evaluate`` -> their means + the trajectory count.  Host arithmetic on a few hundred floats, once per validation."""

from __future__ import annotations

import torch


def count_trajectories(metrics_dict: dict) -> int:
    if not metrics_dict:
        return 0
    first = next(iter(metrics_dict.values()))
    if isinstance(first, torch.Tensor):
        return int(first.shape[0])
    return sum(int(t.shape[0]) if isinstance(t, torch.Tensor) else len(t) for t in first)


def compute_evaluate_metrics(eval_metrics_list: list) -> dict:
    if not eval_metrics_list:
        return {}
    keys: list = []
    for m in eval_metrics_list:
        keys.extend(k for k in m if k not in keys)
    out = {}
    for key in keys:
        shards = [torch.as_tensor(m[key]).reshape(-1).float() for m in eval_metrics_list if key in m]
        stacked = torch.cat(shards) if shards else torch.zeros(0)
        out[key] = float(stacked.mean()) if stacked.numel() > 0 else 0.0
    out["num_trajectories"] = sum(count_trajectories(m) for m in eval_metrics_list)
    return out

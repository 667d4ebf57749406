"""CPU restatement of the reference's bucketed weight sync.  TEST INFRASTRUCTURE ONLY.

SURVEY.md 8f item 3: rlinf/hybrid_engines/weight_syncer/bucket_syncer.py -- ``iter_named_tensor_buckets`` (:33-127: the
bucket plan, the transport dtype per tensor, the metadata the first bucket carries), ``BucketWeightSyncer._bucket_key``
(:166-187), ``_transport_dtype`` (:189-203), ``iter_buckets`` (:205-243) and the receiver loop ``apply`` (:282-339,
``load_state_dict(bucket, strict=False)`` per bucket).  Byte work: compared bit for bit (NaN payloads aside, which
depend on the backend's f32 -> bf16 routine) -- against the real reference classes in tests/test_oracle_vs_reference.py
and against tests/golden/weight_bucket.pt everywhere.

Nothing under ``rlinf_amd/`` imports this file.
"""

from __future__ import annotations

import torch

TOTAL, VERSION = "total_buckets", "syncer_version"


def bucket_key(key: str, has_visual: bool, is_agent: bool):
    if "_extra_state" in key:
        return None
    if has_visual and is_agent and key.startswith("model.language_model."):
        return "model." + key[len("model.language_model."):]
    return key


def transport_dtype(dtype: torch.dtype, bucket_dtype):
    return bucket_dtype if (bucket_dtype is not None and dtype.is_floating_point) else dtype


def make_buckets(state: dict, names: list, version: int, bucket_size: int, bucket_dtype=None, is_agent: bool = False) -> list:
    """-> the list of dicts BucketWeightSyncer.sync would send (CPU tensors)."""
    present = set(names)
    has_visual = any("visual." in k for k in present if k in state)
    plan, cur, held = [], [], 0
    for name in names:
        value = state.get(name)
        if value is None:
            continue
        key = bucket_key(name, has_visual, is_agent)
        if key is None:
            continue
        if key in (TOTAL, VERSION):
            raise ValueError(f"Bucket payload key conflicts with metadata key: {key}")
        tdt = transport_dtype(value.dtype, bucket_dtype)
        cur.append((key, value, tdt))
        held += value.numel() * torch.empty((), dtype=tdt).element_size()
        if held >= bucket_size:  # closed AFTER the tensor that crosses the threshold: a tensor is never split
            plan.append(cur)
            cur, held = [], 0
    if held > 0:
        plan.append(cur)
    if not plan:
        raise ValueError("No parameters to sync")
    out = []
    for k, items in enumerate(plan):
        b = {}
        if k == 0:
            b[TOTAL] = torch.tensor(len(plan), dtype=torch.int32)
            b[VERSION] = torch.as_tensor(version, dtype=torch.int32)
        for key, value, tdt in items:
            b[key] = value.detach().to(dtype=tdt)
        out.append(b)
    return out


def apply_buckets(state: dict, buckets: list) -> int:
    """The receiver: copy every received tensor the target knows into it (dtype of the target), ignore the others, raise on
    a shape mismatch -- load_state_dict(strict=False).  -> the version carried by the first bucket."""
    first = dict(buckets[0])
    total = int(first.pop(TOTAL))
    version = int(first.pop(VERSION))
    assert total == len(buckets)
    for b in [first] + [dict(x) for x in buckets[1:]]:
        for key, value in b.items():
            if key not in state:
                continue
            if state[key].shape != value.shape:
                raise RuntimeError(f"size mismatch for {key}")
            state[key].copy_(value)
    return version

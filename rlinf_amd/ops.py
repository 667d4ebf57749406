"""Tensor-level launchers over the C ABI (include/rlx.h).  torch is plumbing only: device memory,
the current HIP stream and dtype/shape checks.  Every function requires HIP tensors and raises
otherwise -- there is no eager/CPU path.
"""

from __future__ import annotations

from ctypes import byref
from typing import Optional

import torch

from . import _lib
from ._lib import GaeParams, RlxError


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
             variant: int = 0):
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

"""HIP streams restricted to a subset of the compute units (hipExtStreamCreateWithCUMask), wrapped for torch.

Pipeline mode (runner.use_training_pipeline, rollout_epoch > 1) puts the rollout of epoch e + 1 on a stream of its own next to the
training on epoch e.  Both are chains of short, latency-bound launches that want whole CUs (LDS and VGPR footprints that do not
co-reside), so WHERE the second stream's workgroups land decides whether the overlap hides the rollout or merely delays the
training launches.  ``RLX_ROLLOUT_CUS=n`` (or runner.rollout_cus) gives the rollout stream n CUs, spread evenly over the device's
enumeration order; the measurement that decides the default is tools/pipeline_overlap_probe.py."""

from __future__ import annotations

import ctypes
from typing import Optional

import torch

_hip = None


def _libhip():
    global _hip
    if _hip is None:
        for name in ("libamdhip64.so", "libamdhip64.so.7", "libamdhip64.so.6"):
            try:
                _hip = ctypes.CDLL(name)
                break
            except OSError:
                continue
        if _hip is None:
            raise OSError("libamdhip64.so not found")
        _hip.hipExtStreamCreateWithCUMask.restype = ctypes.c_int
        _hip.hipExtStreamCreateWithCUMask.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint32)]
        _hip.hipStreamDestroy.restype = ctypes.c_int
        _hip.hipStreamDestroy.argtypes = [ctypes.c_void_p]
    return _hip


def cu_mask_words(n_cus: int, total_cus: int, offset: int = 0) -> list:
    """Bit i of the mask enables CU i of the device's enumeration; n_cus of total_cus, evenly spread (every XCD / shader engine
    keeps a share whatever the enumeration order is), starting at ``offset``."""
    n_cus = max(1, min(int(n_cus), int(total_cus)))
    words = [0] * ((total_cus + 31) // 32)
    for k in range(n_cus):
        i = (offset + (k * total_cus) // n_cus) % total_cus
        words[i >> 5] |= 1 << (i & 31)
    return words


class MaskedStream:
    """A torch-usable stream that only dispatches to ``n_cus`` compute units (``complement``: to all the others)."""

    def __init__(self, device, n_cus: int, complement: bool = False, offset: int = 0):
        self.device = torch.device(device)
        total = torch.cuda.get_device_properties(self.device).multi_processor_count
        words = cu_mask_words(n_cus, total, offset)
        if complement:
            full = cu_mask_words(total, total)
            words = [f & ~w for f, w in zip(full, words)]
        self.n_cus = sum(bin(w).count("1") for w in words)
        self.mask_words = words
        arr = (ctypes.c_uint32 * len(words))(*words)
        self._raw = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            rc = _libhip().hipExtStreamCreateWithCUMask(ctypes.byref(self._raw), len(words), arr)
        if rc != 0 or not self._raw.value:
            raise RuntimeError(f"hipExtStreamCreateWithCUMask failed (rc {rc})")
        self.stream = torch.cuda.ExternalStream(self._raw.value, device=self.device)

    def close(self):
        if self._raw is not None and self._raw.value:
            try:
                torch.cuda.synchronize(self.device)
                _libhip().hipStreamDestroy(self._raw)
            except Exception:  # noqa: BLE001
                pass
            self._raw = None


def rollout_stream(device, n_cus: Optional[int]):
    """(torch stream, owner): a CU-masked stream when ``n_cus`` is given, else an ordinary one."""
    if n_cus:
        ms = MaskedStream(device, int(n_cus))
        return ms.stream, ms
    return torch.cuda.Stream(device), None

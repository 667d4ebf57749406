"""Metrics on their way from the device.

The runner needs a step's numbers on the host only when it LOGS them; reading them back with ``tensor.tolist()`` at the point
where they are produced stalls the host behind everything queued on the device, and the device then idles while the host
builds dicts and queues the next launches (measured in the benchmark loop: ~0.3 ms of a 9.2 ms iteration, twice per iteration).
``PendingMetrics`` queues the copy into pinned host memory behind the producing kernels, records an event, and finishes the dict
when asked -- the runner asks one iteration late, with the next iteration's launches already queued."""

from __future__ import annotations

from typing import Callable

import torch

_RING = 4  # pinned buffers per (device, dtype, length): at most two iterations are ever in flight
_pool: dict = {}


def _pinned(like: torch.Tensor) -> torch.Tensor:
    key = (like.device, like.dtype, like.numel())
    ring = _pool.setdefault(key, [[], 0])
    if len(ring[0]) < _RING:
        ring[0].append(torch.empty(like.numel(), dtype=like.dtype, pin_memory=True))
    buf = ring[0][ring[1] % len(ring[0])]
    ring[1] += 1
    return buf


class PendingMetrics:
    """``vec``: a 1-d device tensor; ``finish(list_of_floats) -> dict`` runs on the host once the copy has landed."""

    def __init__(self, vec: torch.Tensor, finish: Callable[[list], dict]):
        self._finish = finish
        self._value = None
        if vec.device.type != "cuda":
            self._host, self._event = vec.detach().reshape(-1), None
            return
        self._host = _pinned(vec)
        self._host.copy_(vec.detach().reshape(-1), non_blocking=True)
        self._event = torch.cuda.Event()
        self._event.record(torch.cuda.current_stream(vec.device))

    def result(self) -> dict:
        if self._value is None:
            if self._event is not None:
                self._event.synchronize()
            self._value = self._finish(self._host.tolist())
            self._host = None
        return self._value


def resolve(x):
    """A metrics dict, however it was produced."""
    return x.result() if isinstance(x, PendingMetrics) else x

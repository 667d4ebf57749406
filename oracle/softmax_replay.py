"""TEST INFRASTRUCTURE (never imported by the product): a numpy restatement of what torch.softmax(x, -1) does on the host, operation
for operation -- ATen's vectorised last-dim kernel (aten/src/ATen/native/cpu/SoftMaxKernel.cpp, _vec_softmax_lastdim; torch >= 2.5,
pinned here to the installed 2.10): Sleef_expf{16,8}_u10 of (x - max), the row sum accumulated per SIMD lane over consecutive
W-element chunks and folded by a W-lane butterfly, p = e * (1 / sum); reduced-precision rows (bf16) convert to f32, run the whole
chunks through the vector path with zero-initialised accumulators and the tail through the scalar expf, added one by one behind the
butterfly, and round p to the dtype.  csrc/token_ops.hip's categorical_sample_kernel replays exactly this, which is what makes its
sampled indices bit-exact against the reference's torch.multinomial (openvla_oft_action_model.py:352-414).

Pinned: tests/test_token_host.py::test_cpu_softmax_restatement compares it bit for bit with torch.softmax itself on this host."""

from __future__ import annotations

import numpy as np

F32 = np.float32
_LD = np.longdouble


def _fma(a, b, c):  # one rounding: the product and the sum are exact in the 64-bit significand of x87 long double
    return (np.asarray(a, dtype=_LD) * np.asarray(b, dtype=_LD) + np.asarray(c, dtype=_LD)).astype(F32)


_R_LN2 = F32(1.442695040888963407359924681001892137426645954152985934135449406931)
_L2U, _L2L = F32(0.693145751953125), F32(1.428606765330187045e-06)
_C = [F32(c) for c in (0.000198527617612853646278381, 0.00139304355252534151077271, 0.00833336077630519866943359,
                       0.0416664853692054748535156, 0.166666671633720397949219, 0.5)]


def sleef_expf_u10(d: np.ndarray) -> np.ndarray:
    """Sleef's xexpf (sleefsimdsp.c), the exp behind Vectorized<float>::exp() on AVX2 / AVX-512 builds."""
    d = d.astype(F32)
    dc = np.maximum(d, F32(-120.0))
    q = np.rint((dc * _R_LN2).astype(F32)).astype(F32)
    s = _fma(q, -_L2U, dc)
    s = _fma(q, -_L2L, s)
    u = np.full_like(dc, _C[0])
    for c in _C[1:]:
        u = _fma(u, s, c)
    u = (F32(1.0) + _fma((s * s).astype(F32), u, s)).astype(F32)
    qi = q.astype(np.int32)
    h = qi >> 1
    u = ((u * np.ldexp(F32(1), h).astype(F32)).astype(F32) * np.ldexp(F32(1), qi - h).astype(F32)).astype(F32)
    u = np.where(d < F32(-104.0), F32(0), u)
    return np.where(d > F32(100.0), F32(np.inf), u).astype(F32)


def _butterfly(acc: np.ndarray, lanes: int) -> np.float32:
    v, sh = acc.copy(), lanes // 2
    while sh:
        v = (v + v[np.arange(lanes) ^ sh]).astype(F32)
        sh //= 2
    return v[0]


def to_bf16(x: np.ndarray) -> np.ndarray:
    """float32 -> bf16 (round to nearest even), returned as float32 values."""
    u = x.astype(F32).view(np.uint32).astype(np.uint64)
    return ((u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000).astype(np.uint32).view(F32)


def softmax_row(x: np.ndarray, lanes: int = 16, reduced: bool = False) -> np.ndarray:
    """One row of torch.softmax(x, -1) on a host with ``lanes`` f32 SIMD lanes; ``reduced``: a bf16 row (x holds its f32 values)."""
    x = x.astype(F32)
    K, W = len(x), lanes
    d = (x - x.max()).astype(F32)
    if reduced:
        nv = K - K % W
        e = np.empty(K, F32)
        e[:nv] = sleef_expf_u10(d[:nv])
        e[nv:] = np.exp(d[nv:].astype(np.float64)).astype(F32)  # std::exp(float): correctly rounded
        acc = np.zeros(W, F32)
        for c0 in range(0, nv, W):
            acc = (acc + e[c0:c0 + W]).astype(F32)
        s = _butterfly(acc, W)
        for t in e[nv:]:
            s = F32(s + t)
        return to_bf16((e * (F32(1) / s)).astype(F32))
    e = sleef_expf_u10(d)
    if K < W:  # vec_reduce_all(acc, size): element by element
        s = e[0]
        for t in e[1:]:
            s = F32(s + t)
    else:
        acc, c0 = e[:W].copy(), W
        while c0 < K - K % W:
            acc = (acc + e[c0:c0 + W]).astype(F32)
            c0 += W
        if K - c0 > 0:
            acc[:K - c0] = (acc[:K - c0] + e[c0:]).astype(F32)
        s = _butterfly(acc, W)
    return (e * (F32(1) / s)).astype(F32)


def host_lanes() -> int:
    import torch
    return 8 if torch.backends.cpu.get_cpu_capability().upper() == "AVX2" else 16

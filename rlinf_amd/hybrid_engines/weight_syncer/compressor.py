"""PatchCompressor: the compression stage of the sparse weight-patch transport (mirror of
rlinf/hybrid_engines/weight_syncer/compressor.py: ``PatchCompressor.create`` :70-98, ``IdentityCompressor`` :101-112,
``NVCompCompressor`` :115-250; payload type ``CompressedWeightPatch``, patch_syncer.py:205-250).

Same surface -- ``create(compression_algorithm, transport_device)``, ``compress(WeightPatch) -> transport payload``,
``decompress(payload) -> WeightPatch``, the same CompressedWeightPatch fields and the same dtype-code table -- with the codec
itself written for gfx950: ``ZPlaneCompressor`` (csrc/zplane_codec.hip, byte planes + two-level zero masks by wavefront ballot).
nvCOMP, the reference's only codec, is NVIDIA-only and its container is not a public format: the payload BYTES of
``rlx_zplane`` differ from an nvCOMP stream and a reference peer cannot decode them.  The codec therefore answers to its OWN
name only.  The reference's name ``nvcomp_lz4`` is REFUSED with an error that says so (round 3 accepted it as a silent alias,
which hid the incompatible wire format): a deployment whose sender and receiver both run this package chooses ``rlx_zplane``
explicitly -- or sets ``RLX_NVCOMP_LZ4_AS_ZPLANE=1`` to keep a reference configuration file unchanged, with a loud warning --
and a deployment that shares patches with reference peers uses ``none``, the only byte-compatible transport.
"""

from __future__ import annotations

import ctypes
from abc import ABC, abstractmethod

import torch

from ... import _lib
from ..._lib import RlxError
from ...ops import _stream_ptr

# compressor.py:35-46 (the codes travel in the payload: they are part of the wire contract)
DTYPE_TO_CODE = {torch.uint8: 0, torch.int16: 1, torch.int32: 2, torch.int64: 3, torch.float16: 4, torch.bfloat16: 5,
                 torch.float32: 6, torch.float64: 7}
CODE_TO_DTYPE = {code: dtype for dtype, code in DTYPE_TO_CODE.items()}
ZPLANE_ALGORITHMS = ("rlx_zplane",)   # the gfx950 codec answers to its own name only
NVCOMP_ALGORITHMS = ("nvcomp_lz4",)    # rlinf/hybrid_engines/weight_syncer/compressor.py:30-32: a wire format this build cannot produce


class PatchCompressor(ABC):
    def __init__(self, transport_device):
        self.transport_device = torch.device(transport_device) if transport_device is not None else None

    @abstractmethod
    def compress(self, patch): ...

    @abstractmethod
    def decompress(self, payload): ...

    @classmethod
    def create(cls, compression_algorithm: str, transport_device) -> "PatchCompressor":
        if compression_algorithm == "none":
            return IdentityCompressor(transport_device=transport_device)
        if compression_algorithm in ZPLANE_ALGORITHMS:
            return ZPlaneCompressor(compression_algorithm=compression_algorithm, transport_device=transport_device)
        if compression_algorithm in NVCOMP_ALGORITHMS:
            import os
            import warnings
            if os.environ.get("RLX_NVCOMP_LZ4_AS_ZPLANE", "0") not in ("", "0"):
                warnings.warn(f"compression_algorithm={compression_algorithm}: nvCOMP is NVIDIA-only; sending 'rlx_zplane' streams instead "
                              "(RLX_NVCOMP_LZ4_AS_ZPLANE=1).  A reference (nvCOMP) peer CANNOT decode them: both ends must run rlinf_amd.",
                              stacklevel=2)
                return ZPlaneCompressor(compression_algorithm="rlx_zplane", transport_device=transport_device)
            raise ValueError(f"compression_algorithm={compression_algorithm!r} names the reference's nvCOMP LZ4 container, which this "
                             "gfx950 build cannot produce or read.  Use 'rlx_zplane' when sender and receiver both run rlinf_amd (its own "
                             "wire format), or 'none' for patches shared with reference peers; RLX_NVCOMP_LZ4_AS_ZPLANE=1 maps the name "
                             "to 'rlx_zplane' for configuration files that must stay unchanged.")
        # compressor.py:92-98: an unknown name is ignored with a warning, the flat tensors travel as they are
        import warnings
        warnings.warn("PatchWeightSyncer uses flat tensor transport; "
                      f"compression_algorithm={compression_algorithm} is ignored for now.", stacklevel=2)
        return IdentityCompressor(transport_device=transport_device)


class IdentityCompressor(PatchCompressor):
    def compress(self, patch):
        return patch

    def decompress(self, payload):
        from .patch_syncer import WeightPatch
        assert isinstance(payload, WeightPatch), f"IdentityCompressor expected WeightPatch, got {type(payload)}"
        return payload


class ZPlaneCompressor(PatchCompressor):
    """rows / cols / value bytes -> three "RLXZ" streams (uint8 tensors trimmed to their length: one host read of the three
    lengths per patch, where the reference reads each nvCOMP output's size).  Accelerator tensors only, like the reference's."""

    def __init__(self, compression_algorithm: str, transport_device):
        super().__init__(transport_device=transport_device)
        self.compression_algorithm = compression_algorithm
        if self.transport_device is not None and self.transport_device.type != "cuda":
            raise ValueError(f"{compression_algorithm} compression requires transport_device to be the accelerator")
        self._lib = _lib.load()
        self._pending_status = []  # device status words of the decompress launches of the current payload

    # ---- one stream ---------------------------------------------------------------------------------------------------
    def _launch_compress(self, tensor: torch.Tensor):
        """-> (worst-case sized output buffer, device u64 length) ; asynchronous."""
        t = tensor.contiguous()
        if DTYPE_TO_CODE.get(t.dtype) is None:
            raise TypeError(f"Unsupported patch tensor dtype for compression: {t.dtype}")
        es, n, dev = t.element_size(), t.numel(), t.device
        if t.data_ptr() % 16:
            t = t.clone()
        out = torch.empty(self._lib.rlx_zplane_bound_bytes(n, es), dtype=torch.uint8, device=dev)
        ws = torch.empty(self._lib.rlx_zplane_workspace_bytes(n, es), dtype=torch.uint8, device=dev)
        length = torch.zeros(1, dtype=torch.int64, device=dev)
        with torch.cuda.device(dev):
            _lib.check(self._lib.rlx_zplane_compress(t.data_ptr(), n, es, out.data_ptr(), out.numel(), length.data_ptr(),
                                                     ws.data_ptr(), ws.numel(), _stream_ptr(dev)), "rlx_zplane_compress")
        return out, length, (t, ws)

    def _decompress_tensor(self, stream: torch.Tensor, dtype_code_tensor: torch.Tensor) -> torch.Tensor:
        dtype = CODE_TO_DTYPE[int(dtype_code_tensor.item())]
        if stream.numel() == 0:  # compressor.py:186-187: an empty field travels as an empty tensor
            return torch.empty(0, dtype=dtype, device=stream.device)
        if not stream.is_cuda:
            raise RlxError("patch decompression runs on the accelerator (no CPU fallback)")
        if stream.numel() < 24:  # shorter than the "RLXZ" header: the C side would read 24 bytes out of a shorter buffer
            raise RlxError(f"compressed patch field is truncated: {stream.numel()} bytes, the stream header alone is 24")
        stream = stream.contiguous()
        if stream.data_ptr() % 8:
            stream = stream.clone()
        head = stream[:24].cpu().numpy().tobytes()
        n, es, total = ctypes.c_int64(), ctypes.c_int(), ctypes.c_uint64()
        _lib.check(self._lib.rlx_zplane_parse_header(head, ctypes.byref(n), ctypes.byref(es), ctypes.byref(total)),
                   "rlx_zplane_parse_header")
        itemsize = torch.empty((), dtype=dtype).element_size()
        if es.value != itemsize or total.value != stream.numel():
            raise RlxError(f"compressed patch field: stream of {es.value}-byte elements / {total.value} bytes does not match its "
                           f"dtype code ({dtype}) / length ({stream.numel()})")
        out = torch.empty(n.value, dtype=dtype, device=stream.device)
        status = torch.zeros(1, dtype=torch.int32, device=stream.device)
        with torch.cuda.device(stream.device):
            _lib.check(self._lib.rlx_zplane_decompress(stream.data_ptr(), stream.numel(), out.data_ptr(), n.value, es.value,
                                                       status.data_ptr(), _stream_ptr(stream.device)), "rlx_zplane_decompress")
        self._pending_status.append(status)
        return out

    # ---- the patch ------------------------------------------------------------------------------------------------------
    def compress(self, patch):
        from .patch_syncer import CompressedWeightPatch, WeightPatch
        if not isinstance(patch, WeightPatch):
            return patch  # an EmptyWeightPatch travels as it is
        assert patch.rows.is_cuda, f"{self.compression_algorithm} compression requires patch tensors on the accelerator"
        dev = patch.rows.device
        fields = (patch.rows, patch.cols, patch.values)
        launched = [self._launch_compress(t) if t.numel() else None for t in fields]
        lengths = torch.cat([l[1] for l in launched if l is not None]).tolist() if any(launched) else []  # ONE read-back
        if any(n < 24 for n in lengths):  # the encoder's bounded wait expired (a workgroup never published its offset): no stream
            raise RlxError("rlx_zplane_compress produced no stream (its look-back wait expired); set RLX_ZPLANE_SINGLE_PASS=0 for "
                           "the multi-launch encoder")
        outs, it = [], iter(lengths)
        for t, l in zip(fields, launched):
            outs.append(torch.empty(0, dtype=torch.uint8, device=dev) if l is None else l[0][:next(it)].clone())
        code = lambda t: torch.tensor(DTYPE_TO_CODE[t.dtype], dtype=torch.int8, device=dev)  # noqa: E731
        return CompressedWeightPatch(version=patch.version, ordinals=patch.ordinals, nnz_per_tensor=patch.nnz_per_tensor,
                                     rows_compressed=outs[0], cols_compressed=outs[1], values_compressed=outs[2],
                                     rows_dtype_code=code(patch.rows), cols_dtype_code=code(patch.cols),
                                     values_dtype_code=code(patch.values))

    def decompress(self, payload):
        from .patch_syncer import CompressedWeightPatch, WeightPatch
        assert isinstance(payload, CompressedWeightPatch), f"ZPlaneCompressor expected CompressedWeightPatch, got {type(payload)}"
        self._pending_status = []
        rows = self._decompress_tensor(payload.rows_compressed, payload.rows_dtype_code)
        cols = self._decompress_tensor(payload.cols_compressed, payload.cols_dtype_code)
        values = self._decompress_tensor(payload.values_compressed, payload.values_dtype_code)
        if self._pending_status and int(torch.cat(self._pending_status).abs().max().item()) != 0:
            raise RlxError("compressed patch field is corrupt or truncated (rlx_zplane_decompress status != 0)")
        return WeightPatch(version=payload.version, ordinals=payload.ordinals, nnz_per_tensor=payload.nnz_per_tensor, rows=rows,
                           cols=cols, values=values)

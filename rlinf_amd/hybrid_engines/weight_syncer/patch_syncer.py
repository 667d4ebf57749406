"""Sparse weight-patch sync, same wire format and call surface as rlinf/hybrid_engines/weight_syncer/patch_syncer.py
(``WeightPatch`` fields :130-205, ``PatchBuilder.delta_encode/decode`` :290-370, ``GPUSnapshotPatchBuilder.create_patch``
:648-774, ``PatchWeightSyncer`` :777-1137), with the per-tensor work on weight_patch.hip:

    sender   scan (one read of tensor + snapshot) -> [host reads nnz, as the reference does after nonzero()] -> emit
    receiver decode + scatter in one pass

Mirrored: both patch builders -- the same-device ("GPU snapshot") one and the host-snapshot one (``CPUSnapshotPatchBuilder``
:416-640: the snapshot lives in pinned host memory and is staged tensor by tensor on a copy stream, one tensor ahead of the
scan) --, the compression stage (``PatchCompressor.create`` / ``CompressedWeightPatch``: compressor.py, with a gfx950 codec in place
of nvCOMP's, which is NVIDIA-only) and the init handshake (receiver announces key order, shapes and
dtypes; the sender snapshots in the RECEIVER's dtypes, :950-1010).  Transport is any pair of send / recv callables, as in
the reference; ``rlinf_amd.scheduler.dist.broadcast_weight_patch`` is the RCCL broadcast of a patch from the
actor rank to the rollout ranks.  A patch built here is applied by the
reference's receiver and vice versa (tests/test_gpu_weight_patch.py checks both against the oracle byte for byte).
"""

from __future__ import annotations

from dataclasses import dataclass
from typing import Callable, Optional

import torch

from ... import _lib
from ..._lib import RlxError
from ...ops import _stream_ptr

_FLOAT_CODES = {torch.float32: _lib.DTYPE_F32, torch.bfloat16: _lib.DTYPE_BF16, torch.float16: _lib.DTYPE_F16}
_RAW_BY_SIZE = {1: _lib.DTYPE_RAW8, 2: _lib.DTYPE_RAW16, 4: _lib.DTYPE_RAW32, 8: _lib.DTYPE_RAW64}
_INDEX_CODES = {torch.uint8: 0, torch.int32: 1, torch.int64: 2}


def _dtype_code(dtype: torch.dtype) -> int:
    if dtype in _FLOAT_CODES:
        return _FLOAT_CODES[dtype]
    if dtype.is_floating_point or dtype.is_complex:
        raise RlxError(f"weight patch: unsupported floating dtype {dtype}")
    return _RAW_BY_SIZE[torch.empty((), dtype=dtype).element_size()]


def downscale_nonnegative_indices(tensor: torch.Tensor) -> torch.Tensor:
    """uint8 / int32 / int64 by the largest value (:35-57); empty tensors become uint8."""
    if tensor.numel() == 0:
        return tensor.to(torch.uint8)
    return tensor.to(_index_dtype_for(int(tensor.max().item())))


def _index_dtype_for(max_value: int) -> torch.dtype:
    if max_value <= 255:
        return torch.uint8
    if max_value <= torch.iinfo(torch.int32).max:
        return torch.int32
    return torch.int64


def as_coo_2d_view(tensor: torch.Tensor):
    """(2-D view, original shape) (:60-95): scalar -> (1,1), vector -> (1,N), rank >= 3 -> (shape[0], rest)."""
    shape = tensor.shape
    if tensor.ndim == 0:
        return tensor.unsqueeze(0).unsqueeze(0), shape
    if tensor.ndim == 1:
        return tensor.unsqueeze(0), shape
    if tensor.ndim == 2:
        return tensor, shape
    try:
        return tensor.view(tensor.shape[0], -1), shape
    except RuntimeError as err:
        raise ValueError("PatchWeightSyncer only supports ndim>=3 tensors whose trailing dimensions can be flattened as a "
                         f"view. Got shape={tuple(tensor.shape)}, stride={tuple(tensor.stride())}.") from err


@dataclass
class EmptyWeightPatch:
    version: torch.Tensor

    def to(self, device, non_blocking: bool = False) -> "EmptyWeightPatch":
        return EmptyWeightPatch(self.version.to(device=device, non_blocking=non_blocking))

    def tensors(self) -> list:
        return [self.version]


@dataclass
class WeightPatch:
    version: torch.Tensor          # i64 scalar
    ordinals: torch.Tensor         # i32 [k]   which state-dict entries changed (receiver key order)
    nnz_per_tensor: torch.Tensor   # i32 [k]
    rows: torch.Tensor             # u8 / i32 / i64 [sum nnz]  absolute or delta-encoded
    cols: torch.Tensor
    values: torch.Tensor           # u8, the changed values' bytes in the receiver's dtypes

    def to(self, device, non_blocking: bool = False) -> "WeightPatch":
        return WeightPatch(*[t.to(device=device, non_blocking=non_blocking) for t in self.tensors()])

    def tensors(self) -> list:
        return [self.version, self.ordinals, self.nnz_per_tensor, self.rows, self.cols, self.values]


@dataclass
class CompressedWeightPatch:
    """patch_syncer.py:205-250: the WeightPatch with rows / cols / value bytes as compressed byte tensors + the dtype codes the
    receiver needs to rebuild them (compressor.py:35-46)."""
    version: torch.Tensor
    ordinals: torch.Tensor
    nnz_per_tensor: torch.Tensor
    rows_compressed: torch.Tensor
    cols_compressed: torch.Tensor
    values_compressed: torch.Tensor
    rows_dtype_code: torch.Tensor
    cols_dtype_code: torch.Tensor
    values_dtype_code: torch.Tensor

    def to(self, device, non_blocking: bool = False) -> "CompressedWeightPatch":
        return CompressedWeightPatch(*[t.to(device=device, non_blocking=non_blocking) for t in self.tensors()])

    def tensors(self) -> list:
        return [self.version, self.ordinals, self.nnz_per_tensor, self.rows_compressed, self.cols_compressed,
                self.values_compressed, self.rows_dtype_code, self.cols_dtype_code, self.values_dtype_code]


class PatchBuilder:
    """Same-device snapshot builder (the reference's GPUSnapshotPatchBuilder)."""

    def __init__(self, snapshot: Optional[dict], ordered_keys: list, param_names_need_sync: list, original_shapes: dict,
                 transport_device, delta_encoding: bool):
        if not param_names_need_sync:
            raise ValueError("param_names_need_sync must not be empty")
        if not ordered_keys:
            raise ValueError("ordered_keys must not be empty")
        self.snapshot, self.ordered_keys = snapshot, ordered_keys
        self.param_names_need_sync, self.original_shapes = param_names_need_sync, original_shapes
        self.transport_device = torch.device(transport_device) if transport_device is not None else None
        self.delta_encoding = bool(delta_encoding)
        wanted = set(param_names_need_sync)
        self.param_names_need_sync_ordinals = {k: i for i, k in enumerate(ordered_keys) if k in wanted}
        self._ws: dict = {}

    # the two index codecs, for callers that hold decoded indices (:290-370); the kernels do both on the fly
    @staticmethod
    def delta_encode(rows: torch.Tensor, cols: torch.Tensor):
        assert rows.numel() > 0, "No indices to encode"
        assert rows.numel() == cols.numel(), "Rows and columns must have the same number of elements"
        if rows.numel() == 1:
            return rows, cols
        dr, dc = torch.empty_like(rows), torch.empty_like(cols)
        dr[0], dc[0] = rows[0], cols[0]
        dr[1:] = rows[1:] - rows[:-1]
        dc[1:] = torch.where(rows[1:] == rows[:-1], cols[1:] - cols[:-1], cols[1:])
        return dr, dc

    @staticmethod
    def delta_decode(rows_delta: torch.Tensor, cols_delta: torch.Tensor):
        if rows_delta.numel() == 0:
            raise ValueError("No indices to decode")
        if rows_delta.numel() != cols_delta.numel():
            raise ValueError("Rows and columns must have the same number of elements")
        rows = torch.cumsum(rows_delta, dim=0, dtype=torch.int64)
        start = torch.zeros_like(rows_delta, dtype=torch.bool)
        start[0] = True
        start[1:] = rows_delta[1:] != 0
        idx = torch.arange(rows_delta.numel(), device=rows_delta.device, dtype=torch.int64)
        seg = torch.cummax(torch.where(start, idx, torch.zeros_like(idx)), dim=0).values
        cum = torch.cumsum(cols_delta, dim=0, dtype=torch.int64)
        return rows, cum - (cum - cols_delta)[seg]

    def _workspace(self, dev, nbytes: int) -> torch.Tensor:
        ws = self._ws.get(dev)
        if ws is None or ws.numel() < nbytes:
            ws = self._ws[dev] = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        return ws

    @torch.no_grad()
    def create_patch(self, state_dict: dict, version):
        if set(state_dict.keys()) != set(self.ordered_keys):
            raise ValueError("State dict keys do not match snapshot keys")
        if self.snapshot is None:  # an inactive sender still answers, with an empty patch (:658-669)
            return EmptyWeightPatch(torch.as_tensor(version, dtype=torch.int64, device=self.transport_device))
        lib = _lib.load()
        staged = []  # (ordinal, snapshot 2-D view, value 2-D view, nnz counter, workspace)
        dev = None
        for name in self.param_names_need_sync:
            value = state_dict[name]
            if value.shape != self.original_shapes[name]:
                raise ValueError(f"Shape mismatch for key {name}: expected {self.original_shapes[name]}, got {value.shape}")
            value2d, _ = as_coo_2d_view(value.detach())
            snap = self.snapshot[name]
            if not value2d.is_cuda or not snap.is_cuda:
                raise ValueError(f"PatchBuilder requires sender tensors and snapshots on the accelerator (key={name})")
            if snap.device != value2d.device:
                raise ValueError(f"GPU snapshot and state tensor must be on the same accelerator (key={name})")
            dev = snap.device
            value2d = value2d.contiguous()
            try:
                vcode, scode = _dtype_code(value2d.dtype), _dtype_code(snap.dtype)
                supported = vcode == scode or (vcode == _lib.DTYPE_F32 and scode in (_lib.DTYPE_BF16, _lib.DTYPE_F16))
            except (RlxError, KeyError):
                supported = False
            if not supported:  # rare dtype pairs: cast first (one extra pass), then the same-dtype kernels
                value2d = value2d.to(snap.dtype)
                vcode = scode = _dtype_code(snap.dtype)
            n = value2d.numel()
            if n == 0:
                continue
            nbytes = lib.rlx_patch_workspace_bytes(n)
            ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)  # one per tensor: all scans run before the first readback
            nnz = torch.zeros(1, dtype=torch.int64, device=dev)
            with torch.cuda.device(dev):
                _lib.check(lib.rlx_patch_scan(value2d.data_ptr(), vcode, snap.data_ptr(), scode, n, ws.data_ptr(), nbytes,
                                              nnz.data_ptr(), _stream_ptr(dev)), "rlx_patch_scan")
            staged.append((self.param_names_need_sync_ordinals[name], snap, value2d, vcode, scode, nnz, ws))
        if dev is None:
            raise RuntimeError("Snapshot contains no tensors")
        counts = torch.cat([s[5] for s in staged]).tolist() if staged else []  # ONE readback for the whole state dict
        ords, nnzs, rows_l, cols_l, vals_l = [], [], [], [], []
        maxima = torch.zeros(2, dtype=torch.int64, device=dev)
        for (ordinal, snap, value2d, vcode, scode, _, ws), nnz in zip(staged, counts):
            if nnz == 0:
                continue
            rows = torch.empty(nnz, dtype=torch.int64, device=dev)
            cols = torch.empty(nnz, dtype=torch.int64, device=dev)
            vals = torch.empty(nnz, dtype=snap.dtype, device=dev)
            with torch.cuda.device(dev):
                _lib.check(lib.rlx_patch_emit(value2d.data_ptr(), vcode, snap.data_ptr(), scode, value2d.numel(),
                                              snap.shape[1], int(self.delta_encoding), ws.data_ptr(), nnz, rows.data_ptr(),
                                              cols.data_ptr(), vals.data_ptr(), maxima.data_ptr(), _stream_ptr(dev)),
                           "rlx_patch_emit")
            ords.append(ordinal), nnzs.append(nnz), rows_l.append(rows), cols_l.append(cols)
            vals_l.append(vals.view(torch.uint8))
        tdev = self.transport_device or dev
        if not rows_l:
            return EmptyWeightPatch(torch.tensor(version, dtype=torch.int64, device=tdev))
        max_r, max_c = maxima.tolist()  # the reference reads .max().item() of the concatenated indices here
        return WeightPatch(
            version=torch.tensor(version, dtype=torch.int64, device=tdev),
            ordinals=torch.tensor(ords, dtype=torch.int32, device=tdev),
            nnz_per_tensor=torch.tensor(nnzs, dtype=torch.int32, device=tdev),
            rows=torch.cat(rows_l).to(_index_dtype_for(max_r)).to(tdev),
            cols=torch.cat(cols_l).to(_index_dtype_for(max_c)).to(tdev),
            values=torch.cat(vals_l).to(tdev))


class CPUSnapshotPatchBuilder(PatchBuilder):
    """The snapshot lives on the HOST (patch_syncer.py:416-640, `snapshot_device: cpu`: a sender whose accelerator cannot spare a
    second copy of the weights).  Per tensor: the pinned host snapshot is staged to the accelerator on a copy stream -- one tensor
    AHEAD of the scan, like the reference's _prefetch_snapshot -- scanned / emitted by the same kernels as the same-device builder
    (which also bring the staged copy up to date), and, only when something changed, written back to the host snapshot as ONE
    sequential copy on the copy stream.  (The reference scatters the changed values into the host tensor with index_put: a
    random-access pass over host memory per tensor; a sequential PCIe write of the tensor is cheaper from the first few per cent
    of changed elements on, and nothing at all is written back for an unchanged tensor.)"""

    def __init__(self, snapshot, ordered_keys, param_names_need_sync, original_shapes, transport_device, delta_encoding):
        super().__init__(snapshot, ordered_keys, param_names_need_sync, original_shapes, transport_device, delta_encoding)
        self._copy_streams: dict = {}
        if snapshot is not None:
            for key, t in snapshot.items():
                if t.device.type != "cpu":
                    raise ValueError(f"CPUSnapshotPatchBuilder requires snapshots to be on CPU. Got key={key}, device={t.device}.")

    def _copy_stream(self, dev):
        st = self._copy_streams.get(dev)
        if st is None:
            st = self._copy_streams[dev] = torch.cuda.Stream(dev)
        return st

    def _prefetch(self, state_dict: dict, i: int):
        name = self.param_names_need_sync[i]
        value = state_dict[name]
        if value.shape != self.original_shapes[name]:
            raise ValueError(f"Shape mismatch for key {name}: expected {self.original_shapes[name]}, got {value.shape}")
        value2d, _ = as_coo_2d_view(value.detach())
        if not value2d.is_cuda:
            raise ValueError(f"CPUSnapshotPatchBuilder requires sender state_dict tensors to be on accelerator. Got key={name}, "
                             f"device={value2d.device}.")
        host = self.snapshot[name]
        dev = value2d.device
        cs = self._copy_stream(dev)
        cs.wait_stream(torch.cuda.current_stream(dev))  # the staging buffer's previous user
        with torch.cuda.stream(cs):
            staged = host.to(device=dev, non_blocking=True)
        done = torch.cuda.Event()
        done.record(cs)
        return name, value2d.contiguous(), host, staged, done

    @torch.no_grad()
    def create_patch(self, state_dict: dict, version):
        if set(state_dict.keys()) != set(self.ordered_keys):
            raise ValueError("State dict keys do not match snapshot keys")
        if self.snapshot is None:
            return EmptyWeightPatch(torch.as_tensor(version, dtype=torch.int64, device=self.transport_device))
        lib = _lib.load()
        staged_l, dev = [], None
        nxt = self._prefetch(state_dict, 0)
        for i in range(len(self.param_names_need_sync)):
            name, value2d, host, snap, done = nxt
            nxt = self._prefetch(state_dict, i + 1) if i + 1 < len(self.param_names_need_sync) else None
            if dev is None:
                dev = value2d.device
            elif dev != value2d.device:
                raise ValueError("CPUSnapshotPatchBuilder requires all sender state_dict tensors to be on the same accelerator. "
                                 f"Expected {dev}, got {value2d.device} for key={name}.")
            cur = torch.cuda.current_stream(dev)
            cur.wait_event(done)
            snap.record_stream(cur)
            try:
                vcode, scode = _dtype_code(value2d.dtype), _dtype_code(snap.dtype)
                supported = vcode == scode or (vcode == _lib.DTYPE_F32 and scode in (_lib.DTYPE_BF16, _lib.DTYPE_F16))
            except (RlxError, KeyError):
                supported = False
            if not supported:
                value2d = value2d.to(snap.dtype)
                vcode = scode = _dtype_code(snap.dtype)
            n = value2d.numel()
            if n == 0:
                continue
            nbytes = lib.rlx_patch_workspace_bytes(n)
            ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            nnz = torch.zeros(1, dtype=torch.int64, device=dev)
            with torch.cuda.device(dev):
                _lib.check(lib.rlx_patch_scan(value2d.data_ptr(), vcode, snap.data_ptr(), scode, n, ws.data_ptr(), nbytes,
                                              nnz.data_ptr(), _stream_ptr(dev)), "rlx_patch_scan")
            staged_l.append((self.param_names_need_sync_ordinals[name], snap, value2d, vcode, scode, nnz, ws, host))
        if dev is None:
            raise RuntimeError("Snapshot contains no tensors")
        counts = torch.cat([s[5] for s in staged_l]).tolist() if staged_l else []  # ONE readback for the whole state dict
        ords, nnzs, rows_l, cols_l, vals_l = [], [], [], [], []
        maxima = torch.zeros(2, dtype=torch.int64, device=dev)
        cs = self._copy_stream(dev)
        for (ordinal, snap, value2d, vcode, scode, _, ws, host), nnz in zip(staged_l, counts):
            if nnz == 0:
                continue  # unchanged tensor: the host snapshot is already right, nothing travels back
            rows = torch.empty(nnz, dtype=torch.int64, device=dev)
            cols = torch.empty(nnz, dtype=torch.int64, device=dev)
            vals = torch.empty(nnz, dtype=snap.dtype, device=dev)
            with torch.cuda.device(dev):
                _lib.check(lib.rlx_patch_emit(value2d.data_ptr(), vcode, snap.data_ptr(), scode, value2d.numel(),
                                              snap.shape[1], int(self.delta_encoding), ws.data_ptr(), nnz, rows.data_ptr(),
                                              cols.data_ptr(), vals.data_ptr(), maxima.data_ptr(), _stream_ptr(dev)),
                           "rlx_patch_emit")
            # the emit launch brought the staged copy up to date: it IS the new snapshot -> one sequential write-back
            cs.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(cs):
                host.copy_(snap, non_blocking=True)
            snap.record_stream(cs)
            ords.append(ordinal), nnzs.append(nnz), rows_l.append(rows), cols_l.append(cols)
            vals_l.append(vals.view(torch.uint8))
        cs.synchronize()  # the host snapshot is complete when create_patch returns (the next call may start right away)
        tdev = self.transport_device or dev
        if not rows_l:
            return EmptyWeightPatch(torch.tensor(version, dtype=torch.int64, device=tdev))
        max_r, max_c = maxima.tolist()
        return WeightPatch(
            version=torch.tensor(version, dtype=torch.int64, device=tdev),
            ordinals=torch.tensor(ords, dtype=torch.int32, device=tdev),
            nnz_per_tensor=torch.tensor(nnzs, dtype=torch.int32, device=tdev),
            rows=torch.cat(rows_l).to(_index_dtype_for(max_r)).to(tdev),
            cols=torch.cat(cols_l).to(_index_dtype_for(max_c)).to(tdev),
            values=torch.cat(vals_l).to(tdev))


def create_patch_builder(snapshot, ordered_keys, param_names_need_sync, original_shapes, snapshot_device, transport_device,
                         delta_encoding):
    """PatchBuilder.create (patch_syncer.py:372-404): host snapshot -> CPUSnapshotPatchBuilder, accelerator -> the same-device one."""
    snapshot_device = torch.device(snapshot_device)
    if snapshot_device.type == "cpu":
        return CPUSnapshotPatchBuilder(snapshot, ordered_keys, param_names_need_sync, original_shapes, transport_device, delta_encoding)
    if snapshot_device.type == "cuda":
        return PatchBuilder(snapshot, ordered_keys, param_names_need_sync, original_shapes, transport_device, delta_encoding)
    raise ValueError(f"Unsupported snapshot device: {snapshot_device}")


class PatchWeightSyncer:
    """init_receiver / init_sender / sync / apply with the reference's semantics; send / recv are plain callables (or
    awaitables are not needed here: workers of one rank share a process and cross-rank transport is a collective)."""

    def __init__(self, snapshot_device="cuda", transport_device=None, delta_encoding: bool = True,
                 compression_algorithm: str = "none", init_sync_enabled: bool = False, init_sync_prefixes=None,
                 init_sync_bucket_size: int = 128 * 1024 * 1024):
        from .compressor import PatchCompressor
        self.snapshot_device = torch.device(snapshot_device)
        if self.snapshot_device.type not in ("cuda", "cpu"):
            raise ValueError(f"Unsupported snapshot device: {snapshot_device}")
        self.compression_algorithm = compression_algorithm
        # patch_syncer.py:806-809; the compressed transport needs accelerator tensors (like the reference's nvCOMP one)
        self.compressor = PatchCompressor.create(compression_algorithm=compression_algorithm,
                                                 transport_device=transport_device if transport_device is not None else "cuda")
        self.snapshot = None
        self.ordered_keys: Optional[list] = None
        self.original_shapes: Optional[dict] = None
        self.patch_builder: Optional[PatchBuilder] = None
        self.delta_encoding = bool(delta_encoding)
        self.transport_device = None if transport_device is None else torch.device(transport_device)
        self._sender_initialized = self._receiver_initialized = False
        self._comm_options = None
        # init sync (:784-805): one full copy of (a prefix-selected part of) the weights, in the RECEIVER's dtypes, sent as
        # buckets before the first patch so that sender snapshot and receiver state start out equal
        self.init_sync_enabled = bool(init_sync_enabled)
        self.init_sync_prefixes = None if init_sync_prefixes is None else [str(p) for p in init_sync_prefixes]
        if self.init_sync_enabled and self.init_sync_prefixes == []:
            raise ValueError("Patch init sync prefixes must not be empty")
        self.init_sync_bucket_size = init_sync_bucket_size

    @property
    def comm_options(self):
        return self._comm_options

    def _select_init_sync_weights(self, state_dict: dict) -> list:
        """:811-840 -- every entry, or those whose key is a listed prefix / lies under one; an unmatched prefix is an error."""
        if self.init_sync_prefixes is None:
            return list(state_dict.items())
        matched = dict.fromkeys(self.init_sync_prefixes, False)
        selected = []
        for key, value in state_dict.items():
            for prefix in self.init_sync_prefixes:
                if key == prefix or key.startswith(f"{prefix}."):
                    matched[prefix] = True
                    selected.append((key, value))
                    break
        unmatched = [p for p, hit in matched.items() if not hit]
        if unmatched:
            raise ValueError(f"Patch init sync prefixes did not match any state_dict keys: {unmatched}")
        return selected

    def _sync_init_weights(self, state_dict: dict, receiver_dtypes: dict, send: Callable) -> None:
        from .bucket_syncer import iter_named_tensor_buckets

        selected = self._select_init_sync_weights(state_dict)
        for key, _ in selected:
            if key not in receiver_dtypes:
                raise ValueError(f"Patch init sync sender key {key} does not exist on receiver")
        for bucket in iter_named_tensor_buckets(selected, 0, bucket_size=self.init_sync_bucket_size,
                                                bucket_device=self.transport_device or "cuda",
                                                dtype_resolver=lambda key, _dtype: receiver_dtypes[key]):
            send(bucket)

    def _apply_init_weights(self, state_dict: dict, recv: Callable) -> None:
        from .bucket_syncer import SYNCER_VERSION_KEY, TOTAL_BUCKETS_KEY, load_bucket

        total = None
        while total is None or total > 0:
            bucket = recv()
            if not isinstance(bucket, dict):
                raise TypeError("Patch init sync receiver expected a bucket payload dictionary")
            if total is None:
                total = int(bucket.pop(TOTAL_BUCKETS_KEY).item())
                bucket.pop(SYNCER_VERSION_KEY)
            for key in bucket:
                if key not in state_dict:
                    raise ValueError(f"Patch init sync receiver key {key} does not exist in state_dict")
            load_bucket(state_dict, bucket)
            total -= 1

    def sender_initialized(self) -> bool:
        return self._sender_initialized

    def receiver_initialized(self) -> bool:
        return self._receiver_initialized

    def init_receiver(self, state_dict: dict, recv: Optional[Callable], send: Callable) -> None:
        assert not self._receiver_initialized, "Receiver already initialized"
        if state_dict is None:
            raise ValueError("PatchWeightSyncer receiver init requires a state_dict")
        self.ordered_keys, self.original_shapes, dtypes = [], {}, {}
        for key, tensor in state_dict.items():
            view, shape = as_coo_2d_view(tensor)
            self.ordered_keys.append(key)
            self.original_shapes[key] = shape
            dtypes[key] = view.dtype
        send({"ordered_keys": self.ordered_keys, "original_shapes": self.original_shapes, "receiver_dtypes": dtypes})
        if self.init_sync_enabled:
            self._apply_init_weights(state_dict, recv)
        self._receiver_initialized = True

    @torch.no_grad()
    def init_sender(self, state_dict: dict, param_names_need_sync: list, send: Optional[Callable], recv: Callable,
                    is_sender: bool = True) -> None:
        assert not self._sender_initialized, "Sender already initialized"
        meta = recv()
        self.ordered_keys, self.original_shapes = meta["ordered_keys"], meta["original_shapes"]
        dtypes = meta["receiver_dtypes"]
        if set(state_dict.keys()) != set(self.ordered_keys):
            raise ValueError("Sender state dict keys do not match receiver keys")
        if self.init_sync_enabled:
            self._sync_init_weights(state_dict, dtypes, send)
        snapshot = {}
        for key in param_names_need_sync:
            view, shape = as_coo_2d_view(state_dict[key].detach())
            if shape != self.original_shapes[key]:
                raise ValueError(f"Shape mismatch for key {key}: expected {self.original_shapes[key]}, got {shape}")
            if is_sender:
                snap = view.to(dtype=dtypes[key], copy=True).contiguous()
                if self.snapshot_device.type == "cpu":  # pinned: the per-tensor staging copies are asynchronous
                    host = torch.empty(snap.shape, dtype=snap.dtype, device="cpu", pin_memory=torch.cuda.is_available())
                    host.copy_(snap)
                    snap = host
                snapshot[key] = snap
        self.snapshot = snapshot if is_sender else None
        self.patch_builder = create_patch_builder(self.snapshot, self.ordered_keys, list(param_names_need_sync), self.original_shapes,
                                                  self.snapshot_device, self.transport_device, self.delta_encoding)
        self._sender_initialized = True

    def create_patch(self, state_dict: dict, version):
        if self.patch_builder is None:
            raise RuntimeError("Sender not initialized")
        return self.patch_builder.create_patch(state_dict, version)

    def sync(self, state_dict: dict, send: Callable, version) -> None:
        """patch_syncer.py:1018-1041: build the patch, compress it (an EmptyWeightPatch travels as it is), send."""
        patch = self.create_patch(state_dict, version)
        send(patch if isinstance(patch, EmptyWeightPatch) else self.compressor.compress(patch))

    @torch.no_grad()
    def apply(self, model_or_state_dict, recv: Callable) -> int:
        assert self.ordered_keys is not None and self.original_shapes is not None, "Snapshot info not initialized"
        payload = recv()
        if isinstance(payload, EmptyWeightPatch):
            return int(payload.version.item())
        payload = self.compressor.decompress(payload)  # :1060 (identity for compression "none")
        from .bucket_syncer import target_state, weights_changed
        state = target_state(model_or_state_dict)
        nnzs = payload.nnz_per_tensor.tolist()
        ords = payload.ordinals.tolist()
        total = sum(nnzs)
        assert payload.rows.numel() == payload.cols.numel() == total, "Patch payload size does not match nnz_per_tensor"
        lib = _lib.load()
        off = voff = 0
        for ordinal, nnz in zip(ords, nnzs):
            key = self.ordered_keys[ordinal]
            target, _ = as_coo_2d_view(state[key])
            assert state[key].shape == self.original_shapes[key], f"Shape mismatch for key {key}"
            if not target.is_cuda or not target.is_contiguous():
                raise RlxError(f"patch apply needs contiguous accelerator tensors (key={key})")
            dev = target.device
            rows = payload.rows[off:off + nnz].to(dev)
            cols = payload.cols[off:off + nnz].to(dev)
            es = target.element_size()
            vals = payload.values[voff:voff + nnz * es].to(dev)
            off, voff = off + nnz, voff + nnz * es
            ws_bytes = lib.rlx_patch_apply_workspace_bytes(nnz)
            ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
            with torch.cuda.device(dev):
                _lib.check(lib.rlx_patch_apply(target.data_ptr(), _dtype_code(target.dtype), target.shape[0], target.shape[1],
                                               rows.data_ptr(), _INDEX_CODES[rows.dtype], cols.data_ptr(),
                                               _INDEX_CODES[cols.dtype], int(self.delta_encoding), vals.data_ptr(), nnz,
                                               ws.data_ptr(), ws_bytes, _stream_ptr(dev)), "rlx_patch_apply")
        weights_changed(model_or_state_dict)
        return int(payload.version.item())

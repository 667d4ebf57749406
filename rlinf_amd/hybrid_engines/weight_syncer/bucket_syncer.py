"""Bucketed full-weight sync, same wire format and call surface as rlinf/hybrid_engines/weight_syncer/bucket_syncer.py
(``iter_named_tensor_buckets`` :33-127, ``BucketWeightSyncer`` :130-339).

What a bucket is on the wire is unchanged: a dict name -> tensor, the first one also carrying ``total_buckets`` and
``syncer_version`` (int32).  What changes is how it is made: the reference casts and copies one parameter at a time
(one kernel and one allocation each, :110-121); here every payload tensor of a bucket is a view of ONE flat byte buffer
(``WeightBucket.flat``), filled by one launch of ``rlx_copy_segments`` (bucket_copy.hip) that converts on the way -- and
the receiver's ``load_state_dict`` (:296-323) is one launch of the same kernel in the other direction.  A flat bucket
also travels as a single RCCL broadcast (``rlinf_amd.scheduler.dist.broadcast_weight_bucket``) instead of one
send per tensor.  Buckets made by the reference's sender are applied by this receiver and vice versa.
"""

from __future__ import annotations

import ctypes
from typing import Callable, Iterable, Iterator, Optional

import numpy as np
import torch

from ... import _lib
from ..._lib import RlxError
from ...ops import _stream_ptr
from .patch_syncer import _dtype_code

_STR_DTYPES = {"float32": torch.float32, "fp32": torch.float32, "float16": torch.float16, "fp16": torch.float16,
               "bfloat16": torch.bfloat16, "bf16": torch.bfloat16}
_ALIGN = 256  # byte alignment of every payload tensor inside the flat buffer (16-byte packs, whole cache lines)

TOTAL_BUCKETS_KEY = "total_buckets"
SYNCER_VERSION_KEY = "syncer_version"


def normalize_dtype(dtype):
    """rlinf/utils/utils.py:99-109."""
    if dtype is None or isinstance(dtype, torch.dtype):
        return dtype
    if isinstance(dtype, str) and dtype.lower() in _STR_DTYPES:
        return _STR_DTYPES[dtype.lower()]
    raise TypeError(f"Unsupported dtype: {dtype}")


def normalize_device(device) -> torch.device:
    """rlinf/utils/utils.py:112-116 (the worker device type here is always the HIP device, "cuda" to torch)."""
    if device is None:
        device = "cuda"
    return device if isinstance(device, torch.device) else torch.device(device)


_DTYPE_SIZES: dict = {}


def dtype_size(dtype) -> int:
    size = _DTYPE_SIZES.get(dtype)
    if size is None:
        size = _DTYPE_SIZES[dtype] = torch.empty((), dtype=normalize_dtype(dtype)).element_size()
    return size


class WeightBucket(dict):
    """name -> tensor like the reference's bucket; every payload tensor is a view into ``flat`` (uint8), and ``layout``
    lists (key, dtype, shape, byte offset) so that the flat buffer alone can be shipped and re-viewed."""

    flat: Optional[torch.Tensor] = None
    layout: tuple = ()

    @classmethod
    def from_flat(cls, flat: torch.Tensor, layout, meta: Optional[dict] = None) -> "WeightBucket":
        b = cls(meta or {})
        b.flat, b.layout = flat, tuple(layout)
        typed = {}  # one typed alias of the flat buffer per dtype, then ONE as_strided per tensor
        for key, dtype, shape, off in layout:
            base = typed.get(dtype)
            if base is None:
                es = dtype_size(dtype)
                base = typed[dtype] = (flat[:flat.numel() // es * es].view(dtype), es)
            strides, acc = [], 1
            for d in reversed(shape):
                strides.append(acc)
                acc *= d
            b[key] = base[0].as_strided(shape, strides[::-1], off // base[1])
        return b


def plan_buckets(items: Iterable, bucket_size: int, dtype_resolver: Optional[Callable] = None) -> list:
    """The bucket plan of iter_named_tensor_buckets (:62-86): walk the items in order, close a bucket as soon as its
    transport bytes reach ``bucket_size`` (a tensor is never split).  -> [[(key, tensor, transport dtype), ...], ...]"""
    reserved = {TOTAL_BUCKETS_KEY, SYNCER_VERSION_KEY}
    plan, cur, held = [], [], 0
    for key, value in items:
        if key in reserved:
            raise ValueError(f"Bucket payload key conflicts with metadata key: {key}")
        tdt = dtype_resolver(key, value.dtype) if dtype_resolver is not None else value.dtype
        cur.append((key, value, tdt))
        held += value.numel() * dtype_size(tdt)
        if held >= bucket_size:
            plan.append(cur)
            cur, held = [], 0
    if held > 0:
        plan.append(cur)
    if not plan:
        raise ValueError("No parameters to sync")
    return plan


_SEGMENT = np.dtype([("src", "<u8"), ("dst", "<u8"), ("n", "<i8"), ("src_dtype", "<i4"), ("dst_dtype", "<i4"),
                     ("first_chunk", "<i8")])  # rlx_copy_segment
assert _SEGMENT.itemsize == ctypes.sizeof(_lib.CopySegment)


def _codes(src_dtype: torch.dtype, dst_dtype: torch.dtype, what) -> tuple:
    if src_dtype == dst_dtype:  # raw copy by width, whatever the dtype
        es = dtype_size(src_dtype)
        if es not in (1, 2, 4, 8):
            raise RlxError(f"copy segment {what}: {src_dtype} elements are {es} bytes wide")
        code = _lib.DTYPE_RAW8 + (1, 2, 4, 8).index(es)
        return code, code
    return _dtype_code(src_dtype), _dtype_code(dst_dtype)


def _place_table(table: np.ndarray, dev: torch.device) -> tuple:
    """Plan (host: chunk numbers, validation) and place the table on the device -> (device table, segments, chunks)."""
    lib = _lib.load()
    total = ctypes.c_int64(0)
    _lib.check(lib.rlx_copy_segments_plan(table.ctypes.data_as(ctypes.POINTER(_lib.CopySegment)), len(table), ctypes.byref(total)),
               "rlx_copy_segments_plan")
    return torch.from_numpy(table.view(np.uint8)).to(dev), len(table), total.value


def _launch_placed(placed: tuple, dev: torch.device) -> None:
    table_dev, n, chunks = placed
    with torch.cuda.device(dev):
        _lib.check(_lib.load().rlx_copy_segments(table_dev.data_ptr(), n, chunks, _stream_ptr(dev)), "rlx_copy_segments")
    table_dev.record_stream(torch.cuda.current_stream(dev))


def _launch_table(table: np.ndarray, dev: torch.device) -> None:
    """ONE rlx_copy_segments launch over the table."""
    if len(table):
        _launch_placed(_place_table(table, dev), dev)


def _run_segments(segments: list, dev: torch.device) -> None:
    """segments: (src tensor, dst tensor) pairs of equal numel, both contiguous on ``dev`` -> one launch."""
    table = np.zeros(len(segments), dtype=_SEGMENT)
    for k, (src, dst) in enumerate(segments):
        if src.numel() != dst.numel():
            raise RlxError(f"copy segment {k}: {src.numel()} source elements for {dst.numel()} destination elements")
        table[k] = (src.data_ptr(), dst.data_ptr(), src.numel(), *_codes(src.dtype, dst.dtype, k), 0)
    _launch_table(table, dev)


def _device_contiguous(t: torch.Tensor, what: str) -> torch.Tensor:
    if not t.is_cuda:
        raise RlxError(f"bucket weight sync needs accelerator tensors ({what} is on {t.device}); there is no CPU path")
    return t if t.is_contiguous() else t.contiguous()


class BucketPacker:
    """One bucket of the plan, prepared once: layout, byte offsets, element counts and dtype codes as arrays, so that a
    sync is one allocation, a vectorised table fill, one small upload and one launch -- parameters keep their names,
    shapes and dtypes from sync to sync, only their addresses may move."""

    def __init__(self, items: list):
        layout, off = [], 0
        for key, value, tdt in items:
            layout.append((key, tdt, tuple(value.shape), off))
            off += -(-value.numel() * dtype_size(tdt) // _ALIGN) * _ALIGN
        self.layout = tuple(layout)
        self.nbytes = max(off, _ALIGN)
        self.table = np.zeros(len(items), dtype=_SEGMENT)
        self.table["n"] = [v.numel() for _, v, _ in items]
        codes = [_codes(v.dtype, tdt, key) for key, v, tdt in items]
        self.table["src_dtype"], self.table["dst_dtype"] = [c[0] for c in codes], [c[1] for c in codes]
        self.offsets = np.array([l[3] for l in layout], dtype=np.uint64)
        self.signature = tuple((key, v.dtype, tuple(v.shape), tdt) for key, v, tdt in items)
        self._kept = None  # persistent mode: (device flat, placed table, bucket, source addresses)

    def pack(self, items: list, bucket_device: torch.device, meta: Optional[dict] = None, persistent: bool = False
             ) -> WeightBucket:
        """``persistent``: keep the transport buffer, its views and the device table between syncs -- a sync whose
        parameters sit where they sat last time is then one launch and nothing else.  The bucket handed out views the same
        memory every time: the receiver must have consumed it before the next sync (load_bucket copies out at once)."""
        srcs = [_device_contiguous(v.detach(), key) for key, v, _ in items]  # (temporaries stay alive past the launch)
        dev = srcs[0].device
        ptrs = [t.data_ptr() for t in srcs]
        if persistent and self._kept is not None and self._kept[0].device == dev:
            flat, placed, bucket, old_ptrs = self._kept
            if ptrs != old_ptrs:
                self.table["src"] = ptrs
                placed = _place_table(self.table, dev)
            _launch_placed(placed, dev)
            if bucket_device.type != dev.type:
                bucket.flat.copy_(flat)
            self._kept = (flat, placed, bucket, ptrs)
            return self._handout(bucket, meta)
        flat = torch.empty(self.nbytes, dtype=torch.uint8, device=dev)
        self.table["src"] = ptrs
        self.table["dst"] = self.offsets + np.uint64(flat.data_ptr())
        placed = _place_table(self.table, dev)
        _launch_placed(placed, dev)
        # a host-staged bucket: ONE device-to-host copy of the flat buffer
        out = flat if bucket_device.type == dev.type else flat.to(bucket_device)
        bucket = WeightBucket.from_flat(out, self.layout, meta)  # metadata keys first, as in the reference's first bucket
        if persistent:
            self._kept = (flat, placed, WeightBucket.from_flat(out, self.layout), ptrs)
        return bucket

    @staticmethod
    def _handout(kept: WeightBucket, meta: Optional[dict]) -> WeightBucket:
        """A fresh dict over the kept views (receivers pop the metadata keys off what they are given)."""
        b = WeightBucket(meta or {})
        b.update(kept)
        b.flat, b.layout = kept.flat, kept.layout
        return b


def pack_bucket(items: list, bucket_device: torch.device, meta: Optional[dict] = None) -> WeightBucket:
    """One bucket of the plan [(key, tensor, transport dtype), ...] -> WeightBucket: one allocation, one launch."""
    return BucketPacker(items).pack(items, bucket_device, meta)


def iter_named_tensor_buckets(items: Iterable, version, *, bucket_size: int, bucket_device, dtype_resolver=None,
                              packers: Optional[dict] = None, persistent: bool = False) -> Iterator[WeightBucket]:
    """bucket_syncer.py:33-127.  DTensors have no meaning here (one process per GPU holds whole tensors).
    ``packers``: a dict the caller keeps between syncs; prepared buckets are reused while names / shapes / dtypes repeat."""
    bucket_device = normalize_device(bucket_device)
    plan = plan_buckets(items, bucket_size, dtype_resolver)
    for bucket_items in plan:  # fail before touching the device: the product has no host path
        for key, value, _ in bucket_items:
            if not value.is_cuda:
                raise RlxError(f"bucket weight sync needs accelerator tensors ({key} is on {value.device}); there is no CPU path")
    meta = {TOTAL_BUCKETS_KEY: torch.tensor(len(plan), dtype=torch.int32, device=bucket_device),
            SYNCER_VERSION_KEY: torch.as_tensor(version, dtype=torch.int32, device=bucket_device)}
    for k, bucket_items in enumerate(plan):
        packer = packers.get(k) if packers is not None else None
        if packer is None or packer.signature != tuple((key, v.dtype, tuple(v.shape), tdt) for key, v, tdt in bucket_items):
            packer = BucketPacker(bucket_items)
            if packers is not None:
                packers[k] = packer
        yield packer.pack(bucket_items, bucket_device, meta if k == 0 else None, persistent and packers is not None)


def target_state(model_or_state_dict) -> dict:
    """The tensors an apply() writes into: a module's state_dict() (aliasing its parameters, as torch's does -- MLPPolicy
    hands out reference-named views into its flat buffer) or the dict itself."""
    return model_or_state_dict.state_dict() if hasattr(model_or_state_dict, "state_dict") else model_or_state_dict


def weights_changed(model_or_state_dict) -> None:
    """Raw-pointer kernels wrote into the model's parameters: derived weight images (fragment tiles, packed layouts) are
    stale now.  Models that keep such images expose mark_updated()."""
    mark = getattr(model_or_state_dict, "mark_updated", None)
    if callable(mark):
        mark()


def load_bucket(state: dict, bucket: dict) -> int:
    """load_state_dict(bucket, strict=False) for the tensors of one bucket (:296-323): keys the target does not have are
    ignored, a shape mismatch raises like torch does, the copy converts to the target's dtype -- in one launch."""
    segments, errors, dev = [], [], None
    for key, value in bucket.items():
        target = state.get(key)
        if target is None:
            continue
        if tuple(target.shape) != tuple(value.shape):
            errors.append(f"size mismatch for {key}: copying a param with shape {tuple(value.shape)} from checkpoint, "
                          f"the shape in current model is {tuple(target.shape)}.")
            continue
        if not target.is_cuda or not target.is_contiguous():
            raise RlxError(f"bucket apply needs contiguous accelerator tensors (key={key})")
        dev = target.device
        segments.append((key, value, target.detach()))
    if errors:
        raise RuntimeError("Error(s) in loading state_dict:\n\t" + "\n\t".join(errors))
    if not segments:
        return 0
    flat = getattr(bucket, "flat", None)
    if flat is not None and not flat.is_cuda:  # host-staged flat bucket: one host-to-device copy, then re-view
        moved = WeightBucket.from_flat(flat.to(dev), bucket.layout)
        segments = [(k, moved[k], t) for k, _, t in segments]
    else:
        segments = [(k, v if v.is_cuda else v.to(dev), t) for k, v, t in segments]
    srcs = [_device_contiguous(v, k) for k, v, _ in segments]
    table = np.zeros(len(segments), dtype=_SEGMENT)
    table["src"] = [v.data_ptr() for v in srcs]
    table["dst"] = [t.data_ptr() for _, _, t in segments]
    table["n"] = [t.numel() for _, _, t in segments]
    codes = [_codes(v.dtype, t.dtype, k) for (k, _, t), v in zip(segments, srcs)]
    table["src_dtype"], table["dst_dtype"] = [c[0] for c in codes], [c[1] for c in codes]
    _launch_table(table, dev)
    return len(segments)


class BucketWeightSyncer:
    """bucket_syncer.py:130-339 with synchronous send / recv callables (no Ray event loop here)."""

    _TOTAL_BUCKETS_KEY = TOTAL_BUCKETS_KEY
    _SYNCER_VERSION_KEY = SYNCER_VERSION_KEY

    def __init__(self, bucket_size: int, bucket_dtype, bucket_device, is_agent: bool = False, load_instant: bool = True,
                 persistent_buckets: bool = False):
        self._sender_initialized = False
        self._receiver_initialized = False
        self._comm_options = None
        self.bucket_size = bucket_size
        self.bucket_dtype = normalize_dtype(bucket_dtype)
        self.bucket_device = normalize_device(bucket_device)
        self.is_agent = is_agent
        self.load_instant = load_instant
        self._packers: dict = {}  # bucket index -> BucketPacker, reused while the plan repeats
        # not in the reference: reuse the transport buffers between syncs (see BucketPacker.pack); off = fresh buffers per
        # sync like the reference's
        self.persistent_buckets = bool(persistent_buckets)

    @property
    def comm_options(self):
        return self._comm_options

    def sender_initialized(self) -> bool:
        return self._sender_initialized

    def receiver_initialized(self) -> bool:
        return self._receiver_initialized

    def _bucket_key(self, key: str, has_visual: bool):
        if "_extra_state" in key:
            return None
        if has_visual and self.is_agent and key.startswith("model.language_model."):
            return "model." + key[len("model.language_model."):]
        return key

    def _transport_dtype(self, dtype: torch.dtype) -> torch.dtype:
        if self.bucket_dtype is not None and dtype.is_floating_point:
            return self.bucket_dtype
        return dtype

    def iter_buckets(self, state_dict: dict, version) -> Iterator[WeightBucket]:
        has_visual = any("visual." in key for key in self.param_names_need_sync_set if key in state_dict)

        def named_items():
            for key in self.param_names_need_sync:
                value = state_dict.get(key)
                if value is None:
                    continue
                bucket_key = self._bucket_key(key, has_visual)
                if bucket_key is not None:
                    yield bucket_key, value

        yield from iter_named_tensor_buckets(named_items(), version, bucket_size=self.bucket_size,
                                             bucket_device=self.bucket_device,
                                             dtype_resolver=lambda _, dtype: self._transport_dtype(dtype), packers=self._packers,
                                             persistent=self.persistent_buckets)

    def init_sender(self, state_dict, param_names_need_sync: list, send=None, recv=None, is_sender: bool = True) -> None:
        del state_dict, send, recv, is_sender
        self.param_names_need_sync = param_names_need_sync
        self.param_names_need_sync_set = set(param_names_need_sync)
        if not self.param_names_need_sync_set:
            raise ValueError("param_names_need_sync must not be empty")
        self._sender_initialized = True

    def init_receiver(self, state_dict=None, recv=None, send=None) -> None:
        del state_dict, recv, send
        self._receiver_initialized = True

    def sync(self, state_dict: dict, send: Callable, version) -> None:
        for bucket in self.iter_buckets(state_dict, version):
            send(bucket)
            del bucket

    def apply(self, model_or_state_dict, recv: Callable) -> int:
        """Receive ``total_buckets`` buckets and load each into the model.  ``load_instant=False`` defers the loads until
        every bucket has arrived (the reference stages them in host memory for that; with 288 GB of HBM they simply stay
        where they were received)."""
        state = target_state(model_or_state_dict)
        bucket = recv()
        total_buckets = int(bucket.pop(self._TOTAL_BUCKETS_KEY).item())
        applied_version = int(bucket.pop(self._SYNCER_VERSION_KEY).item())
        held, landed, sent_keys = [], 0, []
        for k in range(total_buckets):
            if k > 0:
                bucket = recv()
            sent_keys.extend(key for key in bucket.keys() if key not in (TOTAL_BUCKETS_KEY, SYNCER_VERSION_KEY))
            if self.load_instant:
                landed += load_bucket(state, bucket)
            else:
                held.append(bucket)
        for bucket in held:
            landed += load_bucket(state, bucket)
        if sent_keys and landed == 0:
            # strict=False ignores unknown keys one by one; a whole sync of which NOTHING lands is a wiring error (e.g. a model
            # whose state_dict does not carry the reference's names): the replica would silently keep its old weights
            raise RlxError(f"weight sync delivered {len(sent_keys)} tensors (first: {sent_keys[0]!r}) but none matches a key of "
                           f"the target state dict (first target keys: {list(state)[:3]})")
        weights_changed(model_or_state_dict)
        return applied_version

"""XgmiAllReduce: the learner's gradient all-reduce as peer reads over xGMI (csrc/xgmi_allreduce.hip, include/rlx.h "e").

Replaces the gradient synchronisation of the reference's NO_SHARD data-parallel learner (rlinf/hybrid_engines/fsdp/strategy/
fsdp.py:480-496) on the critical path of every optimizer step.  One process per GPU: every rank exports a fine-grained device
buffer over HIP IPC, the handles are exchanged ONCE over torch.distributed (any backend), and from then on an all-reduce is two
kernel launches on the caller's stream -- no host round trip, no RCCL call, capturable in the hipGraph of the update phase.

``build()`` never trusts the transport blindly: it runs a seeded all-reduce and compares it with torch.distributed's own
all-reduce on every rank, and all ranks agree (MIN) on the verdict -- a node where IPC mapping, peer access or cross-GPU
visibility does not work falls back to RCCL, loudly, on all ranks at once."""

from __future__ import annotations

import ctypes
import os
from typing import Optional

import torch
import torch.distributed as dist

from .. import _lib
from .._lib import RlxError, XGMI_HANDLE_BYTES, XGMI_MAX_RANKS
from ..ops import _stream_ptr


class XgmiAllReduce:
    """One rank's end of the exchange.  Construction is COLLECTIVE-SAFE: whatever fails locally (allocation, IPC export, a bad
    device), every rank still takes part in the one all_gather_object of the handles and derives the same verdict from the
    gathered list -- a rank never sits in a collective its peers have skipped."""

    def __init__(self, ctx, n_max: int, timeout_ms: Optional[int] = None, mem_kind: int = 0):
        # The wait bound is a safety net against a peer that died, not a performance knob: ranks reach an all-reduce seconds
        # apart (lazy code-object loading on a fresh box, a slow checkpoint write on one rank), so the default is minutes, like
        # a collective library's.  An expired wait sets the status word; the AdamW launch of that all-reduce skips its update
        # and run_training raises.  (Start-up validation runs with its own short bound, see _attempt.)
        if timeout_ms is None:
            timeout_ms = int(os.environ.get("RLX_XGMI_TIMEOUT_MS", "300000"))
        self.ctx, self.n_max = ctx, int(n_max)
        self._lib = _lib.load()
        self._comm = ctypes.c_void_p()
        self.algo = "rsag" if ctx.world_size >= 4 else "direct"  # the library's default (rlx_xgmi_create): configure() changes it only when asked
        self.wait_mode = "inline"
        # True once the start-up validation has seen the ONE-LAUNCH exchange (the gradient pushed as self-validating words inside
        # the optimizer launch: csrc/adamw_clip.hip, XchgPeers) deliver the right mean on every rank; the learners then hand
        # rlx_xgmi_clip_adamw_step their sync words.  Never with ranks that share a device (the kernels spin device-wide).
        self.one_launch = False
        handle = (ctypes.c_char * XGMI_HANDLE_BYTES)()
        err = None
        try:  # local part: may fail on some ranks only
            if ctx.world_size > XGMI_MAX_RANKS:
                raise RlxError(f"xGMI all-reduce is for one node (<= {XGMI_MAX_RANKS} ranks), world_size={ctx.world_size}")
            if ctx.device is None or ctx.device.type != "cuda":
                raise RlxError("xGMI all-reduce needs device tensors")
            with torch.cuda.device(ctx.device):
                _lib.check(self._lib.rlx_xgmi_create(ctx.rank, ctx.world_size, self.n_max, int(timeout_ms), int(mem_kind),
                                                     ctypes.byref(self._comm), handle), "rlx_xgmi_create")
        except Exception as e:  # noqa: BLE001 -- reported through the gather below, raised on every rank alike
            err = f"{type(e).__name__}: {e}"
        if ctx.world_size > 1:
            # collective part: ALWAYS entered.  (handle or None, device identity) per rank
            dev_id = None
            if err is None:
                from .dist import device_identity
                dev_id = device_identity(ctx.device)  # (host, PCI location)
            gathered = [None] * ctx.world_size
            dist.all_gather_object(gathered, (None if err is not None else bytes(handle), dev_id, err))
            bad = [(r, g[2]) for r, g in enumerate(gathered) if g[0] is None]
            if bad:  # the same list on every rank -> every rank raises here
                self.close()
                raise RlxError("xGMI communicator could not be created on rank(s) " + ", ".join(f"{r} ({why})" for r, why in bad))
            # ranks that share a device (the one-GPU test set-up) must not hold it with a device-wide spin: the hand-shake then
            # runs as a one-wave launch of its own.  Derived from the gathered list, so every rank picks the same mode.
            ids = [g[1] for g in gathered]
            self.shared_device = len(set(ids)) < len(ids)
            with torch.cuda.device(ctx.device):
                _lib.check(self._lib.rlx_xgmi_connect(self._comm, b"".join(g[0] for g in gathered)), "rlx_xgmi_connect")
        elif err is not None:
            raise RlxError(err)
        else:
            self.shared_device = False
        self._ws = torch.empty(self._lib.rlx_adamw_workspace_bytes(self.n_max), dtype=torch.uint8, device=ctx.device)
        algo = os.environ.get("RLX_XGMI_ALGO", "auto").lower()
        wait = os.environ.get("RLX_XGMI_WAIT", "auto").lower()
        self.configure(algo=None if algo == "auto" else algo,
                       wait_mode=("kernel" if self.shared_device else "inline") if wait == "auto" else wait)

    _ALGOS = {"direct": 0, "rsag": 1}
    _WAITS = {"inline": 0, "kernel": 1}

    def configure(self, algo: Optional[str] = None, wait_mode: Optional[str] = None, timeout_ms: int = 0):
        """algo "direct" / "rsag" (None: the library's default -- direct below four ranks, reduce-scatter + all-gather from four
        on); wait_mode "inline" / "kernel"; timeout_ms > 0: a new bound for every peer wait.  Same call on every rank."""
        a = -1 if algo is None else self._ALGOS[algo]
        w = -1 if wait_mode is None else self._WAITS[wait_mode]
        _lib.check(self._lib.rlx_xgmi_configure(self._comm, a, w, int(timeout_ms)), "rlx_xgmi_configure")
        if algo is not None:
            self.algo = algo
        if wait_mode is not None:
            self.wait_mode = wait_mode

    @property
    def handle(self):
        return self._comm

    def all_reduce(self, inp: torch.Tensor, out: Optional[torch.Tensor] = None, scale: float = 1.0) -> torch.Tensor:
        """out = scale * sum over ranks of inp ([n] or [slabs, n] f32, slabs summed first); asynchronous on the current stream."""
        if inp.dtype != torch.float32 or not inp.is_contiguous():
            raise RlxError("xgmi all_reduce needs a contiguous float32 tensor")
        n = inp.shape[-1] if inp.dim() > 1 else inp.numel()
        slabs = inp.numel() // max(n, 1)
        out = torch.empty(n, dtype=torch.float32, device=inp.device) if out is None else out
        with torch.cuda.device(inp.device):
            _lib.check(self._lib.rlx_xgmi_allreduce_f32(self._comm, inp.data_ptr(), slabs, out.data_ptr(), n, float(scale),
                                                        self._ws.data_ptr(), self._ws.numel(), _stream_ptr(inp.device)),
                       "rlx_xgmi_allreduce_f32")
        return out

    def status_ok(self) -> bool:
        """False when a peer wait timed out since the last call (clears the word)."""
        with torch.cuda.device(self.ctx.device):
            return self._lib.rlx_xgmi_status(self._comm) == 0

    def status_snapshot(self, dst: torch.Tensor) -> torch.Tensor:
        """dst[0] (device f32) = the status word behind everything queued on the current stream; asynchronous, clears nothing.  The
        run-ahead loop appends it to a step's metric vector and calls check_status() only when it arrives non-zero."""
        with torch.cuda.device(self.ctx.device):
            _lib.check(self._lib.rlx_xgmi_status_snapshot(self._comm, dst.data_ptr(), _stream_ptr(self.ctx.device)), "rlx_xgmi_status_snapshot")
        return dst

    def check_status(self):
        """Raises when a peer wait timed out since the last call (a rank died or never reached the all-reduce)."""
        if not self.status_ok():
            raise RlxError("xGMI all-reduce: a peer did not publish its gradient within the timeout; results are invalid "
                           "(the optimizer skipped the affected steps)")

    def close(self):
        if self._comm:
            self._lib.rlx_xgmi_destroy(self._comm)
            self._comm = ctypes.c_void_p()

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass


def _agree(ok: int, ctx) -> bool:
    verdict = torch.tensor([int(ok)], dtype=torch.int32, device=ctx.device)
    dist.all_reduce(verdict, op=dist.ReduceOp.MIN)
    return int(verdict.item()) == 1


def _attempt(ctx, n_max: int, mem_kind: int, rounds: int, validate_timeout_ms: int = 20000):
    """-> (communicator or None, reason); the verdict is collective: every rank gets the same answer.

    Every rank executes THE SAME sequence of torch.distributed calls whatever happens locally: the constructor's one gather, one
    MIN, then per round one SUM of `want`, then one MIN -- a rank that sees a mismatch or a time-out records it and keeps going
    (it only stops issuing xGMI launches), so the collectives never get out of step (with RCCL a size / op mismatch is undefined
    behaviour, exactly when the transport misbehaves on a subset of ranks)."""
    comm, ok, why = None, 1, ""
    try:
        comm = XgmiAllReduce(ctx, n_max, mem_kind=mem_kind)
    except Exception as e:  # noqa: BLE001 -- every failure mode takes the same collective decision below
        ok, why = 0, f"{type(e).__name__}: {e}"
    if not _agree(ok, ctx):
        if comm is not None:
            comm.close()
        return None, why or "a peer failed to create / connect its communicator"
    product_timeout = int(os.environ.get("RLX_XGMI_TIMEOUT_MS", "300000"))
    algos = ["direct", "rsag"] if ctx.world_size >= 2 and n_max % 4 == 0 else ["direct"]
    preferred = comm.algo
    g = torch.Generator(device=ctx.device).manual_seed(1234 + ctx.rank)
    try:
        comm.configure(timeout_ms=min(validate_timeout_ms, product_timeout))  # a dead transport costs one short wait, not minutes
    except Exception as e:  # noqa: BLE001
        ok, why = 0, f"{type(e).__name__}: {e}"
    for algo in algos:  # both forms are validated: the preferred one runs the product, the other is what RLX_XGMI_ALGO selects
        for k in range(rounds):  # both staging slots, a two-slab input, the scale
            x = torch.randn(2, n_max, device=ctx.device, generator=g)
            want = x.sum(0)
            dist.all_reduce(want)  # unconditional: the ranks stay in step
            if not ok:
                continue
            try:
                comm.configure(algo=algo)
                got = comm.all_reduce(x, scale=1.0 / ctx.world_size)
                torch.cuda.synchronize(ctx.device)
                if not comm.status_ok():
                    ok, why = 0, f"{algo} round {k}: a peer wait timed out"
                elif not torch.allclose(got * ctx.world_size, want, rtol=1e-5, atol=1e-5):
                    ok, why = 0, f"{algo} round {k}: max |diff| {float((got * ctx.world_size - want).abs().max()):.3e}"
            except Exception as e:  # noqa: BLE001
                ok, why = 0, f"{type(e).__name__}: {e}"
    if ok:
        try:
            comm.configure(algo=preferred, timeout_ms=product_timeout)
        except Exception as e:  # noqa: BLE001
            ok, why = 0, f"{type(e).__name__}: {e}"
    if not _agree(ok, ctx):
        if comm is not None:
            comm.close()
        return None, why or "a peer failed the validation"
    # the one-launch form of the exchange (csrc/adamw_clip.hip, XchgPeers) is a protocol of its own -- remote WRITES of
    # self-validating words instead of remote reads behind flags -- and gets its own verdict: failing it costs the form, not the
    # transport (the learners then run the validated launch chain)
    one, one_why = 1, ""
    try:
        one, one_why = _validate_one_launch(comm, ctx, n_max, g, min(validate_timeout_ms, product_timeout), product_timeout)
    except Exception as e:  # noqa: BLE001
        one, one_why = 0, f"{type(e).__name__}: {e}"
    comm.one_launch = _agree(one, ctx)
    comm.one_launch_verdict = "passed" if comm.one_launch else (one_why or "a peer failed it")
    if not comm.one_launch and ctx.rank == 0 and one_why != "off":
        print(f"[rlinf_amd] xGMI one-launch exchange not used ({comm.one_launch_verdict}); the launch chain runs", flush=True)
    return comm, ""


def _validate_one_launch(comm, ctx, n_max: int, g, validate_timeout_ms: int, product_timeout_ms: int, rounds: int = 3):
    """(1 | 0, reason) on this rank; every rank issues the same torch.distributed calls.  A step of rlx_xgmi_clip_adamw_step with
    learning rate 0 and no clipping on scratch buffers: what lands in grad_flat is the exchange's mean gradient, stats[0] its norm."""
    from ..ops import PreparedAdamw, adamw_sync_words
    n, dev, W = int(n_max), ctx.device, ctx.world_size
    usable = (not comm.shared_device) and n % 4 == 0 and W >= 2 and os.environ.get("RLX_XGMI_ONE_LAUNCH", "1") != "0"
    sync = adamw_sync_words(n, dev) if usable else None
    ok, why = (1, "") if sync is not None else (0, "off")
    x = torch.zeros(2, n, device=dev)
    scratch = [torch.zeros(n, device=dev) for _ in range(4)]  # params, exp_avg, exp_avg_sq, grad_flat
    stats, state = torch.zeros(2, device=dev), torch.zeros(2, dtype=torch.int32, device=dev)
    ws = torch.empty(comm._lib.rlx_adamw_workspace_bytes(n), dtype=torch.uint8, device=dev)
    step = None
    if ok:
        comm.one_launch = True
        comm.configure(timeout_ms=validate_timeout_ms)
        step = PreparedAdamw(scratch[0], x, scratch[1], scratch[2], [(0, n, 0.0)], max_grad_norm=0.0, grad_scale=1.0 / W, stats=stats,
                             step_state=state, workspace=ws, xgmi=comm, grad_flat=scratch[3], sync=sync)
    for k in range(rounds):
        x.copy_(torch.randn(2, n, device=dev, generator=g))
        want = x.sum(0)
        dist.all_reduce(want)  # unconditional
        if not ok:
            continue
        with torch.cuda.device(dev):
            step(_stream_ptr(dev))
        torch.cuda.synchronize(dev)
        got = scratch[3] * W
        if not comm.status_ok():
            ok, why = 0, f"one-launch round {k}: a peer's words never arrived"
        elif not torch.allclose(got, want, rtol=1e-5, atol=1e-5):
            ok, why = 0, f"one-launch round {k}: max |diff| {float((got - want).abs().max()):.3e}"
        elif not abs(float(stats[0]) - float((want / W).double().norm())) <= 1e-4 * max(1.0, float(stats[0])):
            ok, why = 0, f"one-launch round {k}: norm {float(stats[0])} vs {float((want / W).double().norm())}"
    comm.one_launch = False
    if step is not None:
        comm.configure(timeout_ms=product_timeout_ms)
    return ok, why


class SelfAliasedXgmi:
    """TIMING TOOL (rlx_xgmi_connect_self; bench.py ``scaling_model``): the communicator of rank 0 of a ``world``-rank job whose
    peers are all its own buffer -- the per-step launch chain of a rank (stage, hand-shake, reduce(-scatter), hand-shake, gather +
    clip + AdamW) runs on one device.  Results are not a valid all-reduce: callers pass scratch parameter / moment buffers."""

    def __init__(self, device, world: int, n_max: int, algo: Optional[str] = None, wait_mode: str = "inline", timeout_ms: int = 5000,
                 timing: bool = False):
        self._lib = _lib.load()
        self.world, self.n_max, self.device = int(world), int(n_max), torch.device(device)
        self._comm = ctypes.c_void_p()
        handle = (ctypes.c_char * XGMI_HANDLE_BYTES)()
        with torch.cuda.device(self.device):
            _lib.check(self._lib.rlx_xgmi_create(0, self.world, self.n_max, int(timeout_ms), 0, ctypes.byref(self._comm), handle),
                       "rlx_xgmi_create")
            _lib.check(self._lib.rlx_xgmi_connect_self(self._comm), "rlx_xgmi_connect_self")
            self.algo = algo or ("rsag" if self.world >= 4 else "direct")
            self.wait_mode = wait_mode
            _lib.check(self._lib.rlx_xgmi_configure(self._comm, XgmiAllReduce._ALGOS[self.algo], XgmiAllReduce._WAITS[wait_mode], 0),
                       "rlx_xgmi_configure")
            if timing:  # the one-launch exchange with a rank's real communication structure instead of exact values (include/rlx.h)
                _lib.check(self._lib.rlx_xgmi_self_timing(self._comm, 1), "rlx_xgmi_self_timing")
        self.shared_device = False
        self.one_launch = wait_mode == "inline"  # (the caller decides per PreparedAdamw whether it passes sync words)

    @property
    def handle(self):
        return self._comm

    def status_ok(self) -> bool:
        with torch.cuda.device(self.device):
            return self._lib.rlx_xgmi_status(self._comm) == 0

    def close(self):
        if self._comm:
            self._lib.rlx_xgmi_destroy(self._comm)
            self._comm = ctypes.c_void_p()


def build(ctx, n_max: int, rounds: int = 4) -> Optional[XgmiAllReduce]:
    """Create, connect and VALIDATE an xGMI communicator; returns None (on every rank alike) when any step fails on any rank.
    Fine-grained device memory first, uncached second (both are coherent across GPUs); plain hipMalloc is never used for the
    product (RLX_XGMI_MEM_KIND=2 forces it for experiments).  RLX_GRAD_ALLREDUCE=rccl skips the attempt."""
    from .dist import forced_exchange
    if ctx.world_size <= 1 and not (getattr(ctx, "force_exchange", False) and forced_exchange() == "xgmi"):
        return None  # (RLX_FORCE_EXCHANGE=xgmi: the one-rank communicator, validated against the one-rank RCCL all-reduce)
    if os.environ.get("RLX_GRAD_ALLREDUCE", "xgmi").lower() in ("rccl", "nccl", "torch"):
        return None
    forced = os.environ.get("RLX_XGMI_MEM_KIND")
    reasons = []
    for kind in ((int(forced),) if forced is not None else (0, 1)):
        comm, why = _attempt(ctx, n_max, kind, rounds)
        if comm is not None:
            comm.mem_kind = kind
            return comm
        reasons.append(f"mem_kind {kind}: {why}")
    print(f"[rlinf_amd] rank {ctx.rank}: xGMI gradient all-reduce unavailable ({'; '.join(reasons)}); using RCCL", flush=True)
    return None


class LocalXgmiGroup:
    """W communicators of ONE process wired to each other directly (rlx_xgmi_connect_local): every "rank" launches on its own
    stream of the same device.  Test / bring-up tool for the exchange protocol at W = 2 .. 8 on a one-GPU box (no IPC, no
    torch.distributed); the hand-shake runs as its own one-wave launch so that W spinning grids never fill the device."""

    def __init__(self, world: int, n_max: int, device, algo: str = "direct", timeout_ms: int = 20000, mem_kind: int = 0,
                 streams=None, one_launch: bool = False):
        """``streams``: reuse these (one per rank) instead of creating new ones -- every "rank" needs a hardware queue of its own
        (GPU_MAX_HW_QUEUES >= world), and a process that keeps creating streams eventually shares queues between them."""
        self._lib = _lib.load()
        self.world, self.n_max, self.device = int(world), int(n_max), torch.device(device)
        self.comms = []
        handle = (ctypes.c_char * XGMI_HANDLE_BYTES)()
        with torch.cuda.device(self.device):
            for r in range(world):
                c = ctypes.c_void_p()
                _lib.check(self._lib.rlx_xgmi_create(r, 1 if world == 1 else world, self.n_max, timeout_ms, mem_kind, ctypes.byref(c),
                                                     handle), "rlx_xgmi_create")
                self.comms.append(c)
            arr = (ctypes.c_void_p * world)(*[c.value for c in self.comms])
            _lib.check(self._lib.rlx_xgmi_connect_local(arr, world), "rlx_xgmi_connect_local")
            # one_launch: the exchange inside the optimizer launch (pushed self-validating words; csrc/adamw_clip.hip, XchgPeers) --
            # inline waits, and all W kernels of this device must be resident together (the library checks: W <= 3 here)
            for c in self.comms:
                _lib.check(self._lib.rlx_xgmi_configure(c, XgmiAllReduce._ALGOS[algo], 0 if one_launch else 1, 0), "rlx_xgmi_configure")
        self.one_launch = bool(one_launch)
        self._permit = torch.zeros(4, dtype=torch.int64, device=self.device) if one_launch else None  # rlx_adamw_params.sync_words
        self.streams = list(streams) if streams is not None else [torch.cuda.Stream(self.device) for _ in range(world)]
        assert len(self.streams) == world
        self._ws = [torch.empty(self._lib.rlx_adamw_workspace_bytes(self.n_max), dtype=torch.uint8, device=self.device)
                    for _ in range(world)]

    def all_reduce(self, inputs, outs, scale: float = 1.0):
        """inputs[r] ([slabs, n] f32) -> outs[r] ([n]) for every rank, each on its own stream; returns after enqueueing."""
        cur = torch.cuda.current_stream(self.device)
        for r in range(self.world):
            self.streams[r].wait_stream(cur)
        for r in range(self.world):
            x = inputs[r]
            n = x.shape[-1]
            with torch.cuda.stream(self.streams[r]):
                _lib.check(self._lib.rlx_xgmi_allreduce_f32(self.comms[r], x.data_ptr(), x.numel() // n, outs[r].data_ptr(), n,
                                                            float(scale), self._ws[r].data_ptr(), self._ws[r].numel(),
                                                            self.streams[r].cuda_stream), "rlx_xgmi_allreduce_f32")
        for r in range(self.world):
            cur.wait_stream(self.streams[r])

    def clip_adamw_step(self, params, grads, grad_flat, exp_avg, exp_avg_sq, groups, stats, step_state, *, betas=(0.9, 0.999),
                        eps: float = 1e-8, weight_decay: float = 0.01, max_grad_norm: float = 0.5):
        """rlx_xgmi_clip_adamw_step on every "rank" (rank r: params[r], its slabs grads[r] [slabs, n], ...), each on its own
        stream: stage + exchange (the group's form: direct, or reduce-scatter + the gather fused into the AdamW launch) + clip +
        AdamW with the gradient mean.  Returns after enqueueing."""
        from ..ops import AdamwGroup, AdamwParams  # (ops imports nothing from here)
        cur = torch.cuda.current_stream(self.device)
        for r in range(self.world):
            self.streams[r].wait_stream(cur)
        keep = []
        for r in range(self.world):
            n = params[r].numel()
            p = AdamwParams()
            p.beta1, p.beta2, p.eps, p.weight_decay = float(betas[0]), float(betas[1]), float(eps), float(weight_decay)
            p.max_grad_norm, p.step, p.n_groups = float(max_grad_norm), 0, len(groups)
            p.grad_partials, p.grad_scale = grads[r].numel() // n, 1.0 / self.world
            for k, (b, e, lr) in enumerate(groups):
                p.groups[k] = AdamwGroup(int(b), int(e), float(lr))
            if self._permit is not None:
                p.sync_words = self._permit.data_ptr()
            keep.append(p)
            with torch.cuda.stream(self.streams[r]):
                _lib.check(self._lib.rlx_xgmi_clip_adamw_step(self.comms[r], params[r].data_ptr(), grads[r].data_ptr(),
                                                              grad_flat[r].data_ptr(), exp_avg[r].data_ptr(), exp_avg_sq[r].data_ptr(),
                                                              n, ctypes.byref(p), stats[r].data_ptr(), step_state[r].data_ptr(),
                                                              self._ws[r].data_ptr(), self._ws[r].numel(),
                                                              self.streams[r].cuda_stream), "rlx_xgmi_clip_adamw_step")
        for r in range(self.world):
            cur.wait_stream(self.streams[r])
        return keep

    def status_ok(self) -> bool:
        with torch.cuda.device(self.device):
            return all(self._lib.rlx_xgmi_status(c) == 0 for c in self.comms)

    def close(self):
        for c in self.comms:
            if c:
                self._lib.rlx_xgmi_destroy(c)
        self.comms = []

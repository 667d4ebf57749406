"""Tensor-level launchers of the token tier (include/rlx.h t1-t3): per-token log-prob / entropy over
vocabulary logits, the reasoning learner's micro-batch loss, GRPO advantages in the [bsz, seq] layout.

Same rules as ``ops.py``: HIP tensors only, no eager path; torch supplies memory, the stream and autograd
plumbing.
"""

from __future__ import annotations

import os

from ctypes import byref
from typing import Optional

import torch

from . import _lib
from ._lib import RlxError, TokenLossParams, TokenRows
from .ops import _as_f32, _as_u8, _dev, _ptr, _stream_ptr, make_ppo_params

_DTYPES = {torch.float32: _lib.DTYPE_F32, torch.bfloat16: _lib.DTYPE_BF16}


def _rows_of(logits: torch.Tensor, temperature: float, round_outputs: bool):
    """Describe ``logits[..., V]`` to the kernels without copying when it is a [bsz, rows, V] slice of a larger
    buffer (the reference's ``logits[:, -resp-1:-1, :]``); anything else is made contiguous."""
    if logits.dtype not in _DTYPES:
        raise RlxError(f"logits must be float32 or bfloat16 (got {logits.dtype})")
    if logits.dim() < 2:
        raise RlxError("logits must be [..., vocab]")
    vocab = logits.shape[-1]
    n = logits.numel() // max(vocab, 1)
    rows = TokenRows()
    rows.vocab, rows.dtype = vocab, _DTYPES[logits.dtype]
    rows.n_tokens, rows.temperature, rows.round_outputs = n, float(temperature), int(bool(round_outputs))
    if logits.dim() == 3 and logits.stride(-1) == 1 and logits.stride(1) >= vocab and not logits.is_contiguous():
        rows.rows_per_seq, rows.seq_stride, rows.row_stride = logits.shape[1], logits.stride(0), logits.stride(1)
        return logits, rows
    logits = logits.contiguous()
    rows.rows_per_seq, rows.seq_stride, rows.row_stride = max(n, 1), 0, vocab
    return logits, rows


def token_logprob_fwd(logits: torch.Tensor, labels: torch.Tensor, temperature: float = 1.0, with_entropy: bool = False,
                      round_outputs: bool = False):
    """-> (logprob, entropy or None, lse), each f32 of shape ``logits.shape[:-1]``."""
    dev = _dev(logits, labels)
    if labels.dtype != torch.int64:
        raise RlxError(f"labels must be int64 (got {labels.dtype})")
    if labels.numel() != logits.numel() // max(logits.shape[-1], 1):
        raise RlxError("labels must have one entry per logits row")
    if not temperature > 0:
        raise RlxError("temperature must be positive")
    lead = logits.shape[:-1]
    x, rows = _rows_of(logits, temperature, round_outputs)
    lab = labels.contiguous()
    logprob = torch.empty(lead, dtype=torch.float32, device=dev)
    lse = torch.empty(lead, dtype=torch.float32, device=dev)
    entropy = torch.empty(lead, dtype=torch.float32, device=dev) if with_entropy else None
    with torch.cuda.device(dev):
        _lib.check(_lib.load().rlx_token_logprob_fwd(x.data_ptr(), lab.data_ptr(), byref(rows), logprob.data_ptr(),
                                                     _ptr(entropy), lse.data_ptr(), _stream_ptr(dev)),
                   "rlx_token_logprob_fwd")
    return logprob, entropy, lse


def token_logprob_bwd(logits: torch.Tensor, labels: torch.Tensor, lse: torch.Tensor, entropy: Optional[torch.Tensor],
                      d_logprob: torch.Tensor, d_entropy: Optional[torch.Tensor], temperature: float = 1.0,
                      out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Gradient w.r.t. the UNSCALED logits, in their dtype.  ``out`` may be ``logits`` itself (in-place)."""
    dev = _dev(logits, labels, lse, d_logprob)
    x, rows = _rows_of(logits, temperature, False)
    if out is None:
        out = torch.empty(logits.shape, dtype=logits.dtype, device=dev)
    if out.shape != logits.shape or out.dtype != logits.dtype:
        raise RlxError("d_logits must match logits in shape and dtype")
    if rows.seq_stride == 0:  # flat contiguous logits: rows are addressed as i * vocab
        if not out.is_contiguous():
            raise RlxError("a strided d_logits needs logits of the same [bsz, rows, V] slicing")
        dss, drs = 0, rows.vocab
    elif out.dim() == 3 and out.stride(-1) == 1 and out.stride(1) >= rows.vocab:
        dss, drs = out.stride(0), out.stride(1)  # dense [bsz, rows, V] or a slice like the logits
    else:
        raise RlxError("d_logits must be contiguous or a [bsz, rows, V] slice like logits")
    with torch.cuda.device(dev):
        _lib.check(_lib.load().rlx_token_logprob_bwd(
            x.data_ptr(), labels.contiguous().data_ptr(), byref(rows), _as_f32(lse, "lse").data_ptr(),
            _ptr(_as_f32(entropy, "entropy")), _as_f32(d_logprob, "d_logprob").data_ptr(),
            _ptr(_as_f32(d_entropy, "d_entropy")), out.data_ptr(), dss, drs, _stream_ptr(dev)), "rlx_token_logprob_bwd")
    return out


class _TokenLogprobFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, labels, temperature, with_entropy, round_outputs, inplace_grad, window):
        # ``window`` = (start, stop) along dim 1 of a [bsz, S, V] tensor: the kernels address that slice in place and the backward
        # hands autograd a gradient for the WHOLE tensor (zeros outside the window, written by two fills of the outside rows) --
        # what SliceBackward would build as zeros(logits.shape) + a strided copy of the window gradient: 2 x [bsz, S, V] of
        # traffic and one more allocation per micro-batch.
        view = logits if window is None else logits[:, window[0]:window[1], :]
        logprob, entropy, lse = token_logprob_fwd(view, labels, temperature, with_entropy, round_outputs)
        ctx.save_for_backward(logits, labels, lse, entropy if with_entropy else lse)
        ctx.cfg = (float(temperature), bool(with_entropy), bool(inplace_grad), window)
        ctx.mark_non_differentiable(lse)
        if with_entropy:
            return logprob, entropy, lse
        return logprob, lse.new_empty(0), lse

    @staticmethod
    def backward(ctx, d_logprob, d_entropy, _d_lse):
        logits, labels, lse, entropy = ctx.saved_tensors
        temperature, with_entropy, inplace, window = ctx.cfg
        if d_logprob is None:
            d_logprob = torch.zeros_like(lse)
        use_ent = with_entropy and d_entropy is not None
        if window is None:
            dx = token_logprob_bwd(logits, labels, lse, entropy if use_ent else None, d_logprob.contiguous(),
                                   d_entropy.contiguous() if use_ent else None, temperature, out=logits if inplace else None)
            return dx, None, None, None, None, None, None
        full = logits if inplace else torch.empty_like(logits)
        token_logprob_bwd(logits[:, window[0]:window[1], :], labels, lse, entropy if use_ent else None, d_logprob.contiguous(),
                          d_entropy.contiguous() if use_ent else None, temperature, out=full[:, window[0]:window[1], :])
        full[:, :window[0], :].zero_()
        full[:, window[1]:, :].zero_()
        return full, None, None, None, None, None, None


class _PackedLogprobFn(torch.autograd.Function):
    """Packed-stream scoring with the unpack fused into the stores (include/rlx.h, rlx_token_logprob_fwd_packed)."""

    @staticmethod
    def forward(ctx, logits, labels, lp_dst, ent_dst, bsz, response_len, temperature, with_entropy, round_outputs, inplace_grad):
        dev = _dev(logits, labels, lp_dst)
        x, rows = _rows_of(logits, temperature, round_outputs)
        n = rows.n_tokens
        if labels.dtype != torch.int64 or labels.numel() != n or lp_dst.numel() != n or lp_dst.dtype != torch.int32:
            raise RlxError("packed scoring needs int64 labels and int32 destination maps with one entry per packed row")
        lab = labels.contiguous().reshape(-1)
        logprob = torch.zeros((bsz, response_len), dtype=torch.float32, device=dev)   # pad_val 0 of unpack_sequences
        entropy = torch.zeros((bsz, response_len), dtype=torch.float32, device=dev) if with_entropy else None
        lse = torch.empty(n, dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            _lib.check(_lib.load().rlx_token_logprob_fwd_packed(x.data_ptr(), lab.data_ptr(), byref(rows), lp_dst.data_ptr(),
                                                                _ptr(ent_dst if with_entropy else None), logprob.data_ptr(),
                                                                _ptr(entropy), lse.data_ptr(), _stream_ptr(dev)),
                       "rlx_token_logprob_fwd_packed")
        ctx.save_for_backward(logits, lab, lse, lp_dst, ent_dst, entropy if with_entropy else lse)
        ctx.cfg = (float(temperature), bool(with_entropy), bool(inplace_grad))
        ctx.out_shape = (int(bsz), int(response_len))
        return (logprob, entropy) if with_entropy else (logprob, lse.new_empty(0))

    @staticmethod
    def backward(ctx, d_logprob, d_entropy):
        logits, lab, lse, lp_dst, ent_dst, entropy = ctx.saved_tensors
        temperature, with_entropy, inplace = ctx.cfg
        dev = logits.device
        # the upstream gradients stay in their unpacked [bsz, response_len] shape: the kernel follows the forward's maps (a packed row
        # that fed nothing is written as zeros without being read)
        use_ent = with_entropy and d_entropy is not None
        if d_logprob is None:
            d_logprob = torch.zeros(ctx.out_shape, dtype=torch.float32, device=dev)
        x, rows = _rows_of(logits, temperature, False)
        out = logits if inplace else torch.empty(logits.shape, dtype=logits.dtype, device=dev)
        if rows.seq_stride == 0:
            if not out.is_contiguous():  # (a transposed lm_head output with inplace_grad: the forward scored a contiguous copy,
                #                          so there is no buffer of the logits' own layout to overwrite -- a fresh gradient instead)
                out = torch.empty(logits.shape, dtype=logits.dtype, device=dev)
            dss, drs = 0, rows.vocab
        else:
            dss, drs = out.stride(0), out.stride(1)
        with torch.cuda.device(dev):
            _lib.check(_lib.load().rlx_token_logprob_bwd_packed(
                x.data_ptr(), lab.data_ptr(), byref(rows), lse.data_ptr(), _ptr(entropy if use_ent else None), lp_dst.data_ptr(),
                _ptr(ent_dst if use_ent else None), _as_f32(d_logprob, "d_logprob").data_ptr(),
                _ptr(_as_f32(d_entropy, "d_entropy") if use_ent else None), out.data_ptr(), dss, drs, _stream_ptr(dev)),
                "rlx_token_logprob_bwd_packed")
        return out, None, None, None, None, None, None, None, None, None


def packed_token_logprobs(logits: torch.Tensor, packed_input_ids: torch.Tensor, idx_starts, idx_ends, *, max_seq_len_unpack: int,
                          response_len: int, eos_token_id: int, temperature: float = 1.0, with_entropy: bool = False,
                          round_outputs: bool = False, inplace_grad: bool = False):
    """``unpack_fsdp_logprobs`` + ``unpack_sequences`` of the entropy + the ``[:, -response_len:]`` slices in one scoring launch
    (rlinf/hybrid_engines/fsdp/utils.py:980-1022, rlinf/workers/actor/fsdp_actor_worker.py:482-503): ``logits`` [1, L, V] (or [L, V])
    of a packed stream, ``packed_input_ids`` [1, L] -> differentiable (logprobs [bsz, response_len], entropy or None), zeros where
    the reference pads."""
    from .hybrid_engines.fsdp.utils import unpack_index_maps
    ids = packed_input_ids.reshape(-1)
    L = ids.numel()
    if logits.numel() // max(logits.shape[-1], 1) != L:
        raise RlxError("packed logits and input_ids disagree on the stream length")
    labels = torch.cat([ids[1:], ids.new_full((1,), int(eos_token_id))]).to(torch.int64)  # token t + 1, eos behind the last (:1000-1009)
    lp_dst, ent_dst = unpack_index_maps(idx_starts, idx_ends, L, int(max_seq_len_unpack), int(response_len), logits.device)
    out = _PackedLogprobFn.apply(logits, labels, lp_dst, ent_dst, len(idx_starts), int(response_len), float(temperature),
                                 bool(with_entropy), bool(round_outputs), bool(inplace_grad))
    return (out[0], out[1]) if with_entropy else (out[0], None)


def token_logprobs(logits: torch.Tensor, labels: torch.Tensor, *, temperature: float = 1.0, with_entropy: bool = False,
                   round_outputs: bool = False, inplace_grad: bool = False, window: Optional[tuple] = None):
    """Differentiable (logprob, entropy or None).  ``inplace_grad`` lets the backward pass overwrite the logits
    buffer with its gradient (saves one [tokens, vocab] allocation; only valid when nothing else reads the logits
    afterwards, which holds for an lm_head output feeding only this op).  ``window`` = (start, stop): ``logits`` is the model's
    whole [bsz, S, V] output and only rows start..stop-1 of every sequence are scored (``labels`` [bsz, stop - start]); the
    gradient comes back for the whole tensor, zero outside the window."""
    if window is not None:
        if logits.dim() != 3 or not (0 <= window[0] < window[1] <= logits.shape[1]):
            raise RlxError(f"window {window} does not fit logits of shape {tuple(logits.shape)}")
        window = (int(window[0]), int(window[1]))
    logprob, entropy, _ = _TokenLogprobFn.apply(logits, labels, float(temperature), bool(with_entropy),
                                                bool(round_outputs), bool(inplace_grad), window)
    return logprob, (entropy if with_entropy else None)


# ---------------------------------------------------------------------------------------------------------
def make_token_loss_params(*, loss_agg: str, clip_ratio_low: float, clip_ratio_high: float, clip_ratio_c=None,
                           clip_log_ratio_min=None, clip_log_ratio_max=None, critic_warmup=False,
                           fast_path_zero_loss_mask=False, kl_penalty_type: Optional[str] = None, kl_beta: float = 0.0,
                           use_entropy: bool = False, entropy_bonus: float = 0.0) -> TokenLossParams:
    if loss_agg not in _lib.LOSS_AGG:
        raise ValueError(f"Unsupported loss aggregation method: {loss_agg}")
    if kl_penalty_type not in _lib.KL_TYPE:
        raise NotImplementedError(f"kl_penalty type {kl_penalty_type!r}")
    if clip_ratio_c is not None and not clip_ratio_c > 1.0:
        raise AssertionError("clip_ratio_c must be greater than 1.0")
    p = TokenLossParams()
    p.ppo = make_ppo_params(clip_ratio_low=clip_ratio_low, clip_ratio_high=clip_ratio_high, clip_ratio_c=clip_ratio_c,
                            clip_log_ratio_min=clip_log_ratio_min, clip_log_ratio_max=clip_log_ratio_max,
                            critic_warmup=critic_warmup, has_critic=False, action_dim=1)
    p.loss_agg = _lib.LOSS_AGG[loss_agg]
    p.fast_path_zero_loss_mask = int(bool(fast_path_zero_loss_mask))
    p.kl_type = _lib.KL_TYPE[kl_penalty_type] if kl_beta > 0 else 0
    p.kl_beta = float(kl_beta)
    p.use_entropy, p.entropy_bonus = int(bool(use_entropy)), float(entropy_bonus)
    return p


class _TokenLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logprobs, entropy, old_logprobs, advantages, ref_logprobs, loss_mask, params: TokenLossParams):
        lib = _lib.load()
        dev = logprobs.device
        bsz, seq = logprobs.shape
        g_lp = torch.empty((bsz, seq), dtype=torch.float32, device=dev)
        needs_gent = bool(params.use_entropy) and entropy is not None and params.entropy_bonus != 0.0
        g_ent = torch.empty((bsz, seq), dtype=torch.float32, device=dev) if needs_gent else None
        row_w = torch.empty((bsz,), dtype=torch.float32, device=dev)
        out = torch.empty((_lib.TOK_OUT_FLOATS,), dtype=torch.float32, device=dev)
        ws_bytes = lib.rlx_token_loss_workspace_bytes(bsz, seq)
        ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=dev)
        with torch.cuda.device(dev):
            _lib.check(lib.rlx_token_loss_fwd(logprobs.data_ptr(), old_logprobs.data_ptr(), advantages.data_ptr(),
                                              _ptr(ref_logprobs), _ptr(entropy), _ptr(loss_mask), bsz, seq, byref(params),
                                              g_lp.data_ptr(), _ptr(g_ent), row_w.data_ptr(), out.data_ptr(),
                                              ws.data_ptr(), ws_bytes, _stream_ptr(dev)), "rlx_token_loss_fwd")
        ctx.save_for_backward(g_lp, g_ent if needs_gent else row_w, row_w)
        ctx.needs_gent = needs_gent
        ctx.ent_given = entropy is not None
        ctx.mark_non_differentiable(out)
        return out[0].clone(), out

    @staticmethod
    def backward(ctx, grad_loss, _grad_out):
        lib = _lib.load()
        g_lp, g_ent, row_w = ctx.saved_tensors
        dev = g_lp.device
        bsz, seq = g_lp.shape
        go = grad_loss.to(dtype=torch.float32).contiguous()
        d_lp = torch.empty_like(g_lp)
        d_ent = torch.empty_like(g_lp) if ctx.needs_gent else None
        with torch.cuda.device(dev):
            _lib.check(lib.rlx_token_loss_bwd(g_lp.data_ptr(), g_ent.data_ptr() if ctx.needs_gent else None,
                                              row_w.data_ptr(), go.data_ptr(), d_lp.data_ptr(), _ptr(d_ent), bsz, seq,
                                              _stream_ptr(dev)), "rlx_token_loss_bwd")
        return d_lp, d_ent, None, None, None, None, None


def token_loss(logprobs: torch.Tensor, old_logprobs: torch.Tensor, advantages: torch.Tensor,
               loss_mask: Optional[torch.Tensor], params: TokenLossParams, *, entropy: Optional[torch.Tensor] = None,
               ref_logprobs: Optional[torch.Tensor] = None):
    """-> (loss scalar with grad into logprobs/entropy, out f32[16] in rlx_tok_out order)."""
    dev = _dev(logprobs, old_logprobs, advantages, loss_mask, entropy, ref_logprobs)
    if logprobs.dim() != 2:
        raise RlxError(f"logprobs must be [bsz, seq]; got {tuple(logprobs.shape)}")
    for name, t in (("logprobs", logprobs), ("old_logprobs", old_logprobs), ("advantages", advantages)):
        if t.dtype != torch.float32:
            raise AssertionError(f"{name} must be float32 to keep numerical stability")
        if t.shape != logprobs.shape:
            raise RlxError(f"{name} shape {tuple(t.shape)} != logprobs shape {tuple(logprobs.shape)}")
    del dev
    mask = _as_u8(loss_mask)
    if mask is not None and mask.shape != logprobs.shape:
        raise RlxError("loss_mask must be [bsz, seq]")
    return _TokenLossFn.apply(logprobs.contiguous(), _as_f32(entropy, "entropy"), old_logprobs.contiguous(),
                              advantages.contiguous(), _as_f32(ref_logprobs, "ref_logprobs"), mask, params)


def grpo_seq_adv(rewards: torch.Tensor, loss_mask: torch.Tensor, group_size: int, eps: float = 1e-6) -> torch.Tensor:
    """rewards [bsz] f32, loss_mask [bsz, seq] bool -> advantages [bsz, seq] f32."""
    dev = _dev(rewards, loss_mask)
    if loss_mask.dim() != 2 or rewards.numel() != loss_mask.shape[0]:
        raise RlxError("grpo_seq_adv: rewards [bsz], loss_mask [bsz, seq]")
    bsz, seq = loss_mask.shape
    if group_size < 1 or bsz % group_size != 0:
        raise RlxError(f"bsz {bsz} is not a multiple of group_size {group_size}")
    r = _as_f32(rewards.reshape(-1), "rewards")
    m = _as_u8(loss_mask)
    adv = torch.empty((bsz, seq), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        _lib.check(_lib.load().rlx_grpo_seq_adv(r.data_ptr(), m.data_ptr(), adv.data_ptr(), bsz, seq, int(group_size),
                                                float(eps), _stream_ptr(dev)), "rlx_grpo_seq_adv")
    return adv


def reinpp_seq_adv(rewards: torch.Tensor, loss_mask: torch.Tensor, logprob: Optional[torch.Tensor] = None,
                   ref_logprob: Optional[torch.Tensor] = None, kl_beta: float = 0.0, kl_penalty_type: Optional[str] = None
                   ) -> torch.Tensor:
    """Reinforce++ (no group baseline): rewards [bsz] f32, loss_mask [bsz, seq] bool, logprob / ref_logprob [bsz, seq] f32
    (only with kl_beta > 0) -> normalised advantages [bsz, seq] f32 (advantages.py:300-364 incl. the pre / post shaping)."""
    dev = _dev(rewards, loss_mask, logprob, ref_logprob)
    if loss_mask.dim() != 2 or rewards.numel() != loss_mask.shape[0]:
        raise RlxError("reinpp_seq_adv: rewards [bsz], loss_mask [bsz, seq]")
    bsz, seq = loss_mask.shape
    r = _as_f32(rewards.reshape(-1), "rewards")
    m = _as_u8(loss_mask)
    lp = rp = None
    kind = 0
    if kl_beta > 0:
        if logprob is None or ref_logprob is None:
            raise RlxError("reinpp_seq_adv: kl_beta > 0 needs logprob and ref_logprob")
        if kl_penalty_type not in _lib.KL_TYPE or kl_penalty_type is None:
            raise NotImplementedError(kl_penalty_type)  # kl_penalty raises the same for "full" / unknown (utils.py:58-64)
        kind = _lib.KL_TYPE[kl_penalty_type]
        lp, rp = _as_f32(logprob, "logprob"), _as_f32(ref_logprob, "ref_logprob")
        if tuple(lp.shape) != (bsz, seq) or tuple(rp.shape) != (bsz, seq):
            raise RlxError("reinpp_seq_adv: logprob / ref_logprob must be [bsz, seq]")
    adv = torch.empty((bsz, seq), dtype=torch.float32, device=dev)
    lib = _lib.load()
    ws_bytes = lib.rlx_reinpp_workspace_bytes(bsz)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        _lib.check(lib.rlx_reinpp_seq_adv(r.data_ptr(), m.data_ptr(), None if lp is None else lp.data_ptr(),
                                          None if rp is None else rp.data_ptr(), kind, float(kl_beta), adv.data_ptr(), bsz, seq,
                                          ws.data_ptr(), ws_bytes, _stream_ptr(dev)), "rlx_reinpp_seq_adv")
    return adv


def reference_softmax_lanes() -> int:
    """f32 lanes of the vector kernels torch's CPU softmax runs with on THIS host: 16 (AVX-512 builds) or 8 (AVX2).  The order of
    the additions in the reference's row sum -- and with it the last bit of softmax(x), and with that a sampled index on a near
    tie -- depends on it; ``categorical_sample`` replays that order (include/rlx.h, rlx_categorical_sample).  Builds whose vector
    width this package has not pinned (NEON, SVE, DEFAULT ...) get 16 with a warning."""
    try:
        cap = torch.backends.cpu.get_cpu_capability().upper()
    except Exception:  # noqa: BLE001
        cap = "?"
    if cap == "AVX2":
        return 8
    if cap != "AVX512":
        import warnings
        warnings.warn(f"torch CPU capability {cap!r}: its softmax summation order is not one this package replays; using 16 lanes")
    return 16


def default_softmax_lanes() -> int:
    """What ``categorical_sample`` uses when the caller names no lane count: a FIXED 16 (the C ABI's own default, the golden
    fixtures' order), so that the same seed samples the same tokens on every node of a job whatever CPUs the hosts have.
    ``RLX_SOFTMAX_LANES=host`` (or 8 / 16) opts in to matching THIS host's torch CPU softmax instead -- for side-by-side parity
    runs against a CPU reference on the same machine."""
    v = os.environ.get("RLX_SOFTMAX_LANES", "16").strip().lower()
    if v == "host":
        return reference_softmax_lanes()
    if v not in ("8", "16"):
        raise RlxError(f"RLX_SOFTMAX_LANES must be 8, 16 or host (got {v!r})")
    return int(v)


def categorical_sample(logits: torch.Tensor, noise: Optional[torch.Tensor] = None, *, temperature: float = 1.0,
                       top_k: int = -1, bin_centers: Optional[torch.Tensor] = None, with_logprob: bool = True,
                       round_outputs: bool = True, softmax_lanes: Optional[int] = None):
    """logits [..., K<=1024] (a view into the model's logits is fine) -> (tokens i64, logprobs f32 or None,
    actions f32 or None), each of shape ``logits.shape[:-1]``.  ``noise``: Exp(1) draws of the logits' dtype and
    shape (what torch.multinomial draws internally); None = argmax.  The sampled indices are bit-exact against the reference's
    CPU path on a host with ``softmax_lanes`` f32 SIMD lanes (None: default_softmax_lanes() -- 16 unless RLX_SOFTMAX_LANES says
    otherwise; ``reference_softmax_lanes()`` is this host's)."""
    dev = _dev(logits, noise, bin_centers)
    lead = logits.shape[:-1]
    x, rows = _rows_of(logits, temperature if noise is not None else 1.0, round_outputs)
    if rows.vocab > 1024:
        raise RlxError(f"categorical_sample handles up to 1024 categories per row (got {rows.vocab})")
    if noise is not None:
        if noise.dtype != logits.dtype or noise.shape != logits.shape:
            raise RlxError("noise must match the logits in dtype and shape")
        noise = noise.contiguous()
    centers = _as_f32(bin_centers, "bin_centers")
    tokens = torch.empty(lead, dtype=torch.int64, device=dev)
    logprob = torch.empty(lead, dtype=torch.float32, device=dev) if with_logprob else None
    actions = torch.empty(lead, dtype=torch.float32, device=dev) if centers is not None else None
    with torch.cuda.device(dev):
        _lib.check(_lib.load().rlx_categorical_sample(x.data_ptr(), byref(rows), _ptr(noise), int(top_k),
                                                      int(softmax_lanes if softmax_lanes is not None else default_softmax_lanes()),
                                                      _ptr(centers),
                                                      0 if centers is None else centers.numel(), tokens.data_ptr(),
                                                      _ptr(logprob), _ptr(actions), _stream_ptr(dev)),
                   "rlx_categorical_sample")
    return tokens, logprob, actions


def gae_seq(values: torch.Tensor, rewards: torch.Tensor, gamma: float = 1.0, gae_lambda: float = 1.0, out=None, workspace=None):
    """Reasoning GAE in the [bsz, seq] layout: values [bsz, seq] f32, rewards [bsz] -> (advantages, returns), un-normalised.
    ``out`` = (advantages, returns) and ``workspace`` (rlx_gae_seq_workspace_bytes) make the call allocation-free."""
    dev = _dev(values, rewards)
    if values.dim() != 2 or rewards.numel() != values.shape[0]:
        raise RlxError("gae_seq: values [bsz, seq], rewards [bsz]")
    v = _as_f32(values, "values")
    r = _as_f32(rewards.reshape(-1), "rewards")
    bsz, seq = v.shape
    adv, ret = (torch.empty_like(v), torch.empty_like(v)) if out is None else out
    lib = _lib.load()
    wsb = lib.rlx_gae_seq_workspace_bytes(bsz, seq)
    ws = workspace if workspace is not None and workspace.numel() >= wsb else torch.empty(wsb, dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        _lib.check(lib.rlx_gae_seq(v.data_ptr(), r.data_ptr(), adv.data_ptr(), ret.data_ptr(), bsz, seq, float(gamma),
                                   float(gamma * gae_lambda), ws.data_ptr(), ws.numel(), _stream_ptr(dev)), "rlx_gae_seq")
    return adv, ret

"""Sequence packing of the reasoning learner (rlinf/hybrid_engines/fsdp/utils.py:812-1010; used by FSDPActor.forward_batch with
``runner.enable_dynamic_batch_size`` or ``actor.model.variable_seq_lengths``, rlinf/workers/actor/fsdp_actor_worker.py:450-505).

A micro-batch arrives padded -- row i = [left pad | prompt_i | response_i | right pad] of ``encoder_seq_length`` tokens with the
prompt right-aligned at ``max_prompt_length`` -- and goes through the model as ONE packed stream of the valid windows
[idx_start_i, idx_end_i) back to back (flash-attention derives the sequence boundaries from ``position_ids``).  The reference then
computes a log-prob for every packed position, shifts it right by one, scatters every segment back into a zero-padded
[bsz, encoder_seq_length] matrix and keeps the last ``response_len`` columns (``unpack_fsdp_logprobs`` / ``unpack_sequences``).  Here
the pack side is host index arithmetic + two gathers, and the unpack side is an index map the scoring kernel stores through
(``rlx_token_logprob_fwd_packed``): the [bsz, response_len] results are written directly, prompt rows are never read.
"""

from __future__ import annotations

from typing import Optional, Sequence

import torch


def prepare_pack_fsdp(m_batch, max_prompt_len: int):
    """:917-934 -> (idx_starts, idx_ends): the valid window of every row, as host lists."""
    idx_starts = (max_prompt_len - m_batch["prompt_lengths"]).tolist()
    idx_ends = (max_prompt_len + m_batch["response_lengths"]).tolist()
    return idx_starts, idx_ends


def _packed_rows(idx_starts: Sequence[int], idx_ends: Sequence[int], device):
    """For every position of the packed stream (the windows back to back): (row, column in that row), built on ``device`` from the two
    window lists -- one small host-to-device copy of the 2 x bsz window bounds, no per-row aranges, no blocking gather."""
    a = torch.tensor([int(x) for x in idx_starts], dtype=torch.int64)
    b = torch.tensor([int(x) for x in idx_ends], dtype=torch.int64)
    lens = b - a
    total = int(lens.sum())
    bounds = torch.stack([a, lens, torch.cumsum(lens, 0) - lens]).to(device, non_blocking=True)
    a_d, lens_d, cu_d = bounds[0], bounds[1], bounds[2]
    row = torch.repeat_interleave(torch.arange(a.numel(), dtype=torch.int64, device=device), lens_d, output_size=total)
    col = torch.arange(total, dtype=torch.int64, device=device) - cu_d[row] + a_d[row]
    return row, col, total


def pack_sequences(input_tensor: torch.Tensor, idx_starts: Sequence[int], idx_ends: Sequence[int], max_seq_len: int, pad_val,
                   pad_to_fixed_len: bool = False) -> torch.Tensor:
    """:812-855: the windows back to back (1-D); ``pad_to_fixed_len`` pads the end to ``max_seq_len`` with ``pad_val``.  One gather
    instead of bsz slices + cat, its index built on the tensor's device."""
    assert input_tensor.dim() == 2
    assert input_tensor.shape[0] == len(idx_starts) == len(idx_ends)
    row, col, total = _packed_rows(idx_starts, idx_ends, input_tensor.device)
    assert total <= max_seq_len
    out = input_tensor.reshape(-1)[row * input_tensor.shape[1] + col]
    if pad_to_fixed_len and max_seq_len > total:
        out = torch.cat([out, torch.full((max_seq_len - total,), pad_val, dtype=input_tensor.dtype, device=input_tensor.device)])
    return out


def pack_fsdp_input(input_ids, position_ids, *, idx_starts, idx_ends, max_seq_len_pack, eos_token_id, pad_to_fixed_len: bool = False):
    """:937-977 -> (input_ids [1, L], position_ids [1, L], attention_mask None)."""
    ids = pack_sequences(input_ids, idx_starts, idx_ends, max_seq_len_pack, eos_token_id, pad_to_fixed_len).unsqueeze(0)
    pos = pack_sequences(position_ids, idx_starts, idx_ends, max_seq_len_pack, 0, pad_to_fixed_len).unsqueeze(0)
    return ids, pos, None


def unpack_index_maps(idx_starts: Sequence[int], idx_ends: Sequence[int], packed_len: int, max_seq_len_unpack: int, response_len: int,
                      device) -> tuple:
    """The unpack of ``unpack_fsdp_logprobs`` (:980-1022) + ``[:, -response_len:]`` as two int32 maps over the packed rows:
    ``lp_dst[t]``: where the log-prob computed FROM row t (of token t + 1) lands -- the reference shifts the packed log-probs
    right by one before scattering, so row t feeds packed position t + 1 --, ``ent_dst[t]``: where row t's entropy lands (not
    shifted: ``unpack_sequences(entropy, ...)``, fsdp_actor_worker.py:497-501).  Flat indices into [bsz, response_len]; -1 = the
    value is dropped (left of the response window, or padding of a fixed-length pack)."""
    first_col = max_seq_len_unpack - response_len
    row, col, total = _packed_rows(idx_starts, idx_ends, device)
    keep = col >= first_col                                   # columns left of the response window are dropped
    dst = torch.where(keep, row * response_len + (col - first_col), torch.full_like(col, -1)).to(torch.int32)
    lp = torch.full((packed_len,), -1, dtype=torch.int32, device=device)
    ent = torch.full((packed_len,), -1, dtype=torch.int32, device=device)
    ent[:total] = dst
    if total > 1:
        lp[:total - 1] = dst[1:]                              # packed position p >= 1 is fed by row p - 1 (position 0: the prepended zero)
    return lp, ent


def get_seqlen_bfd_partitions(seq_len_list: Sequence[int], max_tokens_per_mbs: int) -> list:
    """get_seqlen_BFD_partitions (rlinf/utils/data_iter_utils.py:447-503): best-fit-decreasing bins of at most
    ``max_tokens_per_mbs`` tokens; only the NUMBER of bins is used by the caller."""
    lens = [int(x) for x in seq_len_list]
    if any(x > max_tokens_per_mbs for x in lens):
        raise ValueError(f"Sequence length {max(lens)} exceeds the threshold {max_tokens_per_mbs}")
    order = sorted(range(len(lens)), key=lambda i: lens[i], reverse=True)  # stable, like list.sort(reverse=True)
    parts, room = [], []
    for i in order:
        best, best_left = -1, float("inf")
        for gidx, r in enumerate(room):
            if r >= lens[i] and r - lens[i] < best_left:
                best, best_left = gidx, r - lens[i]
        if best >= 0:
            parts[best].append(i)
            room[best] -= lens[i]
        else:
            parts.append([i])
            room.append(max_tokens_per_mbs - lens[i])
    return parts


def split_dynamic_batch_size(batch: dict, max_tokens_per_mbs: int, balanced_partitions, ctx=None):
    """split_dynamic_batch_size / get_iterator_dynamic for dict batches (data_iter_utils.py:505-600,675-700) -> (micro-batches,
    n_micro_batch, index partitions).  The count starts at the best-fit-decreasing bin count of the effective lengths (MAX over the
    data-parallel group), the sequences are dealt by the Karmarkar-Karp partitions (``equal_size=False``), and the count grows
    until no micro-batch exceeds the token budget on any rank.  ``balanced_partitions``: the learner's own
    ``seqlen_balanced_partitions`` (pinned to the reference's)."""
    import torch.distributed as dist
    multi = ctx is not None and ctx.world_size > 1
    seq_len = batch["attention_mask"].sum(dim=1).tolist()
    max_seq_len = batch["attention_mask"].shape[-1]
    assert max_tokens_per_mbs >= max_seq_len, (
        f"max_tokens_per_mbs must be greater than sequence length. Got {max_tokens_per_mbs=} and {max_seq_len=}")
    n = len(get_seqlen_bfd_partitions(seq_len, max_tokens_per_mbs))
    if multi:
        t = torch.tensor([n], device=batch["attention_mask"].device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        n = int(t.item())
    while True:
        parts = balanced_partitions(seq_len, n, False)
        micro, ok = [], 1
        for part in parts:
            idx = torch.as_tensor(part, dtype=torch.int64)
            cur = {}
            for k, v in batch.items():
                if isinstance(v, torch.Tensor):
                    cur[k] = v[idx.to(v.device)]
                elif isinstance(v, list):
                    cur[k] = [v[i] for i in part]
            micro.append(cur)
            if int(cur["prompt_lengths"].sum()) + int(cur["response_lengths"].sum()) > max_tokens_per_mbs:
                ok = 0
                break
        if multi:
            t = torch.tensor([ok], device=batch["attention_mask"].device)
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            ok = int(t.item())
        if ok:
            return micro, len(micro), parts
        n += 1


def get_reverse_idx(idx_map: Sequence[int]) -> list:
    """data_iter_utils.py:703-718."""
    out = list(idx_map)
    for i, idx in enumerate(idx_map):
        out[idx] = i
    return out

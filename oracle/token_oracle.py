"""CPU restatement of the reference's reasoning (LLM) token path.  TEST INFRASTRUCTURE ONLY.

SURVEY.md 8f item 1: per-token log-prob / entropy from vocabulary logits, the KL penalty, the three loss
aggregations, the reasoning pre/post-processing around the advantage registry and the composition of the
micro-batch loss in the reasoning learner.  Same rules as ``ppo_oracle.py``: stock torch CPU ops in the
reference's evaluation order, each function citing the file:line it follows (paths relative to
/root/reference); pinned against the real reference by ``tests/test_oracle_vs_reference.py`` (build
container) and against ``tests/golden/token_path.pt`` (everywhere).

Nothing under ``rlinf_amd/`` imports this file.
"""

from __future__ import annotations

from typing import Callable, Optional

import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------------------------
# t1  log-prob / entropy from logits     rlinf/utils/utils.py:454-512
# --------------------------------------------------------------------------------------------
def logprobs_from_logits(logits: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
    """op_type="torch": cross entropy in the logits' dtype, result cast to fp32 (utils.py:454-492)."""
    lead = logits.shape[:-1]
    flat = logits.reshape(-1, logits.shape[-1])
    out = -F.cross_entropy(flat, target.reshape(-1), reduction="none")
    return out.view(*lead).float()


def entropy_from_logits(logits: torch.Tensor, dim: int = -1) -> torch.Tensor:
    """H = -sum p log p via log_softmax, zero-probability terms dropped (utils.py:495-512)."""
    logp = F.log_softmax(logits, dim=dim)
    p = logp.exp()
    term = torch.where(p > 0, p * logp, 0.0)
    return -term.sum(dim=dim)


# --------------------------------------------------------------------------------------------
# t2  aggregations                        rlinf/utils/utils.py:307-381
# --------------------------------------------------------------------------------------------
def masked_mean(values, mask, axis=None):
    if mask is None:
        return values.mean() if axis is None else values.mean(dim=axis)
    tot = (values * mask).sum() if axis is None else (values * mask).sum(dim=axis)
    if (~mask).all():
        return tot
    return tot / (mask.sum() if axis is None else mask.sum(dim=axis))


def seq_mean_token_sum(values, mask, dim: int = -1):
    return torch.mean(torch.sum(values * mask, dim=-1))


def seq_mean_token_mean(values, mask, dim: int = -1):
    return torch.mean(torch.sum(values * mask, dim=-1) / torch.sum(mask, dim=-1))


def get_loss_agg_func(name: str) -> Callable:
    table = {"seq-mean-token-sum": seq_mean_token_sum, "seq-mean-token-mean": seq_mean_token_mean,
             "token-mean": masked_mean}
    if name not in table:
        raise ValueError(f"Unsupported loss aggregation method: {name}")
    return table[name]


# --------------------------------------------------------------------------------------------
# t3  KL penalty                          rlinf/algorithms/utils.py:26-64
# --------------------------------------------------------------------------------------------
def kl_penalty(logprob, ref_logprob, kind: str):
    if kind in ("kl", "k1"):
        return logprob - ref_logprob
    if kind == "abs":
        return (logprob - ref_logprob).abs()
    if kind in ("mse", "k2"):
        return 0.5 * (logprob - ref_logprob).square()
    if kind in ("low_var_kl", "k3"):
        kl = torch.clamp(ref_logprob - logprob, min=-20, max=20)
        kld = (torch.exp(kl) - kl - 1).contiguous()
        return torch.clamp(kld, min=-10, max=10)
    raise NotImplementedError(kind)


# --------------------------------------------------------------------------------------------
# t4  actor loss with a caller-chosen aggregation   rlinf/algorithms/losses.py:170-312
# --------------------------------------------------------------------------------------------
ZERO_MASK_KEYS = ("actor/token_num", "actor/policy_loss", "actor/policy_loss_mbs_mean", "actor/policy_loss_abs",
                  "actor/ratio", "actor/clipped_ratio", "actor/dual_cliped_ratio", "actor/approx_kl",
                  "actor/clip_fraction")


def token_actor_loss(logprobs, old_logprobs, advantages, clip_ratio_low, clip_ratio_high, loss_mask=None,
                     clip_ratio_c=None, loss_agg_func=masked_mean, clip_log_ratio_min=None,
                     clip_log_ratio_max=None, fast_path_zero_loss_mask=False, critic_warmup=False):
    if fast_path_zero_loss_mask and loss_mask is not None and loss_mask[0].sum() == 0.0:  # losses.py:206-219
        return torch.tensor(0.0), {k: torch.tensor(0.0) for k in ZERO_MASK_KEYS}
    if loss_mask is None:
        loss_mask = torch.ones_like(logprobs).bool()
    n_valid = loss_mask.count_nonzero() or 1
    log_ratio = logprobs - old_logprobs
    if clip_log_ratio_min is not None:
        log_ratio = torch.clamp(log_ratio, min=clip_log_ratio_min)
    if clip_log_ratio_max is not None:
        log_ratio = torch.clamp(log_ratio, max=clip_log_ratio_max)
    ratio = torch.where(loss_mask, torch.exp(log_ratio), 0)
    kl_terms = torch.where(loss_mask, log_ratio.detach(), 0.0)
    clipped = torch.clamp(ratio, 1.0 - clip_ratio_low, 1.0 + clip_ratio_high)
    surr1 = -advantages * ratio
    surr2 = -advantages * clipped
    is_clipped = surr1.detach() < surr2.detach()
    per_tok = torch.max(surr1, surr2)
    if clip_ratio_c is not None:
        surr3 = torch.sign(advantages) * clip_ratio_c * advantages
        is_dual = surr3.detach() < per_tok.detach()
        per_tok = torch.min(per_tok, surr3)
    else:
        is_dual = torch.zeros_like(is_clipped)
    loss_abs = loss_agg_func(per_tok.abs(), loss_mask, None)
    loss = loss_agg_func(per_tok, loss_mask, None)
    is_dual = (is_dual * loss_mask).bool()
    clip_fraction = (is_clipped * loss_mask).sum() / float(n_valid)
    approx_kl = -torch.sum(kl_terms) / float(n_valid)
    dual_ratio = torch.where(is_dual, ratio, 0)
    if critic_warmup:
        loss = torch.tensor(0.0)
    metrics = {
        "actor/policy_loss": loss.detach(),
        "actor/policy_loss_abs": loss_abs.detach(),
        "actor/ratio": masked_mean(ratio.detach(), loss_mask),
        "actor/ratio_abs": masked_mean((ratio - 1).abs().detach(), loss_mask),
        "actor/clipped_ratio": masked_mean(clipped.detach(), loss_mask),
        "actor/dual_cliped_ratio": masked_mean(dual_ratio.detach(), loss_mask),
        "actor/approx_kl": approx_kl.detach(),
        "actor/clip_fraction": clip_fraction.detach(),
    }
    return loss, metrics


# --------------------------------------------------------------------------------------------
# t5  the reasoning learner's micro-batch loss   rlinf/workers/actor/fsdp_actor_worker.py:476-508,694-790
# --------------------------------------------------------------------------------------------
def reasoning_micro_batch_loss(logits, responses, old_logprobs, advantages, loss_mask, *, temperature=1.0,
                               loss_agg="token-mean", clip_ratio_low=0.2, clip_ratio_high=0.2, clip_ratio_c=3.0,
                               clip_log_ratio_min=None, clip_log_ratio_max=None, calculate_entropy=False,
                               entropy_bonus=0.0, ref_logprobs=None, kl_beta=0.0, kl_penalty_type="low_var_kl",
                               gradient_accumulation=1):
    """logits [bsz, resp, V] are the response-aligned slice (``logits[:, -resp-1:-1]``); returns
    (loss_for_backward, metrics) — the loss already divided by ``gradient_accumulation``."""
    logits = logits / temperature  # the reference divides in place (fsdp_actor_worker.py:478)
    logprobs = logprobs_from_logits(logits, responses)
    entropy = entropy_from_logits(logits) if calculate_entropy else None
    loss, metrics = reasoning_loss_from_logprobs(
        logprobs, entropy, old_logprobs, advantages, loss_mask, loss_agg=loss_agg, clip_ratio_low=clip_ratio_low,
        clip_ratio_high=clip_ratio_high, clip_ratio_c=clip_ratio_c, clip_log_ratio_min=clip_log_ratio_min,
        clip_log_ratio_max=clip_log_ratio_max, entropy_bonus=entropy_bonus, ref_logprobs=ref_logprobs, kl_beta=kl_beta,
        kl_penalty_type=kl_penalty_type, gradient_accumulation=gradient_accumulation)
    return loss, metrics, logprobs, entropy


def reasoning_loss_from_logprobs(logprobs, entropy, old_logprobs, advantages, loss_mask, *, loss_agg="token-mean", clip_ratio_low=0.2,
                                 clip_ratio_high=0.2, clip_ratio_c=3.0, clip_log_ratio_min=None, clip_log_ratio_max=None,
                                 entropy_bonus=0.0, ref_logprobs=None, kl_beta=0.0, kl_penalty_type="low_var_kl", gradient_accumulation=1):
    """The part of training_step behind forward_batch (:694-781): PPO token loss, entropy bonus, KL penalty, / accumulation.
    ``entropy`` None = calculate_entropy off."""
    agg = get_loss_agg_func(loss_agg)
    calculate_entropy = entropy is not None
    loss, metrics = token_actor_loss(
        logprobs, old_logprobs, advantages, clip_ratio_low, clip_ratio_high, loss_mask=loss_mask,
        clip_ratio_c=clip_ratio_c, loss_agg_func=agg, clip_log_ratio_min=clip_log_ratio_min,
        clip_log_ratio_max=clip_log_ratio_max, fast_path_zero_loss_mask=True)
    entropy_loss = torch.tensor(0.0)
    if calculate_entropy:
        entropy_loss = agg(entropy, mask=loss_mask)
        if entropy_bonus != 0.0:  # calculate_entropy_loss (fsdp_actor_worker.py:134-137)
            loss = loss - entropy_bonus * entropy_loss
    kl_loss = torch.tensor(0.0)
    if kl_beta > 0 and ref_logprobs is not None:
        kld = kl_penalty(ref_logprobs, logprobs, kl_penalty_type)
        kl_loss = agg(kld, loss_mask)
        loss = loss + kl_loss * kl_beta
    metrics.update({"actor/final_loss": loss.detach(), "actor/entropy_loss": entropy_loss.detach(),
                    "actor/kl_loss": kl_loss.detach()})
    return loss / gradient_accumulation, metrics


# --------------------------------------------------------------------------------------------
# t6  reasoning advantage shaping        rlinf/algorithms/utils.py:177-277, advantages.py:89-121
# --------------------------------------------------------------------------------------------
def preprocess_reasoning(rewards, loss_mask, adv_type, values=None, group_size=None):
    """-> dict(rewards, loss_mask [seq,bsz], dones [seq+1,bsz], values [seq+1,bsz] or absent)."""
    bsz, seq = loss_mask.shape
    out = {"loss_mask": loss_mask.transpose(0, 1)}
    assert rewards.ndim == 1
    if adv_type == "gae":
        expanded = torch.zeros((seq, bsz), dtype=rewards.dtype)
        expanded[-1] = rewards
        out["rewards"] = expanded
    elif adv_type == "grpo":
        out["rewards"] = rewards.reshape(-1, group_size).contiguous()
    else:
        raise AssertionError(adv_type)
    if values is not None:
        v = values.transpose(0, 1)
        out["values"] = torch.cat([v, torch.zeros((1, v.shape[-1]), dtype=v.dtype)], dim=0)
    dones = torch.zeros(seq + 1, bsz, dtype=torch.bool)
    dones[-1] = True
    out["dones"] = dones
    return out


def grpo_reasoning_advantages(rewards, loss_mask, group_size: int):
    """rewards [bsz], loss_mask [bsz, seq] -> advantages [bsz, seq] (pre + advantages.py:89-121 + post)."""
    pre = preprocess_reasoning(rewards, loss_mask, "grpo", group_size=group_size)
    grouped = pre["rewards"].view(-1, group_size)
    mean = grouped.mean(dim=-1, keepdim=True).expand_as(grouped)
    std = grouped.std(dim=-1, keepdim=True).expand_as(grouped)
    adv = (grouped - mean) / (std + 1e-6)
    adv = (torch.zeros_like(pre["loss_mask"]) + adv.view(1, -1)) * pre["loss_mask"]
    return adv.transpose(0, 1).contiguous()


def reinpp_reasoning_advantages(rewards, loss_mask, group_size: int, use_reinpp_baseline: bool = False, kl_beta: float = 0.0,
                                logprob=None, ref_logprob=None, kl_penalty_type: str = ""):
    """Reinforce++ on reasoning batches: rewards [bsz], loss_mask / logprob / ref_logprob [bsz, seq] -> advantages [bsz, seq]
    (preprocess utils.py:218-219,245-251 + compute_reinpp_advantages advantages.py:300-364 + post utils.py:265-277).

    Restated as written, including what looks unintended: the "position of the last True" is read off ``fliplr`` of the
    [seq, bsz] mask -- a flip of the BATCH axis -- so sequence b's reward lands at seq-1 minus the index of the FIRST True
    of sequence bsz-1-b's mask (seq-1 whenever that mask starts with True, which response masks do); and
    ``use_reinpp_baseline`` ends in a 1-D ``src`` for a 2-D scatter index, which torch rejects."""
    assert rewards.ndim == 1
    mask = loss_mask.transpose(0, 1)  # [seq, bsz]
    rewards = rewards.unsqueeze(0)
    if use_reinpp_baseline:
        grouped = rewards.view(-1, group_size)
        grouped -= grouped.mean(dim=1, keepdims=True)  # (in place: the caller's rewards change, as in the reference)
        rewards = grouped.view(-1)
    r = torch.zeros_like(mask).float()
    seq = mask.size(0)
    eos = seq - 1 - mask.long().fliplr().argmax(dim=0, keepdim=True)
    r = r.scatter_(dim=0, index=eos, src=rewards)
    if kl_beta > 0:
        r -= kl_beta * kl_penalty(logprob.transpose(0, 1), ref_logprob.transpose(0, 1), kl_penalty_type)
    ret = torch.cumsum(r.flip(dims=[0]), dim=0).flip(dims=[0])
    mean = masked_mean(ret, mask)
    var = masked_mean((ret - mean).pow(2), mask)
    adv = (ret - mean) * var.clamp(min=1e-8).rsqrt()
    return adv.transpose(0, 1).contiguous()


# --------------------------------------------------------------------------------------------
# t6b  sequence packing of the reasoning learner   rlinf/hybrid_engines/fsdp/utils.py:812-1022,
#      FSDPActor.forward_batch's packed branch rlinf/workers/actor/fsdp_actor_worker.py:450-503
# --------------------------------------------------------------------------------------------
def prepare_pack(m_batch, max_prompt_len: int):
    """prepare_pack_fsdp :917-934: the valid window [idx_start, idx_end) of every padded row."""
    return (max_prompt_len - m_batch["prompt_lengths"]).tolist(), (max_prompt_len + m_batch["response_lengths"]).tolist()


def pack_sequences(x, idx_starts, idx_ends, max_seq_len: int, pad_val, pad_to_fixed_len: bool = False):
    """pack_sequences :812-855: the windows back to back, optionally padded at the end to max_seq_len."""
    assert sum(idx_ends) - sum(idx_starts) <= max_seq_len
    parts = [x[i, a:b] for i, (a, b) in enumerate(zip(idx_starts, idx_ends))]
    if pad_to_fixed_len:
        pad = max_seq_len - (sum(idx_ends) - sum(idx_starts))
        if pad > 0:
            parts.append(torch.full((pad,), pad_val, dtype=x.dtype))
    return torch.cat(parts)


def unpack_sequences(x, idx_starts, idx_ends, max_seq_len: int, pad_val):
    """unpack_sequences :858-914: packed [1, L] -> [bsz, max_seq_len], segment i at columns [idx_start_i, idx_end_i), pad elsewhere."""
    rows, cu = [], 0
    for a, b in zip(idx_starts, idx_ends):
        seg = x[0, cu:cu + (b - a)]
        cu += b - a
        rows.append(torch.cat([torch.full((a,), pad_val, dtype=x.dtype), seg, torch.full((max_seq_len - b,), pad_val, dtype=x.dtype)]))
    return torch.stack(rows)


def unpack_logprobs(logits, packed_ids, idx_starts, idx_ends, max_seq_len_unpack: int, eos_token_id: int):
    """unpack_fsdp_logprobs :980-1022: targets = the packed ids shifted left (eos behind the last), log-probs of every packed
    position, shifted RIGHT by one (a zero in front), scattered back to the padded layout with zeros."""
    responses = torch.cat([packed_ids[:, 1:], torch.full((1, 1), eos_token_id, dtype=packed_ids.dtype)], dim=1)
    lp = logprobs_from_logits(logits, responses)
    lp = torch.cat([torch.zeros((1, 1), dtype=logits.dtype), lp[:, :-1]], dim=-1)
    return unpack_sequences(lp, idx_starts, idx_ends, max_seq_len_unpack, 0)


def packed_forward(model, m_batch, *, max_prompt_len: int, encoder_seq_length: int, response_len: int, max_tokens_per_mbs: int,
                   variable_seq_lengths: bool, eos_token_id: int, temperature: float, calculate_entropy: bool):
    """forward_batch with packing on (:450-503) -> (logprobs [bsz, max_response_len], entropy [bsz, response_len] | None).  Note
    what the reference does with the entropy: it is computed per PACKED position and unpacked WITHOUT the one-position shift the
    log-probs get, then cut to the last response_len columns."""
    idx_starts, idx_ends = prepare_pack(m_batch, max_prompt_len)
    ids = pack_sequences(m_batch["input_ids"], idx_starts, idx_ends, max_tokens_per_mbs, eos_token_id, not variable_seq_lengths).unsqueeze(0)
    pos = pack_sequences(m_batch["position_ids"], idx_starts, idx_ends, max_tokens_per_mbs, 0, not variable_seq_lengths).unsqueeze(0)
    logits = model(input_ids=ids, attention_mask=None, position_ids=pos, use_cache=False).logits
    logits = logits / temperature
    lp = unpack_logprobs(logits, ids, idx_starts, idx_ends, encoder_seq_length, eos_token_id)[:, -(encoder_seq_length - max_prompt_len):]
    ent = None
    if calculate_entropy:
        ent = unpack_sequences(entropy_from_logits(logits), idx_starts, idx_ends, encoder_seq_length, 0)[:, -response_len:]
    return lp, ent


def bfd_partitions(seq_len_list, max_tokens_per_mbs: int):
    """get_seqlen_BFD_partitions rlinf/utils/data_iter_utils.py:447-503 (best fit decreasing)."""
    order = sorted(range(len(seq_len_list)), key=lambda i: seq_len_list[i], reverse=True)
    parts, room = [], []
    for i in order:
        n = seq_len_list[i]
        if n > max_tokens_per_mbs:
            raise ValueError(f"Sequence length {n} exceeds the threshold {max_tokens_per_mbs}")
        best, left = -1, float("inf")
        for g, r in enumerate(room):
            if r >= n and r - n < left:
                best, left = g, r - n
        if best >= 0:
            parts[best].append(i)
            room[best] -= n
        else:
            parts.append([i])
            room.append(max_tokens_per_mbs - n)
    return parts


# --------------------------------------------------------------------------------------------
# t7  categorical action sampling (K2)   rlinf/models/embodiment/openvla_oft/official/openvla_oft_action_model.py:363-414
#     (_discrete_prediction's sampling branch + _compute_logprobs_and_entropy :258-287)
# --------------------------------------------------------------------------------------------
def categorical_sample(action_logits, noise=None, temperature=1.0, top_k=-1, bin_centers=None):
    """action_logits [..., K] (the n_action_bins window of the vocabulary) -> (tokens, logprobs, processed logits,
    normalized actions or None).  ``noise`` is the Exp(1) draw torch.multinomial makes internally: for num_samples=1 it
    returns argmax(probs / q) with q = empty_like(probs).exponential_(1) (verified against torch.multinomial with the
    same generator state in tests/test_oracle_vs_reference.py), so passing q in makes the choice a pure function.
    ``noise=None`` is do_sample=False (argmax of the raw logits)."""
    if noise is None:
        tokens = action_logits.argmax(dim=-1)
        processed = action_logits
    else:
        assert temperature > 0
        processed = action_logits / temperature
        k = min(top_k, processed.size(-1))
        if k > 0:  # transformers.TopKLogitsWarper: everything below the k-th largest score becomes -inf
            kth = torch.topk(processed, k)[0][..., -1, None]
            processed = processed.masked_fill(processed < kth, float("-inf"))
        probs = torch.softmax(processed, dim=-1)
        tokens = (probs / noise).argmax(dim=-1)
    logprobs = logprobs_from_logits(processed, tokens)
    actions = None
    if bin_centers is not None:  # :397-403: token -> bin index counted from the end of the vocabulary
        nbins = action_logits.shape[-1]
        disc = torch.clamp(nbins - tokens - 1, min=0, max=bin_centers.shape[0] - 1)
        actions = bin_centers[disc]
    return tokens, logprobs, processed, actions

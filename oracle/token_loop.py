"""CPU oracle of ONE reasoning (LLM GRPO / Reinforce++) learner iteration.  TEST INFRASTRUCTURE.

Restates the control flow of FSDPActor.run_training / training_step / forward_batch (rlinf/workers/actor/fsdp_actor_worker.py
:860-939, :659-813, :434-505) with the token arithmetic of token_oracle.py and stock torch CPU ops:
  advantages      compute_advantages_and_returns :941-978 (grpo | reinpp) when the batch carries none
  normalisation   masked_normalization over the whole rank batch (rlinf/utils/distributed.py:866-930; one rank: its own sums)
  shuffle         get_iterator_k_split(shuffle=True, shuffle_seed=actor.seed) -- randperm from a freshly seeded generator
  global batches  n_minibatches equal row ranges, each cut into micro-batches of actor.micro_batch_size
  micro-batch     model(...).logits / temperature -> response window -> log-prob / entropy -> PPO token loss (+ entropy bonus,
                  + KL to the reference policy) / gradient_accumulation -> backward
  optimizer step  clip_grad_norm_ + AdamW, skipped when the norm is not finite (fsdp_model_manager.py:429-463)
  pipeline mode   pipeline_iteration: run_training_pipeline :816-854 over BatchResizingIterator (reasoning_results.py:1424-1600)
tests/test_reference_reasoning_loop.py pins it bit for bit against the reference's own run_training / training_step /
forward_batch compiled from source around the same tiny model; the HIP learner (rlinf_amd.workers.actor.fsdp_actor_worker.
FSDPActor) is compared with THIS loop on the GPU (tests/test_gpu_reasoning_loop.py).  Nothing in rlinf_amd/ imports it."""

from __future__ import annotations

import torch

from . import ppo_oracle as O
from . import token_oracle as TO


class TinyCausalLM(torch.nn.Module):
    """A stand-in for the transformer: token + position embedding, one tanh layer, an untied vocabulary projection.  It only has
    to turn (input_ids, position_ids) into logits [bsz, seq, vocab] that depend on its parameters -- the learner's path starts
    at those logits."""

    def __init__(self, vocab: int, dim: int, max_len: int):
        super().__init__()
        self.tok = torch.nn.Embedding(vocab, dim)
        self.pos = torch.nn.Embedding(max_len, dim)
        self.mix = torch.nn.Linear(dim, dim)
        self.head = torch.nn.Linear(dim, vocab, bias=False)

    def forward(self, input_ids=None, attention_mask=None, position_ids=None, use_cache=False, **_):
        h = torch.tanh(self.mix(self.tok(input_ids) + self.pos(position_ids)))
        return type("Out", (), {"logits": self.head(h)})()


def synthetic_rollout_batch(seed: int, bsz: int, prompt_len: int, response_len: int, vocab: int, group_size: int = 1):
    """One rank's rollout batch in the reference's field names (RolloutResult.to_actor_batch): left-padded prompts, right-padded
    responses, response_mask over the real response tokens, per-sequence rewards, rollout log-probs."""
    g = torch.Generator().manual_seed(seed)
    S = prompt_len + response_len
    input_ids = torch.randint(1, vocab, (bsz, S), generator=g)
    plen = torch.randint(2, prompt_len + 1, (bsz,), generator=g)
    rlen = torch.randint(1, response_len + 1, (bsz,), generator=g)
    pos = torch.arange(S).unsqueeze(0)
    attn = (pos >= (prompt_len - plen).unsqueeze(1)) & (pos < (prompt_len + rlen).unsqueeze(1))
    response_mask = (pos >= prompt_len) & (pos < (prompt_len + rlen).unsqueeze(1))
    input_ids = torch.where(attn, input_ids, torch.zeros_like(input_ids))
    position_ids = (attn.long().cumsum(dim=1) - 1).clamp(min=0)
    rewards = torch.randint(0, 2, (bsz,), generator=g).float() * 5.0 - 2.5 + torch.randn(bsz, generator=g) * 0.1
    return dict(input_ids=input_ids, attention_mask=attn, position_ids=position_ids, response_mask=response_mask,
                rewards=rewards, rollout_logprobs=-torch.rand(bsz, response_len, generator=g) * 2.0,
                prompt_lengths=plen, response_lengths=rlen, is_end=(rlen < response_len))


def forward_logprobs(model, m_batch: dict, response_len: int, temperature: float, calculate_entropy: bool = False):
    """forward_batch's fixed-length branch (:476-505)."""
    logits = model(input_ids=m_batch["input_ids"], attention_mask=m_batch["attention_mask"], position_ids=m_batch["position_ids"],
                   use_cache=False).logits
    logits = logits / temperature
    window = logits[:, -response_len - 1:-1, :]
    lp = TO.logprobs_from_logits(window, m_batch["input_ids"][:, -response_len:])
    return (lp, TO.entropy_from_logits(window)) if calculate_entropy else lp


def dynamic_micro_batches(batch: dict, max_tokens_per_mbs: int, balanced_partitions):
    """split_dynamic_batch_size / get_iterator_dynamic for one rank (rlinf/utils/data_iter_utils.py:505-600,675-700): the number of
    micro-batches starts at the best-fit-decreasing bin count of the effective lengths and grows until none exceeds the token
    budget; the sequences are dealt by the Karmarkar-Karp partitions (equal_size False).  -> (micro-batches, partitions)."""
    lens = batch["attention_mask"].sum(dim=1).tolist()
    assert max_tokens_per_mbs >= batch["attention_mask"].shape[-1]
    n = len(TO.bfd_partitions(lens, max_tokens_per_mbs))
    while True:
        parts = balanced_partitions(lens, n, False)
        micro = [{k: torch.stack([v[i] for i in part]) for k, v in batch.items() if isinstance(v, torch.Tensor)} for part in parts]
        if all(int(m["prompt_lengths"].sum()) + int(m["response_lengths"].sum()) <= max_tokens_per_mbs for m in micro):
            return micro, parts
        n += 1


def advantages(batch: dict, *, response_len: int, adv_type: str, group_size: int, reinpp_kl_beta: float = 0.0,
               kl_penalty_type: str = "low_var_kl"):
    if batch.get("advantages") is None:
        mask = batch["response_mask"][:, -response_len:]
        logprob = batch.get("recomputed_logprobs")
        if logprob is None:
            logprob = batch.get("rollout_logprobs")
        if adv_type == "grpo":
            batch["advantages"] = TO.grpo_reasoning_advantages(batch["rewards"], mask, group_size)
        elif adv_type == "reinpp":
            batch["advantages"] = TO.reinpp_reasoning_advantages(batch["rewards"].clone(), mask, group_size, False, reinpp_kl_beta,
                                                                 logprob, batch.get("ref_logprobs"), kl_penalty_type)
        else:
            raise ValueError(adv_type)
    return batch


def iteration(model, opt, batch: dict, *, response_len: int, micro_batch: int, n_minibatches: int, seed: int, adv_type: str = "grpo",
              group_size: int = 1, normalize_advantages: bool = True, shuffle: bool = True, temperature: float = 1.0,
              loss_agg: str = "token-mean", clip_ratio_low: float = 0.2, clip_ratio_high: float = 0.2, clip_ratio_c: float = 3.0,
              calculate_entropy: bool = False, entropy_bonus: float = 0.0, kl_beta: float = 0.0, kl_penalty_type: str = "low_var_kl",
              clip_grad: float = 1.0, reinpp_kl_beta: float = 0.0, pack: dict | None = None):
    """-> (the shuffled global batch incl. advantages, per-optimizer-step metric dicts).  ``pack``: sequence packing on -- dict(
    max_prompt_len, encoder_seq_length, max_tokens_per_mbs, variable_seq_lengths, eos_token_id[, dynamic: the balanced-partition
    function -> runner.enable_dynamic_batch_size])."""
    batch = advantages(dict(batch), response_len=response_len, adv_type=adv_type, group_size=group_size,
                       reinpp_kl_beta=reinpp_kl_beta, kl_penalty_type=kl_penalty_type)
    if normalize_advantages:
        batch["advantages"] = O.masked_normalization(batch["advantages"], batch["response_mask"][:, -response_len:]).float()
    bsz = batch["input_ids"].shape[0]
    if shuffle:
        perm = torch.randperm(bsz, generator=torch.Generator().manual_seed(seed))
        batch = {k: (v[perm] if isinstance(v, torch.Tensor) and v.shape[:1] == (bsz,) else v) for k, v in batch.items()}
    per = bsz // n_minibatches
    out = []
    for i in range(n_minibatches):
        mini = {k: v[i * per:(i + 1) * per] for k, v in batch.items() if isinstance(v, torch.Tensor)}
        accum = per // micro_batch
        if pack is not None and pack.get("dynamic") is not None:
            micro, _ = dynamic_micro_batches(mini, pack["max_tokens_per_mbs"], pack["dynamic"])
        else:
            micro = [{k: v[j * micro_batch:(j + 1) * micro_batch] for k, v in mini.items()} for j in range(accum)]
        out.append(optimizer_step(model, opt, micro, response_len=response_len, temperature=temperature, loss_agg=loss_agg,
                                  clip_ratio_low=clip_ratio_low, clip_ratio_high=clip_ratio_high, clip_ratio_c=clip_ratio_c,
                                  calculate_entropy=calculate_entropy, entropy_bonus=entropy_bonus, kl_beta=kl_beta,
                                  kl_penalty_type=kl_penalty_type, clip_grad=clip_grad, pack=pack))
    return batch, out


def optimizer_step(model, opt, micro_batches: list, *, response_len: int, temperature: float, loss_agg: str, clip_ratio_low: float,
                   clip_ratio_high: float, clip_ratio_c: float, calculate_entropy: bool, entropy_bonus: float, kl_beta: float,
                   kl_penalty_type: str, clip_grad: float, pack: dict | None = None) -> dict:
    """training_step (:659-813) over the micro-batches of one global batch -> its metric dict."""
    accum = len(micro_batches)
    opt.zero_grad()
    rows = []
    for mb in micro_batches:
        old = mb.get("recomputed_logprobs")
        if old is None:
            old = mb["rollout_logprobs"]
        if pack is not None:  # forward_batch's packed branch: log-probs / entropy arrive already unpacked to [bsz, response_len]
            lp, ent = TO.packed_forward(model, mb, max_prompt_len=pack["max_prompt_len"], encoder_seq_length=pack["encoder_seq_length"],
                                        response_len=response_len, max_tokens_per_mbs=pack["max_tokens_per_mbs"],
                                        variable_seq_lengths=pack["variable_seq_lengths"], eos_token_id=pack["eos_token_id"],
                                        temperature=temperature, calculate_entropy=calculate_entropy)
            loss, metrics = TO.reasoning_loss_from_logprobs(
                lp, ent, old, mb["advantages"], mb["response_mask"][:, -response_len:], loss_agg=loss_agg, clip_ratio_low=clip_ratio_low,
                clip_ratio_high=clip_ratio_high, clip_ratio_c=clip_ratio_c, entropy_bonus=entropy_bonus,
                ref_logprobs=mb.get("ref_logprobs"), kl_beta=kl_beta, kl_penalty_type=kl_penalty_type, gradient_accumulation=accum)
            loss.backward()
            rows.append(metrics)
            continue
        logits = model(input_ids=mb["input_ids"], attention_mask=mb["attention_mask"], position_ids=mb["position_ids"],
                       use_cache=False).logits
        loss, metrics, _, _ = TO.reasoning_micro_batch_loss(
            logits[:, -response_len - 1:-1, :], mb["input_ids"][:, -response_len:], old, mb["advantages"],
            mb["response_mask"][:, -response_len:], temperature=temperature, loss_agg=loss_agg, clip_ratio_low=clip_ratio_low,
            clip_ratio_high=clip_ratio_high, clip_ratio_c=clip_ratio_c, calculate_entropy=calculate_entropy,
            entropy_bonus=entropy_bonus, ref_logprobs=mb.get("ref_logprobs"), kl_beta=kl_beta, kl_penalty_type=kl_penalty_type,
            gradient_accumulation=accum)
        loss.backward()
        rows.append(metrics)
    gn = torch.nn.utils.clip_grad_norm_(model.parameters(), clip_grad)
    if torch.isfinite(gn):
        opt.step()
    m = {k: float(torch.mean(torch.stack([torch.as_tensor(r[k], dtype=torch.float32) for r in rows]))) for k in rows[0]}
    m["actor/grad_norm"] = float(gn)
    return m


def _shuffled_split(batch: dict, num_splits: int, shuffle: bool, seed: int):
    """get_iterator_k_split (rlinf/utils/data_iter_utils.py:129-262): one permutation from a generator seeded HERE (equal sizes ->
    equal permutations), then equal row ranges."""
    tensors = {k: v for k, v in batch.items() if isinstance(v, torch.Tensor)}
    n = next(iter(tensors.values())).shape[0]
    assert n % num_splits == 0
    if shuffle:
        perm = torch.randperm(n, generator=torch.Generator().manual_seed(seed))
        tensors = {k: v[perm] for k, v in tensors.items()}
    per = n // num_splits
    return [{k: v[i * per:(i + 1) * per] for k, v in tensors.items()} for i in range(num_splits)]


def pipeline_iteration(model, opt, received: list, *, total: int, response_len: int, micro_batch: int, n_minibatches: int, seed: int,
                       adv_type: str = "grpo", group_size: int = 1, normalize_advantages: bool = True, shuffle: bool = True,
                       temperature: float = 1.0, loss_agg: str = "token-mean", clip_ratio_low: float = 0.2, clip_ratio_high: float = 0.2,
                       clip_ratio_c: float = 3.0, calculate_entropy: bool = False, entropy_bonus: float = 0.0, kl_beta: float = 0.0,
                       kl_penalty_type: str = "low_var_kl", clip_grad: float = 1.0, reinpp_kl_beta: float = 0.0):
    """FSDPActor.run_training_pipeline (:816-854) around BatchResizingIterator (rlinf/data/schema/reasoning_results.py:1424-1600):
    ``received`` = the rank's rollout pieces in arrival order.  Every piece gets its advantages when it arrives; a piece smaller than
    a global batch (total / n_minibatches sequences) is topped up with the following pieces when advantages are normalised (the
    normalisation needs whole global batches), otherwise handed on as it is; pieces are shuffled and cut into global batches,
    each global batch is normalised ON ITS OWN, shuffled again (same seed) and cut into micro-batches; every optimizer step takes
    global / micro of them.  -> (everything trained on, in consumption order; per-step metric dicts)."""
    gbs = total // n_minibatches
    feed = list(received)

    def receive():
        return advantages(dict(feed.pop(0)), response_len=response_len, adv_type=adv_type, group_size=group_size,
                          reinpp_kl_beta=reinpp_kl_beta, kl_penalty_type=kl_penalty_type)

    def cat(a, b):
        return {k: torch.cat([a[k], b[k]]) for k in a if isinstance(a[k], torch.Tensor)}

    def micro_batches():
        while feed:
            piece = receive()
            n = piece["input_ids"].shape[0]
            if n % gbs != 0 and not normalize_advantages:
                globals_ = [piece]
            else:
                while n < gbs and n % gbs != 0:
                    piece = cat(piece, receive())
                    n = piece["input_ids"].shape[0]
                globals_ = _shuffled_split(piece, n // gbs, shuffle, seed)
            for g in globals_:
                if normalize_advantages:
                    g = dict(g)
                    g["advantages"] = O.masked_normalization(g["advantages"], g["response_mask"][:, -response_len:]).float()
                yield from _shuffled_split(g, g["input_ids"].shape[0] // micro_batch, shuffle, seed)

    stream, seen, out = micro_batches(), [], []
    for _ in range(n_minibatches):
        micro = [next(stream) for _ in range(gbs // micro_batch)]
        seen.extend(micro)
        out.append(optimizer_step(model, opt, micro, response_len=response_len, temperature=temperature, loss_agg=loss_agg,
                                  clip_ratio_low=clip_ratio_low, clip_ratio_high=clip_ratio_high, clip_ratio_c=clip_ratio_c,
                                  calculate_entropy=calculate_entropy, entropy_bonus=entropy_bonus, kl_beta=kl_beta,
                                  kl_penalty_type=kl_penalty_type, clip_grad=clip_grad))
    return {k: torch.cat([m[k] for m in seen]) for k in seen[0]}, out

"""The token tier of the reasoning learner: the part of ``FSDPActor`` (rlinf/workers/actor/fsdp_actor_worker.py)
that starts at the model's logits and ends at ``loss.backward()`` -- lines 476-505 (logits -> log-probs / entropy)
and 694-781 (micro-batch loss) -- on the kernels of token_ops.hip.

The transformer itself, FSDP wrapping and weight sync are outside this path (model backends are out of
scope); sequence packing and dynamic batches are in (``hybrid_engines/fsdp/utils.py``, the reference's own config
keys); ``TokenLearnerStep`` is what a maintainer drops into ``training_step`` in
place of ``forward_batch``'s tail + the loss block:

    step = TokenLearnerStep.from_cfg(cfg)
    logits = model(**inputs).logits                          # [bsz, S, V], bf16 under amp or fp32
    loss, metrics = step(logits, m_batch, gradient_accumulation)
    loss.backward()                                          # d_logits is written by ONE kernel

Per micro-batch the reference runs: div_ (read+write logits), a reshape copy of the response slice, log_softmax
(+exp, *, where, sum for the entropy), cross_entropy, ~40 elementwise [bsz, seq] kernels and ~12 .item()-free
reductions for the loss, and autograd's mirror image of all of it.  Here: one read of the logits forward, one
read + one write backward, three small [bsz, seq] launches each way.
"""

from __future__ import annotations

from dataclasses import dataclass
from typing import Mapping, Optional

import torch

from ... import ops
from ...algorithms.registry import calculate_adv_and_returns, policy_loss
from ...utils.utils import compute_logprobs_and_entropy_from_logits, compute_logprobs_from_logits


def _get(node, key, default=None):
    if node is None:
        return default
    if isinstance(node, Mapping):
        return node.get(key, default)
    return getattr(node, key, default)


@dataclass
class TokenLearnerStep:
    response_len: int
    loss_agg: str = "token-mean"
    loss_type: str = "actor"
    task_type: str = "reasoning"
    clip_ratio_low: float = 0.2
    clip_ratio_high: float = 0.2
    clip_ratio_c: Optional[float] = 3.0
    clip_log_ratio_min: Optional[float] = None
    clip_log_ratio_max: Optional[float] = None
    temperature: float = 1.0
    calculate_entropy: bool = False
    entropy_bonus: float = 0.0
    kl_beta: float = 0.0
    kl_penalty_type: str = "low_var_kl"
    importance_sampling_fix: bool = False
    importance_sampling_clip: Optional[float] = None
    logprob_op_type: str = "torch"   # "torch" rounds like the reference's default, "flash_attn" keeps fp32
    inplace_grad: bool = False       # let backward overwrite the logits buffer with d_logits
    # advantage stage (fsdp_actor_worker.py:900-907,941-978)
    adv_type: str = "grpo"
    group_size: int = 1
    reinpp_kl_beta: float = 0.0
    use_reinpp_baseline: bool = False
    normalize_advantages: bool = False
    # sequence packing (fsdp_actor_worker.py:157-170,450-505): the lengths the unpack needs
    max_prompt_length: int = 0
    encoder_seq_length: int = 0
    eos_token_id: int = 0

    @classmethod
    def from_cfg(cfg_cls, cfg) -> "TokenLearnerStep":
        """Field names of the reference's yaml (fsdp_actor_worker.py:116-160,710-722)."""
        algo, actor, data = _get(cfg, "algorithm"), _get(cfg, "actor"), _get(cfg, "data")
        eps = _get(algo, "ratio_clip_eps", 0.2)
        low, high = _get(algo, "clip_ratio_low"), _get(algo, "clip_ratio_high")
        enc = _get(_get(actor, "model"), "encoder_seq_length")
        return cfg_cls(
            response_len=int(enc - _get(data, "max_prompt_length")),
            loss_agg=_get(algo, "loss_agg_func", "token-mean"), loss_type=_get(algo, "loss_type", "actor"),
            task_type=_get(_get(cfg, "runner"), "task_type", "reasoning"),
            clip_ratio_low=eps if low is None else low, clip_ratio_high=eps if high is None else high,
            clip_ratio_c=_get(algo, "clip_ratio_c", 3.0), clip_log_ratio_min=_get(algo, "clip_log_ratio_min"),
            clip_log_ratio_max=_get(algo, "clip_log_ratio_max"),
            temperature=float(_get(_get(algo, "sampling_params"), "temperature", 1.0)),
            calculate_entropy=bool(_get(algo, "calculate_entropy", False)),
            entropy_bonus=float(_get(algo, "entropy_bonus", 0.0)), kl_beta=float(_get(algo, "kl_beta", 0.0)),
            kl_penalty_type=_get(algo, "kl_penalty_type", "low_var_kl"),
            importance_sampling_fix=bool(_get(algo, "importance_sampling_fix", False)),
            importance_sampling_clip=_get(algo, "importance_sampling_clip"),
            logprob_op_type=_get(algo, "logprob_op_type", "torch"),
            # ours (no reference key): the backward writes d_logits into the logits buffer itself -- valid when the logits feed
            # only this step, which holds for an lm_head output; saves one [bsz, S, V] allocation per micro-batch
            inplace_grad=bool(_get(actor, "inplace_logits_grad", False)),
            adv_type=_get(algo, "adv_type", "grpo"), group_size=int(_get(algo, "group_size", 1)),
            reinpp_kl_beta=float(_get(algo, "reinpp_kl_beta", 0.0)),
            use_reinpp_baseline=bool(_get(algo, "use_reinpp_baseline", False)),
            normalize_advantages=bool(_get(algo, "normalize_advantages", False)),
            max_prompt_length=int(_get(data, "max_prompt_length")), encoder_seq_length=int(enc),
            # (the reference asks its tokenizer; a run without one names the id: actor.tokenizer.eos_token_id / actor.eos_token_id)
            eos_token_id=int(_get(_get(actor, "tokenizer"), "eos_token_id", _get(actor, "eos_token_id", 0)) or 0))

    # fsdp_actor_worker.py:941-978
    def compute_advantages_and_returns(self, batch: dict) -> dict:
        """Fill ``batch["advantages"]`` ([bsz, response_len]) unless the rollout already did: GRPO, or Reinforce++ with its
        per-token KL reward from (recomputed | rollout) log-probs against the reference policy's."""
        if batch.get("advantages") is None:
            mask = batch["response_mask"][:, -self.response_len:]
            logprob = batch.get("recomputed_logprobs")
            if logprob is None:
                logprob = batch.get("rollout_logprobs")
            advantages, _ = calculate_adv_and_returns(
                task_type=self.task_type, adv_type=self.adv_type, rewards=batch["rewards"], loss_mask=mask,
                group_size=self.group_size, kl_beta=self.reinpp_kl_beta, kl_penalty_type=self.kl_penalty_type, logprob=logprob,
                ref_logprob=batch.get("ref_logprobs"), use_reinpp_baseline=self.use_reinpp_baseline)
            batch["advantages"] = advantages
        return batch

    # fsdp_actor_worker.py:900-907
    def normalize_batch_advantages(self, batch: dict, ctx=None) -> dict:
        """``algorithm.normalize_advantages``: masked_normalization over the data-parallel group -- (count, sum, sumsq) of the
        masked advantages in one f64[3] all-reduce (the reference issues three), then one elementwise pass."""
        if self.normalize_advantages:
            from ...scheduler import all_reduce_flat_
            adv = batch["advantages"]
            mask = batch["response_mask"][:, -self.response_len:].contiguous()
            stats = ops.masked_stats(adv, mask)
            if ctx is not None:
                all_reduce_flat_(stats, ctx)
            batch["advantages"] = ops.masked_normalize(adv, mask, stats)
        return batch

    # fsdp_actor_worker.py:476-505 (the fixed-length branch)
    def logprobs_and_entropy(self, logits: torch.Tensor, input_ids: torch.Tensor):
        resp = self.response_len
        responses = input_ids[:, -resp:]
        S = logits.shape[1]
        # (window mode needs the model's dense [bsz, S, V] output: a transposed / expanded lm_head result goes the sliced way below,
        #  whose copy the kernels then address -- the strided in-place gradient of window mode has nothing to alias there)
        if (logits.dim() == 3 and S > resp and logits.is_contiguous()
                and self.logprob_op_type in ("torch", "flash_attn", "liger_kernel")):
            # logits[:, -resp - 1:-1, :] scored in place; the gradient comes back for the whole [bsz, S, V] tensor from the kernel's
            # own launch + two fills of the rows outside the window, instead of autograd's zeros(logits.shape) + strided copy
            from ... import token_ops
            return token_ops.token_logprobs(logits, responses, temperature=self.temperature, with_entropy=self.calculate_entropy,
                                            round_outputs=(self.logprob_op_type != "flash_attn"), inplace_grad=self.inplace_grad,
                                            window=(S - resp - 1, S - 1))
        window = logits[:, -resp - 1:-1, :]  # a strided view; the kernels address it in place
        if self.calculate_entropy:
            return compute_logprobs_and_entropy_from_logits(window, responses, temperature=self.temperature,
                                                            op_type=self.logprob_op_type,
                                                            inplace_grad=self.inplace_grad)
        return compute_logprobs_from_logits(window, responses, self.logprob_op_type, temperature=self.temperature,
                                            inplace_grad=self.inplace_grad), None

    # fsdp_actor_worker.py:450-503 (the packed branch): logits of ONE packed stream -> [bsz, response_len] log-probs / entropy
    def logprobs_and_entropy_packed(self, logits: torch.Tensor, packed_input_ids: torch.Tensor, idx_starts, idx_ends):
        from ... import token_ops
        return token_ops.packed_token_logprobs(
            logits, packed_input_ids, idx_starts, idx_ends, max_seq_len_unpack=self.encoder_seq_length,
            response_len=self.response_len, eos_token_id=self.eos_token_id, temperature=self.temperature,
            with_entropy=self.calculate_entropy, round_outputs=(self.logprob_op_type != "flash_attn"),
            inplace_grad=self.inplace_grad)

    # fsdp_actor_worker.py:694-781
    def __call__(self, logits: torch.Tensor, m_batch: Mapping, gradient_accumulation: int = 1, packed=None):
        """``packed`` = (packed input_ids [1, L], idx_starts, idx_ends) when ``logits`` are those of a packed stream."""
        if packed is not None:
            logprobs, entropy = self.logprobs_and_entropy_packed(logits, *packed)
        else:
            logprobs, entropy = self.logprobs_and_entropy(logits, m_batch["input_ids"])
        old_logprobs = m_batch.get("recomputed_logprobs")
        if old_logprobs is None:
            old_logprobs = m_batch["rollout_logprobs"]
        advantages = m_batch["advantages"]
        ref_logprobs = m_batch.get("ref_logprobs")
        loss_mask = m_batch["response_mask"][:, -self.response_len:]
        if self.importance_sampling_fix:
            if "rollout_logprobs" not in m_batch or "recomputed_logprobs" not in m_batch:
                raise ValueError("importance_sampling_fix requires both rollout_logprobs and recomputed_logprobs")
            advantages = advantages * torch.clamp(
                (m_batch["recomputed_logprobs"] - m_batch["rollout_logprobs"]).exp(), max=self.importance_sampling_clip)
        use_kl = self.kl_beta > 0 and ref_logprobs is not None
        bonus = self.entropy_bonus if (self.entropy_bonus > 0 and self.calculate_entropy) else 0.0
        loss, metrics = policy_loss(
            task_type=self.task_type, loss_type=self.loss_type, loss_agg=self.loss_agg, logprobs=logprobs,
            old_logprobs=old_logprobs, advantages=advantages, clip_ratio_c=self.clip_ratio_c,
            clip_ratio_low=self.clip_ratio_low, clip_ratio_high=self.clip_ratio_high, loss_mask=loss_mask,
            clip_log_ratio_min=self.clip_log_ratio_min, clip_log_ratio_max=self.clip_log_ratio_max,
            fast_path_zero_loss_mask=True,
            # the two terms the reference adds right after policy_loss(), fused into the same kernel:
            entropy=entropy, entropy_bonus=bonus, ref_logprobs=ref_logprobs if use_kl else None,
            kl_beta=self.kl_beta if use_kl else 0.0, kl_penalty_type=self.kl_penalty_type)
        if not self.calculate_entropy:
            metrics.update({"actor/final_loss": loss.detach(), "actor/entropy_loss": 0.0})
        if not use_kl:
            metrics.update({"actor/kl_loss": 0.0})
        return loss / gradient_accumulation, metrics


# =================================================================================================================
# FSDPActor: the loop AROUND the token step (rlinf/workers/actor/fsdp_actor_worker.py:434-1000), NO_SHARD data parallel
# =================================================================================================================
def seqlen_balanced_partitions(seqlen_list: list, k_partitions: int, equal_size: bool = True) -> list:
    """get_seqlen_balanced_partitions (rlinf/utils/data_iter_utils.py:301-444): Karmarkar-Karp largest differencing -- the
    partial partitions with the largest spread are merged first, largest set against smallest -- with the reference's tie rules
    (sum, then size, then the (index, length) lists), so that every rank derives the SAME partitions from the gathered lengths.
    Returns k sorted index lists."""
    import heapq

    assert len(seqlen_list) >= k_partitions, f"number of items:[{len(seqlen_list)}] < k_partitions:[{k_partitions}]"

    class _Set:
        __slots__ = ("sum", "items")

        def __init__(self):
            self.sum, self.items = 0, []

        def key(self):
            return (self.sum, len(self.items), self.items)

    class _State:
        def __init__(self, items):
            self.sets = [_Set() for _ in range(k_partitions)]
            for i, (idx, val) in enumerate(items):
                self.sets[i].items.append((idx, val))
                self.sets[i].sum += val
            self.sets.sort(key=_Set.key, reverse=True)

        def merge(self, other):
            for i in range(k_partitions):
                o = other.sets[k_partitions - 1 - i]
                self.sets[i].items.extend(o.items)
                self.sets[i].sum += o.sum
            self.sets.sort(key=_Set.key, reverse=True)

        def __lt__(self, other):  # a min-heap that pops the largest spread first, then the largest leading set
            a, b = self.sets[0].sum - self.sets[-1].sum, other.sets[0].sum - other.sets[-1].sum
            if a != b:
                return a > b
            return self.sets[0].key() > other.sets[0].key()

    order = sorted((int(v), i) for i, v in enumerate(seqlen_list))
    heap = []
    if equal_size:
        assert len(seqlen_list) % k_partitions == 0, f"{len(seqlen_list)} % {k_partitions} != 0"
        for off in range(0, len(order), k_partitions):
            heapq.heappush(heap, _State([(idx, v) for v, idx in order[off:off + k_partitions]]))
    else:
        for v, idx in order:
            heapq.heappush(heap, _State([(idx, v)]))
    while len(heap) > 1:
        s0, s1 = heapq.heappop(heap), heapq.heappop(heap)
        s0.merge(s1)
        heapq.heappush(heap, s0)
    parts = [sorted(idx for idx, _ in s.items) for s in heap[0].sets]
    assert all(parts) and sorted(sum(parts, [])) == list(range(len(seqlen_list)))
    return parts


class FSDPActor:
    """The reasoning learner around ``TokenLearnerStep``, with the reference's method names and call order
    (fsdp_actor_worker.py): ``run_training`` (:860-939: collect the rank's rollout batch, advantages, optional sequence-length
    balancing over the data-parallel group, masked normalisation, seeded shuffle into ``n_minibatches`` global batches, one
    ``training_step`` each), ``training_step`` (:659-813: micro-batches, forward -> log-prob / entropy -> loss -> backward with
    1 / gradient_accumulation, then clip + AdamW), ``inference_step`` / ``run_inference`` (:509-558: recomputed and reference-policy
    log-probs, micro-batch by micro-batch without a graph), ``compute_advantages_and_returns`` (:941-978), ``_dp_load_balance``
    (:846-858), ``run_training_pipeline`` (:816-854: pipeline mode, micro-batches streamed out of a ``BatchResizingIterator``
    while pieces of the rollout are still arriving).

    What runs where: the transformer is the caller's (any module returning ``.logits`` [bsz, seq, vocab]: model backends are out
    of scope); everything from the logits on is this package's kernels -- one read of the logits forward, one read +
    one write backward (token_ops.hip), the fused token loss, the advantage kernels, the data-parallel gradient mean and
    clip + AdamW on ONE flat f32 parameter buffer (the module's parameters are re-pointed at views of it, its gradients at views of
    a flat gradient buffer: adamw_clip.hip needs no per-tensor launches and the all-reduce is one call).  Sequence packing
    (``runner.enable_dynamic_batch_size`` / ``actor.model.variable_seq_lengths``, :450-505): the micro-batch's valid windows go
    through the module as one packed stream (hybrid_engines/fsdp/utils.py) and the scoring kernel stores its results through the
    unpack map (``rlx_token_logprob_fwd_packed``); dynamic batch sizes cut a global batch by a token budget (:393-414)."""

    ROLE = "actor"

    def __init__(self, cfg, ctx=None, model: Optional[torch.nn.Module] = None):
        from ...scheduler import DistContext
        self.cfg, self.ctx = cfg, ctx or DistContext()
        self._rank, self._world_size, self.device = self.ctx.rank, self.ctx.world_size, self.ctx.device
        algo, actor, data = _get(cfg, "algorithm"), _get(cfg, "actor"), _get(cfg, "data")
        # the reference's own keys (fsdp_actor_worker.py:157-170): runner.enable_dynamic_batch_size, runner.max_tokens_per_mbs,
        # actor.model.variable_seq_lengths.  Either switch sends every micro-batch through the model as ONE packed stream.
        runner = _get(cfg, "runner")
        self.enable_dynamic_batch_size = bool(_get(runner, "enable_dynamic_batch_size", False))
        self.max_tokens_per_mbs = int(_get(runner, "max_tokens_per_mbs", 2048))
        self.variable_seq_lengths = bool(_get(_get(actor, "model"), "variable_seq_lengths", False))
        self.step = TokenLearnerStep.from_cfg(cfg)
        self.response_len = self.step.response_len
        self.micro_batch_size = int(_get(actor, "micro_batch_size"))
        self.n_mini_batches = int(_get(algo, "n_minibatches", 1))
        self.total_batch_size_per_dp = (int(_get(data, "rollout_batch_size")) * int(_get(algo, "group_size", 1))) // self._world_size
        assert int(_get(actor, "global_batch_size")) % (self.micro_batch_size * self._world_size) == 0  # :915-919
        self.enable_dp_load_balance = bool(_get(actor, "enable_dp_load_balance", False))
        # the reference reads this off the placement (``is_disaggregated``, :135); here: ``actor.pipeline`` / ``cluster.pipeline``
        self.is_pipeline = bool(_get(actor, "pipeline", _get(_get(cfg, "cluster") or {}, "pipeline", False)))
        if self.is_pipeline:
            assert not self.enable_dp_load_balance, "DP load balance is not supported in pipeline mode."  # :160-163
            assert not self.enable_dynamic_batch_size, "Dynamic batch size is not supported in pipeline mode."  # :164-166
        self.logprob_forward_micro_batch_size = int(_get(algo, "logprob_forward_micro_batch_size", self.micro_batch_size))
        self.shuffle_rollout = bool(_get(algo, "shuffle_rollout", True))
        self.seed = int(_get(actor, "seed", 1234))
        self.max_prompt_length = int(_get(data, "max_prompt_length"))
        self.gradient_accumulation = 1
        self.model = None
        self.ref_policy_flat: Optional[torch.Tensor] = None
        self.optimizer_steps = 0
        if model is not None:
            self.init_worker(model)

    # ---- set-up: one flat parameter / gradient buffer under the module -------------------------------------------------------
    def init_worker(self, model: torch.nn.Module, keep_reference_policy: Optional[bool] = None):
        o = _get(_get(self.cfg, "actor"), "optim")
        self.model = model.to(self.device)
        params = [p for p in self.model.parameters() if p.requires_grad]
        n = sum(p.numel() for p in params)
        self.flat = torch.empty(n, dtype=torch.float32, device=self.device)
        self.grad_flat = torch.zeros(n, dtype=torch.float32, device=self.device)
        off = 0
        with torch.no_grad():
            for p in params:
                k = p.numel()
                self.flat[off:off + k].copy_(p.detach().reshape(-1).float())
                p.data = self.flat[off:off + k].view_as(p)
                p.grad = self.grad_flat[off:off + k].view_as(p)
                off += k
        self._params = params
        self.exp_avg, self.exp_avg_sq = torch.zeros_like(self.flat), torch.zeros_like(self.flat)
        self.step_state = torch.zeros(2, dtype=torch.int32, device=self.device)
        self.opt_stats = torch.zeros(2, device=self.device)
        self.adamw_ws = torch.empty(ops._lib.load().rlx_adamw_workspace_bytes(n), dtype=torch.uint8, device=self.device)
        from ...scheduler import ranks_share_a_device
        # one-launch optimizer step where the plan allows it (small models; not when ranks share a GPU: see the embodied worker)
        self.adamw_sync = None if ranks_share_a_device(self.ctx) else ops.adamw_sync_words(n, self.device)
        self._optim = dict(betas=(float(_get(o, "adam_beta1", 0.9)), float(_get(o, "adam_beta2", 0.999))), eps=float(_get(o, "adam_eps", 1e-8)),
                           weight_decay=float(_get(o, "weight_decay", 1e-2)), max_grad_norm=float(_get(o, "clip_grad", 1.0)))
        self.lr = float(_get(o, "lr", 1e-6))
        need_ref = (self.step.kl_beta > 0 or self.step.reinpp_kl_beta > 0) if keep_reference_policy is None else keep_reference_policy
        if need_ref:  # ref_policy_state_dict (:262-273): the weights the run starts from
            self.ref_policy_flat = self.flat.clone()

    # ---- forward (fixed-length branch of forward_batch, :434-505) ----------------------------------------------------------
    @property
    def packs_sequences(self) -> bool:
        return self.enable_dynamic_batch_size or self.variable_seq_lengths

    def _model_forward(self, m_batch: Mapping):
        """-> (logits, packed): the module on the padded [bsz, S] micro-batch, or -- packing on (:450-466) -- on ONE packed stream
        [1, L] of the rows' valid windows (attention_mask None: the backend derives the boundaries from position_ids), padded to
        ``max_tokens_per_mbs`` unless ``variable_seq_lengths``; ``packed`` = (packed ids, idx_starts, idx_ends) for the unpack."""
        if not self.packs_sequences:
            outputs = self.model(input_ids=m_batch["input_ids"], attention_mask=m_batch["attention_mask"],
                                 position_ids=m_batch["position_ids"], use_cache=False)
            return outputs.logits, None
        from ...hybrid_engines.fsdp.utils import pack_fsdp_input, prepare_pack_fsdp
        idx_starts, idx_ends = prepare_pack_fsdp(m_batch, self.max_prompt_length)
        ids, pos, mask = pack_fsdp_input(m_batch["input_ids"], m_batch["position_ids"], idx_starts=idx_starts, idx_ends=idx_ends,
                                         max_seq_len_pack=self.max_tokens_per_mbs, eos_token_id=self.step.eos_token_id,
                                         pad_to_fixed_len=not self.variable_seq_lengths)
        outputs = self.model(input_ids=ids, attention_mask=mask, position_ids=pos, use_cache=False)
        return outputs.logits, (ids, idx_starts, idx_ends)

    def forward_batch(self, m_batch: Mapping, calculate_entropy: bool = False):
        logits, packed = self._model_forward(m_batch)
        step = self.step
        if calculate_entropy and not step.calculate_entropy:
            step = TokenLearnerStep(**{**step.__dict__, "calculate_entropy": True})
        if packed is not None:
            logprobs, entropy = step.logprobs_and_entropy_packed(logits, *packed)
        else:
            logprobs, entropy = step.logprobs_and_entropy(logits, m_batch["input_ids"])  # temperature fused (:478)
        return (logprobs, entropy) if calculate_entropy else logprobs

    def _split_to_micro_batch(self, batch: Mapping, split_num: int):
        """-> (micro-batches, count, dynamic-batch index partitions | None).  get_iterator_k_split without shuffle (:391-414,
        data_iter_utils.py:199-243): ``split_num`` equal row ranges, views; with ``runner.enable_dynamic_batch_size``:
        split_dynamic_batch_size -- as many micro-batches as the token budget needs, sequences dealt by length."""
        if self.enable_dynamic_batch_size:
            from ...hybrid_engines.fsdp.utils import split_dynamic_batch_size
            return split_dynamic_batch_size(batch, self.max_tokens_per_mbs, seqlen_balanced_partitions, self.ctx)
        micro, n = self._k_split(batch, split_num)
        return micro, n, None

    @staticmethod
    def _k_split(batch: Mapping, split_num: int):
        """get_iterator_k_split without shuffle (data_iter_utils.py:199-243): ``split_num`` equal row ranges, views."""
        bsz = next(v for v in batch.values() if isinstance(v, torch.Tensor)).shape[0]
        assert bsz % split_num == 0, "Issue with batch size configuration!"
        per = bsz // split_num
        return [{k: (v[i * per:(i + 1) * per] if isinstance(v, (torch.Tensor, list)) else v) for k, v in batch.items()
                 if isinstance(v, (torch.Tensor, list))} for i in range(split_num)], split_num

    @torch.no_grad()
    def inference_step(self, batch: Mapping, num_sequences: int, compute_ref_logprobs: bool):
        """:509-558 -> (recomputed_logprobs, ref_logprobs | None), both [num_sequences, response_len] on the accelerator (the
        reference moves them to the host for its channel; here the next consumer is a kernel on the same device)."""
        micro_batches, _, dbs_indices = self._split_to_micro_batch(batch, num_sequences // self.logprob_forward_micro_batch_size)
        revert = None
        if dbs_indices is not None:  # dynamic batch sizes dealt the sequences by length: back to the batch's order (:522-539)
            from ...hybrid_engines.fsdp.utils import get_reverse_idx
            indices = sum((list(p) for p in dbs_indices), [])
            revert = torch.tensor(get_reverse_idx(indices), dtype=torch.long, device=self.device)
        undo = (lambda t: t[revert]) if revert is not None else (lambda t: t)  # noqa: E731
        recomputed = undo(torch.cat([self.forward_batch(mb) for mb in micro_batches]))
        ref = None
        if compute_ref_logprobs:
            assert self.ref_policy_flat is not None, "Reference policy state dict is None but compute_ref_logprobs is True"
            live = self.flat.clone()  # cpu_weight_swap (:544-548): run the same module on the reference weights, then restore
            self.flat.copy_(self.ref_policy_flat)
            try:
                ref = undo(torch.cat([self.forward_batch(mb) for mb in micro_batches]))
            finally:
                self.flat.copy_(live)
        return recomputed, ref

    def run_inference(self, batch: dict, compute_ref_logprobs: bool) -> dict:
        """run_inference (:560-657) for a batch that is already on this rank: fills ``recomputed_logprobs`` (and
        ``ref_logprobs``) in place of putting RolloutResults on an output channel."""
        self.model.eval()
        n = batch["input_ids"].shape[0]
        batch["recomputed_logprobs"], ref = self.inference_step(batch, n, compute_ref_logprobs)
        if compute_ref_logprobs:
            batch["ref_logprobs"] = ref
        return batch

    # ---- advantages (:941-978) and the data-parallel balancing (:846-858) ------------------------------------------------------
    def compute_advantages_and_returns(self, batch: dict) -> dict:
        return self.step.compute_advantages_and_returns(batch)

    def _dp_load_balance(self, batch: dict) -> dict:
        """RolloutDataBalance.from_rollout_batches (rlinf/utils/distributed.py:309-470): the ranks' samples are re-dealt so that
        every rank trains the same number of sequences with about the same number of TOKENS (attention_mask sums), by
        Karmarkar-Karp partitions computed identically on every rank from the gathered lengths.  One all_gather of the (small,
        [bsz, seq]-shaped) per-sequence tensors; world_size 1: the identity."""
        bsz = batch["input_ids"].shape[0]
        assert bsz == self.total_batch_size_per_dp, (
            f"DP Load balance is only available when a single batch contains all data, e.g., in collocated mode. But got "
            f"batch_size={bsz} and self.total_batch_size_per_dp={self.total_batch_size_per_dp}.")
        if self._world_size == 1:
            return batch
        import torch.distributed as dist
        keys = sorted(k for k, v in batch.items() if isinstance(v, torch.Tensor) and v.shape[:1] == (bsz,))
        gathered = {}
        for k in keys:
            parts = [torch.empty_like(batch[k]) for _ in range(self._world_size)]
            dist.all_gather(parts, batch[k].contiguous())
            gathered[k] = torch.cat(parts)
        lengths = gathered["attention_mask"].sum(dim=1).tolist()
        mine = seqlen_balanced_partitions(lengths, self._world_size, equal_size=True)[self._rank]
        idx = torch.tensor(mine, dtype=torch.int64, device=gathered[keys[0]].device)
        out = dict(batch)
        out.update({k: gathered[k][idx] for k in keys})
        return out

    # ---- one optimizer step (:659-813) -------------------------------------------------------------------------------------------
    def optimizer_step(self):
        """FSDPModelManager.optimizer_step (fsdp_model_manager.py:429-463) on the flat buffers: data-parallel mean of the
        gradient, global-norm clip, AdamW (skipped when the norm is not finite) -- two launches (+ one all-reduce)."""
        from ...scheduler import all_reduce_flat_
        if self._world_size > 1:
            all_reduce_flat_(self.grad_flat, self.ctx)
        ops.clip_adamw_step_(self.flat, self.grad_flat, self.exp_avg, self.exp_avg_sq, [(0, self.flat.numel(), self.lr)], 0,
                             grad_scale=1.0 / self._world_size, stats=self.opt_stats, step_state=self.step_state,
                             workspace=self.adamw_ws, sync=self.adamw_sync, **self._optim)
        self.optimizer_steps += 1
        return self.opt_stats[0], [self.lr]

    def training_step(self, batch) -> dict:
        """``batch``: one global batch (a dict, cut into micro-batches here) or, in pipeline mode, the ``BatchResizingIterator``
        the step pulls its ``global_batch / micro_batch`` micro-batches from (:662-683)."""
        if isinstance(batch, Mapping):
            global_batch_size = batch["input_ids"].shape[0]
            assert global_batch_size % self.micro_batch_size == 0, (
                f"global batch size {global_batch_size} can not divide micro_batch_size {self.micro_batch_size}")
            micro_batches, cnt, _ = self._split_to_micro_batch(batch, global_batch_size // self.micro_batch_size)
        else:
            cnt = (self.total_batch_size_per_dp // self.n_mini_batches) // self.micro_batch_size
            micro_batches = (next(batch) for _ in range(cnt))
        self.gradient_accumulation = cnt
        self.grad_flat.zero_()  # optimizer.zero_grad(): the parameters' .grad are views of this buffer
        rows = []
        for m_batch in micro_batches:
            logits, packed = self._model_forward(m_batch)
            loss, metrics = self.step(logits, m_batch, cnt, packed=packed)
            loss.backward()
            rows.append(metrics)
        grad_norm, lr_list = self.optimizer_step()
        keys = [k for k in rows[0] if all(k in r for r in rows)]
        stacked = torch.stack([torch.stack([torch.as_tensor(r[k], dtype=torch.float32, device=self.device).reshape(()) for k in keys])
                               for r in rows]).mean(dim=0)  # mean over micro-batches (:783-791)
        if self._world_size > 1:
            from ...scheduler import all_reduce_flat_
            all_reduce_flat_(stacked, self.ctx, average=True)
        host = torch.cat([stacked, grad_norm.reshape(1)]).tolist()  # ONE read-back per optimizer step
        out = dict(zip(keys, host[:-1]))
        out["actor/grad_norm"], out["actor/lr"] = host[-1], lr_list[0]
        if ops.check_adamw_sync(self.adamw_sync, host[-1]):  # the one-launch form's exchange expired: two launches from here on
            self._retired_adamw_sync, self.adamw_sync = self.adamw_sync, None
        return out

    # ---- the iteration (:860-939) ----------------------------------------------------------------------------------------------
    def get_batch(self, input_channel):
        """-> (batch dict, number of sequences): the channel hands out dicts of [n, ...] tensors (already on the accelerator)."""
        batch = input_channel.get() if hasattr(input_channel, "get") else input_channel.pop(0)
        return batch, batch["input_ids"].shape[0]

    def run_training_pipeline(self, input_channel):
        """:816-854 -- the rollout side is still producing: every received piece gets its advantages at once, pieces are topped up
        / cut to global batches, each global batch is normalised on its own, and ``n_minibatches`` optimizer steps pull their
        micro-batches from the stream; the rollout metrics are taken over everything that was trained on."""
        from functools import partial

        from ...data.batch_iterator import BatchResizingIterator
        self.model.train()
        it = BatchResizingIterator(cfg=self.cfg, get_batch_fn=partial(self.get_batch, input_channel), micro_batch_size=self.micro_batch_size,
                                   total_batch_size=self.total_batch_size_per_dp, num_global_batches=self.n_mini_batches,
                                   forward_only=False)
        it.register_get_batch_handler(self.compute_advantages_and_returns)
        if self.step.normalize_advantages:
            it.register_global_batch_handler(
                lambda b: self.step.normalize_batch_advantages(b, self.ctx if self._world_size > 1 else None))
        training_metrics_list = [self.training_step(it) for _ in range(self.n_mini_batches)]
        return self.rollout_metrics(it.get_all_batches()), training_metrics_list

    def run_training(self, input_channel):
        if self.is_pipeline:
            return self.run_training_pipeline(input_channel)
        batches, got = [], 0
        while got < self.total_batch_size_per_dp:
            batch, n = self.get_batch(input_channel)
            batches.append(batch)
            got += n
        assert got == self.total_batch_size_per_dp, f"Expected {self.total_batch_size_per_dp} sequences from channel, but got {got}"
        global_batch = batches[0] if len(batches) == 1 else {k: torch.cat([b[k] for b in batches]) for k in batches[0]
                                                             if isinstance(batches[0][k], torch.Tensor)}
        assert "recomputed_logprobs" in global_batch or "rollout_logprobs" in global_batch
        global_batch = self.compute_advantages_and_returns(global_batch)
        if self.enable_dp_load_balance:
            global_batch = self._dp_load_balance(global_batch)
        global_batch = self.step.normalize_batch_advantages(global_batch, self.ctx if self._world_size > 1 else None)
        # get_iterator_k_split(shuffle=True, shuffle_seed=actor.seed) (data_iter_utils.py:148-181): ONE seeded permutation of the
        # rows, re-drawn identically every call (the generator is re-seeded) -- every tensor gathered by the same index
        tensors = {k: v for k, v in global_batch.items() if isinstance(v, torch.Tensor)}
        bsz = tensors["input_ids"].shape[0]
        if self.shuffle_rollout:
            perm = torch.randperm(bsz, generator=torch.Generator().manual_seed(self.seed)).to(self.device)
            tensors = {k: (v[perm] if v.shape[:1] == (bsz,) else v) for k, v in tensors.items()}
        global_batch = {**global_batch, **tensors}
        mini_batches, _ = self._k_split(tensors, self.n_mini_batches)
        self.model.train()
        training_metrics_list = [self.training_step(mb) for mb in mini_batches]
        return self.rollout_metrics(global_batch), training_metrics_list

    def rollout_metrics(self, batch: Mapping) -> dict:
        """compute_math_rollout_metrics (rlinf/utils/distributed.py:186-306): lengths, rewards, end fraction and masked advantage
        statistics of the iteration's batch; sums in one all-reduce, the (-min, max) pair in another, one read-back."""
        import torch.distributed as dist
        mask = batch["response_mask"][:, -self.response_len:]
        adv = batch["advantages"]
        valid = adv[mask]
        plen, rlen = batch["prompt_lengths"].float(), batch["response_lengths"].float()
        sums = torch.stack([plen.sum(), rlen.sum(), batch["rewards"].float().sum(), batch["is_end"].float().sum(),
                            valid.double().sum().float(), torch.tensor(float(plen.numel()), device=adv.device),
                            mask.sum().float(), (rlen * rlen).sum()])
        ext = torch.stack([-valid.min(), valid.max(), -rlen.min(), rlen.max()]) if valid.numel() else torch.zeros(4, device=adv.device)
        if self._world_size > 1:
            dist.all_reduce(sums)
            dist.all_reduce(ext, op=dist.ReduceOp.MAX)
        s_pl, s_rl, s_rw, s_end, s_adv, n_seq, n_tok, s_rl2 = sums.tolist()
        e = ext.tolist()
        mean_rl = s_rl / n_seq
        var_rl = (s_rl2 - n_seq * mean_rl * mean_rl) / (n_seq - 1) if n_seq > 1 else float("nan")  # torch.var: unbiased
        return {"total_num_sequence": n_seq, "prompt_length": s_pl / n_seq, "response_length": s_rl / n_seq,
                "average_response_length": mean_rl, "variance_of_response_length": var_rl, "max_of_response_length": e[3],
                "min_of_response_length": -e[2], "total_length": (s_pl + s_rl) / n_seq, "reward_scores": s_rw / n_seq,
                "fraction_of_samples_properly_ended": s_end / n_seq, "advantages_mean": s_adv / n_tok if n_tok else float("nan"),
                "advantages_max": e[1], "advantages_min": -e[0]}

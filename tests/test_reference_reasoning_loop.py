"""oracle.token_loop.iteration against the REFERENCE's own reasoning learner loop, executed on CPU: FSDPActor.run_training,
training_step, forward_batch, compute_advantages_and_returns (rlinf/workers/actor/fsdp_actor_worker.py:434-505, 659-813, 860-978)
and get_iterator_k_split (rlinf/utils/data_iter_utils.py:129-262) are compiled from their source where they lie and run as plain
functions over a stand-in ``self`` (a tiny real torch model, torch's AdamW, clip_grad_norm_ as the NO_SHARD branch of
FSDPModelManager.optimizer_step): everything that computes -- advantages through its registry, masked_normalization, the seeded
shuffle, mini / micro-batching, policy_loss, entropy bonus, KL penalty, accumulation -- is the reference's code.  The oracle loop
must land on the same parameters bit for bit; the HIP learner is compared with the oracle loop on the GPU
(tests/test_gpu_reasoning_loop.py)."""

import contextlib
import copy
import logging
from collections import UserDict
from types import SimpleNamespace

import pytest
import torch

from conftest import free_port
import torch.distributed as dist

from oracle import token_loop as TL

pytestmark = pytest.mark.reference


class Cfg(dict):
    __getattr__ = dict.__getitem__

    def get(self, k, d=None):
        return self[k] if k in self and self[k] is not None else d


@pytest.fixture(scope="module")
def one_rank_group():
    started = not dist.is_initialized()
    if started:
        dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{free_port()}", rank=0, world_size=1)
    yield
    if started:
        dist.destroy_process_group()


def _packing_helpers(R):
    """The reference's own pack / unpack helpers and its dynamic-batch split, compiled from the files where they lie."""
    import heapq
    import itertools
    fs = "rlinf/hybrid_engines/fsdp/utils.py"
    it_py = "rlinf/utils/data_iter_utils.py"
    pack_sequences = R.load_function(fs, "pack_sequences", torch=torch)
    unpack_sequences = R.load_function(fs, "unpack_sequences", torch=torch)
    helpers = dict(
        prepare_pack_fsdp=R.load_function(fs, "prepare_pack_fsdp"),
        pack_fsdp_input=R.load_function(fs, "pack_fsdp_input", pack_sequences=pack_sequences),
        unpack_fsdp_logprobs=R.load_function(fs, "unpack_fsdp_logprobs", torch=torch, unpack_sequences=unpack_sequences),
        unpack_sequences=unpack_sequences, pack_sequences=pack_sequences)

    class CpuTorch:  # get_iterator_dynamic builds its two one-element tensors on "cuda": the same integers, on the host
        def __getattr__(self, name):
            return getattr(torch, name)

        @staticmethod
        def tensor(data, device=None, **kw):
            return torch.tensor(data, **kw)

    kk = R.load_function(it_py, "karmarkar_karp", heapq=heapq)
    balanced = R.load_function(it_py, "get_seqlen_balanced_partitions", karmarkar_karp=kk)
    bfd = R.load_function(it_py, "get_seqlen_BFD_partitions")
    dyn = R.load_function(it_py, "get_iterator_dynamic", torch=CpuTorch(), dist=dist, UserDict=UserDict, itertools=itertools,
                          get_seqlen_BFD_partitions=bfd, get_seqlen_balanced_partitions=balanced,
                          roundup_divisible=R.load_function(it_py, "roundup_divisible"), Union=None, Optional=None)
    helpers["split_dynamic_batch_size"] = R.load_function(it_py, "split_dynamic_batch_size", get_iterator_dynamic=dyn)
    helpers["balanced"], helpers["bfd"] = balanced, bfd
    return helpers


def reference_learner(ref, model, *, resp, prompt, micro, n_mini, total, case, pack=None):
    from oracle import reference_loader as R
    du = R.load_distributed_utils()
    py = "rlinf/workers/actor/fsdp_actor_worker.py"
    it_py = "rlinf/utils/data_iter_utils.py"
    worker_stub = SimpleNamespace(torch_device_type="cpu", timer=lambda *_a, **_k: (lambda f: f),
                                  torch_platform=SimpleNamespace(current_device=lambda: torch.device("cpu")))
    split_list = R.load_function(it_py, "split_list")
    import itertools
    k_split = R.load_function(it_py, "get_iterator_k_split", torch=torch, UserDict=UserDict, logging=logging, split_list=split_list, itertools=itertools,
                              Union=None, Optional=None, Iterator=None)
    mu = ref.metric_utils

    def masked_normalization_cpu(x, mask):  # the reference's own function moves its inputs to .cuda(): same arithmetic, on the host
        src = R.load_function("rlinf/utils/distributed.py", "masked_normalization", torch=torch, np=__import__("numpy"))
        orig = torch.Tensor.cuda
        torch.Tensor.cuda = lambda self, *a, **k: self
        try:
            return src(x, mask)
        finally:
            torch.Tensor.cuda = orig

    bonus, beta = case.get("entropy_bonus", 0.0), case.get("kl_beta", 0.0)
    alg = Cfg(sampling_params=Cfg(temperature=case.get("temperature", 1.0)), ratio_clip_eps=0.2, clip_ratio_high=0.28, loss_type="actor",
              entropy_bonus=bonus, adv_type=case.get("adv_type", "grpo"), group_size=case.get("group_size", 4),
              normalize_advantages=case.get("normalize", True), n_minibatches=n_mini, shuffle_rollout=True)
    cfg = Cfg(algorithm=alg, actor=Cfg(seed=1234, global_batch_size=total // n_mini, micro_batch_size=micro,
                                       model=Cfg(encoder_seq_length=prompt + resp)), data=Cfg(max_prompt_length=prompt))
    pk = pack or {}
    helpers = _packing_helpers(R) if pack else {}
    opt = torch.optim.AdamW(model.parameters(), lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01)
    me = SimpleNamespace(
        cfg=cfg, model=model, response_len=resp, enable_dynamic_batch_size=bool(pk.get("dynamic", False)),
        variable_seq_lengths=bool(pk.get("variable_seq_lengths", False)), max_tokens_per_mbs=pk.get("max_tokens_per_mbs"),
        tokenizer=SimpleNamespace(eos_token_id=pk.get("eos_token_id", 0)),
        amp_context=contextlib.nullcontext(), before_micro_batch=lambda *_a, **_k: contextlib.nullcontext(), task_type="reasoning",
        loss_agg_func=ref.utils.get_loss_agg_func(case.get("loss_agg", "token-mean")), calculate_entropy=bonus > 0,
        calculate_entropy_loss=bonus > 0, kl_beta=beta, kl_penalty_type=case.get("kl", "low_var_kl"), entropy_op_type="torch",
        micro_batch_size=micro, total_batch_size_per_dp=total, n_mini_batches=n_mini, gradient_accumulation=None,
        lr_sched_sync_with_optim=False, optimizer=opt, grad_scaler=SimpleNamespace(scale=lambda loss: loss), is_pipeline=False,
        enable_dp_load_balance=False, _world_size=1, reinpp_kl_beta=case.get("reinpp_kl_beta", 0.0),
        lr_scheduler=SimpleNamespace(step=lambda: None), worker_timer=lambda *_a: contextlib.nullcontext(),
        _load_weight_and_optimizer=lambda: None)

    def optimizer_step():  # FSDPModelManager.optimizer_step, NO_SHARD: torch's clip_grad_norm_, step unless the norm is not finite
        gn = torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)
        if torch.isfinite(gn):
            opt.step()
        return float(gn), [1e-3]

    me.optimizer_step = optimizer_step
    me.compute_logprobs = lambda logits, target: ref.utils.compute_logprobs_from_logits(logits, target, op_type="torch")
    forward_batch = R.load_function(py, "FSDPActor.forward_batch", torch=torch, Worker=worker_stub,
                                    compute_entropy_from_logits=ref.utils.compute_entropy_from_logits,
                                    **{k: helpers[k] for k in ("prepare_pack_fsdp", "pack_fsdp_input", "unpack_fsdp_logprobs",
                                                               "unpack_sequences") if k in helpers})
    me.forward_batch = lambda m_batch, calculate_entropy=False: forward_batch(me, m_batch, calculate_entropy)
    split = R.load_function(py, "FSDPActor._split_to_micro_batch", get_iterator_k_split=k_split, Optional=None,
                            split_dynamic_batch_size=helpers.get("split_dynamic_batch_size"))
    me._split_to_micro_batch = lambda *a, **k: split(*a, **k)
    training_step = R.load_function(
        py, "FSDPActor.training_step", torch=torch, Worker=worker_stub, policy_loss=ref.registry.policy_loss,
        kl_penalty=ref.algo_utils.kl_penalty, append_to_dict=mu.append_to_dict, compute_rollout_train_kl=lambda *_a: None,
        pop_critic_explained_variance_stats=mu.pop_critic_explained_variance_stats, all_reduce_dict=lambda d, op=None: d,
        compute_critic_explained_variance_from_stats=mu.compute_critic_explained_variance_from_stats,
        CRITIC_EXPLAINED_VARIANCE_KEY=mu.CRITIC_EXPLAINED_VARIANCE_KEY, BatchResizingIterator=type("BatchResizingIterator", (), {}))
    me.training_step = lambda batch: training_step(me, batch)
    cadv = R.load_function(py, "FSDPActor.compute_advantages_and_returns", Worker=worker_stub,
                           calculate_adv_and_returns=ref.registry.calculate_adv_and_returns)
    me.compute_advantages_and_returns = lambda batch: cadv(me, batch)
    rollout_result = SimpleNamespace(merge_batches=staticmethod(lambda batches: batches[0]))
    run_training = R.load_function(py, "FSDPActor.run_training", torch=torch, RolloutResult=rollout_result,
                                   masked_normalization=masked_normalization_cpu, get_iterator_k_split=k_split,
                                   compute_math_rollout_metrics=lambda *a, **k: ({"n": 1}, None, None), Channel=None)
    me.run_training = lambda channel: run_training(me, channel)
    return me


@pytest.mark.parametrize("case", [
    dict(),
    dict(loss_agg="seq-mean-token-sum", temperature=0.7, entropy_bonus=0.01),
    dict(loss_agg="seq-mean-token-mean", temperature=1.3, kl_beta=0.05, kl="low_var_kl"),
    dict(adv_type="reinpp", normalize=False, reinpp_kl_beta=0.0),
    dict(normalize=False, group_size=2),
], ids=lambda c: "-".join(f"{k}={v}" for k, v in c.items()) or "default")
def test_oracle_iteration_matches_the_reference_run_training(ref, one_rank_group, case):
    resp, prompt, vocab, dim = 6, 4, 53, 16
    total, micro, n_mini = 16, 4, 2
    torch.manual_seed(5)
    base = TL.TinyCausalLM(vocab, dim, prompt + resp)
    batch = TL.synthetic_rollout_batch(7, total, prompt, resp, vocab)
    if case.get("kl_beta", 0) > 0:
        with torch.no_grad():
            batch["ref_logprobs"] = TL.forward_logprobs(base, batch, resp, case.get("temperature", 1.0)) + 0.05 * torch.randn(total, resp)
    m_ref, m_ora = copy.deepcopy(base), copy.deepcopy(base)
    me = reference_learner(ref, m_ref, resp=resp, prompt=prompt, micro=micro, n_mini=n_mini, total=total, case=case)
    feed = [{k: (v.clone() if isinstance(v, torch.Tensor) else v) for k, v in batch.items()}]
    me.get_batch = lambda _ch: (feed.pop(0), SimpleNamespace(num_sequence=total))
    _, want_metrics = me.run_training(None)
    opt = torch.optim.AdamW(m_ora.parameters(), lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01)
    _, got_metrics = TL.iteration(
        m_ora, opt, {k: (v.clone() if isinstance(v, torch.Tensor) else v) for k, v in batch.items()}, response_len=resp,
        micro_batch=micro, n_minibatches=n_mini, seed=1234, adv_type=case.get("adv_type", "grpo"), group_size=case.get("group_size", 4),
        normalize_advantages=case.get("normalize", True), temperature=case.get("temperature", 1.0),
        loss_agg=case.get("loss_agg", "token-mean"), clip_ratio_low=0.2, clip_ratio_high=0.28,
        calculate_entropy=case.get("entropy_bonus", 0) > 0, entropy_bonus=case.get("entropy_bonus", 0.0),
        kl_beta=case.get("kl_beta", 0.0), kl_penalty_type=case.get("kl", "low_var_kl"), clip_grad=1.0,
        reinpp_kl_beta=case.get("reinpp_kl_beta", 0.0))
    for (n, a), (_, b) in zip(m_ref.named_parameters(), m_ora.named_parameters()):
        assert torch.equal(a, b), n
    assert not torch.equal(next(m_ref.parameters()), next(base.parameters()))  # it did train
    assert len(want_metrics) == len(got_metrics) == n_mini
    for w, g in zip(want_metrics, got_metrics):
        for k in ("actor/final_loss", "actor/policy_loss", "actor/approx_kl", "actor/clip_fraction", "actor/entropy_loss",
                  "actor/kl_loss", "actor/grad_norm"):
            assert float(w[k]) == pytest.approx(g[k], rel=1e-6, abs=1e-9), k


@pytest.mark.parametrize("pack", [
    dict(variable_seq_lengths=True, max_tokens_per_mbs=40, eos_token_id=3),                      # the packed stream as long as it is
    dict(variable_seq_lengths=True, max_tokens_per_mbs=48, eos_token_id=3, entropy_bonus=0.02),  # + the (unshifted) entropy unpack
    dict(dynamic=True, variable_seq_lengths=False, max_tokens_per_mbs=28, eos_token_id=5),        # micro-batches cut by tokens, the
    dict(dynamic=True, variable_seq_lengths=False, max_tokens_per_mbs=33, eos_token_id=5, entropy_bonus=0.02),  # stream padded to the budget
    dict(dynamic=True, variable_seq_lengths=True, max_tokens_per_mbs=21, eos_token_id=5, temperature=0.8, kl_beta=0.05),
], ids=lambda p: "-".join(f"{k}={v}" for k, v in p.items()))
def test_oracle_iteration_with_sequence_packing_matches_the_reference(ref, one_rank_group, pack):
    """Packing on (runner.enable_dynamic_batch_size / actor.model.variable_seq_lengths): the reference's forward_batch packs the
    micro-batch (prepare_pack_fsdp, pack_fsdp_input), scores the packed stream and unpacks (unpack_fsdp_logprobs, unpack_sequences)
    -- all compiled from rlinf/hybrid_engines/fsdp/utils.py --, its _split_to_micro_batch cuts global batches by the token budget
    (split_dynamic_batch_size -> get_iterator_dynamic: best-fit-decreasing count, Karmarkar-Karp partitions).  The oracle loop
    with the same switches lands on the same parameters bit for bit."""
    resp, prompt, vocab, dim = 6, 4, 53, 16
    total, micro, n_mini = 16, 4, 2
    case = {k: pack[k] for k in ("entropy_bonus", "temperature", "kl_beta") if k in pack}
    torch.manual_seed(5)
    base = TL.TinyCausalLM(vocab, dim, max(prompt + resp, pack["max_tokens_per_mbs"]))
    batch = TL.synthetic_rollout_batch(7, total, prompt, resp, vocab)
    if case.get("kl_beta", 0) > 0:
        with torch.no_grad():
            batch["ref_logprobs"] = TL.forward_logprobs(base, batch, resp, case.get("temperature", 1.0)) + 0.05 * torch.randn(total, resp)
    m_ref, m_ora = copy.deepcopy(base), copy.deepcopy(base)
    me = reference_learner(ref, m_ref, resp=resp, prompt=prompt, micro=micro, n_mini=n_mini, total=total, case=case, pack=pack)
    feed = [{k: (v.clone() if isinstance(v, torch.Tensor) else v) for k, v in batch.items()}]
    me.get_batch = lambda _ch: (feed.pop(0), SimpleNamespace(num_sequence=total))
    _, want_metrics = me.run_training(None)
    from oracle import reference_loader as R
    helpers = _packing_helpers(R)
    opt = torch.optim.AdamW(m_ora.parameters(), lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01)
    opack = dict(max_prompt_len=prompt, encoder_seq_length=prompt + resp, max_tokens_per_mbs=pack["max_tokens_per_mbs"],
                 variable_seq_lengths=pack["variable_seq_lengths"], eos_token_id=pack["eos_token_id"],
                 dynamic=(lambda lens, k, eq: helpers["balanced"](lens, k, eq)) if pack.get("dynamic") else None)
    _, got_metrics = TL.iteration(
        m_ora, opt, {k: (v.clone() if isinstance(v, torch.Tensor) else v) for k, v in batch.items()}, response_len=resp,
        micro_batch=micro, n_minibatches=n_mini, seed=1234, adv_type="grpo", group_size=4, normalize_advantages=True,
        temperature=case.get("temperature", 1.0), loss_agg="token-mean", clip_ratio_low=0.2, clip_ratio_high=0.28,
        calculate_entropy=case.get("entropy_bonus", 0) > 0, entropy_bonus=case.get("entropy_bonus", 0.0),
        kl_beta=case.get("kl_beta", 0.0), kl_penalty_type="low_var_kl", clip_grad=1.0, pack=opack)
    for (n, a), (_, b) in zip(m_ref.named_parameters(), m_ora.named_parameters()):
        assert torch.equal(a, b), n
    assert not torch.equal(next(m_ref.parameters()), next(base.parameters()))
    for w, g in zip(want_metrics, got_metrics):
        for k in ("actor/final_loss", "actor/policy_loss", "actor/approx_kl", "actor/entropy_loss", "actor/kl_loss", "actor/grad_norm"):
            assert float(w[k]) == pytest.approx(g[k], rel=1e-6, abs=1e-9), k


# ---- pipeline mode: run_training_pipeline over the reference's own BatchResizingIterator ---------------------------------------
def _reference_batch_iterator_class(ref):
    from typing import Callable, Optional

    from oracle import reference_loader as R
    it_py = "rlinf/utils/data_iter_utils.py"
    import itertools
    split_list = R.load_function(it_py, "split_list")
    k_split = R.load_function(it_py, "get_iterator_k_split", torch=torch, UserDict=UserDict, logging=logging, split_list=split_list,
                              itertools=itertools, Union=None, Optional=None, Iterator=None)
    merge = R.load_function("rlinf/data/schema/reasoning_results.py", "RolloutResult.merge_batches", torch=torch)
    rollout_result = SimpleNamespace(merge_batches=staticmethod(merge))
    cls = R.load_class("rlinf/data/schema/reasoning_results.py", "BatchResizingIterator", torch=torch, Optional=Optional, Callable=Callable,
                       RolloutResult=rollout_result, get_iterator_k_split=k_split,
                       get_batch_size=lambda batch, key="input_ids": batch[key].size(0))
    return cls, rollout_result


def _pieces(batch, sizes):
    out, lo = [], 0
    for n in sizes:
        out.append({k: (v[lo:lo + n].clone() if isinstance(v, torch.Tensor) else v) for k, v in batch.items()})
        lo += n
    return out


@pytest.mark.parametrize("sizes,handler", [((16,), True), ((8, 8), True), ((4, 4, 8), True), ((4, 4, 4, 4), False), ((8, 8), False),
                                           ((16,), False), ((2, 2, 4, 8), True)], ids=str)
def test_batch_resizing_iterator_hands_out_what_the_reference_class_does(ref, sizes, handler):
    """rlinf_amd.data.batch_iterator.BatchResizingIterator against the reference class compiled from its source: the same
    micro-batches in the same order, the same bookkeeping, for pieces larger than / equal to / smaller than a global batch, with
    and without a global-batch handler."""
    from rlinf_amd.data.batch_iterator import BatchResizingIterator
    ref_cls, _ = _reference_batch_iterator_class(ref)
    total, micro, n_mini = 16, 2, 2
    batch = {k: v for k, v in TL.synthetic_rollout_batch(3, total, 4, 6, 53).items() if isinstance(v, torch.Tensor)}
    cfg = Cfg(algorithm=Cfg(shuffle_rollout=True), actor=Cfg(seed=1234))
    streams = []
    for cls in (ref_cls, BatchResizingIterator):
        feed = _pieces(batch, sizes)

        def get_batch(feed=feed):
            piece = feed.pop(0)
            return piece, SimpleNamespace(num_sequence=piece["input_ids"].shape[0])

        it = cls(cfg=cfg, get_batch_fn=get_batch, micro_batch_size=micro, total_batch_size=total, num_global_batches=n_mini,
                 forward_only=False)
        it.register_get_batch_handler(lambda b: {**b, "tag": b["rewards"] * 2})
        if handler:
            it.register_global_batch_handler(lambda b: {**b, "norm": b["rewards"] - b["rewards"].mean()})
        got, done = [], []
        peek = it.prefetch_one_batch()
        for i in range(total // micro):
            mb = next(it)
            if i == 0:
                assert mb is peek
            got.append(mb)
            done.append(it.global_batch_done)
        it.check_finished_global_batch()
        streams.append((got, done, it.get_all_batches()))
    (want, want_done, want_all), (got, got_done, got_all) = streams
    assert want_done == got_done
    for w, g in zip(want, got):
        assert w.keys() == g.keys()
        for k in w:
            assert torch.equal(w[k], g[k]), k
    for k in want_all:
        assert torch.equal(want_all[k], got_all[k]), k


@pytest.mark.parametrize("case", [
    dict(sizes=(16,)),
    dict(sizes=(8, 8), loss_agg="seq-mean-token-sum", temperature=0.7, entropy_bonus=0.01),
    dict(sizes=(4, 4, 8), kl_beta=0.05),
    dict(sizes=(4, 4, 4, 4), normalize=False, group_size=2),
    dict(sizes=(8, 4, 4), adv_type="reinpp", normalize=False),
], ids=lambda c: "-".join(f"{k}={v}" for k, v in c.items()))
def test_oracle_pipeline_iteration_matches_the_reference_run_training_pipeline(ref, one_rank_group, case):
    from functools import partial

    from oracle import reference_loader as R
    resp, prompt, vocab, dim = 6, 4, 53, 16
    total, micro, n_mini = 16, 4, 2
    torch.manual_seed(5)
    base = TL.TinyCausalLM(vocab, dim, prompt + resp)
    batch = TL.synthetic_rollout_batch(7, total, prompt, resp, vocab)
    if case.get("kl_beta", 0) > 0:
        with torch.no_grad():
            batch["ref_logprobs"] = TL.forward_logprobs(base, batch, resp, case.get("temperature", 1.0)) + 0.05 * torch.randn(total, resp)
    batch = {k: v for k, v in batch.items() if isinstance(v, torch.Tensor)}
    m_ref, m_ora = copy.deepcopy(base), copy.deepcopy(base)
    me = reference_learner(ref, m_ref, resp=resp, prompt=prompt, micro=micro, n_mini=n_mini, total=total, case=case)
    it_cls, rollout_result = _reference_batch_iterator_class(ref)
    # training_step tells a dict from the iterator by isinstance(batch, dict): recompile it next to the real class
    py = "rlinf/workers/actor/fsdp_actor_worker.py"
    mu = ref.metric_utils
    worker_stub = SimpleNamespace(torch_device_type="cpu", torch_platform=SimpleNamespace(current_device=lambda: torch.device("cpu")))
    du = R.load_distributed_utils()

    def masked_normalization_cpu(x, mask):
        src = R.load_function("rlinf/utils/distributed.py", "masked_normalization", torch=torch, np=__import__("numpy"))
        orig = torch.Tensor.cuda
        torch.Tensor.cuda = lambda self, *a, **k: self
        try:
            return src(x, mask)
        finally:
            torch.Tensor.cuda = orig

    seen = {}

    def metrics_probe(b, *_a, **_k):
        seen["batch"] = b
        return {"n": b["input_ids"].shape[0]}, None, None

    pipeline = R.load_function(py, "FSDPActor.run_training_pipeline", torch=torch, BatchResizingIterator=it_cls, partial=partial,
                               masked_normalization=masked_normalization_cpu, compute_math_rollout_metrics=metrics_probe, Channel=None)
    feed = _pieces(batch, case["sizes"])
    me.get_batch = lambda _ch: (lambda p: (p, SimpleNamespace(num_sequence=p["input_ids"].shape[0])))(feed.pop(0))
    me.is_pipeline = True
    rollout_metrics, want_metrics = pipeline(me, None)
    assert rollout_metrics == {"n": total} and not feed
    opt = torch.optim.AdamW(m_ora.parameters(), lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01)
    trained_on, got_metrics = TL.pipeline_iteration(
        m_ora, opt, _pieces(batch, case["sizes"]), total=total, response_len=resp, micro_batch=micro, n_minibatches=n_mini, seed=1234,
        adv_type=case.get("adv_type", "grpo"), group_size=case.get("group_size", 4), normalize_advantages=case.get("normalize", True),
        temperature=case.get("temperature", 1.0), loss_agg=case.get("loss_agg", "token-mean"), clip_ratio_low=0.2, clip_ratio_high=0.28,
        calculate_entropy=case.get("entropy_bonus", 0) > 0, entropy_bonus=case.get("entropy_bonus", 0.0),
        kl_beta=case.get("kl_beta", 0.0), kl_penalty_type=case.get("kl", "low_var_kl"), clip_grad=1.0,
        reinpp_kl_beta=case.get("reinpp_kl_beta", 0.0))
    for (n, a), (_, b) in zip(m_ref.named_parameters(), m_ora.named_parameters()):
        assert torch.equal(a, b), n
    assert not torch.equal(next(m_ref.parameters()), next(base.parameters()))
    for k in ("input_ids", "advantages", "rewards"):
        assert torch.equal(seen["batch"][k], trained_on[k]), k
    assert len(want_metrics) == len(got_metrics) == n_mini
    for w, g in zip(want_metrics, got_metrics):
        for k in ("actor/final_loss", "actor/policy_loss", "actor/approx_kl", "actor/clip_fraction", "actor/grad_norm"):
            assert float(w[k]) == pytest.approx(g[k], rel=1e-6, abs=1e-9), k

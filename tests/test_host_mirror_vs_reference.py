"""The product's torch-only host mirrors (shaping, aggregation, KL) against the reference's own functions on CPU tensors --
no oracle in between.  Runs where /root/reference exists."""

import pytest
import torch

from conftest import free_port

from rlinf_amd.algorithms import utils as AU
from rlinf_amd.utils import utils as UU

pytestmark = pytest.mark.reference


def _same(a, b, what=""):
    if a is None or b is None:
        assert a is None and b is None, what
        return
    if isinstance(a, torch.Tensor):
        assert a.shape == b.shape and a.dtype == b.dtype and torch.equal(a, b), what
    else:
        assert a == b, what


@pytest.mark.parametrize("C", [1, 3])
@pytest.mark.parametrize("reward_type", ["action_level", "chunk_level"])
@pytest.mark.parametrize("adv_type", ["gae", "grpo"])
def test_embodied_advantage_shaping(ref, C, reward_type, adv_type):
    g = torch.Generator().manual_seed(C)
    n, B = 6, 8
    kw = dict(rewards=torch.rand(n, B, C, generator=g), dones=torch.rand(n + 1, B, C, generator=g) < 0.2,
              values=torch.randn(n + 1, B, 1 if reward_type == "chunk_level" else C, generator=g),  # one value per chunk step
              loss_mask=torch.rand(n, B, C, generator=g) < 0.7,
              loss_mask_sum=torch.randint(0, 9, (n, B, C), generator=g), reward_type=reward_type, adv_type=adv_type, group_size=4)
    want = ref.algo_utils.preprocess_embodied_advantages_inputs(**{k: (v.clone() if torch.is_tensor(v) else v) for k, v in kw.items()})
    got = AU.preprocess_embodied_advantages_inputs(**kw)
    assert set(want) == set(got)
    for k in want:
        _same(want[k], got[k], k)
    adv, ret = torch.randn(want["n_steps"], B, generator=g), torch.randn(want["n_steps"], B, generator=g)
    w = ref.algo_utils.postprocess_embodied_advantages_outputs(advantages=adv, returns=ret, **want)
    o = AU.postprocess_embodied_advantages_outputs(advantages=adv, returns=ret, **got)
    assert set(w) == set(o)
    for k in w:
        _same(w[k], o[k], k)


@pytest.mark.parametrize("adv_type", ["gae", "grpo", "reinpp", "raw"])
def test_reasoning_advantage_shaping(ref, adv_type):
    g = torch.Generator().manual_seed(2)
    bsz, seq = 8, 11
    kw = dict(rewards=torch.randn(bsz, generator=g), loss_mask=torch.rand(bsz, seq, generator=g) < 0.7,
              values=torch.randn(bsz, seq, generator=g), logprob=torch.randn(bsz, seq, generator=g),
              ref_logprob=torch.randn(bsz, seq, generator=g), adv_type=adv_type, group_size=4)
    want = ref.algo_utils.preprocess_reasoning_advantages_inputs(**kw)
    got = AU.preprocess_reasoning_advantages_inputs(**kw)
    assert set(want) == set(got)
    for k in want:
        _same(want[k], got[k], k)
    adv = torch.randn(seq, bsz, generator=g)
    for a, b in zip(ref.algo_utils.postprocess_reasoning_advantages_outputs(adv, adv * 2),
                    AU.postprocess_reasoning_advantages_outputs(adv, adv * 2)):
        _same(a, b)
        assert b.is_contiguous()


@pytest.mark.parametrize("logprob_type", ["token_level", "action_level", "chunk_level"])
@pytest.mark.parametrize("reward_type", ["action_level", "chunk_level"])
def test_loss_input_shaping(ref, logprob_type, reward_type):
    g = torch.Generator().manual_seed(4)
    bsz, C, A = 10, 2, 4
    shape = (bsz, 1) if reward_type == "chunk_level" else (bsz, C)
    kw = dict(logprobs=torch.randn(bsz, C * A, generator=g), old_logprobs=torch.randn(bsz, C * A, generator=g),
              advantages=torch.randn(*shape, generator=g), logprob_type=logprob_type, single_action_dim=A,
              loss_mask=torch.rand(*shape, generator=g) < 0.7, loss_mask_sum=torch.randint(1, 9, shape, generator=g),
              values=torch.randn(*shape, generator=g), prev_values=torch.randn(*shape, generator=g),
              returns=torch.randn(*shape, generator=g), reward_type=reward_type, clip_ratio_low=0.2)
    if logprob_type != "chunk_level" and reward_type == "chunk_level":
        pytest.skip("per-step advantages cannot be expanded to per-chunk log-probs: the reference fails downstream too")
    want = ref.algo_utils.preprocess_loss_inputs(**kw)
    got = AU.preprocess_loss_inputs(**kw)
    for k in ("logprobs", "old_logprobs", "advantages", "loss_mask", "loss_mask_sum", "values", "prev_values", "returns"):
        _same(want[k], got[k], k)
    assert got["clip_ratio_low"] == 0.2


def test_aggregations_and_kl(ref):
    g = torch.Generator().manual_seed(6)
    v = torch.randn(7, 9, generator=g)
    for mask in (torch.rand(7, 9, generator=g) < 0.6, torch.zeros(7, 9, dtype=torch.bool), None):
        _same(ref.utils.masked_mean(v, mask), UU.masked_mean(v, mask))
        if mask is not None:
            _same(ref.utils.masked_mean(v, mask, axis=-1), UU.masked_mean(v, mask, axis=-1))
            _same(ref.utils.masked_sum(v, mask), UU.masked_sum(v, mask))
            _same(ref.utils.seq_mean_token_sum(v, mask), UU.seq_mean_token_sum(v, mask))
    mask = torch.rand(7, 9, generator=g) < 0.6
    mask[:, 0] = True
    _same(ref.utils.seq_mean_token_mean(v, mask), UU.seq_mean_token_mean(v, mask))
    ratio = torch.rand(7, 9, generator=g) + 0.1
    _same(ref.utils.masked_mean_ratio(v, mask, ratio), UU.masked_mean_ratio(v, mask, ratio))
    for name in ("token-mean", "seq-mean-token-sum", "seq-mean-token-mean"):
        _same(ref.utils.get_loss_agg_func(name)(v, mask), UU.get_loss_agg_func(name)(v, mask), name)
    with pytest.raises(Exception):
        UU.get_loss_agg_func("median")
    a, b = torch.randn(5, 6, generator=g), torch.randn(5, 6, generator=g) * 30
    for kind in ("kl", "k1", "abs", "mse", "k2", "low_var_kl", "k3"):
        _same(ref.algo_utils.kl_penalty(a, b, kind), AU.kl_penalty(a, b, kind), kind)


# ---- sequence packing of the reasoning learner (rlinf/hybrid_engines/fsdp/utils.py:812-1022, rlinf/utils/data_iter_utils.py:447-700) ----
def _pack_case(seed, bsz, prompt, resp):
    g = torch.Generator().manual_seed(seed)
    plen = torch.randint(0, prompt + 1, (bsz,), generator=g)   # (0: a row without a prompt -- its first response log-prob is the
    rlen = torch.randint(1, resp + 1, (bsz,), generator=g)     #  prepended zero when it opens the packed stream)
    m_batch = {"prompt_lengths": plen, "response_lengths": rlen}
    ids = torch.randint(1, 97, (bsz, prompt + resp), generator=g)
    return m_batch, ids


@pytest.mark.parametrize("seed,bsz,prompt,resp", [(0, 5, 4, 6), (1, 1, 3, 3), (2, 8, 7, 9), (3, 4, 0, 5)])
def test_sequence_packing_host_side_matches_the_reference(ref, seed, bsz, prompt, resp):
    """The product's pack gathers and -- the part the scoring kernel stores through -- its unpack index maps, against the reference's
    own pack_sequences / prepare_pack_fsdp / unpack_fsdp_logprobs / unpack_sequences compiled from source: the log-prob of packed
    row t must land exactly where the reference's shift-right + scatter + [:, -response_len:] puts it, the entropy where its
    unshifted unpack puts it, and nothing else may be written."""
    from oracle import reference_loader as R
    from rlinf_amd.hybrid_engines.fsdp import utils as PK
    fs = "rlinf/hybrid_engines/fsdp/utils.py"
    r_pack = R.load_function(fs, "pack_sequences", torch=torch)
    r_unpack = R.load_function(fs, "unpack_sequences", torch=torch)
    r_prepare = R.load_function(fs, "prepare_pack_fsdp")
    r_unpack_lp = R.load_function(fs, "unpack_fsdp_logprobs", torch=torch, unpack_sequences=r_unpack)
    m_batch, ids = _pack_case(seed, bsz, prompt, resp)
    S = prompt + resp
    want_se = r_prepare(m_batch, prompt)
    got_se = PK.prepare_pack_fsdp(m_batch, prompt)
    assert want_se == got_se
    idx_starts, idx_ends = got_se
    total = sum(idx_ends) - sum(idx_starts)
    for fixed, budget in ((False, total), (True, total + 5), (True, total)):
        want = r_pack(ids, idx_starts, idx_ends, budget, 7, fixed)
        got = PK.pack_sequences(ids, idx_starts, idx_ends, budget, 7, fixed)
        _same(want, got, f"pack_sequences fixed={fixed}")
        L = got.numel()
        # the unpack: give every packed row a recognisable value and push it through the reference's own functions
        lp_rows = torch.arange(1, L + 1, dtype=torch.float32).unsqueeze(0) * -1.0      # "log-prob computed from row t" = -(t + 1)
        ent_rows = torch.arange(1, L + 1, dtype=torch.float32).unsqueeze(0) * 0.5
        want_lp = r_unpack_lp(torch.zeros(1, L, 3), got.unsqueeze(0), idx_starts=idx_starts, idx_ends=idx_ends, max_seq_len_unpack=S,
                              eos_token_id=7, compute_logprobs_fn=lambda _l, _t: lp_rows)[:, -resp:]
        want_ent = r_unpack(ent_rows, idx_starts, idx_ends, S, pad_val=0)[:, -resp:]
        lp_dst, ent_dst = PK.unpack_index_maps(idx_starts, idx_ends, L, S, resp, "cpu")
        got_lp, got_ent = torch.zeros(bsz * resp), torch.zeros(bsz * resp)
        keep = lp_dst >= 0
        got_lp[lp_dst[keep].long()] = lp_rows[0][keep]
        keep = ent_dst >= 0
        got_ent[ent_dst[keep].long()] = ent_rows[0][keep]
        _same(want_lp, got_lp.view(bsz, resp), "log-prob unpack map")
        _same(want_ent, got_ent.view(bsz, resp), "entropy unpack map")
        assert lp_dst[lp_dst >= 0].unique().numel() == int((lp_dst >= 0).sum())     # no destination written twice


@pytest.mark.parametrize("seed", range(6))
def test_dynamic_batch_split_matches_the_reference(ref, seed):
    """runner.enable_dynamic_batch_size: get_seqlen_BFD_partitions and split_dynamic_batch_size (-> get_iterator_dynamic) of the
    reference, compiled from source (its two device="cuda" one-element tensors built on the host), against the product's."""
    import heapq
    import itertools
    from collections import UserDict

    import torch.distributed as dist

    from oracle import reference_loader as R
    from rlinf_amd.hybrid_engines.fsdp import utils as PK
    from rlinf_amd.workers.actor.fsdp_actor_worker import seqlen_balanced_partitions
    it_py = "rlinf/utils/data_iter_utils.py"
    started = not dist.is_initialized()
    if started:
        dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{free_port()}", rank=0, world_size=1)
    try:
        class CpuTorch:
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
        split = R.load_function(it_py, "split_dynamic_batch_size", get_iterator_dynamic=dyn)
        g = torch.Generator().manual_seed(seed)
        bsz, prompt, resp = 12, 6, 10
        plen = torch.randint(1, prompt + 1, (bsz,), generator=g)
        rlen = torch.randint(1, resp + 1, (bsz,), generator=g)
        pos = torch.arange(prompt + resp).unsqueeze(0)
        attn = (pos >= (prompt - plen).unsqueeze(1)) & (pos < (prompt + rlen).unsqueeze(1))
        batch = dict(input_ids=torch.randint(1, 50, (bsz, prompt + resp), generator=g), attention_mask=attn, prompt_lengths=plen,
                     response_lengths=rlen, tags=[f"s{i}" for i in range(bsz)])
        budget = int(torch.randint(prompt + resp, 3 * (prompt + resp), (1,), generator=g))
        lens = attn.sum(dim=1).tolist()
        assert PK.get_seqlen_bfd_partitions(lens, budget) == bfd(lens, budget)
        it, _, n_want, parts_want = split(batch=dict(batch), cp_world_size=1, vpp_world_size=1, max_tokens_per_mbs=budget,
                                          microbatch_group_size_per_vp_stage=1)
        micro_want = list(it)
        micro_got, n_got, parts_got = PK.split_dynamic_batch_size(dict(batch), budget, seqlen_balanced_partitions, None)
        assert n_got == n_want and [list(p) for p in parts_got] == [list(p) for p in parts_want]
        for w, o in zip(micro_want, micro_got):
            assert set(w) == set(o)
            for k in w:
                _same(w[k], o[k], k)
        assert PK.get_reverse_idx(sum((list(p) for p in parts_got), [])) == R.load_function(it_py, "get_reverse_idx", copy=__import__("copy"))(
            sum((list(p) for p in parts_want), []))
    finally:
        if started:
            dist.destroy_process_group()

"""The product's torch-only host mirrors (shaping, aggregation, KL) against the reference's own functions on CPU tensors --
no oracle in between.  Runs where /root/reference exists."""

import pytest
import torch

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

"""Pin the CPU oracle (oracle/ppo_oracle.py) against the REAL reference executed on CPU.

Runs only where /root/reference exists (the build container).  Everything here must be bit-exact:
the restatement uses the same torch ops in the same order as the reference.
"""

import copy

import pytest
import torch

from conftest import free_port, synth_rollout
from oracle import ppo_oracle as O

pytestmark = pytest.mark.reference


def _eq(a, b):
    assert a.shape == b.shape, (a.shape, b.shape)
    assert a.dtype == b.dtype, (a.dtype, b.dtype)
    assert torch.equal(a, b), (a - b).abs().max() if a.is_floating_point() else "mismatch"


@pytest.mark.parametrize("C", [1, 4])
@pytest.mark.parametrize("p_done", [0.0, 0.05, 0.5])
def test_loss_mask(ref, C, p_done):
    d = synth_rollout(T=12, B=16, C=C, p_done=p_done)["dones"]
    m0, s0 = ref.metric_utils.compute_loss_mask(d)
    m1, s1 = O.loss_mask_from_dones(d)
    _eq(m0, m1)
    _eq(s0, s1)


@pytest.mark.parametrize("C", [1, 2])
@pytest.mark.parametrize("norm", [True, False])
@pytest.mark.parametrize("gl", [(0.8, 0.9), (0.99, 0.95)])
@pytest.mark.parametrize("use_mask", [False, True])
def test_gae_embodied(ref, C, norm, gl, use_mask):
    r = synth_rollout(T=16, B=32, C=C, p_done=0.05)
    lm, lms = (ref.metric_utils.compute_loss_mask(r["dones"]) if use_mask else (None, None))
    kw = dict(task_type="embodied", adv_type="gae", rewards=r["rewards"], dones=r["dones"],
              values=r["values"], gamma=gl[0], gae_lambda=gl[1], group_size=8,
              reward_type="action_level", loss_mask=lm, loss_mask_sum=lms,
              normalize_advantages=norm)
    want = ref.registry.calculate_adv_and_returns(**kw)
    got = O.embodied_adv_and_returns(adv_type="gae", rewards=r["rewards"], dones=r["dones"],
                                     values=r["values"], gamma=gl[0], gae_lambda=gl[1], loss_mask=lm,
                                     loss_mask_sum=lms, normalize_advantages=norm)
    _eq(want["advantages"], got["advantages"])
    _eq(want["returns"], got["returns"])


def test_gae_critic_free_and_chunk_level(ref):
    r = synth_rollout(T=10, B=8, C=4, p_done=0.1)
    want = ref.advantages.compute_gae_advantages_and_returns(
        rewards=r["rewards"][..., 0], values=None, dones=r["dones"][..., 0], gamma=0.9, gae_lambda=0.8)
    got = O.gae_tb(r["rewards"][..., 0], r["dones"][..., 0], None, 0.9, 0.8)
    _eq(want[0], got[0])
    _eq(want[1], got[1])
    kw = dict(task_type="embodied", adv_type="gae", rewards=r["rewards"], dones=r["dones"],
              values=r["values"][..., :1], gamma=0.99, gae_lambda=0.95, reward_type="chunk_level",
              loss_mask=None, loss_mask_sum=None)
    want = ref.registry.calculate_adv_and_returns(**kw)
    got = O.embodied_adv_and_returns(adv_type="gae", rewards=r["rewards"], dones=r["dones"],
                                     values=r["values"][..., :1], gamma=0.99, gae_lambda=0.95,
                                     reward_type="chunk_level")
    _eq(want["advantages"], got["advantages"])
    _eq(want["returns"], got["returns"])


@pytest.mark.parametrize("C", [1, 2])
@pytest.mark.parametrize("G", [2, 8])
def test_grpo_embodied(ref, C, G):
    r = synth_rollout(T=12, B=32, C=C, p_done=0.08)
    lm, lms = ref.metric_utils.compute_loss_mask(r["dones"])
    kw = dict(task_type="embodied", adv_type="grpo", rewards=r["rewards"], dones=r["dones"],
              values=None, gamma=1.0, gae_lambda=1.0, group_size=G, reward_type="action_level",
              loss_mask=lm, loss_mask_sum=lms)
    want = ref.registry.calculate_adv_and_returns(**kw)
    got = O.embodied_adv_and_returns(adv_type="grpo", rewards=r["rewards"], dones=r["dones"],
                                     loss_mask=lm, loss_mask_sum=lms, group_size=G)
    _eq(want["advantages"], got["advantages"])
    assert "returns" not in want and "returns" not in got


def _loss_inputs(seed, mb=64, C=1, A=8, masked=False):
    g = torch.Generator().manual_seed(seed)
    lp = (torch.randn(mb, C * A, generator=g) * 0.3 - 1.0).requires_grad_(True)
    old = lp.detach() + torch.randn(mb, C * A, generator=g) * 0.1
    adv = torch.randn(mb, C, generator=g)
    v = torch.randn(mb, C, generator=g).requires_grad_(True)
    pv = v.detach() + torch.randn(mb, C, generator=g) * 0.7
    ret = torch.randn(mb, C, generator=g) * 3
    lm = (torch.rand(mb, C, generator=g) < 0.7) if masked else None
    lms = (torch.randint(1, 50, (mb, 1), generator=g).expand(mb, C)) if masked else None
    return lp, old, adv, v, pv, ret, lm, lms


@pytest.mark.parametrize("logprob_type", ["action_level", "token_level", "chunk_level"])
@pytest.mark.parametrize("masked", [False, True])
@pytest.mark.parametrize("variant", ["plain", "dual", "logclip", "warmup", "ratio_agg"])
def test_actor_critic_loss(ref, logprob_type, masked, variant):
    C = 2
    lp, old, adv, v, pv, ret, lm, lms = _loss_inputs(7, C=C, masked=masked)
    if logprob_type == "chunk_level":
        adv, v, pv, ret = adv[:, 0], v[:, :1].detach().squeeze(-1).requires_grad_(True), pv[:, 0], ret[:, 0]
        lm = None if lm is None else lm[:, 0]
        lms = None if lms is None else lms[:, 0]
    extra = {}
    if variant == "dual":
        extra["clip_ratio_c"] = 3.0
    if variant == "logclip":
        extra.update(clip_log_ratio_min=-0.05, clip_log_ratio_max=0.05)
    if variant == "warmup":
        extra["critic_warmup"] = True
    mes = 50 if variant == "ratio_agg" else None
    if variant == "ratio_agg" and not masked:
        pytest.skip("ratio aggregation needs a loss mask")
    kw = dict(loss_type="actor_critic", task_type="embodied", logprob_type=logprob_type,
              reward_type="action_level", single_action_dim=8, logprobs=lp, values=v, old_logprobs=old,
              advantages=adv, returns=ret, prev_values=pv, clip_ratio_high=0.2, clip_ratio_low=0.2,
              value_clip=1.0, huber_delta=10.0, loss_mask=lm, loss_mask_sum=lms,
              max_episode_steps=mes, **extra)
    loss0, m0 = ref.registry.policy_loss(**kw)
    g0 = torch.autograd.grad(loss0, [lp, v], allow_unused=True)

    shaped = O.shape_loss_inputs(lp, old, adv, logprob_type, 8, loss_mask=lm, loss_mask_sum=lms,
                                 values=v, prev_values=pv, returns=ret)
    loss1, m1 = O.ppo_actor_critic_loss(clip_ratio_low=0.2, clip_ratio_high=0.2, value_clip=1.0,
                                        huber_delta=10.0, max_episode_steps=mes, **extra, **shaped)
    g1 = torch.autograd.grad(loss1, [lp, v], allow_unused=True)
    assert torch.equal(loss0.detach(), loss1.detach())
    for a, b in zip(g0, g1):
        if a is None:
            assert b is None
        else:
            _eq(a, b)
    for k in ("actor/policy_loss", "actor/policy_loss_abs", "actor/ratio", "actor/ratio_abs",
              "actor/clipped_ratio", "actor/dual_cliped_ratio", "actor/approx_kl", "actor/clip_fraction",
              "critic/value_loss", "critic/value_clip_ratio"):
        assert m0[k] == pytest.approx(float(m1[k]), rel=0, abs=0), k
    ev_ref = {k.split("/")[-1]: val for k, val in m0.items() if "explained_variance" in k}
    for k in O.EV_KEYS:
        assert ev_ref[k] == float(m1[f"ev/{k}"]), k


@pytest.mark.parametrize("logprob_type", ["action_level", "token_level"])
@pytest.mark.parametrize("masked", [False, True])
@pytest.mark.parametrize("ratio_agg", [False, True])
def test_chunk_level_reward_shapes_against_the_reference(ref, logprob_type, masked, ratio_agg):
    """reward_type='chunk_level' (utils.py:296-308: advantages / mask / values flattened to [bsz]) under a [bsz, C] (action_level)
    or [bsz, C, A] (token_level) ratio, C = 3: the un-broadcast mask count in the action_level metrics (losses.py:288-290)."""
    if ratio_agg and not masked:
        pytest.skip("ratio aggregation needs a loss mask")
    g = torch.Generator().manual_seed(5)
    bsz, C, A = 48, 3, 4
    lp = (torch.randn(bsz, C * A, generator=g) * 0.3).requires_grad_(True)
    old = lp.detach() + 0.1 * torch.randn(bsz, C * A, generator=g)
    adv, pv, ret = (torch.randn(bsz, 1, generator=g) for _ in range(3))
    v = torch.randn(bsz, 1, generator=g).requires_grad_(True)
    lm = (torch.rand(bsz, 1, generator=g) < 0.7) if masked else None
    lms = torch.randint(1, 50, (bsz, 1), generator=g) if masked else None
    mes = 50 if ratio_agg else None
    kw = dict(loss_type="actor_critic", task_type="embodied", logprob_type=logprob_type, reward_type="chunk_level",
              single_action_dim=A, logprobs=lp, values=v, old_logprobs=old, advantages=adv, returns=ret, prev_values=pv,
              clip_ratio_high=0.2, clip_ratio_low=0.2, value_clip=1.0, huber_delta=10.0, loss_mask=lm, loss_mask_sum=lms,
              max_episode_steps=mes)
    loss0, m0 = ref.registry.policy_loss(**kw)
    g0 = torch.autograd.grad(loss0, [lp, v])
    shaped = O.shape_loss_inputs(lp, old, adv, logprob_type, A, loss_mask=lm, loss_mask_sum=lms, values=v, prev_values=pv,
                                 returns=ret, reward_type="chunk_level")
    loss1, m1 = O.ppo_actor_critic_loss(clip_ratio_low=0.2, clip_ratio_high=0.2, value_clip=1.0, huber_delta=10.0,
                                        max_episode_steps=mes, **shaped)
    g1 = torch.autograd.grad(loss1, [lp, v])
    assert torch.equal(loss0.detach(), loss1.detach())
    for a, b in zip(g0, g1):
        _eq(a, b)
    for k in ("actor/policy_loss", "actor/policy_loss_abs", "actor/ratio", "actor/ratio_abs", "actor/clipped_ratio",
              "actor/approx_kl", "actor/clip_fraction", "critic/value_loss"):
        assert m0[k] == pytest.approx(float(m1[k]), rel=0, abs=0), k


def test_grpo_actor_loss_name(ref):
    lp, old, adv, *_ = _loss_inputs(3)
    kw = dict(loss_type="actor", task_type="embodied", logprob_type="action_level",
              reward_type="action_level", single_action_dim=8, logprobs=lp, old_logprobs=old,
              advantages=adv, clip_ratio_high=0.28, clip_ratio_low=0.2, loss_mask=None,
              loss_mask_sum=None, max_episode_steps=None)
    loss0, m0 = ref.registry.policy_loss(**kw)
    shaped = O.shape_loss_inputs(lp, old, adv, "action_level", 8)
    loss1, m1 = O.ppo_actor_loss(shaped["logprobs"], shaped["old_logprobs"], shaped["advantages"], 0.2, 0.28)
    assert torch.equal(loss0.detach(), loss1.detach())


def _paired_policies(ref, seed=0):
    torch.manual_seed(seed)
    theirs = ref.mlp_policy.MLPPolicy(42, 8, 1, True, False)
    ours = O.OracleMLPPolicy(42, 8, 1)
    missing = ours.load_state_dict(copy.deepcopy(theirs.state_dict()), strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    return theirs, ours


def test_policy_param_names_and_init_distribution(ref):
    theirs, ours = _paired_policies(ref)
    assert [n for n, _ in theirs.named_parameters()] == [n for n, _ in ours.named_parameters()]
    assert sum(p.numel() for p in ours.parameters()) == 287504
    torch.manual_seed(5)
    a = ref.mlp_policy.MLPPolicy(42, 8, 1, True, False)
    torch.manual_seed(5)
    b = O.OracleMLPPolicy(42, 8, 1)
    # same construction order + same seed -> identical initial weights
    for (n, p), (m, q) in zip(a.named_parameters(), b.named_parameters()):
        assert n == m
        _eq(p.detach(), q.detach())


def test_value_free_policy_matches_the_reference(ref):
    """add_value_head False (mlp_policy.py:56-62, :283-286): no value-head parameters, the same init stream for the rest, zeros as
    prev_values."""
    torch.manual_seed(5)
    theirs = ref.mlp_policy.MLPPolicy(42, 8, 1, False, False)
    torch.manual_seed(5)
    ours = O.OracleMLPPolicy(42, 8, 1, add_value_head=False)
    assert [n for n, _ in theirs.named_parameters()] == [n for n, _ in ours.named_parameters()]
    assert not any("value_head" in n for n, _ in ours.named_parameters())
    for (n, p), (m, q) in zip(theirs.named_parameters(), ours.named_parameters()):
        _eq(p.detach(), q.detach())
    states = torch.randn(16, 42, generator=torch.Generator().manual_seed(3))
    torch.manual_seed(7)
    acts0, res0 = theirs.predict_action_batch({"states": states}, mode="train")
    torch.manual_seed(7)
    a1, lp1, v1 = ours.act(states, eps=torch.randn(16, 8), mode="train")
    _eq(acts0.reshape(16, 8), a1)
    _eq(res0["prev_logprobs"], lp1)
    _eq(res0["prev_values"], v1)
    assert v1.shape == (16, 1) and not v1.any()
    with pytest.raises(NotImplementedError):
        theirs.default_forward({"states": states, "action": a1})


def test_policy_rollout_with_injected_noise(ref):
    theirs, ours = _paired_policies(ref)
    g = torch.Generator().manual_seed(11)
    states = torch.randn(64, 42, generator=g)
    torch.manual_seed(99)
    acts0, res0 = theirs.predict_action_batch({"states": states}, mode="train")
    torch.manual_seed(99)
    eps = torch.randn(64, 8)  # same generator state -> the draw torch.normal(mean, std) consumed
    a1, lp1, v1 = ours.act(states, eps=eps, mode="train")
    _eq(acts0.reshape(64, 8), a1)
    _eq(res0["prev_logprobs"], lp1)
    _eq(res0["prev_values"], v1)
    _eq(res0["forward_inputs"]["action"], a1)
    acts0, res0 = theirs.predict_action_batch({"states": states}, mode="eval")
    a1, lp1, v1 = ours.act(states, mode="eval")
    _eq(acts0.reshape(64, 8), a1)
    _eq(res0["prev_logprobs"], lp1)


def test_policy_training_forward_and_optimizer_step(ref):
    theirs, ours = _paired_policies(ref)
    g = torch.Generator().manual_seed(21)
    mb = 128
    states = torch.randn(mb, 42, generator=g)
    action = torch.randn(mb, 8, generator=g) * 0.5
    o0 = theirs.default_forward({"states": states, "action": action})
    o1 = ours.evaluate(states, action)
    for k in ("logprobs", "entropy", "values"):
        _eq(o0[k].detach(), o1[k].detach())
    # one full optimizer step, reference-style, against the oracle's
    batch = dict(states=states, action=action, prev_logprobs=o0["logprobs"].detach() + 0.05,
                 advantages=torch.randn(mb, 1, generator=g), prev_values=torch.randn(mb, 1, generator=g),
                 returns=torch.randn(mb, 1, generator=g))
    opt0 = torch.optim.AdamW(
        [{"params": [p for n, p in theirs.named_parameters() if "value_head" not in n], "lr": 3e-4,
          "betas": (0.9, 0.999)},
         {"params": [p for n, p in theirs.named_parameters() if "value_head" in n], "lr": 3e-4,
          "betas": (0.9, 0.999)}], eps=1e-8, weight_decay=0.01)
    opt0.zero_grad()
    out = theirs.default_forward({"states": states, "action": action})
    loss, _ = ref.registry.policy_loss(
        loss_type="actor_critic", task_type="embodied", logprob_type="action_level",
        reward_type="action_level", single_action_dim=8, logprobs=out["logprobs"], values=out["values"],
        old_logprobs=batch["prev_logprobs"], advantages=batch["advantages"], returns=batch["returns"],
        prev_values=batch["prev_values"], clip_ratio_high=0.2, clip_ratio_low=0.2, value_clip=1.0,
        huber_delta=10.0, loss_mask=None, loss_mask_sum=None, max_episode_steps=50)
    loss.backward()
    gn0 = torch.nn.utils.clip_grad_norm_(theirs.parameters(), 0.5)
    opt0.step()
    opt1 = O.build_adamw(ours)
    m = O.ppo_minibatch_step(ours, opt1, batch)
    assert float(gn0) == float(m["actor/grad_norm"])
    for (n, p), (_, q) in zip(theirs.named_parameters(), ours.named_parameters()):
        _eq(p.detach(), q.detach())


def test_flatten_and_shuffle(ref):
    g = torch.Generator().manual_seed(2)
    T, B = 6, 10
    batch = dict(rewards=torch.rand(T, B, 1, generator=g), dones=torch.rand(T + 1, B, 1, generator=g) < 0.1,
                 prev_values=torch.randn(T + 1, B, 1, generator=g),
                 forward_inputs=dict(states=torch.randn(T, B, 42, generator=g)))
    perm = torch.randperm(T * B, generator=torch.Generator().manual_seed(1234))
    want = ref.nested.process_nested_dict_for_train(batch, perm)
    got = O.flatten_and_shuffle(batch, perm)
    _eq(want["rewards"], got["rewards"])
    _eq(want["dones"], got["dones"])
    _eq(want["prev_values"], got["prev_values"])
    _eq(want["forward_inputs"]["states"], got["forward_inputs"]["states"])
    w = ref.nested.split_dict_to_chunk(want, 4)
    o = O.chunk_batch(got, 4)
    for i in range(4):
        _eq(w[i]["forward_inputs"]["states"], o[i]["forward_inputs"]["states"])
        _eq(w[i]["rewards"], o[i]["rewards"])


# ---- token tier (oracle/token_oracle.py) -----------------------------------------------------------------
from oracle import token_oracle as TO  # noqa: E402
from oracle.make_golden import token_batch  # noqa: E402


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("vocab", [7, 517, 4096])
def test_token_logprob_entropy(ref, dtype, vocab):
    b = token_batch(101, 4, 6, vocab)
    x = b["logits"].to(dtype)
    _eq(ref.utils.compute_logprobs_from_logits(x, b["labels"]), TO.logprobs_from_logits(x, b["labels"]))
    _eq(ref.utils.compute_entropy_from_logits(x), TO.entropy_from_logits(x))


@pytest.mark.parametrize("kind", ["kl", "k1", "abs", "mse", "k2", "low_var_kl", "k3"])
def test_token_kl_penalty(ref, kind):
    b = token_batch(102, 4, 16, 11)
    a, c = b["ref_logprobs"] * 9, b["old_logprobs"]  # wide enough to hit both clamps of k3
    _eq(ref.algo_utils.kl_penalty(a, c, kind), TO.kl_penalty(a, c, kind))


@pytest.mark.parametrize("agg", ["token-mean", "seq-mean-token-sum", "seq-mean-token-mean"])
@pytest.mark.parametrize("zero_first", [False, True])
@pytest.mark.parametrize("clip_c", [None, 3.0])
def test_token_actor_loss(ref, agg, zero_first, clip_c):
    b = token_batch(103, 8, 10, 13, zero_first=zero_first)
    lp0 = b["old_logprobs"] + 0.4 * torch.randn(8, 10, generator=torch.Generator().manual_seed(5))
    outs = []
    for mod, fn, aggs in ((ref, ref.losses.compute_ppo_actor_loss, ref.utils.get_loss_agg_func),
                          (TO, TO.token_actor_loss, TO.get_loss_agg_func)):
        lp = lp0.clone().requires_grad_(True)
        loss, metrics = fn(logprobs=lp, old_logprobs=b["old_logprobs"], advantages=b["advantages"], clip_ratio_low=0.2,
                           clip_ratio_high=0.28, loss_mask=b["loss_mask"], clip_ratio_c=clip_c,
                           loss_agg_func=aggs(agg), fast_path_zero_loss_mask=True, clip_log_ratio_max=0.5)
        g = torch.autograd.grad(loss, lp)[0] if loss.requires_grad else None
        outs.append((loss.detach(), metrics, g))
    (l0, m0, g0), (l1, m1, g1) = outs
    _eq(l0, l1)
    assert set(m0) == set(m1)
    for k in m0:
        _eq(m0[k], m1[k])
    assert (g0 is None) == (g1 is None)
    if g0 is not None:
        _eq(g0, g1)


def test_token_reasoning_shaping(ref):
    b = token_batch(104, 8, 10, 5)
    values = torch.randn(8, 10)
    for adv_type in ("gae", "grpo"):
        want = ref.algo_utils.preprocess_reasoning_advantages_inputs(
            rewards=b["rewards"], loss_mask=b["loss_mask"], values=values, adv_type=adv_type, group_size=4)
        got = TO.preprocess_reasoning(b["rewards"], b["loss_mask"], adv_type, values=values, group_size=4)
        for k in ("rewards", "loss_mask", "dones", "values"):
            _eq(want[k], got[k])
    want = ref.registry.calculate_adv_and_returns(task_type="reasoning", adv_type="grpo", rewards=b["rewards"],
                                                  loss_mask=b["loss_mask"], group_size=4)
    _eq(want[0], TO.grpo_reasoning_advantages(b["rewards"], b["loss_mask"], 4))
    assert want[1] is None


@pytest.mark.parametrize("kl,beta", [("", 0.0), ("kl", 0.01), ("abs", 0.3), ("mse", 0.3), ("low_var_kl", 0.001)])
@pytest.mark.parametrize("masks", ["prefix", "ragged", "empty_rows", "all_false"])
def test_reinpp_reasoning(ref, kl, beta, masks):
    from oracle.make_golden import reinpp_batch
    rewards, mask, lp, rlp = reinpp_batch(21, 10, 29, masks)
    want = ref.registry.calculate_adv_and_returns(task_type="reasoning", adv_type="reinpp", rewards=rewards.clone(), loss_mask=mask,
                                                  group_size=5, kl_beta=beta, logprob=lp, ref_logprob=rlp, kl_penalty_type=kl)
    _eq(want[0], TO.reinpp_reasoning_advantages(rewards.clone(), mask, 5, False, beta, lp, rlp, kl))
    assert want[1] is None
    # the group baseline cannot run in the reference (1-D src against a 2-D scatter index); the restatement fails alike
    for fn in (lambda: ref.registry.calculate_adv_and_returns(task_type="reasoning", adv_type="reinpp", rewards=rewards.clone(),
                                                              loss_mask=mask, group_size=5, use_reinpp_baseline=True),
               lambda: TO.reinpp_reasoning_advantages(rewards.clone(), mask, 5, True)):
        with pytest.raises(IndexError):
            fn()


@pytest.mark.parametrize("masked", [False, True])
def test_masked_normalization(ref, masked, monkeypatch):
    """rlinf/utils/distributed.py:866-937 moves its inputs with .cuda(); neutralised here so that it runs on CPU tensors."""
    from oracle import reference_loader
    du = reference_loader.load_distributed_utils()
    monkeypatch.setattr(torch.Tensor, "cuda", lambda self, *a, **k: self)
    g = torch.Generator().manual_seed(6)
    x = torch.randn(400, 3, generator=g) * 2 + 0.5
    mask = (torch.rand(400, 3, generator=g) < 0.6) if masked else None
    _eq(du.masked_normalization(x, mask), O.masked_normalization(x, mask))


@pytest.mark.parametrize("splits", [1, 2])
def test_trajectory_buffer_row_alignment(ref, splits):
    """a7 / SURVEY A.1: the same sequence of ChunkStepResults -- a bootstrap row without reward, T rows that pair step t's
    policy output with step t-1's env output, one closing row with env output and a value only -- appended to the
    reference's EmbodiedTrajectoryBuilder and to the resident TrajectoryBuffer (on CPU tensors here); trajectories, their
    batch-dim splits and the converted batch agree field by field, and the oracle's rollout() lays its rows out the same."""
    from oracle import ppo_loop as L
    from oracle import reference_loader
    from rlinf_amd.data import embodied_types as mine
    m = reference_loader.load_trajectory_builder()
    T, B, A, D = 5, 4, 8, 42
    env = L.synthetic_env_tensors(0, T, B, D, max_episode_steps=3)
    torch.manual_seed(5)
    pol = O.OracleMLPPolicy(D, A, 1)
    eps = torch.randn(T, B, A, generator=torch.Generator().manual_seed(1))
    ref_builder = m.builder.EmbodiedTrajectoryBuilder(max_episode_length=3)
    buf = mine.TrajectoryBuffer(T, B, D, A, 1, device="cpu", max_episode_length=3)
    obs = env["obs"][0]
    prev_env = dict(dones=torch.zeros(B, 1, dtype=torch.bool), terminations=torch.zeros(B, 1, dtype=torch.bool),
                    truncations=torch.zeros(B, 1, dtype=torch.bool), rewards=None)  # bootstrap_step: no reward yet
    for t in range(T + 1):
        fields = dict(prev_env)
        if t < T:
            a, lp, v = pol.act(obs, eps=eps[t], mode="train")
            fields.update(actions=a, prev_logprobs=lp, prev_values=v, versions=torch.full_like(lp, 2.0),
                          forward_inputs={"states": obs.clone(), "action": a.clone()})
        else:
            fields.update(prev_values=pol.value_head.mlp(obs).detach())
        ref_builder.append_step_result(m.types.ChunkStepResult(**{k: (v.clone() if torch.is_tensor(v) else v) for k, v in fields.items()}))
        buf.append_step_result(mine.ChunkStepResult(**fields))
        if t < T:
            d = env["dones"][t + 1].unsqueeze(-1)
            prev_env = dict(dones=d, terminations=d.clone(), truncations=torch.zeros_like(d), rewards=env["rewards"][t].unsqueeze(-1).clone())
            obs = env["obs"][t + 1]
    want_trajs = ref_builder.to_splited_trajectories(splits)
    got_trajs = buf.to_splited_trajectories(splits)
    names = ("actions", "rewards", "terminations", "truncations", "dones", "prev_logprobs", "prev_values", "versions")
    for w, g in zip(want_trajs, got_trajs):
        for n in names:
            _eq(getattr(w, n), getattr(g, n).contiguous())
        for k in ("states", "action"):
            _eq(w.forward_inputs[k], g.forward_inputs[k].contiguous())
    want = m.types.convert_trajectories_to_batch(want_trajs)
    got = mine.convert_trajectories_to_batch(got_trajs)
    for n in names:
        _eq(want[n], got[n].contiguous())
    assert got["prev_values"].shape[0] == T + 1 and got["rewards"].shape[0] == T and not got["dones"][0].any()
    batch = L.rollout(pol, env, eps, 0.8, auto_reset=False)  # no bootstrap term: rewards are the env's
    for n in ("rewards", "dones", "prev_values", "prev_logprobs"):
        _eq(batch[n], got[n].contiguous())
    _eq(batch["forward_inputs"]["states"], got["forward_inputs"]["states"].contiguous())


@pytest.mark.parametrize("warmup", [False, True])
def test_build_optimizer_groups(ref, warmup):
    """FSDPModelManager.build_optimizer (fsdp_model_manager.py:501-590), the method compiled on its own with a stand-in
    ``self``: two AdamW groups -- names containing value_head at value_lr, the rest at lr -- or the value head alone while
    the critic warms up (everything else frozen); its state-initialising empty step leaves parameters and step counts as
    they were, so a fresh torch AdamW over the same groups (the oracle's build_adamw) is the same optimizer."""
    from types import SimpleNamespace

    from oracle import reference_loader

    class Cfg(dict):
        __getattr__ = dict.__getitem__

    fn = reference_loader.load_function(
        "rlinf/hybrid_engines/fsdp/fsdp_model_manager.py", "FSDPModelManager.build_optimizer", torch=torch,
        warmup_optimizer_state=ref.utils.warmup_optimizer_state,
        Worker=SimpleNamespace(torch_device_type="cpu", torch_platform=SimpleNamespace(is_available=lambda: False)))
    me = SimpleNamespace(_cfg=Cfg(optim=Cfg(adam_beta1=0.9, adam_beta2=0.999, lr=3e-4, value_lr=1e-3),
                                  fsdp_config={"sharding_strategy": "no_shard"}),
                         _logger=SimpleNamespace(info=lambda *a: None), store_requires_grad_param_name=[])
    torch.manual_seed(0)
    pol = O.OracleMLPPolicy(42, 8, 1)
    before = [p.detach().clone() for p in pol.parameters()]
    opt = fn(me, pol, enable_critic_warmup=warmup)
    assert all(torch.equal(a, b) for a, b in zip(before, pol.parameters()))
    assert all(int(opt.state[p]["step"]) == 0 and not opt.state[p]["exp_avg"].any() for g in opt.param_groups for p in g["params"])
    names = {id(p): n for n, p in pol.named_parameters()}
    got = [(sorted(names[id(p)] for p in g["params"]), g["lr"], g["betas"], g["eps"], g["weight_decay"]) for g in opt.param_groups]
    critic = sorted(n for n in names.values() if "value_head" in n)
    actor = sorted(n for n in names.values() if "value_head" not in n)
    if warmup:
        assert got == [(critic, 1e-3, (0.9, 0.999), 1e-8, 1e-2)]
        assert all(p.requires_grad == ("value_head" in n) for n, p in pol.named_parameters())
        return
    ours = O.build_adamw(pol, lr=3e-4, value_lr=1e-3)
    want = [(sorted(names[id(p)] for p in g["params"]), g["lr"], g["betas"], g["eps"], g["weight_decay"]) for g in ours.param_groups]
    assert got == want == [(actor, 3e-4, (0.9, 0.999), 1e-8, 1e-2), (critic, 1e-3, (0.9, 0.999), 1e-8, 1e-2)]


@pytest.mark.parametrize("bootstrap_type", ["always", "standard"])
def test_bootstrap_rewards(ref, bootstrap_type):
    """EnvWorker.compute_bootstrap_rewards (env_worker.py:718-758), the method compiled on its own and called with a stand-in
    ``self``: r[:, -1] += gamma * V(final_obs) where the last sub-step was done ("always") / truncated ("standard")."""
    from types import SimpleNamespace

    from oracle import reference_loader
    fn = reference_loader.load_function("rlinf/workers/env/env_worker.py", "EnvWorker.compute_bootstrap_rewards", torch=torch)
    g = torch.Generator().manual_seed(8)
    B, C = 12, 3
    rewards = torch.rand(B, C, generator=g)
    term, trunc = torch.rand(B, C, generator=g) < 0.3, torch.rand(B, C, generator=g) < 0.3
    values = torch.randn(B, 1, generator=g)
    alg = {"bootstrap_type": bootstrap_type}
    me = SimpleNamespace(cfg=SimpleNamespace(env=SimpleNamespace(train=SimpleNamespace(auto_reset=True)),
                                             algorithm=SimpleNamespace(get=alg.get, gamma=0.8)))
    env_out = SimpleNamespace(rewards=rewards, dones=term | trunc, truncations=trunc)
    want = fn(me, env_out, values, None)
    flags = env_out.dones if bootstrap_type == "always" else trunc
    _eq(want, O.bootstrap_rewards(rewards, flags, values, 0.8))
    assert fn(me, SimpleNamespace(rewards=None), values, None) is None  # the bootstrap row carries no reward (A.1)


@pytest.mark.parametrize("masked", [False, True, "empty"])
def test_rollout_metrics(ref, masked, monkeypatch):
    """a15: compute_rollout_metrics (metric_utils.py:422-506) in a one-rank gloo group against the learner's
    _metrics_from_reductions over the (sum, count, -min, max) rows that rlx_rollout_metrics produces on the device (formed with
    torch here; the kernel itself is checked against the same boolean-index form in tests/test_gpu_advantages.py): masked mean /
    min / max, NaN when nothing is selected, the [T, B] mask broadcast over the trailing dim."""
    import math
    import sys
    import types

    import torch.distributed as dist

    from rlinf_amd.workers.actor import EmbodiedFSDPActor
    from test_end_to_end import make_cfg
    platform = types.SimpleNamespace(current_device=lambda: torch.device("cpu"))
    wmod = types.ModuleType("rlinf.scheduler.worker.worker")
    wmod.Worker = types.SimpleNamespace(torch_platform=platform)
    monkeypatch.setitem(sys.modules, "rlinf.scheduler.worker", types.ModuleType("rlinf.scheduler.worker"))
    monkeypatch.setitem(sys.modules, "rlinf.scheduler.worker.worker", wmod)
    started = False
    if not dist.is_initialized():
        dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{free_port()}", rank=0, world_size=1)
        started = True
    try:
        g = torch.Generator().manual_seed(3)
        T, B = 7, 6
        batch = dict(rewards=torch.rand(T, B, 1, generator=g), advantages=torch.randn(T, B, 1, generator=g),
                     returns=torch.randn(T, B, 1, generator=g))
        if masked:
            batch["loss_mask"] = (torch.rand(T, B, 1, generator=g) < 0.5) if masked is True else torch.zeros(T, B, 1, dtype=torch.bool)
        want = ref.metric_utils.compute_rollout_metrics(dict(batch))
        names, rows = ["rewards", "advantages", "returns"], []
        for k in names:
            v = batch[k]
            sel = v.reshape(-1) if "loss_mask" not in batch else v[batch["loss_mask"].expand_as(v)]
            rows.append([float(sel.double().sum()), float(sel.numel()), float(-sel.min()) if sel.numel() else float("-inf"),
                         float(sel.max()) if sel.numel() else float("-inf")])
        got = EmbodiedFSDPActor(make_cfg())._metrics_from_reductions(names, torch.tensor(rows, dtype=torch.float64))
        assert set(want) == set(got) == {"rewards", "advantages_mean", "advantages_max", "advantages_min", "returns_mean",
                                         "returns_max", "returns_min"}
        for k, v in want.items():
            assert (math.isnan(v) and math.isnan(got[k])) or got[k] == pytest.approx(v, rel=1e-6, abs=1e-7), (k, v, got[k])
    finally:
        if started:
            dist.destroy_process_group()


def test_pipeline_stage_shuffles(ref):
    """EnvWorker.pack_pipeline_micro_batches (env_worker.py:1519-1537), compiled on its own: every stage batch is flattened
    and shuffled with the rank's stateful generator, stage after stage -- the row order oracle.ppo_loop.pipeline_permutation
    hands the learner, as indices into the rank's [T, B] buffer whose stages are contiguous env blocks."""
    from types import SimpleNamespace

    from oracle import ppo_loop as L
    from oracle import reference_loader
    fn = reference_loader.load_function("rlinf/workers/env/env_worker.py", "EnvWorker.pack_pipeline_micro_batches", torch=torch,
                                        flatten_embodied_batch=ref.utils.flatten_embodied_batch, pack_batch=ref.utils.pack_batch,
                                        split_dict_to_chunk=ref.nested.split_dict_to_chunk)
    T, B, stages, micro = 6, 8, 2, 12
    n = B // stages
    ids = torch.arange(T * B, dtype=torch.float32).view(T, B, 1)  # every row of the rank's buffer carries its flat index
    me = SimpleNamespace(shuffle_rollout=True, shuffle_generators={0: torch.Generator().manual_seed(1234)},
                         cfg=SimpleNamespace(actor=SimpleNamespace(micro_batch_size=micro)))
    seen = []
    for st in range(stages):
        stage_batch = {"prev_logprobs": ids[:, st * n:(st + 1) * n], "forward_inputs": {"states": ids[:, st * n:(st + 1) * n].clone()},
                       "prev_values": torch.zeros(T + 1, n, 1), "loss_mask": None}
        for mb in fn(me, stage_batch, 0):
            assert set(mb) == {"prev_logprobs", "forward_inputs::states", "prev_values"} and mb["prev_values"].shape[0] == micro
            seen.append(mb["prev_logprobs"].reshape(-1))
    want = torch.cat(seen).long()
    got = L.pipeline_permutation(T, B, stages, torch.Generator().manual_seed(1234))
    assert torch.equal(want, got)
    me.shuffle_rollout = False
    plain = torch.cat([mb["prev_logprobs"].reshape(-1) for mb in fn(me, {"prev_logprobs": ids[:, :n]}, 0)]).long()
    assert torch.equal(plain, ids[:, :n].reshape(-1).long())


def test_async_flatten_rollout_batch(ref):
    """flatten_rollout_batch_for_train (async_ppo_fsdp_worker.py:42-69): its module needs the whole FSDP stack, so the function
    is compiled from its source on its own; it is the same flatten-and-gather the synchronous learner uses."""
    from typing import Optional

    from oracle import reference_loader
    fn = reference_loader.load_function("rlinf/workers/actor/async_ppo_fsdp_worker.py", "flatten_rollout_batch_for_train",
                                        torch=torch, Optional=Optional)
    d = synth_rollout(T=6, B=5, C=1)
    batch = dict(rewards=d["rewards"], dones=d["dones"], prev_values=d["values"], prev_logprobs=torch.randn(6, 5, 8),
                 versions=torch.full((6, 5, 8), 3.0), loss_mask=None,
                 forward_inputs=dict(states=torch.randn(6, 5, 42), action=torch.randn(6, 5, 8)))
    perm = torch.randperm(30, generator=torch.Generator().manual_seed(2))
    want, got = fn(batch, perm), O.flatten_and_shuffle(batch, perm)
    assert set(want) == set(got) and want["loss_mask"] is None and got["loss_mask"] is None
    for k in ("rewards", "dones", "prev_values", "prev_logprobs", "versions"):
        _eq(want[k], got[k])
    for k in ("states", "action"):
        _eq(want["forward_inputs"][k], got["forward_inputs"][k])
    plain = fn(batch, None)
    _eq(plain["prev_values"], d["values"][:-1].reshape(30, 1))


# ---- decoupled PPO (oracle a19b) ---------------------------------------------------------------------------
@pytest.mark.parametrize("logprob_type", ["action_level", "token_level", "chunk_level"])
@pytest.mark.parametrize("prox_mode", ["given", "old", "versions"])
@pytest.mark.parametrize("masked,thr", [(False, None), (True, None), (True, 1.05)])
def test_decoupled_actor_critic_loss(ref, logprob_type, prox_mode, masked, thr):
    g = torch.Generator().manual_seed(17)
    bsz, C, A = 48, 2, 4
    lp0 = torch.randn(bsz, C * A, generator=g) * 0.3
    old = lp0 + 0.1 * torch.randn(bsz, C * A, generator=g)
    prox = lp0 + 0.05 * torch.randn(bsz, C * A, generator=g) if prox_mode == "given" else None
    versions = None
    if prox_mode == "versions":
        versions = torch.randint(-1, 6, (bsz, 1), generator=g).float().expand(bsz, C * A).contiguous()
    n_adv = (bsz,) if logprob_type == "chunk_level" else (bsz, C)
    adv = torch.randn(*n_adv, generator=g)
    vals, pv, ret = (torch.randn(*n_adv, generator=g) for _ in range(3))
    lm = lms = None
    if masked:
        lm = torch.rand(*n_adv, generator=g) < 0.7
        lms = lm.sum(dim=0, keepdim=True).expand_as(lm).contiguous()
    common = dict(clip_ratio_low=0.2, clip_ratio_high=0.28, clip_ratio_c=3.0, value_clip=0.5, huber_delta=1.0,
                  max_episode_steps=80, critic_warmup=False)
    outs = []
    for which in ("ref", "oracle"):
        lp = lp0.clone().requires_grad_(True)
        v = vals.clone().requires_grad_(True)
        if which == "ref":
            kw = ref.algo_utils.preprocess_loss_inputs(
                logprobs=lp, old_logprobs=old, advantages=adv, logprob_type=logprob_type, single_action_dim=A,
                loss_mask=lm, loss_mask_sum=lms, values=v, prev_values=pv, returns=ret, versions=versions,
                proximal_logprobs=prox, current_version=5, behave_weight_threshold=thr, **common)
            loss, m = ref.losses.compute_decoupled_ppo_actor_critic_loss(**kw)
        else:
            shaped = O.shape_loss_inputs(lp, old, adv, logprob_type, A, loss_mask=lm, loss_mask_sum=lms, values=v,
                                         prev_values=pv, returns=ret)
            p2, v2 = O.shape_decoupled_inputs(prox, versions, logprob_type, A, bsz, shaped["logprobs"].shape)
            loss, m = O.decoupled_actor_critic_loss(proximal_logprobs=p2, versions=v2, current_version=5,
                                                    behave_weight_threshold=thr, **common, **shaped)
        g_lp, g_v = torch.autograd.grad(loss, [lp, v])
        outs.append((loss.detach(), m, g_lp, g_v))
    (l0, m0, a0, b0), (l1, m1, a1, b1) = outs
    _eq(l0, l1), _eq(a0, a1), _eq(b0, b1)
    actor = lambda m: {k for k in m if k.startswith("actor/")}  # noqa: E731  (the critic's keys are pinned elsewhere)
    assert actor(m0) == actor(m1)
    for k in sorted(actor(m0)) + ["critic/value_loss", "critic/value_clip_ratio"]:
        x, y = m0[k], m1[k]
        if isinstance(x, torch.Tensor):
            _eq(x, y)
        else:
            assert x == y, k


def test_fold_rollout_epochs(ref):
    """oracle a8 against process_nested_dict_for_adv (rlinf/utils/nested_dict_process.py:251-269)."""
    g = torch.Generator().manual_seed(0)
    E, n, B = 3, 5, 4
    nested = {"rewards": torch.randn(E * n, B, 2, generator=g), "dones": torch.rand(E * (n + 1), B, 2, generator=g) < 0.3,
              "forward_inputs": {"states": torch.randn(E * n, B, 7, generator=g)}}
    want = ref.nested.process_nested_dict_for_adv(nested, E)
    got = O.fold_rollout_epochs(nested, E)
    _eq(want["rewards"], got["rewards"]), _eq(want["dones"], got["dones"])
    _eq(want["forward_inputs"]["states"], got["forward_inputs"]["states"])
    # epoch e of the stacked time axis lands in batch columns [e*B, (e+1)*B): the layout the env worker writes in place
    assert torch.equal(got["rewards"][:, B:2 * B], nested["rewards"][n:2 * n])


def test_stats_normalisation(ref):
    """oracle a12b against rlinf/utils/distributed.py:942-965."""
    from oracle import reference_loader
    du = reference_loader.load_distributed_utils()
    g = torch.Generator().manual_seed(2)
    x = torch.randn(9, 33, 1, generator=g) * 2 + 0.5
    for mask in (None, torch.rand(9, 33, 1, generator=g) < 0.6, torch.zeros(9, 33, 1, dtype=torch.bool)):
        s0, s1 = du.masked_stats(x, mask), O.masked_stats(x, mask)
        _eq(s0, s1)
        _eq(du.normalize_from_stats(x, s0), O.normalize_from_stats(x, s1))
    two = du.masked_stats(x[:4]) + du.masked_stats(x[4:])
    _eq(du.normalize_from_stats(x, two), O.normalize_from_stats(x, O.masked_stats(x[:4]) + O.masked_stats(x[4:])))


@pytest.mark.parametrize("etype,C", [("action_level", 1), ("action_level", 2), ("chunk_level", 1), ("token_level", 1)])
@pytest.mark.parametrize("masked", [False, True])
def test_entropy_bonus_shapes_against_the_reference(ref, etype, C, masked):
    """reshape_entropy (rlinf/utils/utils.py:384-408) + masked_mean (:323-330) as train_micro_batch chains them
    (embodied_fsdp_actor_worker.py:679-690).  chunk_level with a [bsz, 1] mask broadcasts into an outer product: the result is
    the SUM of the row entropies; with C > 1 torch refuses the broadcast."""
    g = torch.Generator().manual_seed(2)
    bsz, A = 24, 4
    ent = torch.rand(bsz, C * A, generator=g)
    mask = (torch.rand(bsz, C, generator=g) < 0.6) if masked else None
    want = ref.utils.masked_mean(ref.utils.reshape_entropy(ent, entropy_type=etype, action_dim=A, batch_size=bsz), mask=mask)
    got = O.masked_mean(O.reshape_entropy(ent, etype, A, bsz), mask)
    _eq(want, got)
    if etype == "chunk_level" and masked:
        assert float(want) == pytest.approx(float(ent.sum()), rel=1e-6)  # bsz times the mean, whatever the mask holds
    if etype == "chunk_level":
        with pytest.raises(RuntimeError, match="must match the size of tensor"):
            ref.utils.masked_mean(ref.utils.reshape_entropy(torch.rand(bsz, 3 * A), entropy_type=etype, action_dim=A, batch_size=bsz),
                                  mask=torch.ones(bsz, 3, dtype=torch.bool))

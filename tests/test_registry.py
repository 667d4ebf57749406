"""Registry boundary (mirror of rlinf/algorithms/registry.py): host-side semantics on CPU, dispatch on GPU."""

import os

import pytest
import torch

from conftest import GOLDEN_DIR, synth_rollout
from oracle import ppo_oracle as O


def test_registry_names_errors_and_override_rule():
    import rlinf_amd.algorithms as A
    from rlinf_amd.algorithms import registry as R

    assert {"gae", "grpo"} <= set(A.ADV_REGISTRY) and {"actor_critic", "actor"} <= set(A.LOSS_REGISTRY)
    with pytest.raises(ValueError, match="not registered"):
        A.get_adv_and_returns("does-not-exist")
    with pytest.raises(ValueError, match="not registered"):
        A.get_policy_loss("Actor_Critic")  # lookups of losses are case-sensitive upstream (registry.py:71-74)
    assert A.get_adv_and_returns("GAE") is A.ADV_REGISTRY["gae"]  # advantage names are lower-cased
    with pytest.raises(KeyError):
        A.policy_loss(task_type="embodied")  # loss_type is required (registry.py:81)

    calls = []

    @A.register_advantage("My_Adv")
    def my_adv(rewards, dones, **kw):  # receives the reference's flattened [T,B] views
        calls.append((tuple(rewards.shape), tuple(dones.shape), kw["n_steps"]))
        return rewards * 2, None

    # last registration wins and drops the native fast path
    saved, saved_native = A.ADV_REGISTRY["gae"], R._NATIVE_ADV.get("gae")
    try:
        @A.register_advantage("gae")
        def fake_gae(rewards, values, dones, **kw):
            calls.append(("gae", tuple(rewards.shape), tuple(values.shape), tuple(dones.shape)))
            return rewards, rewards
        res = A.calculate_adv_and_returns(task_type="embodied", adv_type="gae", rewards=torch.ones(3, 4, 2),
                                          dones=torch.zeros(4, 4, 2, dtype=torch.bool), values=torch.zeros(4, 4, 2),
                                          reward_type="action_level")
        assert calls[-1] == ("gae", (6, 4), (7, 4), (7, 4))
        assert res["advantages"].shape == (3, 4, 2) and res["returns"].shape == (3, 4, 2)
    finally:
        A.ADV_REGISTRY["gae"] = saved
        R._NATIVE_ADV["gae"] = saved_native
        A.ADV_REGISTRY.pop("my_adv", None)


def test_cpu_inputs_without_gpu_fail_loudly():
    import rlinf_amd.algorithms as A
    from rlinf_amd._lib import RlxError
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    r = synth_rollout(T=4, B=8)
    with pytest.raises(RlxError, match="no CPU fallback"):
        A.calculate_adv_and_returns(task_type="embodied", adv_type="gae", rewards=r["rewards"], dones=r["dones"],
                                    values=r["values"], reward_type="action_level")


@pytest.mark.gpu
@pytest.mark.parametrize("where", ["cpu", "cuda"])
def test_calculate_adv_and_returns_embodied(where):
    import rlinf_amd.algorithms as A
    for case in torch.load(os.path.join(GOLDEN_DIR, "advantages.pt"), weights_only=False):
        p = case["params"]
        mv = (lambda t: None if t is None else t.to(where))
        lms = case["loss_mask_sum"]
        kw = dict(task_type="embodied", adv_type=p["adv_type"], rewards=mv(case["rewards"]), dones=mv(case["dones"]),
                  values=mv(case["values"]) if p["adv_type"] == "gae" else None, gamma=p["gamma"],
                  gae_lambda=p["gae_lambda"], group_size=p["group_size"], reward_type=p["reward_type"],
                  loss_mask=mv(case["loss_mask"]), loss_mask_sum=None if lms is None else mv(lms))
        if p["adv_type"] == "gae":
            kw["normalize_advantages"] = p["normalize_advantages"]
        out = A.calculate_adv_and_returns(**kw)
        assert out["advantages"].device.type == where
        torch.testing.assert_close(out["advantages"].cpu(), case["advantages"], rtol=1e-5, atol=5e-6)
        if case["returns"] is not None:
            torch.testing.assert_close(out["returns"].cpu(), case["returns"], rtol=1e-5, atol=5e-6)
        else:
            assert "returns" not in out


@pytest.mark.gpu
def test_registered_callees_keep_the_reference_contract():
    """The [T,B] callee contract (what rlinf_amd.ext re-registers inside a real RLinf)."""
    import rlinf_amd.algorithms as A
    r = synth_rollout(seed=3, T=24, B=64, p_done=0.05)
    rew, val, don = r["rewards"][..., 0], r["values"][..., 0], r["dones"][..., 0]
    want = O.gae_tb(rew, don, val, 0.99, 0.95)
    adv, ret = A.get_adv_and_returns("gae")(rewards=rew, values=val, dones=don, gamma=0.99, gae_lambda=0.95,
                                            some_unknown_kwarg=1)
    torch.testing.assert_close(adv, want[0], rtol=1e-5, atol=5e-6)
    torch.testing.assert_close(ret, want[1], rtol=1e-5, atol=5e-6)
    lm = O.loss_mask_from_dones(r["dones"])[0][..., 0]
    scores = O.first_episode_scores(rew, don)
    want = O.grpo_tb(scores, lm, 8)
    adv, none = A.get_adv_and_returns("grpo")(rewards=scores.reshape(-1, 8), loss_mask=lm, group_size=8)
    assert none is None
    torch.testing.assert_close(adv, want, rtol=2e-5, atol=2e-5)


@pytest.mark.gpu
def test_policy_loss_entry_and_lazy_metrics():
    import rlinf_amd.algorithms as A
    from rlinf_amd.algorithms.losses import EV_PREFIX, explained_variance_from_stats
    case = torch.load(os.path.join(GOLDEN_DIR, "losses.pt"), weights_only=False)[0]
    p = case["params"]
    lp = case["logprobs"].cuda().requires_grad_(True)
    v = case["values"].cuda().requires_grad_(True)
    loss, metrics = A.policy_loss(loss_type="actor_critic", task_type="embodied", logprob_type=p["logprob_type"],
                                  reward_type="action_level", single_action_dim=p["action_dim"], logprobs=lp, values=v,
                                  old_logprobs=case["old_logprobs"].cuda(), advantages=case["advantages"].cuda(),
                                  returns=case["returns"].cuda(), prev_values=case["prev_values"].cuda(),
                                  clip_ratio_high=0.2, clip_ratio_low=0.2, value_clip=1.0, huber_delta=10.0,
                                  loss_mask=None, loss_mask_sum=None, max_episode_steps=None)
    loss.backward()
    torch.testing.assert_close(loss.detach().cpu(), case["loss"], rtol=1e-5, atol=1e-6)
    assert set(k for k in case["metrics"] if not k.startswith("__")) <= set(metrics.keys())
    for k, want in case["metrics"].items():
        assert metrics[k] == pytest.approx(want, rel=2e-5, abs=1e-4), k
    metrics["actor/entropy_loss"] = 0.0
    stats = {k: metrics.pop(k) for k in list(metrics) if k.startswith(EV_PREFIX)}
    assert len(stats) == 5 and EV_PREFIX + "count" not in metrics
    want_ev = O.explained_variance({k.split("/")[-1]: val for k, val in stats.items()})
    assert explained_variance_from_stats(stats) == pytest.approx(want_ev, rel=1e-5)


# ---- routing maps: the known answers of the reference's tests/unit_tests/test_comm_mapper.py:40-150 -------------------------
_SEND = [("env", "rollout", 0, 2, 3, 12, [(0, 4), (1, 2)]), ("env", "rollout", 1, 2, 3, 12, [(1, 2), (2, 4)]),
         ("rollout", "env", 0, 3, 2, 12, [(0, 4)]), ("rollout", "env", 1, 3, 2, 12, [(0, 2), (1, 2)]),
         ("rollout", "env", 2, 3, 2, 12, [(1, 4)])]
_RECV = [("env", "rollout", 0, 2, 3, 12, [(0, 4)]), ("env", "rollout", 1, 2, 3, 12, [(0, 2), (1, 2)]),
         ("env", "rollout", 2, 2, 3, 12, [(1, 4)])]


def test_route_plans_known_answers():
    from rlinf_amd.scheduler import build_recv_plan, build_send_plan
    for src, dst, rank, sw, dw, bs, want in _SEND:
        plan = build_send_plan(src_group_name=src, dst_group_name=dst, src_rank=rank, src_world_size=sw, dst_world_size=dw,
                               tag="train", batch_size=bs)
        assert [(e.peer_rank, e.batch_size) for e in plan.entries] == want
        assert [e.offset for e in plan.entries] == [sum(s for _, s in want[:i]) for i in range(len(want))]
    for src, dst, rank, sw, dw, bs, want in _RECV:
        plan = build_recv_plan(src_group_name=src, dst_group_name=dst, dst_rank=rank, src_world_size=sw, dst_world_size=dw,
                               tag="train", batch_size=bs)
        assert [(e.peer_rank, e.batch_size) for e in plan.entries] == want


@pytest.mark.reference
def test_comm_mapper_vs_reference_on_a_grid():
    """CommMapper.get_dst_ranks / get_src_ranks (rlinf/scheduler/worker/routing.py:132-196), the reference's static methods
    compiled from their source, on every (batch, src world, dst world, rank) of a grid; and what a source sends to a
    destination is what that destination expects from it."""
    from oracle import reference_loader
    from rlinf_amd.scheduler import CommMapper
    if not reference_loader.available():
        pytest.skip("reference tree not present")
    dst_ref = reference_loader.load_function("rlinf/scheduler/worker/routing.py", "CommMapper.get_dst_ranks")
    src_ref = reference_loader.load_function("rlinf/scheduler/worker/routing.py", "CommMapper.get_src_ranks",
                                             CommMapper=type("CommMapper", (), {"get_dst_ranks": staticmethod(dst_ref)}))
    for sw in (1, 2, 3, 4, 6, 8):
        for dw in (1, 2, 3, 4, 6, 8):
            for mult in (1, 5):
                bs = sw * dw * mult * 2
                sent = {}
                for r in range(sw):
                    got = CommMapper.get_dst_ranks(bs, sw, dw, r)
                    assert got == dst_ref(bs, sw, dw, r)
                    sent.update({(r, d): n for d, n in got})
                for r in range(dw):
                    got = CommMapper.get_src_ranks(bs, sw, dw, r)
                    assert got == src_ref(bs, sw, dw, r)
                    assert all(sent[(s, r)] == n for s, n in got)
    with pytest.raises(AssertionError, match="must be divisible by src_world_size"):
        CommMapper.get_dst_ranks(10, 4, 2, 0)

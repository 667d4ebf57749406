"""rlinf_amd.ext (the RLINF_EXT_MODULE hook, rlinf/scheduler/cluster/utils.py:81-110) against the reference's REAL registry on
the GPU box: the registry, its pre/post-processing and its built-in callees are the reference's own files (oracle/_ref, staged by
oracle/stage_reference.py and shipped with the snapshot; /root/reference itself in the build container).

    expected   reference registry.policy_loss / calculate_adv_and_returns with its OWN callees, CPU tensors
    got        the same reference dispatcher functions AFTER rlinf_amd.ext.register() re-registered gae / grpo / actor_critic /
               actor / decoupled_actor_critic, fed raw embodied kwargs on DEVICE tensors:
               preprocess_loss_inputs (reference) -> HIP callee -> postprocess_loss_metric (reference)"""

import pytest
import torch

from conftest import synth_rollout
from oracle import reference_loader as RL

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module")
def hooked():
    if not RL.available():
        pytest.skip("neither /root/reference nor the staged oracle/_ref copy is present")
    ref = RL.load()
    mr = RL.load_models_registry()  # (register() also re-registers the model builder: the real rlinf/models/__init__.py must be there)
    reg = ref.registry
    saved = dict(reg.ADV_REGISTRY), dict(reg.LOSS_REGISTRY)
    saved_models = dict(mr.models._MODEL_REGISTRY)
    builtin = SimpleRegistry(reg, *saved)
    from rlinf_amd import ext
    ext.register()  # `from rlinf.algorithms import registry` inside resolves to the real module loaded above
    assert reg.ADV_REGISTRY["gae"] is not saved[0]["gae"] and reg.LOSS_REGISTRY["actor_critic"] is not saved[1]["actor_critic"]
    yield ref, builtin
    reg.ADV_REGISTRY.clear(), reg.ADV_REGISTRY.update(saved[0])
    reg.LOSS_REGISTRY.clear(), reg.LOSS_REGISTRY.update(saved[1])
    mr.models._MODEL_REGISTRY.clear(), mr.models._MODEL_REGISTRY.update(saved_models)


class SimpleRegistry:
    """The reference dispatchers run over a saved copy of the built-in registries (the expected side)."""

    def __init__(self, reg, adv, loss):
        self.reg, self.adv, self.loss = reg, adv, loss

    def _with(self, fn, **kw):
        cur = dict(self.reg.ADV_REGISTRY), dict(self.reg.LOSS_REGISTRY)
        self.reg.ADV_REGISTRY.clear(), self.reg.ADV_REGISTRY.update(self.adv)
        self.reg.LOSS_REGISTRY.clear(), self.reg.LOSS_REGISTRY.update(self.loss)
        try:
            return fn(**kw)
        finally:
            self.reg.ADV_REGISTRY.clear(), self.reg.ADV_REGISTRY.update(cur[0])
            self.reg.LOSS_REGISTRY.clear(), self.reg.LOSS_REGISTRY.update(cur[1])

    def policy_loss(self, **kw):
        return self._with(self.reg.policy_loss, **kw)

    def calculate_adv_and_returns(self, **kw):
        return self._with(self.reg.calculate_adv_and_returns, **kw)


def _dev(kw):
    out = {}
    for k, v in kw.items():
        if isinstance(v, torch.Tensor):
            t = v.detach().to(DEV)
            out[k] = t.requires_grad_(True) if v.requires_grad else t
        else:
            out[k] = v
    return out


@pytest.mark.parametrize("C", [1, 2])
@pytest.mark.parametrize("norm", [True, False])
@pytest.mark.parametrize("use_mask", [False, True])
@pytest.mark.parametrize("where", ["cuda", "cpu"])
def test_real_dispatcher_gae(hooked, C, norm, use_mask, where):
    """``where="cpu"``: the reference's trajectories are CPU tensors (embodied_types.py forces .cpu()); the registered callee
    stages them to the accelerator and hands the result back where the inputs live."""
    ref, builtin = hooked
    r = synth_rollout(T=16, B=32, C=C, p_done=0.05)
    lm, lms = (ref.metric_utils.compute_loss_mask(r["dones"]) if use_mask else (None, None))
    kw = dict(task_type="embodied", adv_type="gae", rewards=r["rewards"], dones=r["dones"], values=r["values"], gamma=0.8,
              gae_lambda=0.9, group_size=8, reward_type="action_level", loss_mask=lm, loss_mask_sum=lms, normalize_advantages=norm)
    want = builtin.calculate_adv_and_returns(**kw)
    got = ref.registry.calculate_adv_and_returns(**(_dev(kw) if where == "cuda" else dict(kw)))
    assert got["advantages"].device.type == where and got["advantages"].shape == want["advantages"].shape
    torch.testing.assert_close(got["advantages"].cpu(), want["advantages"], rtol=1e-5, atol=2e-6)
    torch.testing.assert_close(got["returns"].cpu(), want["returns"], rtol=1e-5, atol=2e-6)


@pytest.mark.parametrize("G", [2, 8])
def test_real_dispatcher_grpo(hooked, G):
    ref, builtin = hooked
    r = synth_rollout(T=12, B=32, C=1, p_done=0.08)
    lm, lms = ref.metric_utils.compute_loss_mask(r["dones"])
    kw = dict(task_type="embodied", adv_type="grpo", rewards=r["rewards"], dones=r["dones"], values=None, gamma=1.0, gae_lambda=1.0,
              group_size=G, reward_type="action_level", loss_mask=lm, loss_mask_sum=lms)
    want = builtin.calculate_adv_and_returns(**kw)
    # the reference's calculate_scores allocates its scores with torch.zeros(batch_size) on the CPU whatever the inputs' device
    # (algorithms/utils.py:134-152): its embodied GRPO dispatcher only works on CPU trajectories, which is what
    # the reference hands it; the registered callee stages them to the accelerator and returns CPU tensors
    got = ref.registry.calculate_adv_and_returns(**dict(kw))
    assert got["advantages"].device.type == "cpu"
    torch.testing.assert_close(got["advantages"].cpu(), want["advantages"], rtol=2e-5, atol=2e-5)
    assert "returns" not in got and "returns" not in want


def _loss_inputs(seed, mb=96, C=2, A=8, masked=False):
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


@pytest.mark.parametrize("loss_type", ["actor_critic", "actor"])
@pytest.mark.parametrize("logprob_type", ["action_level", "token_level"])
@pytest.mark.parametrize("variant", ["plain", "masked", "ratio_agg", "dual"])
def test_real_dispatcher_policy_loss(hooked, loss_type, logprob_type, variant):
    ref, builtin = hooked
    masked = variant in ("masked", "ratio_agg")
    lp, old, adv, v, pv, ret, lm, lms = _loss_inputs(7, masked=masked)
    extra = {"clip_ratio_c": 3.0} if variant == "dual" else {}
    kw = dict(loss_type=loss_type, task_type="embodied", logprob_type=logprob_type, reward_type="action_level", single_action_dim=8,
              logprobs=lp, old_logprobs=old, advantages=adv, clip_ratio_high=0.2, clip_ratio_low=0.2, loss_mask=lm,
              loss_mask_sum=lms, max_episode_steps=50 if variant == "ratio_agg" else None, **extra)
    if loss_type == "actor_critic":
        kw.update(values=v, returns=ret, prev_values=pv, value_clip=1.0, huber_delta=10.0)
    want_loss, want_m = builtin.policy_loss(**kw)
    want_g = torch.autograd.grad(want_loss, [lp] + ([v] if loss_type == "actor_critic" else []))
    dkw = _dev(kw)
    loss, metrics = ref.registry.policy_loss(**dkw)  # the reference's own policy_loss: preprocess -> HIP callee -> postprocess
    assert loss.is_cuda
    got_g = torch.autograd.grad(loss, [dkw["logprobs"]] + ([dkw["values"]] if loss_type == "actor_critic" else []))
    torch.testing.assert_close(loss.detach().cpu(), want_loss.detach(), rtol=1e-5, atol=1e-6)
    for a, b in zip(got_g, want_g):
        torch.testing.assert_close(a.cpu(), b, rtol=1e-5, atol=1e-7)
    for k, w in want_m.items():
        assert isinstance(metrics[k], float) or not isinstance(metrics[k], torch.Tensor), (k, type(metrics[k]))  # postprocess .item()s
        assert float(metrics[k]) == pytest.approx(float(w), rel=1e-5, abs=1e-6), k

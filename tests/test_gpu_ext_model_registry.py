"""The MODEL leg of rlinf_amd.ext (RLINF_EXT_MODULE, rlinf/scheduler/cluster/utils.py:81-110) against the reference's own code on
the GPU box: after ``ext.register()``

    rlinf.models.get_model(cfg)                      the reference's registry (rlinf/models/__init__.py:337-352) builds the HIP module
    model.predict_action_batch(env_obs=..., ...)     what MultiStepRolloutWorker.predict calls (huggingface_worker.py:500-530)
    EmbodiedFSDPActor.run_training / train_micro_batch + FSDPModelManager.build_optimizer / optimizer_step
                                                     compiled from the reference's files (oracle/_ref on the GPU box) and run over the
                                                     stand-in learner of test_reference_learner_loop.py with model = the HIP module on
                                                     the device and the registries re-registered by the hook

and the expected side is the same reference code with the reference's MLPPolicy and built-in callees on CPU."""

import copy
import sys

import pytest
import torch

from oracle import ppo_loop as L
from oracle import ppo_oracle as O
from oracle import reference_loader as RL
from test_reference_learner_loop import _learner, one_rank_group  # noqa: F401  (fixture)

pytestmark = pytest.mark.gpu
DEV = "cuda"


class Cfg(dict):
    __getattr__ = dict.__getitem__


@pytest.fixture(scope="module")
def hooked():
    if not RL.available():
        pytest.skip("neither /root/reference nor the staged oracle/_ref copy is present")
    ref = RL.load()
    mr = RL.load_models_registry()
    reg = ref.registry
    saved = dict(reg.ADV_REGISTRY), dict(reg.LOSS_REGISTRY), dict(mr.models._MODEL_REGISTRY)
    worker = sys.modules["rlinf.scheduler"].Worker
    was = getattr(worker, "torch_platform", None), getattr(worker, "torch_device_type", "cpu")
    worker.torch_platform, worker.torch_device_type = torch.cuda, "cuda"  # what Worker holds on a GPU node (worker.py:214-260)
    from rlinf_amd import ext
    ext.register()
    yield ref, mr, saved
    worker.torch_platform, worker.torch_device_type = was
    reg.ADV_REGISTRY.clear(), reg.ADV_REGISTRY.update(saved[0])
    reg.LOSS_REGISTRY.clear(), reg.LOSS_REGISTRY.update(saved[1])
    mr.models._MODEL_REGISTRY.clear(), mr.models._MODEL_REGISTRY.update(saved[2])


def _cfg(**kw):
    return Cfg(dict(model_type="mlp_policy", precision="32", obs_dim=42, action_dim=8, num_action_chunks=1, add_value_head=True,
                    add_q_head=False, is_lora=False), **kw)


def test_reference_get_model_builds_the_hip_module_on_the_device(hooked):
    from rlinf_amd.models.embodiment.mlp_policy_module import ReferenceNamedMLPPolicy

    ref, mr, _ = hooked
    torch.manual_seed(3)
    model = mr.models.get_model(_cfg())
    torch.manual_seed(3)
    theirs = ref.mlp_policy.MLPPolicy(42, 8, 1, True, False)
    assert isinstance(model, ReferenceNamedMLPPolicy)
    for (n, p), (m, q) in zip(theirs.named_parameters(), model.named_parameters()):
        assert n == m and q.is_cuda and torch.equal(p.detach(), q.detach().cpu()), n


@pytest.mark.parametrize("precision,tol", [("32", 1e-4), ("bf16", 3e-2)])
@pytest.mark.parametrize("value_head", [True, False])
def test_predict_action_batch_through_the_registered_module(hooked, precision, tol, value_head):
    """eval mode is deterministic (a = mu): compare with the reference module itself; train mode with this package's injected
    noise against the oracle (the reference draws inside Normal.sample)."""
    ref, mr, _ = hooked
    torch.manual_seed(7)
    model = mr.models.get_model(_cfg(precision=precision, add_value_head=value_head))
    theirs = ref.mlp_policy.MLPPolicy(42, 8, 1, value_head, False)
    theirs.load_state_dict({k: v.cpu() for k, v in model.state_dict().items()})
    obs = {"states": torch.randn(300, 42, generator=torch.Generator().manual_seed(1))}
    with torch.no_grad():
        want_a, want = theirs.predict_action_batch(env_obs=obs, mode="eval", return_obs=True)
    got_a, got = model.predict_action_batch(env_obs=obs, mode="eval", return_obs=True)
    assert got_a.is_cuda and got_a.shape == want_a.shape
    torch.testing.assert_close(got_a.cpu(), want_a, rtol=tol, atol=tol)
    torch.testing.assert_close(got["prev_logprobs"].cpu(), want["prev_logprobs"], rtol=tol, atol=max(tol, 2e-3 if precision == "bf16" else tol))
    torch.testing.assert_close(got["prev_values"].cpu(), want["prev_values"], rtol=tol, atol=tol)
    assert set(got["forward_inputs"]) == set(want["forward_inputs"])
    # the weights change behind the module's back (optimizer / weight syncer write in place): the next step must see them
    with torch.no_grad():
        for p in model.parameters():
            p.mul_(0.5)
        for p in theirs.parameters():
            p.mul_(0.5)
        want_a2, _ = theirs.predict_action_batch(env_obs=obs, mode="eval")
    got_a2, _ = model.predict_action_batch(env_obs=obs, mode="eval")
    torch.testing.assert_close(got_a2.cpu(), want_a2, rtol=tol, atol=tol)
    assert not torch.allclose(got_a2, got_a)
    if precision == "32" and value_head:
        ora = O.OracleMLPPolicy(42, 8, 1)
        ora.load_state_dict({k: v.cpu() for k, v in model.state_dict().items()})
        eps = torch.randn(300, 8, generator=torch.Generator().manual_seed(2))
        a, lp, v = ora.act(obs["states"], eps=eps, mode="train")
        got_a, got = model.predict_action_batch(env_obs=obs, mode="train", eps=eps)
        torch.testing.assert_close(got_a.reshape(300, 8).cpu(), a, rtol=tol, atol=tol)
        torch.testing.assert_close(got["prev_logprobs"].cpu(), lp, rtol=tol, atol=tol)


@pytest.mark.parametrize("shape", [dict(global_batch=40, micro_batch=40), dict(global_batch=80, micro_batch=20, entropy_bonus=0.01),
                                   dict(global_batch=40, micro_batch=40, auto_reset=False),
                                   dict(global_batch=40, micro_batch=20, critic_warmup_steps=3)])
def test_reference_learner_trains_the_registered_module(hooked, one_rank_group, shape):  # noqa: F811
    """The zero-patch route end to end: the reference's run_training -> train_micro_batch -> model(...) -> policy_loss ->
    backward -> clip_grad_norm_ -> AdamW (two learning-rate groups sorted by NAME) on the module its registry built."""
    ref, mr, saved = hooked
    T, B, epochs = 10, 16, 2
    auto_reset = shape.get("auto_reset", True)
    env = L.synthetic_env_tensors(0, T, B, 42, max_episode_steps=5)
    torch.manual_seed(11)
    theirs = ref.mlp_policy.MLPPolicy(42, 8, 1, True, False)
    ours = mr.models.get_model(_cfg())
    ours.load_state_dict(copy.deepcopy(theirs.state_dict()), strict=True)
    ora = O.OracleMLPPolicy(42, 8, 1)
    ora.load_state_dict(copy.deepcopy(theirs.state_dict()), strict=True)
    eps = torch.randn(T, B, 8, generator=torch.Generator().manual_seed(100))
    batch = L.advantages(L.rollout(ora, env, eps, 0.8, auto_reset), 0.8, 0.9, auto_reset)
    kw = dict(global_batch=shape["global_batch"], micro_batch=shape["micro_batch"], update_epoch=epochs,
              entropy_bonus=shape.get("entropy_bonus", 0.0), critic_warmup_steps=shape.get("critic_warmup_steps", 0),
              auto_reset=auto_reset)
    # expected: the reference's module and BUILT-IN losses on the CPU
    reg = ref.registry
    hooked_losses = dict(reg.LOSS_REGISTRY)
    reg.LOSS_REGISTRY.clear(), reg.LOSS_REGISTRY.update(saved[1])
    try:
        want = _learner(ref, theirs, copy.deepcopy(batch), **kw).run_training()
    finally:
        reg.LOSS_REGISTRY.clear(), reg.LOSS_REGISTRY.update(hooked_losses)
    me = _learner(ref, ours, copy.deepcopy(batch), device=DEV, **kw)
    lrs = sorted((len(g["params"]), g["lr"]) for g in me.optimizer.param_groups)
    if not kw["critic_warmup_steps"]:
        assert [n for n, _ in lrs] == [7, 9]  # value_head.* in the critic group, the rest in the actor group
    got = me.run_training()
    assert me.optimizer_steps == (T * B // shape["global_batch"]) * epochs
    for (n, p), (_, q) in zip(theirs.named_parameters(), ours.named_parameters()):
        torch.testing.assert_close(q.detach().cpu(), p.detach(), rtol=2e-4, atol=3e-6, msg=lambda m, n=n: f"{n}: {m}")
    for k, w in want.items():
        assert k in got, k
        assert got[k] == pytest.approx(w, rel=2e-3, abs=2e-5), k

"""The model leg of the RLINF_EXT_MODULE hook, host side (no launches): rlinf_amd.ext.register() must put the HIP MLP policy's
builder into the reference's OWN model registry (rlinf/models/__init__.py:31-53, executed from the reference tree), and the
module that builder returns must look to the reference's learner like the reference's MLPPolicy: the same named_parameters()
(names, order, shapes), the same initial values from the same RNG stream, the same state_dict keys."""

import pytest
import torch

from oracle import reference_loader as RL

pytestmark = pytest.mark.reference


class Cfg(dict):
    __getattr__ = dict.__getitem__


@pytest.fixture(scope="module")
def registry(ref):
    mr = RL.load_models_registry()
    reg = ref.registry  # ext.register() also re-registers the advantage / loss callees: everything is put back afterwards
    saved = dict(mr.models._MODEL_REGISTRY), dict(reg.ADV_REGISTRY), dict(reg.LOSS_REGISTRY)
    yield mr
    mr.models._MODEL_REGISTRY.clear(), mr.models._MODEL_REGISTRY.update(saved[0])
    reg.ADV_REGISTRY.clear(), reg.ADV_REGISTRY.update(saved[1])
    reg.LOSS_REGISTRY.clear(), reg.LOSS_REGISTRY.update(saved[2])


def _register():
    from rlinf_amd import ext

    try:
        ext.register()
    except RuntimeError as e:  # without a HIP device the hook raises AFTER registering (RLinf logs it: cluster/utils.py:100-107)
        assert "no HIP device" in str(e)


def test_register_puts_the_hip_builder_into_the_reference_registry(ref, registry):
    from rlinf_amd.models.embodiment.mlp_policy_module import ReferenceNamedMLPPolicy, build_reference_named_mlp_policy

    builtin = registry.models._MODEL_REGISTRY["mlp_policy"]
    _register()
    assert registry.models._MODEL_REGISTRY["mlp_policy"] is build_reference_named_mlp_policy is not builtin
    assert registry.config.SupportedModel("mlp_policy") in registry.config.EMBODIED_MODEL
    cfg = Cfg(model_type="mlp_policy", precision="32", obs_dim=42, action_dim=8, num_action_chunks=1, add_value_head=True,
              add_q_head=False, is_lora=False, load_to_device=False)
    model = registry.models.get_model(cfg)  # the reference's own get_model (rlinf/models/__init__.py:337-352)
    assert isinstance(model, ReferenceNamedMLPPolicy) and model.compute_dtype == torch.float32
    assert registry.models.get_model(Cfg(cfg, precision="bf16")).compute_dtype == torch.bfloat16


@pytest.mark.parametrize("shape", [dict(obs_dim=42, action_dim=8, chunks=1, value=True), dict(obs_dim=9, action_dim=7, chunks=2, value=True),
                                   dict(obs_dim=42, action_dim=8, chunks=1, value=False)])
def test_module_is_the_reference_module_to_a_learner(ref, shape):
    from rlinf_amd.models.embodiment.mlp_policy_module import ReferenceNamedMLPPolicy

    args = (shape["obs_dim"], shape["action_dim"], shape["chunks"], shape["value"], False)
    torch.manual_seed(5)
    theirs = ref.mlp_policy.MLPPolicy(*args)
    after_theirs = torch.rand(4)
    torch.manual_seed(5)
    ours = ReferenceNamedMLPPolicy(*args)
    after_ours = torch.rand(4)
    assert torch.equal(after_theirs, after_ours)  # same number of RNG draws, in the same order
    want, got = list(theirs.named_parameters()), list(ours.named_parameters())
    assert [n for n, _ in got] == [n for n, _ in want]
    for (n, p), (_, q) in zip(want, got):
        assert p.shape == q.shape and p.requires_grad == q.requires_grad and torch.equal(p.detach(), q.detach()), n
    assert list(ours.state_dict()) == list(theirs.state_dict())
    # the optimizer groups the reference builds from the names (fsdp_model_manager.py:533-560)
    critic = [n for n, _ in got if "value_head" in n]
    assert len(critic) == (7 if shape["value"] else 0)
    # a checkpoint of the reference loads, strictly, and invalidates the kernels' weight image
    torch.manual_seed(6)
    other = ref.mlp_policy.MLPPolicy(*args)
    ours._dirty = False
    ours.load_state_dict(other.state_dict(), strict=True)
    assert ours._dirty and all(torch.equal(a, b) for a, b in zip(ours.state_dict().values(), other.state_dict().values()))
    assert ours.forward.__func__ is not None
    with pytest.raises(NotImplementedError):
        ours(forward_type="sac", forward_inputs={})

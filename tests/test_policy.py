"""MLP policy: host-side parity on CPU (init stream, state-dict layout) and kernel parity on the GPU."""

import os

import pytest
import torch

from conftest import GOLDEN_DIR
from oracle import ppo_oracle as O
from oracle.make_golden import perturbation

FWD_RTOL, FWD_ATOL = 1e-4, 1e-5  # exact-f32 MFMA vs CPU sgemm: summation order only


def _golden():
    return torch.load(os.path.join(GOLDEN_DIR, "policy.pt"), weights_only=False)


def _make(sd=None, device="cpu"):
    from rlinf_amd.models.embodiment.mlp_policy import MLPPolicy
    pol = MLPPolicy(42, 8, 1, True, False)
    if sd is not None:
        pol.load_reference_state_dict(sd)
    return pol.to(device)


def test_init_stream_names_and_layout_match_reference():
    G = _golden()
    torch.manual_seed(4321)  # the seed make_golden.py built the reference policy with
    pol = _make()
    assert list(pol.shapes) == G["param_names"]
    assert pol.n_params == 287504
    sd = pol.reference_state_dict()
    for name, want in G["state_dict"].items():
        assert torch.equal(sd[name], want), name
    flat = torch.cat([G["state_dict"][n].reshape(-1) for n in G["param_names"]])
    assert torch.equal(pol.flat.detach(), flat)
    assert pol.group_ranges(3e-4, 1e-3) == [(0, 8, 3e-4), (8, 142856, 1e-3), (142856, 287504, 3e-4)]
    with pytest.raises(RuntimeError, match="mismatch"):
        pol.load_reference_state_dict({k: v for k, v in sd.items() if k != "actor_mean.bias"})


def test_value_free_policy_exposes_the_reference_parameter_set():
    """add_value_head False: the reference's names / shapes / init stream without the value head (pinned to the reference itself in
    test_oracle_vs_reference.py), a phantom all-zero value net behind them for the kernels, untouched by the optimizer's groups."""
    from rlinf_amd.models.embodiment.mlp_policy import MLPPolicy
    torch.manual_seed(5)
    ora = O.OracleMLPPolicy(42, 8, 1, add_value_head=False)
    torch.manual_seed(5)
    pol = MLPPolicy(42, 8, 1, False, False)
    assert list(pol.shapes) == [n for n, _ in ora.named_parameters()] and list(pol.state_dict()) == list(pol.shapes)
    for n, p in ora.named_parameters():
        assert torch.equal(pol.view(n), p.detach()), n
    assert pol.n_exposed == sum(p.numel() for p in ora.parameters()) == 144656 and pol.n_params > pol.n_exposed
    assert not pol.flat.detach()[pol.n_exposed:].any()                      # the phantom value net is all zeros
    lay = pol.layout
    assert all(lay.off_w[0][l] >= pol.n_exposed and lay.off_w[0][l] % 4 == 0 for l in range(4)) and lay.off_b[0][3] == -1
    assert pol.group_ranges(3e-4, 1e-3, train_value_head=False) == [(0, pol.n_exposed, 3e-4)]
    pol.load_state_dict(ora.state_dict())                                      # the reference's key set round-trips
    with pytest.raises(NotImplementedError):
        pol.default_forward({"states": torch.zeros(2, 42), "action": torch.zeros(2, 8)})
    with pytest.raises(NotImplementedError):
        MLPPolicy(42, 8, 1, False, True)                                       # the Q head (SAC) stays out of scope


@pytest.mark.parametrize("add_value_head", [True, False])
def test_odd_action_dim_layout_is_16_byte_aligned(add_value_head):
    """action_dim 7 (the reference's LIBERO MLP configurations): every tensor of the flat buffer starts on a 16-byte boundary -- the
    kernels' requirement for the weight matrices --, the up-to-three unowned floats behind actor_logstd / actor_mean.bias are zero,
    belong to no state_dict entry, and the AdamW ranges run across them (they have zero gradients, so they stay zero); the
    reference's parameter set, values and init stream are unchanged."""
    from rlinf_amd.models.embodiment.mlp_policy import MLPPolicy
    torch.manual_seed(5)
    ora = O.OracleMLPPolicy(42, 7, 1, add_value_head=add_value_head)
    torch.manual_seed(5)
    pol = MLPPolicy(42, 7, 1, add_value_head, False)
    assert list(pol.shapes) == [n for n, _ in ora.named_parameters()]
    owned = torch.zeros(pol.n_params, dtype=torch.bool)
    for n, p in ora.named_parameters():
        assert pol.offsets[n] % 4 == 0, n
        assert torch.equal(pol.view(n), p.detach()), n
        owned[pol.offsets[n]:pol.offsets[n] + p.numel()] = True
    assert pol.exposed_numel == sum(p.numel() for p in ora.parameters()) and pol.n_exposed > pol.exposed_numel
    assert torch.equal(pol.exposed_flat(), torch.cat([p.detach().reshape(-1) for p in ora.parameters()]))
    assert not pol.flat.detach()[~owned].any()
    lay = pol.layout
    assert all(lay.off_w[y][l] % 4 == 0 for y in range(2) for l in range(4)) and lay.off_logstd % 4 == 0
    groups = pol.group_ranges(3e-4, 1e-3, train_value_head=add_value_head)
    assert len(groups) <= 3 and groups[0][0] == 0                       # merged across the alignment gaps
    covered = torch.zeros(pol.n_params, dtype=torch.bool)
    for b, e, _ in groups:
        covered[b:e] = True
    assert bool(covered[owned].all())                                    # every owned element is in a range
    if add_value_head:                                                   # value-head ranges carry value_lr, nothing else does
        for b, e, lr in groups:
            names = [n for n in pol.shapes if b <= pol.offsets[n] < e]
            assert all(("value_head" in n) == (lr == 1e-3) for n in names), (b, e, lr)


def test_model_registry_boundary():
    from rlinf_amd import models
    m = models.get_model(dict(model_type="mlp_policy", obs_dim=42, action_dim=8, num_action_chunks=1,
                              add_value_head=True, precision="32", load_to_device=False))
    assert m.n_params == 287504
    assert models.get_model(dict(model_type="unknown")) is None
    with pytest.raises(ValueError, match="already registered"):
        models.register_model("mlp_policy", lambda c, d: None)
    with pytest.raises(TypeError):
        models.register_model("x", None)


@pytest.mark.gpu
def test_rollout_golden_injected_noise():
    G = _golden()
    pol = _make(G["state_dict"], "cuda")
    acts, res = pol.predict_action_batch({"states": G["states"]}, mode="train", eps=G["eps"])
    assert acts.shape == (96, 1, 8)
    torch.testing.assert_close(acts.reshape(96, 8).cpu(), G["action"], rtol=FWD_RTOL, atol=FWD_ATOL)
    torch.testing.assert_close(res["prev_logprobs"].cpu(), G["prev_logprobs"], rtol=FWD_RTOL, atol=FWD_ATOL)
    torch.testing.assert_close(res["prev_values"].cpu(), G["prev_values"], rtol=FWD_RTOL, atol=FWD_ATOL)
    assert torch.equal(res["forward_inputs"]["action"], acts.reshape(96, 8))
    acts, res = pol.predict_action_batch({"states": G["states"]}, mode="eval")
    torch.testing.assert_close(acts.reshape(96, 8).cpu(), G["eval_action"], rtol=FWD_RTOL, atol=FWD_ATOL)
    torch.testing.assert_close(res["prev_logprobs"].cpu(), G["eval_logprobs"], rtol=FWD_RTOL, atol=FWD_ATOL)


@pytest.mark.gpu
@pytest.mark.parametrize("M", [1024, 8192, 33, 1, 1000])
def test_rollout_vs_oracle_sizes(M):
    torch.manual_seed(M)
    ora = O.OracleMLPPolicy(42, 8, 1)
    with torch.no_grad():
        for p in ora.parameters():
            p.add_(torch.randn_like(p) * 0.05)  # non-trivial biases / logstd
    pol = _make(ora.state_dict(), "cuda")
    g = torch.Generator().manual_seed(1)
    states, eps = torch.randn(M, 42, generator=g), torch.randn(M, 8, generator=g)
    a0, lp0, v0 = ora.act(states, eps=eps)
    _, res = pol.predict_action_batch({"states": states}, eps=eps)
    torch.testing.assert_close(res["forward_inputs"]["action"].cpu(), a0, rtol=FWD_RTOL, atol=FWD_ATOL)
    torch.testing.assert_close(res["prev_logprobs"].cpu(), lp0, rtol=FWD_RTOL, atol=FWD_ATOL)
    torch.testing.assert_close(res["prev_values"].cpu(), v0, rtol=FWD_RTOL, atol=FWD_ATOL)


@pytest.mark.gpu
def test_training_forward_backward_and_step_golden():
    from rlinf_amd import ops
    G = _golden()
    pol = _make(G["state_dict"], "cuda")
    with torch.no_grad():
        for name, shp in pol.shapes.items():
            pol.view(name).add_(perturbation(torch.Size(shp)).cuda())
    pol.mark_updated()
    out = pol.default_forward({"states": G["states"].cuda(), "action": G["action"].cuda()})
    torch.testing.assert_close(out["logprobs"].detach().cpu(), G["train_logprobs"], rtol=FWD_RTOL, atol=FWD_ATOL)
    torch.testing.assert_close(out["entropy"].detach().cpu(), G["train_entropy"], rtol=1e-6, atol=1e-6)
    torch.testing.assert_close(out["values"].detach().cpu(), G["train_values"], rtol=FWD_RTOL, atol=FWD_ATOL)
    loss, _ = ops.ppo_loss(out["logprobs"], G["prev_logprobs"].cuda(), G["advantages"].cuda(), clip_ratio_low=0.2,
                           clip_ratio_high=0.2, values=out["values"], prev_values=G["prev_values"].cuda(),
                           returns=G["returns"].cuda(), value_clip=1.0, huber_delta=10.0, max_episode_steps=50)
    torch.testing.assert_close(loss.detach().cpu(), G["loss"], rtol=1e-4, atol=1e-5)
    loss.backward()
    want = torch.cat([G["grads"][n].reshape(-1) for n in G["param_names"]])
    got = pol.flat.grad.cpu()
    scale = float(want.abs().max())
    for name in G["param_names"]:  # per-tensor, relative to the tensor's own gradient scale
        o = pol.offsets[name]
        w = G["grads"][name].reshape(-1)
        gt = got[o:o + w.numel()]
        tol = 2e-4 * max(float(w.abs().max()), 1e-3 * scale)
        assert float((gt - w).abs().max()) <= tol, (name, float((gt - w).abs().max()), tol)
    # one full clip + AdamW step against the reference's parameters after its own step
    m, v = torch.zeros_like(pol.flat.data), torch.zeros_like(pol.flat.data)
    stats = ops.clip_adamw_step_(pol.flat.data, pol.flat.grad.clone(), m, v, pol.group_ranges(3e-4, 3e-4), 1,
                                 max_grad_norm=0.5)
    assert float(stats[0]) == pytest.approx(float(G["grad_norm"]), rel=1e-4)
    flat = pol.flat.detach().cpu()
    for name, want in G["params_after_step_stride16"].items():
        o = pol.offsets[name]
        got = flat[o:o + int(torch.tensor(pol.shapes[name]).prod())][::16]
        torch.testing.assert_close(got, want, rtol=1e-5, atol=1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("M", [8192, 700])
def test_training_gradients_vs_oracle_minibatch(M):
    """Full minibatch (8192 = the per-rank PPO minibatch of the bench config) incl. entropy gradient."""
    from rlinf_amd import ops
    torch.manual_seed(7)
    ora = O.OracleMLPPolicy(42, 8, 1)
    with torch.no_grad():
        for p in ora.parameters():
            p.add_(torch.randn_like(p) * 0.02)
    pol = _make(ora.state_dict(), "cuda")
    g = torch.Generator().manual_seed(2)
    states, action = torch.randn(M, 42, generator=g), torch.randn(M, 8, generator=g) * 0.6
    w_lp, w_ent, w_v = torch.randn(M, 8, generator=g), torch.randn(M, 8, generator=g), torch.randn(M, 1, generator=g)
    o0 = ora.evaluate(states, action)
    ((o0["logprobs"] * w_lp).sum() / M + (o0["entropy"] * w_ent).sum() / M + (o0["values"] * w_v).sum() / M).backward()
    o1 = pol.default_forward({"states": states.cuda(), "action": action.cuda()})
    ((o1["logprobs"] * w_lp.cuda()).sum() / M + (o1["entropy"] * w_ent.cuda()).sum() / M
     + (o1["values"] * w_v.cuda()).sum() / M).backward()
    got = pol.flat.grad.cpu()
    for name, p in ora.named_parameters():
        o = pol.offsets[name]
        w = p.grad.reshape(-1)
        tol = 3e-4 * float(w.abs().max()) + 1e-7
        assert float((got[o:o + w.numel()] - w).abs().max()) <= tol, name

"""The two fused hot launches (rlinf_amd/csrc/ppo_step.hip) through the C ABI against the CPU oracle:
rlx_mlp_rollout_step (policy + value-only jobs in one launch) and rlx_ppo_step (forward + loss + backward)."""

import math

import pytest
import torch

from oracle import ppo_oracle as O

pytestmark = pytest.mark.gpu

FWD = dict(rtol=1e-4, atol=1e-5)  # exact-f32 MFMA vs CPU sgemm: summation order only


def _policies(seed=7, jitter=0.02):
    from rlinf_amd.models.embodiment.mlp_policy import MLPPolicy
    torch.manual_seed(seed)
    ora = O.OracleMLPPolicy(42, 8, 1)
    with torch.no_grad():
        for p in ora.parameters():
            p.add_(torch.randn_like(p) * jitter)
    pol = MLPPolicy(42, 8, 1, True, False)
    pol.load_reference_state_dict(ora.state_dict())
    return ora, pol.to("cuda")


@pytest.mark.parametrize("M", [1024, 16, 37, 1])
def test_rollout_step_policy_and_value_jobs(M):
    from rlinf_amd import ops
    ora, pol = _policies()
    g = torch.Generator().manual_seed(3)
    states, eps = torch.randn(M, 42, generator=g), torch.randn(M, 8, generator=g)
    fin, last = torch.randn(M, 42, generator=g), torch.randn(M, 42, generator=g)
    rewards = torch.rand(M, 1, generator=g)
    flags = torch.rand(M, 1, generator=g) < 0.3
    want_a, want_lp, want_v = ora.act(states, eps=eps, mode="train")
    want_boot = ora.value_head.mlp(fin).detach()[:, :1]
    want_r = O.bootstrap_rewards(rewards.clone(), flags, want_boot, 0.8)
    want_last = ora.value_head.mlp(last).detach()

    r_dev = rewards.cuda()
    copy = torch.zeros(M, 42, device="cuda")
    last_v = torch.empty(M, 1, device="cuda")
    a, lp, v = ops.mlp_rollout_step(
        pol.flat.data, pol.tiles(), pol.layout, states.cuda(), eps.cuda(), states_copy=copy,
        value_jobs=(dict(states=fin.cuda(), rewards=r_dev, flags=flags.cuda(), gamma=0.8),
                    dict(states=last.cuda(), values=last_v)))
    torch.testing.assert_close(a.cpu(), want_a, **FWD)
    torch.testing.assert_close(lp.cpu(), want_lp, rtol=2e-4, atol=2e-5)
    torch.testing.assert_close(v.cpu(), want_v, **FWD)
    assert torch.equal(copy.cpu(), states)
    torch.testing.assert_close(r_dev.cpu(), want_r, **FWD)
    assert torch.equal(r_dev.cpu()[~flags], rewards[~flags])  # untouched where the env did not finish
    torch.testing.assert_close(last_v.cpu(), want_last, **FWD)
    # the same bootstrap job with the env-row store fused in (what EnvWorker queues): rows written, reward = env + fold
    r2 = torch.full((M, 1), float("nan"), device="cuda")
    rows = [torch.zeros(M, 1, dtype=torch.bool, device="cuda") for _ in range(3)]
    term = torch.zeros(M, 1, dtype=torch.bool)
    ops.mlp_rollout_step(pol.flat.data, pol.tiles(), pol.layout, None, None,
                         value_jobs=(dict(states=fin.cuda(), rewards=r2, flags=None, gamma=0.8,
                                          env=(rewards.cuda(), term.cuda(), flags.cuda()), rows=tuple(rows)),))
    assert torch.equal(r2, r_dev)
    assert torch.equal(rows[0].cpu(), flags) and torch.equal(rows[1].cpu(), term) and torch.equal(rows[2].cpu(), flags)
    # eval mode: action == mean, value-only launch without a policy job
    a2, _, _ = ops.mlp_rollout_step(pol.flat.data, pol.tiles(), pol.layout, states.cuda(), None)
    torch.testing.assert_close(a2.cpu(), ora.act(states, eps=None, mode="eval")[0], **FWD)
    only_v = torch.empty(M, 1, device="cuda")
    ops.mlp_rollout_step(pol.flat.data, pol.tiles(), pol.layout, None, None, value_jobs=(dict(states=last.cuda(), values=only_v),))
    assert torch.equal(only_v, last_v)  # same tile, same arithmetic, whatever else rides in the grid


def _minibatch(M, g, with_mask):
    mb = dict(states=torch.randn(M, 42, generator=g), action=torch.randn(M, 8, generator=g) * 0.6,
              prev_logprobs=torch.randn(M, 8, generator=g) * 0.1 - 1.0, advantages=torch.randn(M, 1, generator=g),
              prev_values=torch.randn(M, 1, generator=g), returns=torch.randn(M, 1, generator=g))
    if with_mask:
        mb["loss_mask"] = torch.rand(M, 1, generator=g) < 0.7
    return mb


@pytest.mark.parametrize("M,with_mask", [(8192, False), (700, True), (32, False), (5, True)])
def test_ppo_step_gradients_and_metrics_vs_oracle(M, with_mask):
    from rlinf_amd import ops
    from rlinf_amd._lib import PPO_OUT_FLOATS, PPO_OUT_NAMES
    ora, pol = _policies(seed=11)
    g = torch.Generator().manual_seed(5)
    mb = _minibatch(M, g, with_mask)
    # make the old log-probs close to the current ones so that ratios straddle the clip range
    with torch.no_grad():
        cur = ora.evaluate(mb["states"], mb["action"])["logprobs"]
    mb["prev_logprobs"] = cur + torch.randn(M, 8, generator=g) * 0.08

    out = ora.evaluate(mb["states"], mb["action"])
    shaped = O.shape_loss_inputs(out["logprobs"], mb["prev_logprobs"], mb["advantages"], "action_level", 8,
                                 loss_mask=mb.get("loss_mask"), values=out["values"], prev_values=mb["prev_values"],
                                 returns=mb["returns"])
    loss, metrics = O.ppo_actor_critic_loss(clip_ratio_low=0.2, clip_ratio_high=0.2, value_clip=1.0, huber_delta=10.0, **shaped)
    (loss * 0.5).backward()  # grad_out = 1 / gradient_accumulation = 0.5

    lay = pol.layout
    lp = ops.make_ppo_params(logprob_type="action_level", action_dim=8, chunks=1, clip_ratio_low=0.2, clip_ratio_high=0.2,
                             value_clip=1.0, huber_delta=10.0, max_episode_steps=50, has_critic=True)
    slabs = ops.ppo_step_slabs(lay, M)
    grads = torch.full((slabs, lay.n_params), float("nan"), device="cuda")  # every element must be written
    ws = torch.empty(ops.ppo_step_workspace_bytes(lay, M), dtype=torch.uint8, device="cuda")
    row = torch.zeros(PPO_OUT_FLOATS, device="cuda")
    dev_mb = {k: v.cuda().contiguous() for k, v in mb.items()}
    if with_mask:
        dev_mb["loss_mask"] = dev_mb["loss_mask"].view(torch.uint8)
    ops.ppo_step(pol.flat.data, lay, lp, dev_mb, grads, row, ws, grad_out=0.5)
    got = grads.sum(dim=0).cpu()
    assert torch.isfinite(got).all()
    scale = max(float(p.grad.abs().max()) for p in ora.parameters())
    for name, p in ora.named_parameters():
        o = pol.offsets[name]
        w = p.grad.reshape(-1)
        tol = 3e-4 * max(float(w.abs().max()), 1e-3 * scale) + 1e-7
        err = float((got[o:o + w.numel()] - w).abs().max())
        assert err <= tol, (name, err, tol)
    host = row.cpu()
    assert float(host[PPO_OUT_NAMES["loss"]]) == pytest.approx(float(loss.detach()), rel=2e-4, abs=2e-5)
    for key in ("actor/policy_loss", "actor/ratio", "actor/clipped_ratio", "actor/approx_kl", "actor/clip_fraction",
                "critic/value_loss"):
        assert float(host[PPO_OUT_NAMES[key]]) == pytest.approx(float(metrics[key]), rel=5e-4, abs=5e-5), key


@pytest.mark.parametrize("M,case,bf16", [
    (2048, ("action_level", "action_level", True, False), False),   # [M, C] ratios under [M, C] advantages, C values per row
    (700, ("action_level", "action_level", True, True), False),     # ... under a [M, C] loss mask
    (333, ("action_level", "token_level", True, True), False),      # [M, C, A] ratios, advantages / mask unsqueezed
    (512, ("chunk_level", "chunk_level", False, True), False),      # value-free (GRPO): one ratio per row
    (512, ("chunk_level", "action_level", False, True), False),     # [M, C] ratios under [M, 1] advantages
    (2048, ("action_level", "action_level", True, False), True),    # bf16 operands
])
def test_ppo_step_two_action_chunks(M, case, bf16):
    """num_action_chunks = 2 (act_dim = 16 head outputs, 2 value columns) through the FUSED step: every loss shaping of SURVEY.md
    8c that exists for C > 1, gradients of every parameter and the metric row against the oracle."""
    from rlinf_amd import ops
    from rlinf_amd._lib import PPO_OUT_FLOATS, PPO_OUT_NAMES
    from rlinf_amd.models.embodiment.mlp_policy import MLPPolicy
    reward_type, logprob_type, critic, with_mask = case
    C, A = 2, 8
    torch.manual_seed(21)
    ora = O.OracleMLPPolicy(42, A, C)
    with torch.no_grad():
        for p in ora.parameters():
            p.add_(torch.randn_like(p) * 0.02)
    pol = MLPPolicy(42, A, C, True, False, compute_dtype=torch.bfloat16 if bf16 else torch.float32)
    pol.load_reference_state_dict(ora.state_dict())
    pol = pol.to("cuda")
    g = torch.Generator().manual_seed(8)
    nadv = 1 if reward_type == "chunk_level" else C
    mb = dict(states=torch.randn(M, 42, generator=g), action=torch.randn(M, C * A, generator=g) * 0.6,
              advantages=torch.randn(M, nadv, generator=g), prev_values=torch.randn(M, C, generator=g),
              returns=torch.randn(M, C, generator=g))
    if with_mask:
        mb["loss_mask"] = torch.rand(M, nadv, generator=g) < 0.7
    with torch.no_grad():
        cur = ora.evaluate(mb["states"], mb["action"])["logprobs"]
    mb["prev_logprobs"] = cur + torch.randn(M, C * A, generator=g) * 0.05
    out = ora.evaluate(mb["states"], mb["action"])
    shaped = O.shape_loss_inputs(out["logprobs"], mb["prev_logprobs"], mb["advantages"], logprob_type, A,
                                 loss_mask=mb.get("loss_mask"), values=out["values"] if critic else None,
                                 prev_values=mb["prev_values"] if critic else None, returns=mb["returns"] if critic else None,
                                 reward_type=reward_type)
    if critic:
        loss, metrics = O.ppo_actor_critic_loss(clip_ratio_low=0.2, clip_ratio_high=0.2, value_clip=1.0, huber_delta=10.0, **shaped)
    else:
        loss, metrics = O.ppo_actor_loss(shaped["logprobs"], shaped["old_logprobs"], shaped["advantages"], 0.2, 0.2,
                                         loss_mask=shaped["loss_mask"])
    loss.backward()
    lay = pol.layout
    lp = ops.make_ppo_params(logprob_type=logprob_type, action_dim=A, chunks=C, clip_ratio_low=0.2, clip_ratio_high=0.2,
                             value_clip=1.0, huber_delta=10.0, max_episode_steps=50, has_critic=critic, reward_type=reward_type)
    grads = torch.full((ops.ppo_step_slabs(lay, M, bf16=bf16), lay.n_params), float("nan"), device="cuda")
    ws = torch.empty(ops.ppo_step_workspace_bytes(lay, M), dtype=torch.uint8, device="cuda")
    row = torch.zeros(PPO_OUT_FLOATS, device="cuda")
    dev_mb = {k: v.cuda().contiguous() for k, v in mb.items()}
    if not critic:
        dev_mb.pop("returns")
    if with_mask:
        dev_mb["loss_mask"] = dev_mb["loss_mask"].view(torch.uint8)
    ops.ppo_step(pol.flat.data, lay, lp, dev_mb, grads, row, ws, grad_out=1.0, bf16=bf16)
    got = grads.sum(dim=0).cpu()
    assert torch.isfinite(got).all()
    want = torch.zeros_like(got)
    for name, p in ora.named_parameters():
        if p.grad is not None:
            want[pol.offsets[name]:pol.offsets[name] + p.numel()] = p.grad.reshape(-1)
    if bf16:  # DERIVED bound, as for C = 1: no farther from the f32 gradient than twice the reference arithmetic under bf16 autocast
        for p in ora.parameters():
            p.grad = None
        with torch.autocast("cpu", dtype=torch.bfloat16):
            out16 = ora.evaluate(mb["states"], mb["action"])
        out16 = {k: v.float() for k, v in out16.items()}
        sh16 = O.shape_loss_inputs(out16["logprobs"], mb["prev_logprobs"], mb["advantages"], logprob_type, A, loss_mask=mb.get("loss_mask"),
                                   values=out16["values"], prev_values=mb["prev_values"], returns=mb["returns"], reward_type=reward_type)
        O.ppo_actor_critic_loss(clip_ratio_low=0.2, clip_ratio_high=0.2, value_clip=1.0, huber_delta=10.0, **sh16)[0].backward()
        auto = torch.zeros_like(got)
        for name, p in ora.named_parameters():
            auto[pol.offsets[name]:pol.offsets[name] + p.numel()] = p.grad.reshape(-1)
        rel, rel_auto = float((got - want).norm() / want.norm()), float((auto - want).norm() / want.norm())
        assert rel <= 2.0 * rel_auto + 1e-3, (rel, rel_auto)
    else:
        scale = float(want.abs().max())
        for name, p in ora.named_parameters():
            o = pol.offsets[name]
            w = want[o:o + p.numel()]
            tol = 3e-4 * max(float(w.abs().max()), 1e-3 * scale) + 1e-7
            err = float((got[o:o + p.numel()] - w).abs().max())
            assert err <= tol, (name, err, tol)
    host = row.cpu()
    tol = dict(rel=2e-2, abs=2e-3) if bf16 else dict(rel=5e-4, abs=5e-5)
    assert float(host[PPO_OUT_NAMES["loss"]]) == pytest.approx(float(loss.detach()), **tol)
    for key in ("actor/policy_loss", "actor/ratio", "actor/clipped_ratio", "actor/approx_kl", "actor/clip_fraction") + (
            ("critic/value_loss",) if critic else ()):
        assert float(host[PPO_OUT_NAMES[key]]) == pytest.approx(float(metrics[key]), **tol), key


def test_f32_weight_gradients_split_vs_exact_mfma():
    """The f32 weight-gradient launch runs its products as 3 x bf16 splits on the bf16 matrix pipe (six of the nine partial
    products: what is dropped lies below 2^-24 of |a||b|); RLX_F32_EXACT_MFMA=1 selects the exact f32 MFMA.  Same minibatch through
    both, in two processes (the switch is read once per process): each within the f32 tolerance of the oracle's autograd, and
    within 2e-6 relative L2 of each other -- the split is f32-accurate, not bf16-accurate."""
    import os
    import subprocess
    import sys
    import tempfile

    from conftest import ROOT
    code = """
import sys, torch
sys.path.insert(0, %r)
from oracle import ppo_oracle as O
from rlinf_amd import ops
from rlinf_amd._lib import PPO_OUT_FLOATS
from rlinf_amd.models.embodiment.mlp_policy import MLPPolicy
torch.manual_seed(11)
ora = O.OracleMLPPolicy(42, 8, 1)
pol = MLPPolicy(42, 8, 1, True, False)
pol.load_reference_state_dict(ora.state_dict())
pol = pol.to("cuda")
g = torch.Generator().manual_seed(5)
M = 8192
mb = dict(states=torch.randn(M, 42, generator=g), action=torch.randn(M, 8, generator=g) * 0.6, advantages=torch.randn(M, 1, generator=g),
          prev_values=torch.randn(M, 1, generator=g), returns=torch.randn(M, 1, generator=g))
with torch.no_grad():
    mb["prev_logprobs"] = ora.evaluate(mb["states"], mb["action"])["logprobs"] + torch.randn(M, 8, generator=g) * 0.08
out = ora.evaluate(mb["states"], mb["action"])
shaped = O.shape_loss_inputs(out["logprobs"], mb["prev_logprobs"], mb["advantages"], "action_level", 8, values=out["values"],
                             prev_values=mb["prev_values"], returns=mb["returns"])
loss, _ = O.ppo_actor_critic_loss(clip_ratio_low=0.2, clip_ratio_high=0.2, value_clip=1.0, huber_delta=10.0, **shaped)
loss.backward()
want = torch.cat([p.grad.reshape(-1) for p in ora.parameters()])
lay = pol.layout
lp = ops.make_ppo_params(logprob_type="action_level", action_dim=8, chunks=1, clip_ratio_low=0.2, clip_ratio_high=0.2, value_clip=1.0,
                         huber_delta=10.0, max_episode_steps=50, has_critic=True)
grads = torch.full((ops.ppo_step_slabs(lay, M), lay.n_params), float("nan"), device="cuda")
ws = torch.empty(ops.ppo_step_workspace_bytes(lay, M), dtype=torch.uint8, device="cuda")
row = torch.zeros(PPO_OUT_FLOATS, device="cuda")
ops.ppo_step(pol.flat.data, lay, lp, {k: v.cuda().contiguous() for k, v in mb.items()}, grads, row, ws, grad_out=1.0)
got = grads.sum(dim=0).cpu()
torch.save(dict(got=got, want=want), sys.argv[1])
""" % ROOT
    res = {}
    with tempfile.TemporaryDirectory() as d:
        for name, flag in (("split", "0"), ("exact", "1")):
            out = os.path.join(d, name + ".pt")
            r = subprocess.run([sys.executable, "-c", code, out], env=dict(os.environ, RLX_F32_EXACT_MFMA=flag), capture_output=True,
                               text=True, timeout=300)
            assert r.returncode == 0, r.stderr[-2000:]
            res[name] = torch.load(out)
    want = res["split"]["want"]
    for name in ("split", "exact"):
        rel = float((res[name]["got"] - want).norm() / want.norm())
        assert rel <= 2e-5, (name, rel)
    between = float((res["split"]["got"] - res["exact"]["got"]).norm() / want.norm())
    assert 0 < between <= 2e-6, between   # different arithmetic (not the same kernel twice), f32-level agreement


def test_ppo_step_matches_unfused_chain():
    """Same micro-batch through the stage-by-stage entry points (train_fwd -> ppo_loss -> train_bwd)."""
    from rlinf_amd import ops
    from rlinf_amd._lib import PPO_OUT_FLOATS
    _, pol = _policies(seed=3)
    M = 1000
    g = torch.Generator().manual_seed(9)
    mb = {k: v.cuda() for k, v in _minibatch(M, g, True).items()}
    lay = pol.layout
    o = pol.default_forward({"states": mb["states"], "action": mb["action"]})
    loss, out_ref = ops.ppo_loss(o["logprobs"], mb["prev_logprobs"], mb["advantages"], clip_ratio_low=0.2,
                                 clip_ratio_high=0.2, values=o["values"], prev_values=mb["prev_values"],
                                 returns=mb["returns"], value_clip=1.0, huber_delta=10.0, loss_mask=mb["loss_mask"],
                                 max_episode_steps=50)
    loss.backward()
    want = pol.flat.grad.clone()
    lp = ops.make_ppo_params(logprob_type="action_level", action_dim=8, chunks=1, clip_ratio_low=0.2, clip_ratio_high=0.2,
                             value_clip=1.0, huber_delta=10.0, max_episode_steps=50, has_critic=True)
    grads = torch.empty((ops.ppo_step_slabs(lay, M), lay.n_params), device="cuda")
    ws = torch.empty(ops.ppo_step_workspace_bytes(lay, M), dtype=torch.uint8, device="cuda")
    row = torch.zeros(PPO_OUT_FLOATS, device="cuda")
    dev_mb = dict(mb)
    dev_mb["loss_mask"] = mb["loss_mask"].view(torch.uint8)
    ops.ppo_step(pol.flat.data, lay, lp, dev_mb, grads, row, ws, grad_out=1.0)
    got = grads.sum(dim=0)
    torch.testing.assert_close(got, want, rtol=1e-3, atol=float(want.abs().max()) * 2e-4)
    torch.testing.assert_close(row[:16], out_ref[:16], rtol=2e-4, atol=2e-5)


@pytest.mark.parametrize("bf16", [False, True])
def test_optimizer_keeps_tile_image_in_step(bf16):
    """rlx_clip_adamw_step with tile_layout/tiles must leave exactly the image rlx_mlp_pack_tiles[_bf16] would build."""
    from rlinf_amd import ops
    _, pol = _bf16_policy(seed=5) if bf16 else _policies(seed=5)
    lay = pol.layout
    n = lay.n_params
    tiles = pol.tiles().clone()
    g = torch.randn(3, n, device="cuda") * 1e-2
    m, v = torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
    before = pol.flat.data.clone()
    ops.clip_adamw_step_(pol.flat.data, g, m, v, pol.group_ranges(3e-3, 1e-3), 1, max_grad_norm=0.5, tile_layout=lay, tiles=tiles)
    assert not torch.equal(before, pol.flat.data)
    fresh = ops.mlp_pack_tiles(pol.flat.data, lay, bf16=bf16)
    bits = lambda t: t.view(torch.int16 if bf16 else torch.int32)  # noqa: E731 -- the f32 image is opaque bytes (bf16 planes): compare bits
    assert tiles.dtype == fresh.dtype and torch.equal(bits(tiles), bits(fresh))
    # a skipped step (non-finite norm) leaves parameters and tiles alone
    g[0, 7] = float("inf")
    snap = pol.flat.data.clone()
    ops.clip_adamw_step_(pol.flat.data, g, m, v, pol.group_ranges(3e-3, 1e-3), 2, max_grad_norm=0.5, tile_layout=lay, tiles=tiles)
    assert torch.equal(snap, pol.flat.data) and torch.equal(bits(tiles), bits(fresh))


# ---- bf16 MFMA operands ("PPO bf16", BASELINE.json configs[1]): parity against the oracle under bf16 autocast --------------
BF16 = dict(rtol=2e-2, atol=2e-2)  # "bf16 policy forward compared against the oracle run in bf16 with rtol 2e-2"


def _bf16_policy(seed=7):
    from rlinf_amd.models.embodiment.mlp_policy import MLPPolicy
    torch.manual_seed(seed)
    ora = O.OracleMLPPolicy(42, 8, 1)
    with torch.no_grad():
        for p in ora.parameters():
            p.add_(torch.randn_like(p) * 0.02)
    pol = MLPPolicy(42, 8, 1, True, False, compute_dtype=torch.bfloat16)
    pol.load_reference_state_dict(ora.state_dict())
    return ora, pol.to("cuda")


@pytest.mark.parametrize("M", [1024, 37])
def test_rollout_step_bf16_vs_autocast_oracle(M):
    from rlinf_amd import ops
    ora, pol = _bf16_policy()
    assert pol.tiles().dtype == torch.bfloat16
    g = torch.Generator().manual_seed(3)
    states, eps = torch.randn(M, 42, generator=g), torch.randn(M, 8, generator=g)
    fin = torch.randn(M, 42, generator=g)
    rewards, flags = torch.rand(M, 1, generator=g), torch.rand(M, 1, generator=g) < 0.3
    with torch.autocast("cpu", dtype=torch.bfloat16):
        want_a, want_lp, want_v = ora.act(states, eps=eps, mode="train")
        want_boot = ora.value_head.mlp(fin).detach().float()[:, :1]
    f32_a, f32_lp, f32_v = ora.act(states, eps=eps, mode="train")
    want_r = O.bootstrap_rewards(rewards.clone(), flags, want_boot, 0.8)
    r_dev = rewards.cuda()
    a, lp, v = ops.mlp_rollout_step(pol.flat.data, pol.tiles(), pol.layout, states.cuda(), eps.cuda(),
                                    value_jobs=(dict(states=fin.cuda(), rewards=r_dev, flags=flags.cuda(), gamma=0.8),))
    torch.testing.assert_close(a.cpu(), want_a.float(), **BF16)
    torch.testing.assert_close(v.cpu(), want_v.float(), **BF16)
    torch.testing.assert_close(r_dev.cpu(), want_r, **BF16)
    # the f32 heads / log-prob epilogue keep us at least as close to the f32 reference as its own autocast run is
    assert float((a.cpu() - f32_a).abs().max()) <= 1.5 * float((want_a.float() - f32_a).abs().max()) + 1e-3
    # log-prob of the sampled action: (a - mean) / std is eps itself, whatever the mean's rounding
    torch.testing.assert_close(lp.cpu(), f32_lp, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("launch", ["rows", "cols32", "cols64"])
@pytest.mark.parametrize("M,with_mask", [(8192, False), (700, True), (704, False), (5, False), (1000, True)])
def test_ppo_step_bf16_gradients_vs_autocast_oracle(M, with_mask, launch, monkeypatch):
    """Every form of the fused bf16 launch: the row-split one (the default where its shapes allow: a wave per 16 rows, weights through
    an LDS ring) and both tilings of the column-split one (RLX_FUSED_ROWS=0; 32 rows per workgroup, or 64: RLX_FUSED_RT=4).

    bf16 operands perturb the log-probs by ~1e-2, which moves samples across PPO's clip boundary: the actor gradient of
    ANY bf16 implementation differs from the f32 one by a few percent (norm-wise), discontinuously.  The yardstick is
    therefore the reference arithmetic itself under bf16 autocast: we must be no further from the f32 gradient than
    twice its distance; the critic (smooth Huber loss) is held to a tight bound, tensor by tensor."""
    from rlinf_amd import ops
    from rlinf_amd._lib import PPO_OUT_FLOATS, PPO_OUT_NAMES
    if launch != "cols32":
        from conftest import need_dev_variants
        need_dev_variants(f"fused bf16 launch form {launch!r}")
    monkeypatch.setenv("RLX_FUSED_ROWS", "1" if launch == "rows" else "0")  # read at every plan / launch (development builds)
    monkeypatch.setenv("RLX_FUSED_RT", "4" if launch == "cols64" else "2")
    ora, pol = _bf16_policy(seed=11)
    g = torch.Generator().manual_seed(5)
    mb = _minibatch(M, g, with_mask)
    with torch.no_grad():
        cur = ora.evaluate(mb["states"], mb["action"])["logprobs"]
    mb["prev_logprobs"] = cur + torch.randn(M, 8, generator=g) * 0.08

    def oracle_grads(autocast):
        ora.zero_grad()
        with torch.autocast("cpu", dtype=torch.bfloat16, enabled=autocast):
            out = ora.evaluate(mb["states"], mb["action"])
        out = {k: v.float() for k, v in out.items()}
        shaped = O.shape_loss_inputs(out["logprobs"], mb["prev_logprobs"], mb["advantages"], "action_level", 8,
                                     loss_mask=mb.get("loss_mask"), values=out["values"], prev_values=mb["prev_values"],
                                     returns=mb["returns"])
        loss, metrics = O.ppo_actor_critic_loss(clip_ratio_low=0.2, clip_ratio_high=0.2, value_clip=1.0, huber_delta=10.0, **shaped)
        loss.backward()
        return {n: p.grad.clone() for n, p in ora.named_parameters()}, float(loss), {k: float(v) for k, v in metrics.items()}

    g32, loss32, metrics32 = oracle_grads(False)
    g16, _, _ = oracle_grads(True)
    lay = pol.layout
    lp = ops.make_ppo_params(logprob_type="action_level", action_dim=8, chunks=1, clip_ratio_low=0.2, clip_ratio_high=0.2,
                             value_clip=1.0, huber_delta=10.0, max_episode_steps=50, has_critic=True)
    grads = torch.full((ops.ppo_step_slabs(lay, M, bf16=True), lay.n_params), float("nan"), device="cuda")  # every element must be written
    ws = torch.empty(ops.ppo_step_workspace_bytes(lay, M), dtype=torch.uint8, device="cuda")
    row = torch.zeros(PPO_OUT_FLOATS, device="cuda")
    dev_mb = {k: v.cuda().contiguous() for k, v in mb.items()}
    if with_mask:
        dev_mb["loss_mask"] = dev_mb["loss_mask"].view(torch.uint8)
    ops.ppo_step(pol.flat.data, lay, lp, dev_mb, grads, row, ws, grad_out=1.0, bf16=True)
    got = grads.sum(dim=0).cpu()
    assert torch.isfinite(got).all()
    cat = lambda d: torch.cat([d[n].reshape(-1) for n in d])  # noqa: E731
    w32, w16 = cat(g32), cat(g16)
    rel_ours = float((got - w32).norm() / w32.norm())
    rel_auto = float((w16 - w32).norm() / w32.norm())
    cos = float(torch.dot(got, w32) / (got.norm() * w32.norm()))
    assert rel_ours <= max(2.0 * rel_auto, 0.05) and cos > 0.99, (rel_ours, rel_auto, cos)
    for name, w in g32.items():
        o = pol.offsets[name]
        gt = got[o:o + w.numel()]
        err = float((gt - w.reshape(-1)).norm() / (w.norm() + 1e-12))
        ref = float((g16[name] - w).norm() / (w.norm() + 1e-12))
        if w.numel() < 64:  # actor_logstd / actor_mean.bias: a handful of +/- sums, bounded against the whole gradient
            assert float((gt - w.reshape(-1)).norm()) <= 0.05 * float(w32.norm()), name
            continue
        bound = 0.05 if name.startswith("value_head") else max(2.5 * ref, 0.08)  # catches a mis-indexed gradient tile
        assert err <= bound, (name, err, ref)
    host = row.cpu()
    assert float(host[PPO_OUT_NAMES["loss"]]) == pytest.approx(loss32, rel=2e-2, abs=2e-3)
    for key in ("actor/ratio", "actor/clipped_ratio", "critic/value_loss"):
        assert float(host[PPO_OUT_NAMES[key]]) == pytest.approx(metrics32[key], rel=2e-2, abs=2e-3), key


@pytest.mark.parametrize("M,with_mask", [(8192, False), (700, True), (1000, True), (5, False), (64, False)])
def test_ppo_step_bf16_pinned_to_the_operand_rounded_restatement(M, with_mask):
    """The autocast yardstick above BOUNDS the bf16 launches; this test PINS them.  oracle/bf16_operand_model.py writes out where the
    launches round (hidden-layer operands and activations, the backward sweep's dZ) and evaluates exactly that in float64 -- with
    its roundings off it is the oracle's arithmetic, forward and backward (tests/test_bf16_operand_model.py).

    What can still differ: f32 summation order and the hardware exp2 / rcp inside tanh move a pre-rounding value by ~1e-7, which
    now and then lands an activation on the other side of a bf16 rounding boundary; that one-ulp flip (2^-9) is re-rounded by
    the layers above it, so a few per cent of the ROWS carry output differences up to ~1e-3 -- and none at all on a small
    batch: there the gradient agrees to 1e-8.  The minibatch keeps every row 0.04 away from the PPO ratio clip and 0.02 from the value
    clip (in the model's own outputs), so both sides take the same branches and the comparison is of arithmetic, not of which
    side of a discontinuity a sample fell on: gradient within 1e-3 norm-wise (5e-2 for the autocast yardstick), loss to 1e-4,
    the clip fraction equal."""
    from oracle import bf16_operand_model as BM
    from rlinf_amd import ops
    from rlinf_amd._lib import PPO_OUT_FLOATS, PPO_OUT_NAMES
    ora, pol = _bf16_policy(seed=11)
    g = torch.Generator().manual_seed(5)
    mb = _minibatch(M, g, with_mask)
    with torch.no_grad():
        cur = BM.evaluate(ora, mb["states"], mb["action"])
    delta = torch.randn(M, 8, generator=g) * 0.08
    for edge in (math.log(0.8), math.log(1.2)):  # action_level: a ROW's log-ratio is -sum(delta): 0.04 away from both clip edges
        near = (-delta.sum(1) - edge).abs() < 0.04
        delta[:, 0] = torch.where(near, delta[:, 0] + 0.1, delta[:, 0])
    mb["prev_logprobs"] = cur["logprobs"] + delta
    gap = (cur["values"] - mb["prev_values"]).abs()
    mb["prev_values"] = torch.where((gap - 1.0).abs() < 0.02, mb["prev_values"] + 0.05, mb["prev_values"])  # value_clip = 1.0
    ora.zero_grad()
    out = BM.evaluate(ora, mb["states"], mb["action"])
    shaped = O.shape_loss_inputs(out["logprobs"], mb["prev_logprobs"], mb["advantages"], "action_level", 8,
                                 loss_mask=mb.get("loss_mask"), values=out["values"], prev_values=mb["prev_values"], returns=mb["returns"])
    loss, metrics = O.ppo_actor_critic_loss(clip_ratio_low=0.2, clip_ratio_high=0.2, value_clip=1.0, huber_delta=10.0, **shaped)
    loss.backward()
    want = {n: p.grad.clone() for n, p in ora.named_parameters()}
    lay = pol.layout
    lp = ops.make_ppo_params(logprob_type="action_level", action_dim=8, chunks=1, clip_ratio_low=0.2, clip_ratio_high=0.2,
                             value_clip=1.0, huber_delta=10.0, max_episode_steps=50, has_critic=True)
    grads = torch.full((ops.ppo_step_slabs(lay, M, bf16=True), lay.n_params), float("nan"), device="cuda")
    ws = torch.empty(ops.ppo_step_workspace_bytes(lay, M), dtype=torch.uint8, device="cuda")
    row = torch.zeros(PPO_OUT_FLOATS, device="cuda")
    dev_mb = {k: v.cuda().contiguous() for k, v in mb.items()}
    if with_mask:
        dev_mb["loss_mask"] = dev_mb["loss_mask"].view(torch.uint8)
    ops.ppo_step(pol.flat.data, lay, lp, dev_mb, grads, row, ws, grad_out=1.0, bf16=True)
    got = grads.sum(dim=0).cpu()
    cat = torch.cat([want[n].reshape(-1) for n in want])
    flat = torch.cat([got[pol.offsets[n]:pol.offsets[n] + want[n].numel()] for n in want])
    rel = float((flat - cat).norm() / cat.norm())
    worst = max((float((got[pol.offsets[n]:pol.offsets[n] + w.numel()] - w.reshape(-1)).norm() / (w.norm() + 1e-12)), n)
                for n, w in want.items() if w.numel() >= 64)
    host = row.cpu()
    print(f"[bf16 pinned M={M} mask={with_mask}] gradient rel-L2 {rel:.2e}, worst tensor {worst[1]} {worst[0]:.2e}, "
          f"loss {float(host[PPO_OUT_NAMES['loss']]):.7f} vs {float(loss):.7f}")
    small = M <= 8  # no flipped rounding in so few rows (seeded): the arithmetic itself, to f32 summation order
    assert rel <= (2e-6 if small else 1e-3), rel
    assert worst[0] <= (1e-5 if small else 3e-3), worst
    assert float(host[PPO_OUT_NAMES["loss"]]) == pytest.approx(float(loss), rel=1e-4, abs=1e-6)
    assert float(host[PPO_OUT_NAMES["actor/clip_fraction"]]) == float(metrics["actor/clip_fraction"])  # the same branches, row by row
    for key in ("actor/ratio", "actor/clipped_ratio", "critic/value_loss", "actor/approx_kl"):
        assert float(host[PPO_OUT_NAMES[key]]) == pytest.approx(float(metrics[key]), rel=5e-4, abs=2e-5), key


@pytest.mark.parametrize("M", [1024, 37])
def test_rollout_step_bf16_pinned_to_the_operand_rounded_restatement(M):
    """Rollout launch, bf16 tiles, against the operand-rounding model: at least 85 % of the elements (96 % at 1024 rows) agree to 2e-6 (f32 summation
    order); the rest carry a flipped bf16 rounding of some activation through the layers above it -- bounded by 5e-3, where the
    autocast yardstick allows 2e-2 everywhere."""
    from oracle import bf16_operand_model as BM
    from rlinf_amd import ops
    ora, pol = _bf16_policy()
    g = torch.Generator().manual_seed(3)
    states, eps = torch.randn(M, 42, generator=g), torch.randn(M, 8, generator=g)
    want_a, want_lp, want_v = BM.act(ora, states, eps)
    a, lp, v = ops.mlp_rollout_step(pol.flat.data, pol.tiles(), pol.layout, states.cuda(), eps.cuda())
    for name, got, want in (("action", a.cpu(), want_a), ("value", v.cpu(), want_v)):
        d = (got - want).abs()
        exact = float((d <= 2e-6).float().mean())
        print(f"[bf16 rollout pinned M={M}] {name}: {exact:.3f} of the elements within 2e-6, max {float(d.max()):.2e}")
        assert exact >= 0.85 and float(d.max()) <= 5e-3, (name, exact, float(d.max()))
    torch.testing.assert_close(lp.cpu(), want_lp, rtol=1e-4, atol=1e-4)  # (a - mean) / std is eps itself, whatever the mean's rounding


@pytest.mark.parametrize("M,shift", [(37, 3), (1024, 1), (45, 0), (16, 5)])
def test_states_tile_as_float4_views_and_ragged_ends(M, shift):
    """The bf16 launches fetch a tile's states as float4s (rows are contiguous; a global 16-byte load needs dword alignment only)
    and the (at most three) floats behind the array's last whole float4 with a dword load: a view that starts `shift` rows into
    an allocation (8-byte aligned at best) and row counts whose float total is no multiple of four must give the bits an aligned,
    padded copy gives -- rollout outputs, the f32 states copy in the trajectory row, and the fused step's gradients."""
    from rlinf_amd import ops
    from rlinf_amd._lib import PPO_OUT_FLOATS
    _, pol = _bf16_policy(seed=3)
    lay = pol.layout
    g = torch.Generator().manual_seed(M)
    base = torch.randn(M + shift + 2, 42, generator=g).cuda()
    view = base[shift:shift + M]
    aligned = view.clone()
    assert view.data_ptr() % 16 == (shift * 42 * 4) % 16
    eps = torch.randn(M, 8, generator=g).cuda()
    outs = []
    for st in (aligned, view):
        copy = torch.full((M, 42), float("nan"), device="cuda")
        a, lp, v = ops.mlp_rollout_step(pol.flat.data, pol.tiles(), lay, st, eps, states_copy=copy)
        outs.append((a, lp, v, copy))
    for x, y in zip(*outs):
        assert torch.equal(x, y)
    assert torch.equal(outs[0][3], aligned)
    mb = _minibatch(M, g, False)
    lp_params = ops.make_ppo_params(logprob_type="action_level", action_dim=8, chunks=1, clip_ratio_low=0.2, clip_ratio_high=0.2,
                                    value_clip=1.0, huber_delta=10.0, max_episode_steps=50, has_critic=True)
    grads = []
    for st in (aligned, view):
        dev_mb = {k: v.cuda().contiguous() for k, v in mb.items()}
        dev_mb["states"] = st
        gbuf = torch.full((ops.ppo_step_slabs(lay, M, bf16=True), lay.n_params), float("nan"), device="cuda")
        ws = torch.empty(ops.ppo_step_workspace_bytes(lay, M), dtype=torch.uint8, device="cuda")
        row = torch.zeros(PPO_OUT_FLOATS, device="cuda")
        ops.ppo_step(pol.flat.data, lay, lp_params, dev_mb, gbuf, row, ws, grad_out=1.0, bf16=True)
        grads.append((gbuf.sum(0), row))
    assert torch.equal(grads[0][0], grads[1][0]) and torch.equal(grads[0][1], grads[1][1])


@pytest.mark.parametrize("M,prox_mode,with_mask,thr,bf16", [
    (8192, "versions", False, 1.03, False),   # the shipped async configs: interpolated proximal policy + behaviour threshold
    (700, "versions", True, 1.03, False),
    (1000, "given", True, 1.05, False),
    (333, "old", True, None, False),
    (5, "versions", True, 1.03, False),
    (8192, "versions", False, 1.03, True),    # bf16 operands
    (700, "given", True, 1.05, True),
])
def test_ppo_step_decoupled_loss_vs_oracle(M, prox_mode, with_mask, thr, bf16):
    """rlx_ppo_step with rlx_ppo_step_args.decoupled (async PPO, losses.py:27-167 + :383-393) against the oracle: the rlx_dppo_out
    row, and the gradients once the ACTOR network's slabs -- left in sum form: their denominator, the behaviour-mask count, needs
    every tile's forward -- are multiplied by out[RLX_PPO_ACTOR_GRAD_SCALE]; the value network's are final as they leave.
    bf16: the yardstick of test_ppo_step_bf16_gradients_vs_autocast_oracle (no farther from the f32 gradient than twice the
    reference arithmetic under bf16 autocast)."""
    from rlinf_amd import ops
    from rlinf_amd._lib import DPPO_OUT_NAMES, PPO_ACTOR_GRAD_SCALE, PPO_OUT_FLOATS
    ora, pol = _bf16_policy(seed=13) if bf16 else _policies(seed=13)
    g = torch.Generator().manual_seed(17)
    mb = _minibatch(M, g, with_mask)
    with torch.no_grad():
        cur = ora.evaluate(mb["states"], mb["action"])["logprobs"]
    mb["prev_logprobs"] = cur + torch.randn(M, 8, generator=g) * 0.08
    versions = torch.randint(2, 6, (M, 1), generator=g).float().expand(M, 8).contiguous()  # behaviour versions 2..5, current 6
    versions[:: 7] = -1.0  # "unknown": alpha = 0
    prox = cur + torch.randn(M, 8, generator=g) * 0.02
    kw = dict(proximal_logprobs=prox if prox_mode == "given" else None, versions=versions if prox_mode == "versions" else None)

    def oracle(autocast):
        ora.zero_grad()
        with O.amp(autocast):
            out = ora.evaluate(mb["states"], mb["action"])
        shaped = O.shape_loss_inputs(out["logprobs"].float(), mb["prev_logprobs"], mb["advantages"], "action_level", 8,
                                     loss_mask=mb.get("loss_mask"), values=out["values"].float(), prev_values=mb["prev_values"],
                                     returns=mb["returns"])
        p2, v2 = O.shape_decoupled_inputs(kw["proximal_logprobs"], kw["versions"], "action_level", 8, M, shaped["logprobs"].shape)
        loss, metrics = O.decoupled_actor_critic_loss(
            proximal_logprobs=p2, versions=v2, current_version=6 if prox_mode != "old" else None, behave_weight_threshold=thr,
            clip_ratio_low=0.2, clip_ratio_high=0.2, clip_ratio_c=3.0, value_clip=1.0, huber_delta=10.0, **shaped)
        (loss * 0.5).backward()
        return {n: p.grad.clone() for n, p in ora.named_parameters()}, float(loss.detach()), {k: float(v) for k, v in metrics.items()}

    g32, loss32, metrics = oracle(False)
    g16 = oracle(True)[0] if bf16 else None

    lay = pol.layout
    lp = ops.make_ppo_params(logprob_type="action_level", action_dim=8, chunks=1, clip_ratio_low=0.2, clip_ratio_high=0.2,
                             value_clip=1.0, huber_delta=10.0, max_episode_steps=50, clip_ratio_c=3.0, has_critic=True)
    slabs = ops.ppo_step_slabs(lay, M, bf16)
    grads = torch.full((slabs, lay.n_params), float("nan"), device="cuda")
    ws = torch.empty(ops.ppo_step_workspace_bytes(lay, M), dtype=torch.uint8, device="cuda")
    row = torch.zeros(PPO_OUT_FLOATS, device="cuda")
    dev_mb = {k: v.cuda().contiguous() for k, v in mb.items()}
    if with_mask:
        dev_mb["loss_mask"] = dev_mb["loss_mask"].view(torch.uint8)
    if kw["proximal_logprobs"] is not None:
        dev_mb["proximal_logprobs"] = prox.cuda()
    if kw["versions"] is not None:
        dev_mb["versions"] = versions.cuda()
    version_dev = torch.tensor([6.0], device="cuda")  # read at execution time; the host value below is deliberately stale
    dec = ops.decoupled_step_args(lp, dev_mb, current_version=(-100 if prox_mode != "old" else None), behave_weight_threshold=thr,
                                  current_version_dev=version_dev if prox_mode != "old" else None)
    ops.ppo_step(pol.flat.data, lay, lp, dev_mb, grads, row, ws, grad_out=0.5, bf16=bf16, decoupled=dec)
    host = row.cpu()
    got = grads.sum(dim=0).cpu()
    assert torch.isfinite(got).all()
    for b, e in ops.actor_param_ranges(lay):
        got[b:e] *= host[PPO_ACTOR_GRAD_SCALE]
    if bf16:
        cat = lambda d: torch.cat([d[n].reshape(-1) for n in d])  # noqa: E731
        order = torch.cat([torch.arange(pol.offsets[n], pol.offsets[n] + g32[n].numel()) for n in g32])
        w32, w16, ours = cat(g32), cat(g16), got[order]
        rel_ours = float((ours - w32).norm() / w32.norm())
        rel_auto = float((w16 - w32).norm() / w32.norm())
        cos = float(torch.dot(ours, w32) / (ours.norm() * w32.norm()))
        assert rel_ours <= max(2.0 * rel_auto, 0.05) and cos > 0.99, (rel_ours, rel_auto, cos)
    else:
        scale = max(float(w.abs().max()) for w in g32.values())
        for name, w in g32.items():
            o = pol.offsets[name]
            w = w.reshape(-1)
            tol = 3e-4 * max(float(w.abs().max()), 1e-3 * scale) + 1e-7
            err = float((got[o:o + w.numel()] - w).abs().max())
            assert err <= tol, (name, err, tol)
    mtol = dict(rel=2e-2, abs=2e-3) if bf16 else dict(rel=5e-4, abs=5e-5)
    assert float(host[DPPO_OUT_NAMES["loss"]]) == pytest.approx(loss32, **mtol)
    for key in ("actor/policy_loss", "actor/proximal_ratio", "actor/clipped_proximal_ratio", "actor/clip_fraction",
                "actor/dual_clip_fraction", "actor/behav_clip_fraction", "actor/proximal_approx_kl", "actor/behav_approx_kl",
                "critic/value_loss"):
        if bf16 and (key.endswith("fraction") or key.endswith("_kl")):  # counts of elements on a clip edge / sums of ~1e-2 log-ratios
            assert float(host[DPPO_OUT_NAMES[key]]) == pytest.approx(metrics[key], abs=0.02), key
        else:
            assert float(host[DPPO_OUT_NAMES[key]]) == pytest.approx(metrics[key], **mtol), key
    if prox_mode == "versions" and "actor/average_version" in metrics:
        assert float(host[DPPO_OUT_NAMES["actor/average_version"]]) == pytest.approx(metrics["actor/average_version"], rel=1e-5)
    if not bf16:  # the fused launch and the staged entry points (rlx_decoupled_loss_fwd on the same forward) agree on the row
        lpg, _, val, _, _ = ops.mlp_train_fwd(pol.flat.data, pol.packed(), lay, dev_mb["states"], dev_mb["action"])
        _, staged = ops.ppo_loss(lpg, dev_mb["prev_logprobs"], dev_mb["advantages"], logprob_type="action_level", action_dim=8,
                                 clip_ratio_low=0.2, clip_ratio_high=0.2, values=val, prev_values=dev_mb["prev_values"],
                                 returns=dev_mb["returns"], value_clip=1.0, huber_delta=10.0, loss_mask=dev_mb.get("loss_mask"),
                                 max_episode_steps=50, clip_ratio_c=3.0, has_critic=True,
                                 decoupled=dict(proximal_logprobs=dev_mb.get("proximal_logprobs"), versions=dev_mb.get("versions"),
                                                current_version=6 if prox_mode != "old" else None, behave_weight_threshold=thr))
        torch.testing.assert_close(row, staged, rtol=2e-4, atol=2e-5)


@pytest.mark.parametrize("groups,per,n_act", [(1, 10, 8), (2, 3, 8), (3, 1, 7)])
def test_clip_adamw_deferred_actor_scale_is_the_scaled_slab_sum(groups, per, n_act):
    """rlx_adamw_params.deferred_scale: the slab sum multiplies the actor ranges of every micro-batch's slab group by that
    micro-batch's device-side scale -- bit for bit the step a caller gets from forming that gradient itself."""
    from rlinf_amd import ops
    from rlinf_amd._lib import PPO_OUT_FLOATS
    from rlinf_amd.models.embodiment.mlp_policy import MLPPolicy
    pol = MLPPolicy(42, n_act, 1, True, False).to("cuda")
    lay, n = pol.layout, pol.n_params
    g = torch.Generator(device="cuda").manual_seed(3)
    slabs = torch.randn(groups * per, n, generator=g, device="cuda") * 1e-2
    rows = torch.zeros(groups, PPO_OUT_FLOATS + 2, device="cuda")
    rows[:, 16] = torch.tensor([1.0 / 913.0, 1.0 / 1024.0, 1.0 / 77.0], device="cuda")[:groups]
    ranges = ops.actor_param_ranges(lay)
    factor = torch.ones(groups, n, device="cuda")
    for b, e in ranges:
        factor[:, b:e] = rows[:, 16:17]
    want_g = None
    for q in range(groups):
        x = slabs[q * per].clone()
        for k in range(1, per):
            x = x + slabs[q * per + k]
        t = x * factor[q]
        want_g = t if want_g is None else want_g + t
    results = []
    for deferred in (True, False):
        params = pol.flat.data.clone()
        m, v = torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
        grads = slabs.clone() if deferred else want_g.clone().view(1, n)
        stats = ops.clip_adamw_step_(params, grads, m, v, [(0, n, 3e-4)], 1, max_grad_norm=0.5,
                                     deferred=ops.deferred_actor_scale(lay, rows[:, :PPO_OUT_FLOATS], groups) if deferred else None)
        results.append((params, m, v, stats))
    for a, b in zip(*results):
        assert torch.equal(a, b)
    assert float(results[0][3][1]) == 1.0

"""Parity of the fused PPO loss (fwd+bwd), the shuffle-gather and the clip+AdamW kernels with the CPU
oracle / committed reference outputs.  `pytest -m gpu`.

Bars: gather is bit-exact; loss, metrics and gradients within rtol 1e-5 / atol 1e-6 of the fp32 CPU result
(device expf and a different summation order are the only differences); AdamW state within rtol 1e-5 /
atol 1e-7 after several steps.
"""

import os

import math

import pytest
import torch

from conftest import GOLDEN_DIR
from oracle import ppo_oracle as O

pytestmark = pytest.mark.gpu
RTOL, ATOL = 1e-5, 1e-6


def _golden(name):
    return torch.load(os.path.join(GOLDEN_DIR, name), weights_only=False)


def _c(t):
    return None if t is None else t.cuda()


_METRICS = ("actor/policy_loss", "actor/policy_loss_abs", "actor/ratio", "actor/ratio_abs", "actor/clipped_ratio",
            "actor/dual_cliped_ratio", "actor/approx_kl", "actor/clip_fraction", "critic/value_loss",
            "critic/value_clip_ratio")


def test_ppo_loss_golden_all_variants():
    from rlinf_amd import _lib, ops
    cases = _golden("losses.pt")
    assert len(cases) == 27
    for case in cases:
        p = case["params"]
        lp = case["logprobs"].cuda().requires_grad_(True)
        v = case["values"].cuda().requires_grad_(True)
        loss, out = ops.ppo_loss(
            lp, _c(case["old_logprobs"]), _c(case["advantages"]), logprob_type=p["logprob_type"],
            action_dim=p["action_dim"], clip_ratio_low=p["clip_ratio_low"], clip_ratio_high=p["clip_ratio_high"],
            values=v, prev_values=_c(case["prev_values"]), returns=_c(case["returns"]), value_clip=p["value_clip"],
            huber_delta=p["huber_delta"], loss_mask=_c(case["loss_mask"]), loss_mask_sum=_c(case["loss_mask_sum"]),
            max_episode_steps=p["max_episode_steps"], clip_ratio_c=p.get("clip_ratio_c"),
            clip_log_ratio_min=p.get("clip_log_ratio_min"), clip_log_ratio_max=p.get("clip_log_ratio_max"),
            critic_warmup=p.get("critic_warmup", False))
        (loss * 0.5).backward()  # non-unit upstream gradient, like loss /= grad_accum
        tag = (p["logprob_type"], p["masked"], p["variant"])
        torch.testing.assert_close(loss.detach().cpu(), case["loss"], rtol=RTOL, atol=ATOL, msg=lambda m: f"{tag}: {m}")
        want_glp = case["grad_logprobs"]
        if want_glp is None:  # critic warm-up: the actor term is a constant
            assert float(lp.grad.abs().max()) == 0.0
        else:
            torch.testing.assert_close(lp.grad.cpu(), 0.5 * want_glp, rtol=RTOL, atol=1e-7, msg=lambda m: f"{tag}: {m}")
        torch.testing.assert_close(v.grad.cpu(), 0.5 * case["grad_values"], rtol=RTOL, atol=1e-7,
                                   msg=lambda m: f"{tag}: {m}")
        o = out.cpu()
        for k in _METRICS:
            got, want = float(o[_lib.PPO_OUT_NAMES[k]]), case["metrics"][k]
            assert got == pytest.approx(want, rel=2e-5, abs=2e-6), (tag, k, got, want)
        ev = {k.split("/")[-1]: val for k, val in case["metrics"].items() if "explained_variance" in k}
        for k in O.EV_KEYS:
            got = float(o[_lib.PPO_OUT_NAMES[f"ev/{k}"]])
            assert got == pytest.approx(ev[k], rel=2e-5, abs=1e-4), (tag, k)


@pytest.mark.parametrize("mb", [8192, 1000, 1])
def test_ppo_loss_vs_oracle_minibatch_shapes(mb):
    """The shipped shape: action_level, A = 8, C = 1, no mask (auto_reset) -- against the oracle."""
    from rlinf_amd import _lib, ops
    g = torch.Generator().manual_seed(mb)
    lp = (torch.randn(mb, 8, generator=g) * 0.3 - 1.0)
    old = lp + torch.randn(mb, 8, generator=g) * 0.05
    adv = torch.randn(mb, 1, generator=g)
    v = torch.randn(mb, 1, generator=g)
    pv = v + torch.randn(mb, 1, generator=g) * 0.7
    ret = torch.randn(mb, 1, generator=g) * 3
    lp0, v0 = lp.clone().requires_grad_(True), v.clone().requires_grad_(True)
    shaped = O.shape_loss_inputs(lp0, old, adv, "action_level", 8, values=v0, prev_values=pv, returns=ret)
    want, wm = O.ppo_actor_critic_loss(clip_ratio_low=0.2, clip_ratio_high=0.2, value_clip=1.0, huber_delta=10.0,
                                       max_episode_steps=50, **shaped)
    want.backward()
    lp1, v1 = lp.cuda().requires_grad_(True), v.cuda().requires_grad_(True)
    loss, out = ops.ppo_loss(lp1, old.cuda(), adv.cuda(), logprob_type="action_level", action_dim=8, clip_ratio_low=0.2,
                             clip_ratio_high=0.2, values=v1, prev_values=pv.cuda(), returns=ret.cuda(), value_clip=1.0,
                             huber_delta=10.0, max_episode_steps=50)
    loss.backward()
    torch.testing.assert_close(loss.detach().cpu(), want.detach(), rtol=RTOL, atol=ATOL)
    torch.testing.assert_close(lp1.grad.cpu(), lp0.grad, rtol=RTOL, atol=1e-8)
    torch.testing.assert_close(v1.grad.cpu(), v0.grad, rtol=RTOL, atol=1e-8)
    assert float(out[_lib.PPO_OUT_NAMES["actor/clip_fraction"]]) == pytest.approx(float(wm["actor/clip_fraction"]), abs=1e-6)


def test_actor_only_loss_and_all_false_mask():
    from rlinf_amd import ops
    g = torch.Generator().manual_seed(4)
    lp = torch.randn(64, 8, generator=g) * 0.2
    old = lp + torch.randn(64, 8, generator=g) * 0.05
    adv = torch.randn(64, 1, generator=g)
    shaped = O.shape_loss_inputs(lp, old, adv, "action_level", 8)
    want, _ = O.ppo_actor_loss(shaped["logprobs"], shaped["old_logprobs"], shaped["advantages"], 0.2, 0.28)
    loss, _ = ops.ppo_loss(lp.cuda(), old.cuda(), adv.cuda(), clip_ratio_low=0.2, clip_ratio_high=0.28, has_critic=False)
    torch.testing.assert_close(loss.cpu(), want, rtol=RTOL, atol=ATOL)
    mask = torch.zeros(64, 1, dtype=torch.bool)
    shaped = O.shape_loss_inputs(lp, old, adv, "action_level", 8, loss_mask=mask)
    want, wm = O.ppo_actor_loss(shaped["logprobs"], shaped["old_logprobs"], shaped["advantages"], 0.2, 0.2, loss_mask=mask)
    loss, out = ops.ppo_loss(lp.cuda(), old.cuda(), adv.cuda(), clip_ratio_low=0.2, clip_ratio_high=0.2, has_critic=False,
                             loss_mask=mask.cuda())
    assert float(loss) == float(want) == 0.0
    assert float(out[3]) == float(wm["actor/ratio"]) == 0.0
    with pytest.raises(AssertionError):
        ops.ppo_loss(lp.cuda(), old.cuda(), adv.cuda(), clip_ratio_low=0.2, clip_ratio_high=0.2, has_critic=False,
                     clip_ratio_c=0.9)


def test_gather_rows_golden_and_large():
    from rlinf_amd import ops
    G = _golden("shuffle.pt")
    b = G["batch"]
    fields = [b["rewards"], b["dones"][:-1], b["prev_values"][:-1], b["prev_logprobs"], b["forward_inputs"]["states"],
              b["forward_inputs"]["action"]]
    flat = [f.reshape(-1, *f.shape[2:]).contiguous().cuda() for f in fields]
    outs = ops.gather_rows(flat, G["perm"].cuda())
    want = [G["out"]["rewards"], G["out"]["dones"], G["out"]["prev_values"], G["out"]["prev_logprobs"],
            G["out"]["forward_inputs"]["states"], G["out"]["forward_inputs"]["action"]]
    for o, w in zip(outs, want):
        assert o.dtype == w.dtype and torch.equal(o.cpu(), w)
    # the north-star buffer: 131 072 samples, every field of the rollout batch
    N = 128 * 1024
    g = torch.Generator().manual_seed(1)
    big = [torch.randn(N, 42, generator=g), torch.randn(N, 8, generator=g), torch.randn(N, 8, generator=g),
           torch.randn(N, 1, generator=g), torch.rand(N, 1, generator=g) < 0.5, torch.randn(N, 3, generator=g).to(torch.float16),
           torch.randint(0, 255, (N, 5), generator=g).to(torch.uint8)]
    perm = torch.randperm(N, generator=torch.Generator().manual_seed(1234))
    outs = ops.gather_rows([t.cuda() for t in big], perm.cuda())
    for o, t in zip(outs, big):
        assert torch.equal(o.cpu(), t[perm])
    # a permutation is invertible: gathering with the inverse restores the buffer
    inv = torch.empty_like(perm)
    inv[perm] = torch.arange(N)
    back = ops.gather_rows(outs, inv.cuda())
    for o, t in zip(back, big):
        assert torch.equal(o.cpu(), t)
    assert ops.gather_rows([torch.zeros(0, 4).cuda()], torch.zeros(0, dtype=torch.int64).cuda())[0].shape == (0, 4)


def _flat_from(pol):
    names, sizes = [], []
    for n, p in pol.named_parameters():
        names.append(n)
        sizes.append(p.numel())
    return names, sizes


@pytest.mark.parametrize("one_launch", [False, True])
@pytest.mark.parametrize("slabs", [1, 3])
def test_clip_adamw_matches_torch(slabs, one_launch):
    from rlinf_amd import ops
    torch.manual_seed(0)
    pol = O.OracleMLPPolicy(42, 8, 1)
    opt = O.build_adamw(pol, lr=3e-4, value_lr=1e-3)
    names, sizes = _flat_from(pol)
    n = sum(sizes)
    assert n == 287504
    flat = torch.cat([p.detach().reshape(-1) for p in pol.parameters()]).cuda()
    m = torch.zeros(n, device="cuda")
    v = torch.zeros(n, device="cuda")
    # parameter order: actor_logstd, value_head.*, backbone.*, actor_mean.* -> critic group is one range
    off, groups, lo = 0, [], None
    crit = [i for i, nm in enumerate(names) if "value_head" in nm]
    starts = [sum(sizes[:i]) for i in range(len(sizes))]
    c0, c1 = starts[crit[0]], starts[crit[-1]] + sizes[crit[-1]]
    groups = [(0, c0, 3e-4), (c0, c1, 1e-3), (c1, n, 3e-4)]
    g = torch.Generator().manual_seed(1)
    sync = ops.adamw_sync_words(n, "cuda") if one_launch else None
    for step in range(1, 5):
        grads = [torch.randn(p.shape, generator=g) * (0.05 if step % 2 else 0.0005) for p in pol.parameters()]
        for p, gr in zip(pol.parameters(), grads):
            p.grad = gr.clone()
        gn = torch.nn.utils.clip_grad_norm_(pol.parameters(), 0.5)
        opt.step()
        gflat = torch.cat([gr.reshape(-1) for gr in grads])
        if slabs > 1:  # split-K slabs that sum to the gradient, plus a 1/world scale
            parts = torch.randn(slabs - 1, n, generator=g) * 0.01
            gdev = torch.cat([parts, (gflat * 2.0 - parts.sum(0))[None]], 0).cuda().contiguous()
            scale = 0.5
        else:
            gdev, scale = gflat.cuda().clone(), 1.0
        stats = ops.clip_adamw_step_(flat, gdev, m, v, groups, step, max_grad_norm=0.5, grad_scale=scale, sync=sync)
        assert float(stats[0]) == pytest.approx(float(gn), rel=1e-5)
        assert float(stats[1]) == 1.0
        want = torch.cat([p.detach().reshape(-1) for p in pol.parameters()])
        torch.testing.assert_close(flat.cpu(), want, rtol=1e-5, atol=2e-7)
    st = opt.state[next(iter(pol.value_head.parameters()))]
    torch.testing.assert_close(m[c0:c0 + 42 * 256].cpu(), st["exp_avg"].reshape(-1), rtol=1e-5, atol=1e-9)
    # non-finite gradient norm: the update is skipped (fsdp_model_manager.py:444-449)
    before = flat.clone()
    bad = torch.zeros(n, device="cuda")
    bad[5] = float("inf")
    stats = ops.clip_adamw_step_(flat, bad, m, v, groups, 5, max_grad_norm=0.5, sync=sync)
    assert float(stats[1]) == 0.0 and torch.equal(flat, before)


@pytest.mark.parametrize("image", ["none", "bf16", "f32"])
@pytest.mark.parametrize("slabs,deferred", [(1, False), (10, False), (8, True)])
def test_one_launch_optimizer_step(slabs, deferred, image):
    """rlx_adamw_params.sync_words: slab sum + norm + clip + AdamW as one launch around a device-side exchange of the norm
    partials.  Same element arithmetic as the two launches; the norm's f64 partials are formed over different blocks (2048
    parameters, hidden matrices as segments of their own), so the comparison allows the last bit of the norm to move --
    parameters, moments, clipped gradient, norm and the device step counter over a run of steps, eager and as a replayed hipGraph
    (the exchange's epoch advances on the device), with a skipped step (non-finite norm) in the middle.  The weight image the
    optimizer keeps (8-byte forward stores, the transposed image assembled through LDS) must equal a fresh pack of the final
    parameters bit for bit."""
    from rlinf_amd import ops
    from rlinf_amd.models.embodiment.mlp_policy import MLPPolicy
    if ops.adamw_sync_words(8, "cuda") is None:
        pytest.skip("RLX_ADAMW_ONE_LAUNCH=0")
    torch.manual_seed(3)
    dt = torch.bfloat16 if image == "bf16" else torch.float32
    runs = []
    for one in (False, True):
        pol = MLPPolicy(42, 8, 1, True, False, compute_dtype=dt).to("cuda")
        torch.manual_seed(4)
        with torch.no_grad():
            pol.flat.copy_(torch.randn_like(pol.flat) * 0.1)
        n, lay = pol.n_params, pol.layout
        tiles = pol.tiles() if image != "none" else None
        m, v = torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
        state = torch.zeros(2, dtype=torch.int32, device="cuda")
        stats = torch.zeros(2, device="cuda")
        g = torch.Generator(device="cuda").manual_seed(5)
        grads = torch.empty(slabs, n, device="cuda")
        rows = torch.zeros(4, ops._lib.PPO_OUT_FLOATS, device="cuda")
        rows[:, ops._lib.PPO_ACTOR_GRAD_SCALE] = torch.tensor([0.5, 0.25, 2.0, 1.5], device="cuda")
        dfr = ops.deferred_actor_scale(lay, rows, 4) if deferred else None
        ws = torch.empty(ops._lib.load().rlx_adamw_workspace_bytes(n), dtype=torch.uint8, device="cuda")
        step = ops.PreparedAdamw(pol.flat.data, grads, m, v, pol.group_ranges(3e-4, 1e-3), betas=(0.9, 0.999), eps=1e-8,
                                 weight_decay=0.01, max_grad_norm=0.5, grad_scale=0.5, stats=stats, step_state=state, workspace=ws,
                                 tile_layout=lay if tiles is not None else None, tiles=tiles, deferred=dfr,
                                 sync=ops.adamw_sync_words(n, "cuda") if one else None)
        trace = []
        stream = torch.cuda.current_stream().cuda_stream
        for it in range(6):
            grads.normal_(generator=g).mul_(0.02 if it % 2 else 0.0002)
            if it == 3:
                grads[0, 17] = float("nan")
            step(stream)
            trace.append((stats.clone(), state.clone(), grads[0].clone()))
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())  # (the eager steps' trace clones still read `grads` on the default stream)
        with torch.cuda.stream(side):
            grads.normal_(generator=g).mul_(0.01)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=side):
                for _ in range(8):
                    step(side.cuda_stream)
            for _ in range(25):
                graph.replay()
        side.synchronize()
        torch.cuda.synchronize()
        if tiles is not None:  # the image the optimizer kept == the image of its parameters
            fresh = ops.mlp_pack_tiles(pol.flat.data, lay, None, bf16=image == "bf16")
            iv = torch.int16 if image == "bf16" else torch.int32
            assert torch.equal(tiles.view(iv), fresh.view(iv)), ("weight image out of step with the parameters", one)
        runs.append(dict(p=pol.flat.data.clone(), m=m, v=v, trace=trace, stats=stats.clone(), state=state.clone()))
    a, b = runs
    assert torch.equal(a["state"], b["state"]) and float(a["stats"][1]) == float(b["stats"][1]) == 1.0
    assert int(b["state"][0]) + int(b["state"][1]) == 5 + 8 * 25  # one eager step skipped (nan), every replayed step applied
    for k in ("p", "m", "v"):
        torch.testing.assert_close(b[k], a[k], rtol=1e-5, atol=1e-9, msg=lambda t, k=k: f"{k}: {t}")
    for it, ((s0, t0, g0), (s1, t1, g1)) in enumerate(zip(a["trace"], b["trace"])):
        assert torch.equal(t0, t1) and float(s0[1]) == float(s1[1]) == (0.0 if it == 3 else 1.0)
        if it != 3:
            assert float(s1[0]) == pytest.approx(float(s0[0]), rel=1e-6)
            torch.testing.assert_close(g1, g0, rtol=1e-6, atol=0.0)
        else:
            assert not math.isfinite(float(s0[0])) and not math.isfinite(float(s1[0]))


def _one_launch_setup(image="bf16"):
    from rlinf_amd import ops
    from rlinf_amd.models.embodiment.mlp_policy import MLPPolicy
    torch.manual_seed(4)
    pol = MLPPolicy(42, 8, 1, True, False, compute_dtype=torch.bfloat16 if image == "bf16" else torch.float32).to("cuda")
    n = pol.n_params
    tiles = pol.tiles()
    m, v = torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
    state, stats = torch.zeros(2, dtype=torch.int32, device="cuda"), torch.zeros(2, device="cuda")
    grads = torch.randn(10, n, device="cuda", generator=torch.Generator(device="cuda").manual_seed(5)) * 0.01
    ws = torch.empty(ops._lib.load().rlx_adamw_workspace_bytes(n), dtype=torch.uint8, device="cuda")
    sync = ops.adamw_sync_words(n, "cuda")
    mk = lambda sy: ops.PreparedAdamw(pol.flat.data, grads, m, v, pol.group_ranges(3e-4, 1e-3), betas=(0.9, 0.999), eps=1e-8,  # noqa: E731
                                      weight_decay=0.01, max_grad_norm=0.5, grad_scale=1.0, stats=stats, step_state=state,
                                      workspace=ws, tile_layout=pol.layout, tiles=tiles, sync=sy)
    return ops, pol, tiles, m, v, state, stats, grads, sync, mk


def test_one_launch_sticky_word_skips_every_later_step_as_a_whole():
    """Word 1 of the sync buffer (set by a launch whose exchange expired) is read by every block before anything else: the step
    then skips as a whole -- parameters, moments, weight image and step counter untouched, a NaN norm reported -- however often
    it is called, eagerly or replayed; ops.check_adamw_sync turns that into a warning and tells the caller to continue on two
    launches, which then apply the update the sticky launches withheld."""
    ops, pol, tiles, m, v, state, stats, grads, sync, mk = _one_launch_setup()
    if sync is None:
        pytest.skip("RLX_ADAMW_ONE_LAUNCH=0")
    step = mk(sync)
    stream = torch.cuda.current_stream().cuda_stream
    step(stream)  # a normal step first: epoch 1, slots published
    torch.cuda.synchronize()
    assert float(stats[1]) == 1.0 and ops.check_adamw_sync(sync, float(stats[0])) is False
    before = [t.clone() for t in (pol.flat.data, m, v, tiles.view(torch.int16), state)]
    sync[1] = 1  # what an expired poll leaves behind
    for _ in range(3):
        step(stream)
    torch.cuda.synchronize()
    assert not math.isfinite(float(stats[0])) and float(stats[1]) == 0.0
    for a, b in zip(before[:4], (pol.flat.data, m, v, tiles.view(torch.int16))):
        assert torch.equal(a, b)
    assert int(state[0]) == 1 and int(state[1]) == 0  # one applied step on record, none pending
    with pytest.warns(RuntimeWarning, match="timed out waiting for its own workgroups"):
        assert ops.check_adamw_sync(sync, float(stats[0])) is True
    assert ops.check_adamw_sync(None, float("nan")) is False and ops.check_adamw_sync(sync, 1.0) is False
    mk(None)(stream)  # the fallback: two launches, same buffers
    torch.cuda.synchronize()
    assert math.isfinite(float(stats[0])) and float(stats[1]) == 1.0 and int(state[0]) + int(state[1]) == 2
    assert not torch.equal(before[0], pol.flat.data)


def test_one_launch_expiry_on_a_stream_that_cannot_hold_the_plan(monkeypatch):
    """A REAL expiry: the launch is forced onto a stream of 8 compute units (16 resident blocks of the plan's 144; the library's
    own residency bound -- taken against the stream's CU mask -- would pick two launches, RLX_ONE_LAUNCH_TEST_OVERSUBSCRIBE
    overrides it) with a 20 ms poll bound.  The resident blocks expire, set the sticky word and poison their slots; the blocks
    that become resident afterwards read the word first thing: nobody applies anything.  Without the override the same call on
    the same stream is the two-launch form and simply works."""
    from rlinf_amd.utils.streams import MaskedStream
    ops, pol, tiles, m, v, state, stats, grads, sync, mk = _one_launch_setup()
    if sync is None:
        pytest.skip("RLX_ADAMW_ONE_LAUNCH=0")
    small = MaskedStream("cuda", 8)
    try:
        before = [t.clone() for t in (pol.flat.data, m, v, tiles.view(torch.int16))]
        torch.cuda.synchronize()
        with torch.cuda.stream(small.stream):
            mk(sync)(small.stream.cuda_stream)  # capacity 16 < 144 blocks -> two launches
        small.stream.synchronize()
        assert float(stats[1]) == 1.0 and int(sync[1]) == 0 and int(sync[0]) == 0  # (the exchange words were not used)
        after_two = pol.flat.data.clone()
        assert not torch.equal(before[0], after_two)
        monkeypatch.setenv("RLX_ONE_LAUNCH_TEST_OVERSUBSCRIBE", "1")
        monkeypatch.setenv("RLX_ONE_LAUNCH_POLL_MS", "20")
        held = [t.clone() for t in (pol.flat.data, m, v, tiles.view(torch.int16))]
        with torch.cuda.stream(small.stream):
            mk(sync)(small.stream.cuda_stream)
        small.stream.synchronize()
        assert int(sync[1]) == 1, "the poll should have expired"
        assert not math.isfinite(float(stats[0])) and float(stats[1]) == 0.0
        for a, b in zip(held, (pol.flat.data, m, v, tiles.view(torch.int16))):
            assert torch.equal(a, b), "a block applied its update although the exchange expired"
        monkeypatch.delenv("RLX_ONE_LAUNCH_TEST_OVERSUBSCRIBE")
        with pytest.warns(RuntimeWarning):
            assert ops.check_adamw_sync(sync, float(stats[0])) is True
    finally:
        small.close()


# ---- decoupled (async) PPO loss: registry name "decoupled_actor_critic" -------------------------------------
@pytest.mark.parametrize("logprob_type", ["action_level", "token_level", "chunk_level"])
@pytest.mark.parametrize("prox_mode", ["given", "old", "versions"])
@pytest.mark.parametrize("masked,thr,warmup", [(False, None, False), (True, None, False), (True, 1.05, False),
                                               (True, 1.02, True)])
def test_decoupled_loss_vs_oracle(logprob_type, prox_mode, masked, thr, warmup):
    from rlinf_amd.algorithms import registry

    g = torch.Generator().manual_seed(17)
    bsz, C, A = 300, 2, 4
    lp0 = torch.randn(bsz, C * A, generator=g) * 0.3
    old = lp0 + 0.1 * torch.randn(bsz, C * A, generator=g)
    prox = lp0 + 0.05 * torch.randn(bsz, C * A, generator=g) if prox_mode == "given" else None
    versions = None
    if prox_mode != "old":
        versions = torch.randint(-1, 6, (bsz, 1), generator=g).float().expand(bsz, C * A).contiguous()
    n_adv = (bsz,) if logprob_type == "chunk_level" else (bsz, C)
    adv = torch.randn(*n_adv, generator=g)
    vals, pv, ret = (torch.randn(*n_adv, generator=g) for _ in range(3))
    lm = lms = None
    if masked:
        lm = torch.rand(*n_adv, generator=g) < 0.7
        lms = lm.sum(dim=0, keepdim=True).expand_as(lm).contiguous()
    common = dict(clip_ratio_low=0.2, clip_ratio_high=0.28, clip_ratio_c=3.0, value_clip=0.5, huber_delta=1.0,
                  max_episode_steps=80, critic_warmup=warmup)
    lp = lp0.clone().requires_grad_(True)
    v = vals.clone().requires_grad_(True)
    shaped = O.shape_loss_inputs(lp, old, adv, logprob_type, A, loss_mask=lm, loss_mask_sum=lms, values=v, prev_values=pv,
                                 returns=ret)
    p2, v2 = O.shape_decoupled_inputs(prox, versions, logprob_type, A, bsz, shaped["logprobs"].shape)
    wloss, wm = O.decoupled_actor_critic_loss(proximal_logprobs=p2, versions=v2, current_version=5,
                                              behave_weight_threshold=thr, **common, **shaped)
    (wloss * 0.5).backward()

    dlp = lp0.cuda().requires_grad_(True)
    dv = vals.cuda().requires_grad_(True)
    loss, metrics = registry.policy_loss(
        task_type="embodied", loss_type="decoupled_actor_critic", logprob_type=logprob_type, reward_type=
        "chunk_level" if logprob_type == "chunk_level" else "action_level", single_action_dim=A, logprobs=dlp, values=dv,
        old_logprobs=_c(old), advantages=_c(adv), returns=_c(ret), prev_values=_c(pv), proximal_logprobs=_c(prox),
        versions=_c(versions), current_version=5, behave_weight_threshold=thr, loss_mask=_c(lm), loss_mask_sum=_c(lms),
        **common)
    (loss * 0.5).backward()
    tag = (logprob_type, prox_mode, masked, thr, warmup)
    torch.testing.assert_close(loss.detach().cpu(), wloss.detach(), rtol=RTOL, atol=ATOL, msg=lambda m: f"{tag}: {m}")
    want_g = torch.zeros_like(lp0) if lp.grad is None else lp.grad
    torch.testing.assert_close(dlp.grad.cpu(), want_g, rtol=RTOL, atol=1e-7, msg=lambda m: f"{tag} d_logprobs: {m}")
    torch.testing.assert_close(dv.grad.cpu(), v.grad, rtol=RTOL, atol=1e-7, msg=lambda m: f"{tag} d_values: {m}")
    want_keys = {k for k in wm if k.startswith("actor/")}
    got_keys = {k for k in metrics if k.startswith("actor/")}
    assert want_keys == got_keys, (tag, want_keys ^ got_keys)
    for k in sorted(want_keys) + ["critic/value_loss", "critic/value_clip_ratio"]:
        torch.testing.assert_close(torch.tensor(float(metrics[k])), torch.as_tensor(wm[k]).float(), rtol=RTOL, atol=ATOL,
                                   msg=lambda m: f"{tag} {k}: {m}")


@pytest.mark.parametrize("loss_type", ["actor_critic", "actor", "decoupled_actor_critic"])
@pytest.mark.parametrize("masked", [False, True])
@pytest.mark.parametrize("logprob_type", ["token_level", "action_level"])
def test_chunk_level_reward_with_token_level_logprobs(loss_type, masked, logprob_type):
    """reward_type='chunk_level' + logprob_type='token_level' (six shipped configs, e.g. the OpenVLA GRPO ones): advantages,
    mask and values are per env-step ([bsz, 1] flattened, utils.py:296-308) while every action dimension of every chunk
    keeps its own ratio.  With logprob_type='action_level' and C > 1 every chunk keeps a ratio (sum over action_dim) under the
    one advantage / mask element of its env step, and the ratio metrics divide by the UN-broadcast mask count (the reference
    only expands the mask for 3-D ratios, losses.py:288-290)."""
    from rlinf_amd.algorithms import registry

    g = torch.Generator().manual_seed(23)
    bsz, C, A = 200, 3, 4
    lp0 = torch.randn(bsz, C * A, generator=g) * 0.3
    old = lp0 + 0.1 * torch.randn(bsz, C * A, generator=g)
    adv, vals, pv, ret = (torch.randn(bsz, 1, generator=g) for _ in range(4))
    lm = lms = None
    if masked:
        lm = torch.rand(bsz, 1, generator=g) < 0.7
        lms = lm.sum(dim=0, keepdim=True).expand_as(lm).contiguous()
    common = dict(clip_ratio_low=0.2, clip_ratio_high=0.28, clip_ratio_c=3.0, max_episode_steps=80)
    critic = dict(value_clip=0.5, huber_delta=1.0)
    lp = lp0.clone().requires_grad_(True)
    v = vals.clone().requires_grad_(True)
    shaped = O.shape_loss_inputs(lp, old, adv, logprob_type, A, loss_mask=lm, loss_mask_sum=lms, values=v, prev_values=pv,
                                 returns=ret, reward_type="chunk_level")
    if loss_type == "actor_critic":
        wloss, wm = O.ppo_actor_critic_loss(**common, **critic, **shaped)
    elif loss_type == "actor":
        wloss, wm = O.ppo_actor_loss(**{k: shaped[k] for k in ("logprobs", "old_logprobs", "advantages", "loss_mask",
                                                               "loss_mask_sum")}, **common)
    else:
        wloss, wm = O.decoupled_actor_critic_loss(behave_weight_threshold=1.05, **common, **critic, **shaped)
    wloss.backward()
    dlp = lp0.cuda().requires_grad_(True)
    dv = vals.cuda().requires_grad_(True)
    kw = dict(task_type="embodied", loss_type=loss_type, logprob_type=logprob_type, reward_type="chunk_level",
              single_action_dim=A, logprobs=dlp, old_logprobs=_c(old), advantages=_c(adv), loss_mask=_c(lm),
              loss_mask_sum=_c(lms), **common)
    if loss_type != "actor":
        kw.update(values=dv, returns=_c(ret), prev_values=_c(pv), **critic)
    if loss_type == "decoupled_actor_critic":
        kw.update(behave_weight_threshold=1.05)
    loss, metrics = registry.policy_loss(**kw)
    loss.backward()
    torch.testing.assert_close(loss.detach().cpu(), wloss.detach(), rtol=RTOL, atol=ATOL)
    torch.testing.assert_close(dlp.grad.cpu(), lp.grad, rtol=RTOL, atol=1e-7)
    if loss_type != "actor":
        torch.testing.assert_close(dv.grad.cpu(), v.grad, rtol=RTOL, atol=1e-7)
    for k, want in wm.items():
        if k.startswith("actor/") or k in ("critic/value_loss", "critic/value_clip_ratio"):
            torch.testing.assert_close(torch.tensor(float(metrics[k])), torch.as_tensor(want).float(), rtol=RTOL, atol=ATOL,
                                       msg=lambda m: f"{k}: {m}")


def test_ext_hook_callees_behind_a_reference_shaped_registry():
    """RLINF_EXT_MODULE route: rlinf_amd.ext.register() re-registers its callees in (a stand-in for) RLinf's registry;
    RLinf then calls them with the tensors its own preprocess_loss_inputs shaped and, for reasoning, with its own
    loss_agg_func objects -- exactly what the oracle's shaping produces here."""
    import sys
    import types

    from oracle import token_oracle as TO
    from oracle.make_golden import token_batch

    captured = {"adv": {}, "loss": {}, "model": {}}
    fake_models = types.ModuleType("rlinf.models")
    fake_models.register_model = lambda name, builder, category="embodied", force=False: captured["model"].__setitem__(name, (builder, category, force))
    fake = types.ModuleType("rlinf.algorithms.registry")
    fake.register_advantage = lambda name: (lambda fn: captured["adv"].__setitem__(name, fn) or fn)
    fake.register_policy_loss = lambda name: (lambda fn: captured["loss"].__setitem__(name, fn) or fn)
    saved = {k: sys.modules.get(k) for k in ("rlinf", "rlinf.algorithms", "rlinf.algorithms.registry", "rlinf.models")}
    pkg, sub = types.ModuleType("rlinf"), types.ModuleType("rlinf.algorithms")
    pkg.__path__, sub.__path__ = [], []
    pkg.algorithms, sub.registry, pkg.models = sub, fake, fake_models
    sys.modules.update({"rlinf": pkg, "rlinf.algorithms": sub, "rlinf.algorithms.registry": fake, "rlinf.models": fake_models})
    try:
        from rlinf_amd import ext
        ext.register()
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    assert set(captured["adv"]) == {"gae", "grpo", "reinpp"}
    assert set(captured["loss"]) == {"actor_critic", "actor", "decoupled_actor_critic"}
    from rlinf_amd.models.embodiment.mlp_policy_module import build_reference_named_mlp_policy
    assert captured["model"] == {"mlp_policy": (build_reference_named_mlp_policy, "embodied", True)}  # the model leg (rlinf/models/__init__.py:31-53)

    # reasoning "actor": the learner's kwargs, the REFERENCE-named aggregation function object
    b = token_batch(91, 8, 21, 5)
    lp0 = b["old_logprobs"] + 0.3 * torch.randn(8, 21, generator=torch.Generator().manual_seed(1))
    for agg in (TO.masked_mean, TO.seq_mean_token_sum, TO.seq_mean_token_mean):
        lp = lp0.clone().requires_grad_(True)
        wloss, wm = TO.token_actor_loss(lp, b["old_logprobs"], b["advantages"], 0.2, 0.28, loss_mask=b["loss_mask"],
                                        clip_ratio_c=3.0, loss_agg_func=agg, fast_path_zero_loss_mask=True)
        wloss.backward()
        dlp = lp0.cuda().requires_grad_(True)
        loss, metrics = captured["loss"]["actor"](
            task_type="reasoning", loss_type="actor", loss_agg_func=agg, logprobs=dlp, old_logprobs=_c(b["old_logprobs"]),
            advantages=_c(b["advantages"]), clip_ratio_c=3.0, clip_ratio_low=0.2, clip_ratio_high=0.28,
            loss_mask=_c(b["loss_mask"]), clip_log_ratio_min=None, clip_log_ratio_max=None, fast_path_zero_loss_mask=True)
        loss.backward()
        torch.testing.assert_close(loss.detach().cpu(), wloss.detach(), rtol=RTOL, atol=ATOL)
        torch.testing.assert_close(dlp.grad.cpu(), lp.grad, rtol=RTOL, atol=1e-8)
        assert set(metrics) == set(wm)
        for k in wm:
            torch.testing.assert_close(metrics[k].cpu(), wm[k].float(), rtol=RTOL, atol=ATOL, msg=lambda m: f"{k}: {m}")

    # decoupled_actor_critic on reference-shaped tensors (action level: [bsz, C])
    g = torch.Generator().manual_seed(4)
    bsz, C, A = 120, 2, 4
    raw, old = torch.randn(bsz, C * A, generator=g) * 0.3, None
    old = raw + 0.1 * torch.randn(bsz, C * A, generator=g)
    versions = torch.randint(0, 6, (bsz, 1), generator=g).float().expand(bsz, C * A).contiguous()
    adv, vals, pv, ret = (torch.randn(bsz, C, generator=g) for _ in range(4))
    lm = torch.rand(bsz, C, generator=g) < 0.7
    lp = raw.clone().requires_grad_(True)
    shaped = O.shape_loss_inputs(lp, old, adv, "action_level", A, loss_mask=lm, values=vals, prev_values=pv, returns=ret)
    _, v2 = O.shape_decoupled_inputs(None, versions, "action_level", A, bsz, shaped["logprobs"].shape)
    common = dict(clip_ratio_low=0.2, clip_ratio_high=0.28, clip_ratio_c=3.0, value_clip=0.5, huber_delta=1.0)
    wloss, wm = O.decoupled_actor_critic_loss(versions=v2, current_version=5, behave_weight_threshold=1.04, **common, **shaped)
    wloss.backward()
    d_raw = raw.cuda().requires_grad_(True)
    dshaped = O.shape_loss_inputs(d_raw, old.cuda(), adv.cuda(), "action_level", A, loss_mask=lm.cuda(), values=vals.cuda(),
                                  prev_values=pv.cuda(), returns=ret.cuda())
    loss, metrics = captured["loss"]["decoupled_actor_critic"](versions=v2.cuda(), current_version=5,
                                                              behave_weight_threshold=1.04, **common, **dshaped)
    loss.backward()
    torch.testing.assert_close(loss.detach().cpu(), wloss.detach(), rtol=RTOL, atol=ATOL)
    torch.testing.assert_close(d_raw.grad.cpu(), lp.grad, rtol=RTOL, atol=1e-7)
    for k in wm:
        if k.startswith("actor/"):
            torch.testing.assert_close(metrics[k].cpu().float(), torch.as_tensor(wm[k]).float(), rtol=RTOL, atol=ATOL,
                                       msg=lambda m: f"{k}: {m}")

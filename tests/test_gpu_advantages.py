"""Parity of the HIP advantage/return/mask kernels (through the C ABI) with the CPU oracle and with the
committed reference outputs.  Needs a real MI355X: run with `pytest -m gpu`.

Bars: bool/int outputs bit-exact; the streaming scan (nseg == 1) un-normalised adv/ret bit-exact;
segmented scans and everything that passes through a mean/std reduction within rtol 1e-5 / atol 5e-6
of the fp32 CPU result (different summation order only).
"""

import os

import pytest
import torch

from conftest import GOLDEN_DIR, synth_rollout
from oracle import ppo_oracle as O

pytestmark = pytest.mark.gpu

# |err| <= ATOL + RTOL*|ref|.  The accumulators reach |g| ~ 10 here (1 ulp = 1e-6), the segmented scan
# and the moment reductions re-associate a handful of f32 additions -> a few ulp of the accumulator.
RTOL, ATOL = 1e-5, 5e-6
VARIANTS = [(1 | (s << 8)) for s in (1, 2, 4, 8, 16)]


def _ops():
    from rlinf_amd import ops
    return ops


def _golden(name):
    return torch.load(os.path.join(GOLDEN_DIR, name), weights_only=False)


def _cuda(t):
    return None if t is None else t.cuda()


def test_loss_mask_golden_bit_exact():
    ops = _ops()
    for case in _golden("loss_mask.pt"):
        mask, cnt = ops.done_prefix_mask(case["dones"].cuda())
        assert torch.equal(mask.cpu(), case["loss_mask"])
        assert torch.equal(cnt.cpu(), case["loss_mask_sum_row"][:, 0])


@pytest.mark.parametrize("T,B,C,p", [(128, 1024, 1, 0.02), (50, 1000, 1, 0.05), (7, 3, 1, 0.3), (16, 100, 4, 0.05),
                                     (1, 64, 1, 0.5), (128, 8, 1, 0.02), (128, 65536, 1, 0.02)])
def test_loss_mask_vs_oracle(T, B, C, p):
    ops = _ops()
    d = synth_rollout(seed=T * 7 + B, T=T, B=B, C=C, p_done=p)["dones"]
    mask, cnt = ops.done_prefix_mask(d.cuda())
    m0, s0 = O.loss_mask_from_dones(d)
    assert torch.equal(mask.cpu(), m0)
    assert torch.equal(cnt.cpu(), s0[0, :, 0])
    # properties: monotone non-increasing along time, sum == first done index
    mm = mask.cpu().transpose(1, 2).reshape(-1, B)
    assert bool((mm[1:] <= mm[:-1]).all())


def test_advantages_golden():
    ops = _ops()
    for case in _golden("advantages.pt"):
        p = case["params"]
        rewards, values, dones, lm = case["rewards"], case["values"], case["dones"], case["loss_mask"]
        if p["reward_type"] == "chunk_level":  # reductions over the chunk dim (utils.py:80-89) are host-side views
            rewards = rewards.sum(dim=-1, keepdim=True)
            dones = dones.max(dim=-1, keepdim=True)[0]
        if p["adv_type"] == "gae":
            for variant in (0, 1):
                adv, ret = ops.gae_scan(_cuda(rewards), _cuda(values), _cuda(dones), _cuda(lm), p["gamma"],
                                        p["gae_lambda"], normalize_advantages=p["normalize_advantages"],
                                        variant=variant if rewards.shape[-1] == 1 else 0)
                if variant == 1 or rewards.shape[-1] != 1:
                    assert torch.equal(ret.cpu(), case["returns"]), p  # streaming scan: bit-exact
                    if not p["normalize_advantages"]:
                        assert torch.equal(adv.cpu(), case["advantages"]), p
                torch.testing.assert_close(ret.cpu(), case["returns"], rtol=RTOL, atol=ATOL)
                torch.testing.assert_close(adv.cpu(), case["advantages"], rtol=RTOL, atol=ATOL)
        else:
            adv, scores = ops.grpo_group_adv(_cuda(rewards), _cuda(dones), _cuda(lm), p["group_size"])
            torch.testing.assert_close(adv.cpu(), case["advantages"], rtol=RTOL, atol=ATOL)


@pytest.mark.parametrize("variant", VARIANTS)
@pytest.mark.parametrize("T,B", [(128, 1024), (50, 256), (17, 68), (128, 8), (300, 64)])
def test_gae_variants_vs_oracle(variant, T, B):
    ops = _ops()
    vec, nseg = variant & 0xff, variant >> 8
    if B % vec:
        pytest.skip("batch not divisible by vec")
    r = synth_rollout(seed=3 + T + B, T=T, B=B, p_done=0.03)
    if nseg > 1 and (T * 64 * vec * 8 + nseg * 64 * vec * 8 + 40 * nseg > 160 * 1024 or -(-T // nseg) > 64):
        from rlinf_amd._lib import RlxError
        with pytest.raises(RlxError):  # the slab does not fit in the CU's 160 KB of LDS: rejected, not wrong
            ops.gae_scan(r["rewards"].cuda(), r["values"].cuda(), r["dones"].cuda(), None, 0.99, 0.95, variant=variant)
        return
    want_adv, want_ret = O.gae_tb(r["rewards"][..., 0], r["dones"][..., 0], r["values"][..., 0], 0.99, 0.95,
                                  normalize_advantages=False)
    adv, ret = ops.gae_scan(r["rewards"].cuda(), r["values"].cuda(), r["dones"].cuda(), None, 0.99, 0.95,
                            normalize_advantages=False, variant=variant)
    adv, ret = adv.cpu()[..., 0], ret.cpu()[..., 0]
    if nseg == 1:
        assert torch.equal(adv, want_adv) and torch.equal(ret, want_ret)
    torch.testing.assert_close(adv, want_adv, rtol=RTOL, atol=ATOL)
    torch.testing.assert_close(ret, want_ret, rtol=RTOL, atol=ATOL)
    # the reference's own identity adv = ret - V[:-1], computed in f32 (advantages.py:79)
    assert torch.equal(adv, ret - r["values"][:-1, :, 0])


@pytest.mark.parametrize("masked", [False, True])
@pytest.mark.parametrize("gl", [(0.8, 0.9), (0.99, 0.95)])
def test_gae_normalised_north_star_shape(masked, gl):
    """1024 envs x 128 steps (BASELINE.json configs[1]) against the oracle, auto variant."""
    ops = _ops()
    r = synth_rollout(seed=1234, T=128, B=1024, p_done=0.02)
    lm = O.loss_mask_from_dones(r["dones"])[0] if masked else None
    want = O.embodied_adv_and_returns(adv_type="gae", rewards=r["rewards"], dones=r["dones"], values=r["values"],
                                      gamma=gl[0], gae_lambda=gl[1], loss_mask=lm)
    adv, ret = ops.gae_scan(r["rewards"].cuda(), r["values"].cuda(), r["dones"].cuda(), _cuda(lm), gl[0], gl[1])
    torch.testing.assert_close(adv.cpu(), want["advantages"], rtol=RTOL, atol=ATOL)
    torch.testing.assert_close(ret.cpu(), want["returns"], rtol=RTOL, atol=ATOL)
    adv_r, ret_r = ops.gae_scan(r["rewards"].cuda(), r["values"].cuda(), r["dones"].cuda(), _cuda(lm), gl[0], gl[1],
                                normalize_returns=True)
    sel = ret_r if lm is None else ret_r[lm.cuda()]
    assert abs(float(sel.mean())) < 1e-4 and abs(float(sel.std()) - 1.0) < 1e-3


def test_gae_edge_cases():
    ops = _ops()
    # critic-free (values=None): gamma/lambda forced to 1 (advantages.py:61-64)
    r = synth_rollout(seed=5, T=20, B=64, p_done=0.1)
    want = O.gae_tb(r["rewards"][..., 0], r["dones"][..., 0], None, 0.9, 0.8, normalize_advantages=False)
    for variant in (1, 1 | (4 << 8)):
        adv, ret = ops.gae_scan(r["rewards"].cuda(), None, r["dones"].cuda(), None, 0.9, 0.8,
                                normalize_advantages=False, variant=variant)
        torch.testing.assert_close(adv.cpu()[..., 0], want[0], rtol=RTOL, atol=ATOL)
        torch.testing.assert_close(ret.cpu()[..., 0], want[1], rtol=RTOL, atol=ATOL)
    # done at the very last row, all-done, no-done
    for p in (0.0, 1.0):
        r = synth_rollout(seed=6, T=9, B=70, p_done=p)
        r["dones"][-1] = True
        want = O.gae_tb(r["rewards"][..., 0], r["dones"][..., 0], r["values"][..., 0], 0.99, 0.95)
        adv, ret = ops.gae_scan(r["rewards"].cuda(), r["values"].cuda(), r["dones"].cuda(), None, 0.99, 0.95)
        torch.testing.assert_close(adv.cpu()[..., 0], want[0], rtol=RTOL, atol=ATOL)
    # all-False loss mask: safe_normalize leaves the array untouched (utils.py:399)
    r = synth_rollout(seed=7, T=8, B=64, p_done=0.0)
    lm = torch.zeros(8, 64, 1, dtype=torch.bool)
    adv, _ = ops.gae_scan(r["rewards"].cuda(), r["values"].cuda(), r["dones"].cuda(), lm.cuda(), 0.99, 0.95, variant=1)
    raw, _ = ops.gae_scan(r["rewards"].cuda(), r["values"].cuda(), r["dones"].cuda(), None, 0.99, 0.95,
                          normalize_advantages=False, variant=1)
    assert torch.equal(adv, raw)
    # empty buffers
    e_adv, e_ret = ops.gae_scan(torch.zeros(0, 64, 1).cuda(), torch.zeros(1, 64, 1).cuda(),
                                torch.zeros(1, 64, 1, dtype=torch.bool).cuda())
    assert e_adv.numel() == 0 and e_ret.numel() == 0
    # chunked layout (C = 4), masked
    r = synth_rollout(seed=8, T=12, B=40, C=4, p_done=0.05)
    lm, lms = O.loss_mask_from_dones(r["dones"])
    want = O.embodied_adv_and_returns(adv_type="gae", rewards=r["rewards"], dones=r["dones"], values=r["values"],
                                      gamma=0.99, gae_lambda=0.95, loss_mask=lm, loss_mask_sum=lms)
    adv, ret = ops.gae_scan(r["rewards"].cuda(), r["values"].cuda(), r["dones"].cuda(), lm.cuda(), 0.99, 0.95)
    torch.testing.assert_close(adv.cpu(), want["advantages"].contiguous(), rtol=RTOL, atol=ATOL)
    assert torch.equal(ret.cpu(), want["returns"].contiguous())


def test_masked_standardize_standalone():
    ops = _ops()
    g = torch.Generator().manual_seed(3)
    x = torch.randn(1000, 37, generator=g) * 3 + 5
    m = torch.rand(1000, 37, generator=g) < 0.4
    for mask in (None, m):
        want = O.masked_standardize(x, mask)
        got = ops.masked_standardize_(x.clone().cuda(), None if mask is None else mask.cuda())
        torch.testing.assert_close(got.cpu(), want, rtol=RTOL, atol=1e-5)


def test_standardize_published_moments_do_not_leak_between_launches():
    """The standardize pass reads mean / denominator / skip from words its block 0 publishes in front of the moment partials
    (gae_scan.hip); the launch that writes the partials clears them.  One stream, back-to-back launches whose answers differ --
    a normal array, an all-False mask (array untouched: utils.py:399), one valid element (NaN: unbiased std of one sample), a
    16-byte-misaligned view, a single-block array -- and a hipGraph whose replays see new data each time."""
    ops = _ops()
    g = torch.Generator().manual_seed(11)
    big = torch.randn(4096 * 70 + 3, generator=g) * 2 - 1
    cases = [(big, None), (big, torch.zeros_like(big, dtype=torch.bool)), (big * 3 + 4, torch.rand(big.shape, generator=g) < 0.3),
             (big, torch.arange(big.numel()) == 17), (torch.randn(100, generator=g), None), (big[:1000] + 9, None)]
    for rep in range(3):
        for x, m in cases:
            want = O.masked_standardize(x, m)
            got = ops.masked_standardize_(x.clone().cuda(), None if m is None else m.cuda()).cpu()
            torch.testing.assert_close(got, want, rtol=RTOL, atol=1e-5, equal_nan=True)
    # misaligned start (the kernel's scalar path): a view one float into an allocation
    base = torch.empty(big.numel() + 1).cuda()
    view = base[1:]
    view.copy_(big)
    torch.testing.assert_close(ops.masked_standardize_(view).cpu(), O.masked_standardize(big, None), rtol=RTOL, atol=1e-5)
    # the scan's normalisation under graph replay: both arrays, new rewards per replay, against the eager launches
    r = synth_rollout(seed=21, T=64, B=1024, p_done=0.03)
    rew, val, don = r["rewards"].cuda(), r["values"].cuda(), r["dones"].cuda()
    out = (torch.empty_like(rew), torch.empty_like(rew))
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        ops.gae_scan(rew, val, don, None, 0.99, 0.95, normalize_returns=True, out=out)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            ops.gae_scan(rew, val, don, None, 0.99, 0.95, normalize_returns=True, out=out)
    torch.cuda.current_stream().wait_stream(side)
    for k in range(4):
        rew.copy_(torch.randn(rew.shape, generator=g) * (k + 1))
        graph.replay()
        torch.cuda.synchronize()
        adv, ret = ops.gae_scan(rew, val, don, None, 0.99, 0.95, normalize_returns=True)
        assert torch.equal(out[0], adv) and torch.equal(out[1], ret)
        want = O.embodied_adv_and_returns(adv_type="gae", rewards=rew.cpu(), dones=r["dones"], values=r["values"], gamma=0.99,
                                          gae_lambda=0.95, loss_mask=None)
        torch.testing.assert_close(adv.cpu(), want["advantages"], rtol=RTOL, atol=ATOL)


@pytest.mark.parametrize("T,B,C,G", [(128, 1024, 1, 8), (80, 256, 1, 8), (12, 24, 2, 4), (20, 64, 1, 2), (5, 6, 1, 3)])
def test_grpo_vs_oracle(T, B, C, G):
    ops = _ops()
    r = synth_rollout(seed=11 + T, T=T, B=B, C=C, p_done=0.05)
    lm, lms = O.loss_mask_from_dones(r["dones"])
    want = O.embodied_adv_and_returns(adv_type="grpo", rewards=r["rewards"], dones=r["dones"], loss_mask=lm,
                                      loss_mask_sum=lms, group_size=G)
    adv, scores = ops.grpo_group_adv(r["rewards"].cuda(), r["dones"].cuda(), lm.cuda(), G)
    f = O.flatten_embodied_inputs(r["rewards"], r["dones"], None, lm, lms, want_values=False)
    want_scores = O.first_episode_scores(f["rewards"], f["dones"])
    torch.testing.assert_close(scores.cpu(), want_scores, rtol=RTOL, atol=ATOL)
    torch.testing.assert_close(adv.cpu(), want["advantages"].contiguous(), rtol=2e-5, atol=2e-5)
    # property: group members sum to ~0 wherever the whole group is unmasked at t = 0
    a0 = adv[0, :, 0].reshape(-1, G)
    full = lm[0, :, 0].reshape(-1, G).all(dim=1).cuda()
    if bool(full.any()):
        assert float(a0[full].sum(dim=1).abs().max()) < 1e-3


def test_scaled_buffer_streaming_vs_segmented_agree():
    """65536 x 128 (the shape the HBM roofline is judged on): every variant must agree with the
    bit-exact streaming scan, and the streaming scan with the oracle."""
    ops = _ops()
    r = synth_rollout(seed=99, T=128, B=65536, p_done=0.02)
    want_adv, want_ret = O.gae_tb(r["rewards"][..., 0], r["dones"][..., 0], r["values"][..., 0], 0.99, 0.95,
                                  normalize_advantages=False)
    rc, vc, dc = r["rewards"].cuda(), r["values"].cuda(), r["dones"].cuda()
    base_adv, base_ret = ops.gae_scan(rc, vc, dc, None, 0.99, 0.95, normalize_advantages=False, variant=1)
    assert torch.equal(base_adv.cpu()[..., 0], want_adv) and torch.equal(base_ret.cpu()[..., 0], want_ret)
    for variant in (1 | (2 << 8), 1 | (4 << 8), 1 | (8 << 8), 1 | (16 << 8), 0):
        adv, ret = ops.gae_scan(rc, vc, dc, None, 0.99, 0.95, normalize_advantages=False, variant=variant)
        if (variant >> 8) <= 1:
            # the streaming scan in every tuning variant -- variant 0 is what the auto heuristic picks at this shape, the
            # `gae_scan_c1<1,1,64,nt>` kernel bench.py's roofline is quoted on -- runs the CPU loop's own operation order:
            # bit for bit
            assert torch.equal(adv, base_adv) and torch.equal(ret, base_ret), variant
        else:  # segmented scans compose affine maps: same recurrence, different association
            torch.testing.assert_close(adv, base_adv, rtol=RTOL, atol=ATOL)
            torch.testing.assert_close(ret, base_ret, rtol=RTOL, atol=ATOL)


@pytest.mark.parametrize("normalize", [False, True])
def test_scaled_buffer_with_a_loss_mask(normalize):
    """65536 x 128 WITH a loss mask -- what `roofline_normalised`'s second row times: the auto heuristic picks the masked streaming
    scan (32-row non-temporal register batches, env groups paired per XCD) + the batched standardize pass.  The scan is the CPU
    loop's operation order (bit for bit against the plain variant and the oracle); the normalised advantages follow the
    reference's safe_normalize over the masked elements."""
    ops = _ops()
    r = synth_rollout(seed=98, T=128, B=65536, p_done=0.004)
    lm, _ = O.loss_mask_from_dones(r["dones"])
    assert 0.3 < float(lm.float().mean()) < 0.95
    rc, vc, dc, mc = r["rewards"].cuda(), r["values"].cuda(), r["dones"].cuda(), lm.cuda()
    adv, ret = ops.gae_scan(rc, vc, dc, mc, 0.99, 0.95, normalize_advantages=normalize)  # variant 0: auto
    base_adv, base_ret = ops.gae_scan(rc, vc, dc, mc, 0.99, 0.95, normalize_advantages=False, variant=1)
    assert torch.equal(ret, base_ret)
    want_adv, want_ret = O.gae_tb(r["rewards"][..., 0], r["dones"][..., 0], r["values"][..., 0], 0.99, 0.95,
                                  normalize_advantages=normalize, loss_mask=lm[..., 0])
    assert torch.equal(ret.cpu()[..., 0], want_ret)
    if not normalize:
        assert torch.equal(adv, base_adv) and torch.equal(adv.cpu()[..., 0], want_adv)
    else:
        torch.testing.assert_close(adv.cpu()[..., 0], want_adv, rtol=RTOL, atol=ATOL)


@pytest.mark.parametrize("T,B", [(128, 65536), (128, 1024), (100, 700), (130, 129), (40, 64), (512, 256)])
@pytest.mark.parametrize("masked", [False, True])
def test_gae_handoff_variant_is_bit_identical(T, B, masked):
    """gae_scan_handoff (register-resident time segments, the recurrence handed down a chain BEFORE any store, then every segment's
    full pass at once: variant bit 28, segment length in bits 29-30): the sequential loop's own operations in its own order, so
    bit-identical to the streaming scan and to the CPU loop -- ragged last segments, more / fewer envs than a wave, with and
    without a loss mask and the normalisation behind it.  (Built to test whether the 64-deep vmcnt window bounds the streaming
    scan at the roofline shape; it does not -- profiles/r03_gae_handoff_sweep_*.txt -- the variant stays as a tested alternative.)"""
    from conftest import need_dev_variants
    need_dev_variants("gae_scan_handoff")
    ops = _ops()
    r = synth_rollout(seed=7, T=T, B=B, p_done=0.03)
    lm = None
    if masked:
        lm, _ = O.loss_mask_from_dones(r["dones"])
    rc, vc, dc = r["rewards"].cuda(), r["values"].cuda(), r["dones"].cuda()
    lmc = None if lm is None else lm.cuda()
    base = ops.gae_scan(rc, vc, dc, lmc, 0.99, 0.95, normalize_advantages=False, variant=1)
    base_n = ops.gae_scan(rc, vc, dc, lmc, 0.99, 0.95, normalize_advantages=True, variant=1)
    for seg_code, seg in ((2, 64), (1, 32), (0, 16)):
        if (T + seg - 1) // seg > 8:
            continue
        for nt in (0, 1):
            variant = 1 | (1 << 8) | (nt << 25) | (1 << 28) | (seg_code << 29)
            adv, ret = ops.gae_scan(rc, vc, dc, lmc, 0.99, 0.95, normalize_advantages=False, variant=variant)
            assert torch.equal(adv, base[0]) and torch.equal(ret, base[1]), (seg, nt)
            adv_n, _ = ops.gae_scan(rc, vc, dc, lmc, 0.99, 0.95, normalize_advantages=True, variant=variant)
            torch.testing.assert_close(adv_n, base_n[0], rtol=RTOL, atol=ATOL)  # the moment partials sum in another order


@pytest.mark.parametrize("C", [1, 3])
@pytest.mark.parametrize("masked", [False, True])
def test_reward_filter_mask(C, masked):
    """embodied_fsdp_actor_worker.py:235-281 restated with the reference's own tensor expressions."""
    from rlinf_amd import ops
    g = torch.Generator().manual_seed(5)
    n, B, G = 17, 64, 8
    rewards = torch.rand(n, B, C, generator=g)
    rewards[:, 8:16] *= 3.0   # one group clearly above the upper bound
    rewards[:, 24:32] *= 0.1  # one clearly below the lower bound
    lm = (torch.rand(n, B, C, generator=g) < 0.8) if masked else None
    lo, hi = 0.2 * n * C, 0.7 * n * C
    r = rewards * lm if masked else rewards
    per_env = r.transpose(0, 1).reshape(B, -1).reshape(B // G, G, -1).sum(dim=-1)
    keep = ((per_env.mean(dim=1) >= lo) & (per_env.mean(dim=1) <= hi)).repeat_interleave(G)
    want = keep.unsqueeze(0).expand(n, -1).unsqueeze(-1)
    want = (want & lm) if masked else want
    got = ops.reward_filter_mask(rewards.cuda(), None if lm is None else lm.cuda(), G, lo, hi)
    assert got.dtype == torch.bool and got.shape == want.shape
    assert torch.equal(got.cpu(), want)
    assert 0 < int(keep.sum()) < B


@pytest.mark.parametrize("masked", [False, True])
def test_global_advantage_stats(masked):
    """masked_stats / normalize_from_stats (rlinf/utils/distributed.py:942-965), restated with the reference's tensor
    expressions: two 'ranks' reduce their own batches, the three numbers are summed, both normalise with the global ones."""
    from rlinf_amd import ops
    g = torch.Generator().manual_seed(9)
    parts = [torch.randn(37, 64, 1, generator=g) * 3 + 1, torch.randn(50, 16, 1, generator=g) - 2]
    masks = [(torch.rand_like(p) < 0.7) if masked else None for p in parts]

    def ref_stats(x, m):
        x = x.double()
        x = x[m.bool()] if m is not None else x.reshape(-1)
        return torch.tensor([x.numel(), x.sum(), x.square().sum()], dtype=torch.float64)

    want_stats = sum(ref_stats(p, m) for p, m in zip(parts, masks))
    dev_stats = torch.zeros(3, dtype=torch.float64, device="cuda")
    for p, m in zip(parts, masks):
        ops.masked_stats(p.cuda(), None if m is None else m.cuda(), out=dev_stats, accumulate=True)
    torch.testing.assert_close(dev_stats.cpu(), want_stats, rtol=1e-12, atol=1e-9)
    count = want_stats[0].clamp_min(1.0)
    mean = want_stats[1] / count
    var = want_stats[2] / count - mean.square()
    for p in parts:
        want = ((p.double() - mean) * torch.rsqrt(var.clamp_min(0.0) + 1e-5)).float()
        got = ops.normalize_from_stats(p.cuda(), dev_stats)
        torch.testing.assert_close(got.cpu(), want, rtol=1e-6, atol=1e-7)
    empty = ops.masked_stats(parts[0].cuda(), torch.zeros_like(parts[0], dtype=torch.bool).cuda())
    assert empty.tolist() == [0.0, 0.0, 0.0]
    torch.testing.assert_close(ops.normalize_from_stats(parts[0].cuda(), empty).cpu(),
                               (parts[0].double() * torch.rsqrt(torch.tensor(1e-5, dtype=torch.float64))).float())


@pytest.mark.parametrize("C,mask_kind", [(1, None), (1, "full"), (3, "full"), (3, "per_step"), (2, "empty")])
def test_rollout_metrics_on_device(C, mask_kind):
    """a15 compute_rollout_metrics (metric_utils.py:422-506): masked sum / count / min / max of rewards, advantages, returns in
    one pass, against the reference's boolean-index form ``v[mask.expand_as(v)]``."""
    ops = _ops()
    g = torch.Generator().manual_seed(4)
    T, B = 37, 50
    arrs = [torch.randn(T, B, C, generator=g) * s for s in (1.0, 3.0, 0.2)]
    mask = None
    if mask_kind == "full":
        mask = torch.rand(T, B, C, generator=g) < 0.6
    elif mask_kind == "per_step":
        mask = torch.rand(T, B, 1, generator=g) < 0.6  # chunk_level rewards: one mask element per env step
    elif mask_kind == "empty":
        mask = torch.zeros(T, B, C, dtype=torch.bool)
    out = ops.rollout_metrics([a.cuda() for a in arrs], None if mask is None else mask.cuda()).cpu()
    for k, v in enumerate(arrs):
        sel = v.reshape(-1) if mask is None else v[mask.expand_as(v)]
        assert float(out[k, 1]) == sel.numel()
        if sel.numel():
            assert float(out[k, 0]) == pytest.approx(float(sel.double().sum()), rel=1e-12, abs=1e-9)
            assert float(-out[k, 2]) == float(sel.min()) and float(out[k, 3]) == float(sel.max())  # selections: exact
        else:
            assert float(out[k, 2]) == float("-inf") and float(out[k, 3]) == float("-inf")

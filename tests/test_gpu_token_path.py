"""GPU parity of the token tier (token_ops.hip through the C ABI) against the CPU oracle, the committed reference
outputs (tests/golden/token_path.pt) and size-independent properties at LLM vocabulary sizes.

Tolerances (floating point, stated per comparison):
  f32 logits   logprob / entropy / lse: 2e-5 absolute (one f32 ulp at |lse| ~ 30 is 2e-6; sums of ~1e5 terms);
               d_logits: 1e-6 absolute + 2e-5 relative
  bf16 logits  against the oracle evaluated in f32 ON THE SAME bf16 LOGITS: the f32 tolerances (the kernel computes
               in f32); against the reference's own bf16 arithmetic: logprob within one bf16 ulp (round_outputs on),
               entropy 2% (the reference sums bf16-rounded products), d_logits one bf16 ulp of the row's largest entry
  loss scalars / metrics: 1e-5 relative + 1e-6 absolute (double accumulation here vs f32 in torch)
"""

import os

import pytest
import torch

from conftest import GOLDEN_DIR
from oracle import token_oracle as TO
from oracle.make_golden import token_batch
from rlinf_amd import token_ops
from rlinf_amd.algorithms import registry
from rlinf_amd.utils import utils as UU
from rlinf_amd.workers.actor.fsdp_actor_worker import TokenLearnerStep

pytestmark = pytest.mark.gpu
DEV = "cuda"


def close(got, want, atol, rtol=0.0, what=""):
    wide = torch.float64 if (got.dtype == torch.float64 or want.dtype == torch.float64) else torch.float32
    got, want = got.detach().to(wide).cpu(), want.detach().to(wide).cpu()
    assert got.shape == want.shape, (what, got.shape, want.shape)
    bad = ~torch.isclose(got, want, atol=atol, rtol=rtol, equal_nan=True)
    assert not bad.any(), (what, float((got - want).abs().max()), int(bad.sum()))


def oracle_f32(logits, labels, temperature=1.0):
    x = logits.float() / temperature if logits.dtype == torch.float32 else (logits / temperature).float()
    return TO.logprobs_from_logits(x, labels), TO.entropy_from_logits(x), torch.logsumexp(x, -1)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("vocab", [1, 5, 8, 517, 4096, 32003])
@pytest.mark.parametrize("temperature", [1.0, 0.7])
def test_logprob_entropy_vs_oracle(dtype, vocab, temperature):
    b = token_batch(300 + vocab, 3, 5, vocab)
    x = b["logits"].to(dtype)
    lp, ent, lse = token_ops.token_logprob_fwd(x.to(DEV), b["labels"].to(DEV), temperature, with_entropy=True)
    wlp, went, wlse = oracle_f32(x, b["labels"], temperature)
    close(lp, wlp, 2e-5, what="logprob")
    close(ent, went, 2e-5, what="entropy")
    close(lse, wlse, 2e-5, what="lse")
    lp2, none, _ = token_ops.token_logprob_fwd(x.to(DEV), b["labels"].to(DEV), temperature, with_entropy=False)
    assert none is None
    assert torch.equal(lp2, lp)  # the entropy accumulation does not perturb the log-prob


def test_unaligned_rows_and_strided_slice():
    """vocab*sizeof not a multiple of 16 shifts every row's alignment; the response window of a larger
    [bsz, S, V] buffer is addressed in place."""
    bsz, S, resp, V = 3, 11, 6, 1001
    g = torch.Generator().manual_seed(7)
    for dtype in (torch.float32, torch.bfloat16):
        full = (torch.randn(bsz, S, V, generator=g) * 3).to(dtype)
        ids = torch.randint(0, V, (bsz, S), generator=g)
        window, labels = full[:, -resp - 1:-1, :], ids[:, -resp:]
        dfull = full.to(DEV)
        dwin = dfull[:, -resp - 1:-1, :]
        assert not dwin.is_contiguous()
        lp, ent, lse = token_ops.token_logprob_fwd(dwin, labels.to(DEV), 1.0, with_entropy=True)
        wlp, went, _ = oracle_f32(window, labels)
        close(lp, wlp, 2e-5)
        close(ent, went, 2e-5)
        # backward into a dense tensor and in place into the window; the rest of the buffer is untouched
        dlp = torch.randn(bsz, resp, generator=g).to(DEV)
        dense = token_ops.token_logprob_bwd(dwin, labels.to(DEV), lse, None, dlp, None)
        before = dfull.clone()
        token_ops.token_logprob_bwd(dwin, labels.to(DEV), lse, None, dlp, None, out=dwin)
        assert torch.equal(dfull[:, -resp - 1:-1, :], dense)
        assert torch.equal(dfull[:, :-resp - 1], before[:, :-resp - 1]) and torch.equal(dfull[:, -1], before[:, -1])


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("inplace", [False, True])
def test_window_mode_gradient_equals_the_sliced_path(dtype, inplace):
    """token_logprobs(window=(a, b)) on the model's whole [bsz, S, V] output (what TokenLearnerStep uses): outputs and the gradient
    of the whole tensor are bit-identical to slicing first and letting autograd's SliceBackward pad the window gradient with
    zeros -- also when the gradient overwrites the logits buffer."""
    bsz, S, resp, V = 3, 13, 7, 1001
    g = torch.Generator().manual_seed(9)
    base = (torch.randn(bsz, S, V, generator=g) * 3).to(dtype).to(DEV)
    labels = torch.randint(0, V, (bsz, resp), generator=g).to(DEV)
    w_lp, w_ent = torch.randn(bsz, resp, generator=g).to(DEV), torch.randn(bsz, resp, generator=g).to(DEV)
    a, b = S - resp - 1, S - 1
    x0 = base.clone().requires_grad_(True)
    lp0, ent0 = token_ops.token_logprobs(x0[:, a:b, :], labels, temperature=1.3, with_entropy=True, round_outputs=True)
    ((lp0 * w_lp).sum() + (ent0 * w_ent).sum()).backward()
    leaf = base.clone().requires_grad_(True)
    x1 = leaf * 1.0  # a non-leaf like an lm_head output: its buffer may be overwritten by the gradient
    lp1, ent1 = token_ops.token_logprobs(x1, labels, temperature=1.3, with_entropy=True, round_outputs=True, inplace_grad=inplace,
                                         window=(a, b))
    ((lp1 * w_lp).sum() + (ent1 * w_ent).sum()).backward()
    assert torch.equal(lp0, lp1) and torch.equal(ent0, ent1)
    assert torch.equal(x0.grad, leaf.grad)
    assert float(leaf.grad[:, :a].abs().max()) == 0.0 and float(leaf.grad[:, b:].abs().max()) == 0.0
    with pytest.raises(Exception, match="window"):
        token_ops.token_logprobs(base, labels, window=(5, 99))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("with_entropy", [False, True])
@pytest.mark.parametrize("temperature", [1.0, 1.3])
def test_backward_vs_autograd_oracle(dtype, with_entropy, temperature):
    b = token_batch(41, 4, 7, 777)
    g = torch.Generator().manual_seed(3)
    dlp, dent = torch.randn(4, 7, generator=g), torch.randn(4, 7, generator=g)
    dlp[1, 2] = 0.0
    dent[1, 2] = 0.0  # a fully masked token: its row must come back as exact zeros
    x = b["logits"].to(dtype)
    # the oracle differentiates in f32 on the logits the kernel sees (for bf16: after div_'s bf16 rounding)
    if dtype == torch.float32:
        leaf = x.clone().requires_grad_(True)
        xs, post = leaf / temperature, 1.0
    else:
        leaf = (x / temperature).float().requires_grad_(True)
        xs, post = leaf, 1.0 / temperature
    lp, ent = TO.logprobs_from_logits(xs, b["labels"]), TO.entropy_from_logits(xs)
    obj = (lp * dlp).sum() + ((ent * dent).sum() if with_entropy else 0.0)
    want = torch.autograd.grad(obj, leaf)[0] * post
    dx = x.to(DEV).requires_grad_(True)
    glp, gent = token_ops.token_logprobs(dx, b["labels"].to(DEV), temperature=temperature, with_entropy=with_entropy)
    gobj = (glp * dlp.to(DEV)).sum() + ((gent * dent.to(DEV)).sum() if with_entropy else 0.0)
    gobj.backward()
    assert dx.grad.dtype == dtype
    if dtype == torch.float32:
        close(dx.grad, want, 1e-6, 2e-5, "d_logits f32")
    else:
        ulp = want.abs().amax(-1, keepdim=True) * 2.0 ** -8
        assert ((dx.grad.float().cpu() - want).abs() <= ulp + 1e-7).all()
    assert torch.count_nonzero(dx.grad[1, 2]) == 0


def test_special_values():
    """-inf logits (masked vocabulary entries), torch's ignore_index, an out-of-range label."""
    V = 300
    g = torch.Generator().manual_seed(11)
    x = torch.randn(4, V, generator=g) * 2
    x[0, 5:200] = float("-inf")
    x[1, :] = float("-inf")
    x[1, 17] = 0.5  # a one-hot distribution: entropy 0, logprob 0
    labels = torch.tensor([3, 17, -100, V + 4])
    lp, ent, lse = token_ops.token_logprob_fwd(x.to(DEV), labels.to(DEV), 1.0, with_entropy=True)
    wlp, went, _ = oracle_f32(x[:2], labels[:2])
    close(lp[:2], wlp, 2e-5)
    close(ent[:2], went, 2e-5)
    assert abs(float(lp[1])) < 1e-6 and abs(float(ent[1])) < 1e-6
    assert float(lp[2]) == 0.0 and torch.isnan(lp[3])
    want = -torch.nn.functional.cross_entropy(x[2:3], labels[2:3], reduction="none")
    assert float(want) == 0.0
    d = token_ops.token_logprob_bwd(x.to(DEV), labels.to(DEV), lse, ent, torch.ones(4, device=DEV),
                                    torch.ones(4, device=DEV))
    assert torch.isfinite(d[:2]).all() and float(d[0, 5:200].abs().max()) == 0.0


@pytest.mark.parametrize("tag", ["f32", "bf16"])
def test_micro_batch_loss_vs_reference_golden(tag):
    """The whole token path -- logits -> loss -> d_logits -- through TokenLearnerStep against the reference's
    outputs for the same seeded micro-batch."""
    dt = torch.float32 if tag == "f32" else torch.bfloat16
    for case in torch.load(os.path.join(GOLDEN_DIR, "token_path.pt"), weights_only=False):
        p, want = case["params"], case["out"][tag]
        b = token_batch(p["seed"], p["bsz"], p["seq"], p["vocab"], zero_first=p["zero_first"])
        bsz, seq, V = p["bsz"], p["seq"], p["vocab"]
        # embed the response window in a longer sequence the way the model's output holds it
        prompt = 3
        full = torch.zeros(bsz, prompt + seq, V, dtype=dt)
        full[:, prompt - 1:-1] = b["logits"].to(dt)
        ids = torch.zeros(bsz, prompt + seq, dtype=torch.int64)
        ids[:, prompt:] = b["labels"]
        mask = torch.zeros(bsz, prompt + seq, dtype=torch.bool)
        mask[:, prompt:] = b["loss_mask"]
        step = TokenLearnerStep(
            response_len=seq, loss_agg=p["loss_agg"], clip_ratio_low=p["clip_ratio_low"],
            clip_ratio_high=p["clip_ratio_high"], clip_ratio_c=p["clip_ratio_c"],
            clip_log_ratio_min=p["clip_log_ratio_min"], clip_log_ratio_max=p["clip_log_ratio_max"],
            temperature=p["temperature"], calculate_entropy=True, entropy_bonus=p["entropy_bonus"],
            kl_beta=p["kl_beta"], kl_penalty_type=p["kl_penalty_type"])
        dlogits = full.to(DEV).requires_grad_(True)
        m_batch = dict(input_ids=ids.to(DEV), rollout_logprobs=b["old_logprobs"].to(DEV),
                       advantages=b["advantages"].to(DEV), ref_logprobs=b["ref_logprobs"].to(DEV),
                       response_mask=mask.to(DEV))
        loss, metrics = step(dlogits, m_batch, p["gradient_accumulation"])
        loss.backward()
        lp, ent = step.logprobs_and_entropy(dlogits.detach(), ids.to(DEV))
        rel = 1e-5 if tag == "f32" else 1e-2
        if tag == "f32":
            close(lp, want["logprobs"], 2e-5, what="logprobs")
            close(ent, want["entropy"], 2e-5, what="entropy")
        else:
            # torch's CPU bf16 log_softmax rounds logsumexp itself to bf16, so the reference's fixture values sit up to
            # ~1.3 bf16 ulps OF |lse| away from the f32 evaluation of the same bf16 logits (measured when the fixture was
            # made); the kernel's own contract is tighter: identical to the f32 oracle on those logits rounded once
            scaled = (b["logits"].to(dt) / p["temperature"]).float()
            lse = torch.logsumexp(scaled, -1)
            assert ((lp.cpu() - want["logprobs"]).abs() <= lse.abs().clamp(min=1.0) * 2.0 ** -7 + 1e-6).all()
            exact = TO.logprobs_from_logits(scaled, b["labels"]).bfloat16().float()
            assert (lp.cpu() == exact).float().mean() >= 0.95
            assert ((lp.cpu() - exact).abs() <= exact.abs() * 2.0 ** -7 + 1e-6).all()
            close(ent, want["entropy"], 0.0, 4e-2, what="entropy bf16")
        atol = 1e-6 if tag == "f32" else 5e-3  # bf16: the fixture's log-probs carry the lse rounding above
        close(torch.tensor(metrics["actor/final_loss"]), want["final_loss"], atol, rel, "final")
        close(torch.tensor(metrics["actor/entropy_loss"]), want["entropy_loss"], atol, rel if tag == "f32" else 4e-2,
              "entropy_loss")
        close(torch.tensor(metrics["actor/kl_loss"]), want["kl_loss"], atol, rel, "kl_loss")
        assert set(want["metrics"]) <= set(metrics), (set(want["metrics"]) - set(metrics))
        for k, v in want["metrics"].items():
            close(torch.tensor(float(metrics[k])), v.float(), atol, rel, k)
        got = dlogits.grad[:, prompt - 1:-1]
        assert float(dlogits.grad[:, :prompt - 1].abs().max()) == 0.0 and float(dlogits.grad[:, -1].abs().max()) == 0.0
        if tag == "f32":
            close(got, want["d_logits"], 1e-7, 1e-4, "d_logits")
        else:
            # PPO's clip decisions are discontinuous in the log-prob: a token whose ratio sits within bf16 noise of a
            # clip boundary may legitimately flip; everything else agrees to a bf16 ulp of the row's largest entry
            w = want["d_logits"].float()
            err = (got.float().cpu() - w).abs()
            tol = w.abs().amax(-1, keepdim=True) * 2.0 ** -6 + 1e-6
            row_ok = (err <= tol).all(-1)
            assert row_ok.float().mean() >= 0.85, float(row_ok.float().mean())


@pytest.mark.parametrize("agg", ["token-mean", "seq-mean-token-sum", "seq-mean-token-mean"])
@pytest.mark.parametrize("zero_first", [False, True])
@pytest.mark.parametrize("kl", [None, "k1", "abs", "k2", "k3"])
def test_token_loss_vs_oracle(agg, zero_first, kl):
    b = token_batch(77, 8, 33, 5, zero_first=zero_first)
    g = torch.Generator().manual_seed(9)
    lp0 = b["old_logprobs"] + 0.4 * torch.randn(8, 33, generator=g)
    ent0 = torch.rand(8, 33, generator=g) * 3
    ref_lp = b["ref_logprobs"] + (25.0 * (torch.rand(8, 33, generator=g) < 0.05))  # some tokens beyond k3's clamp
    lp = lp0.clone().requires_grad_(True)
    ent = ent0.clone().requires_grad_(True)
    aggf = TO.get_loss_agg_func(agg)
    wloss, wm = TO.token_actor_loss(lp, b["old_logprobs"], b["advantages"], 0.2, 0.28, loss_mask=b["loss_mask"],
                                    clip_ratio_c=3.0, loss_agg_func=aggf, clip_log_ratio_max=0.6,
                                    fast_path_zero_loss_mask=True)
    went = aggf(ent, mask=b["loss_mask"])
    wloss = wloss - 0.01 * went
    wkl = torch.tensor(0.0)
    if kl:
        wkl = aggf(TO.kl_penalty(ref_lp, lp, kl), b["loss_mask"])
        wloss = wloss + 0.1 * wkl
    (wloss / 4).backward()

    dlp = lp0.to(DEV).requires_grad_(True)
    dent = ent0.to(DEV).requires_grad_(True)
    loss, metrics = registry.policy_loss(
        task_type="reasoning", loss_type="actor", loss_agg_func=UU.get_loss_agg_func(agg), logprobs=dlp,
        old_logprobs=b["old_logprobs"].to(DEV), advantages=b["advantages"].to(DEV), clip_ratio_c=3.0, clip_ratio_low=0.2,
        clip_ratio_high=0.28, loss_mask=b["loss_mask"].to(DEV), clip_log_ratio_max=0.6, fast_path_zero_loss_mask=True,
        entropy=dent, entropy_bonus=0.01, ref_logprobs=ref_lp.to(DEV) if kl else None, kl_beta=0.1 if kl else 0.0,
        kl_penalty_type=kl)
    (loss / 4).backward()
    close(loss, wloss, 1e-6, 1e-5, "loss")
    close(torch.tensor(metrics["actor/entropy_loss"]), went, 1e-6, 1e-5)
    close(torch.tensor(metrics["actor/kl_loss"]), wkl, 1e-6, 1e-5)
    assert set(wm) <= set(metrics)
    for k, v in wm.items():
        close(torch.tensor(float(metrics[k])), v.float(), 1e-6, 1e-5, k)
    if zero_first and agg == "seq-mean-token-mean":
        # an all-masked sequence makes this aggregation 0/0: the loss is NaN on both sides and the gradients are
        # NaN-patterned by autograd's where()/mul choices -- nothing further to compare
        assert torch.isnan(loss) and torch.isnan(wloss)
        return
    zeros = torch.zeros_like(lp0)
    close(dlp.grad, zeros if lp.grad is None else lp.grad, 1e-9, 2e-5, "d_logprobs")
    close(dent.grad, ent.grad, 1e-9, 2e-5, "d_entropy")


def test_token_loss_unmasked_and_all_masked():
    b = token_batch(78, 4, 10, 5)
    lp0 = b["old_logprobs"] + 0.1
    for mask in (None, torch.zeros(4, 10, dtype=torch.bool)):
        lp = lp0.clone().requires_grad_(True)
        wloss, wm = TO.token_actor_loss(lp, b["old_logprobs"], b["advantages"], 0.2, 0.2, loss_mask=mask,
                                        loss_agg_func=TO.masked_mean)
        dlp = lp0.to(DEV).requires_grad_(True)
        loss, metrics = registry.policy_loss(task_type="reasoning", loss_type="actor", loss_agg_func=UU.masked_mean,
                                             logprobs=dlp, old_logprobs=b["old_logprobs"].to(DEV),
                                             advantages=b["advantages"].to(DEV), clip_ratio_low=0.2, clip_ratio_high=0.2,
                                             loss_mask=None if mask is None else mask.to(DEV))
        loss.backward()
        wloss.backward()
        close(loss, wloss, 1e-7, 1e-5)
        close(dlp.grad, lp.grad, 1e-9, 2e-5)
        for k, v in wm.items():
            close(torch.tensor(float(metrics[k])), v.float(), 1e-6, 1e-5, k)


@pytest.mark.parametrize("group", [1, 4, 8])
def test_grpo_seq_adv(group):
    b = token_batch(79, 16, 37, 5)
    want = TO.grpo_reasoning_advantages(b["rewards"], b["loss_mask"], group)
    adv, ret = registry.calculate_adv_and_returns(task_type="reasoning", adv_type="grpo", rewards=b["rewards"].to(DEV),
                                                  loss_mask=b["loss_mask"].to(DEV), group_size=group)
    assert ret is None and adv.is_contiguous()
    close(adv, want, 1e-6, 1e-5)  # group mean/std: sequential f32 sums here, pairwise in torch


def test_reasoning_gae_through_the_registry():
    """adv_type='gae' for reasoning runs the reference's shaping (transposes + zero bootstrap row) around gae_scan."""
    from oracle import ppo_oracle as PO
    b = token_batch(80, 8, 21, 5)
    values = torch.randn(8, 21)
    pre = TO.preprocess_reasoning(b["rewards"], b["loss_mask"], "gae", values=values)
    wadv, wret = PO.gae_tb(pre["rewards"], pre["dones"], values=pre["values"], gamma=1.0, gae_lambda=0.95,
                           normalize_advantages=True, loss_mask=pre["loss_mask"])
    adv, ret = registry.calculate_adv_and_returns(
        task_type="reasoning", adv_type="gae", rewards=b["rewards"].to(DEV), loss_mask=b["loss_mask"].to(DEV),
        values=values.to(DEV), gamma=1.0, gae_lambda=0.95, normalize_advantages=True)
    close(adv, wadv.transpose(0, 1), 2e-5, 1e-5)
    close(ret, wret.transpose(0, 1), 1e-5, 1e-5)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_llm_vocab_properties(dtype):
    """Qwen-size vocabulary (151936), sizes the oracle would need minutes for: properties that hold exactly or to
    rounding -- probabilities sum to one (sum_v d_logits == 0 when only d_logprob flows), 0 <= H <= ln V,
    logprob <= 0, lse - logprob is the label's logit, a spot-checked row against the oracle, in-place == out-of-place."""
    V, N = 151936, 96
    g = torch.Generator(device=DEV).manual_seed(5)
    x = (torch.randn(N, V, device=DEV, generator=g) * 4).to(dtype)
    labels = torch.randint(0, V, (N,), device=DEV, generator=g)
    lp, ent, lse = token_ops.token_logprob_fwd(x, labels, 1.0, with_entropy=True)
    assert (lp <= 0).all() and (ent >= 0).all() and (ent <= torch.log(torch.tensor(float(V)))).all()
    xy = x.gather(1, labels[:, None])[:, 0].float()
    close(lse + lp, xy, 2e-5, 1e-6)
    rows = [0, 41, N - 1]
    # at 1.5e5 terms and |lse| ~ 20 the reference's own f32 evaluation carries ~1e-5 of rounding (every log p inherits
    # the rounding of lse); the yardstick is the same formula in f64, and the kernel must be at least as close to it
    # as the f32 oracle is (plus 1e-5 of slack)
    x64 = x[rows].cpu().double()
    elp = torch.log_softmax(x64, -1)
    went64 = -(elp.exp() * elp).sum(-1)
    wlp64 = elp.gather(1, labels[rows].cpu()[:, None])[:, 0]
    wlp, went, _ = oracle_f32(x[rows].cpu(), labels[rows].cpu())
    err_ref = max(float((wlp.double() - wlp64).abs().max()), float((went.double() - went64).abs().max()))
    close(lp[rows].double(), wlp64, err_ref + 1e-5, what="logprob vs f64")
    close(ent[rows].double(), went64, err_ref + 1e-5, what="entropy vs f64")
    print(f"[llm-vocab {dtype}] f32-oracle err vs f64 {err_ref:.2e}; kernel err "
          f"{float((lp[rows].cpu().double() - wlp64).abs().max()):.2e} / {float((ent[rows].cpu().double() - went64).abs().max()):.2e}")
    dlp = torch.randn(N, device=DEV, generator=g)
    d = token_ops.token_logprob_bwd(x, labels, lse, None, dlp, None)
    rowsum = d.float().sum(-1)
    scale = dlp.abs() * (1e-4 if dtype == torch.float32 else 2e-2)
    assert (rowsum.abs() <= scale + 1e-6).all(), float((rowsum.abs() - scale).max())
    x2 = x.clone()
    token_ops.token_logprob_bwd(x2, labels, lse, None, dlp, None, out=x2)
    assert torch.equal(x2, d)


# ---- packed / variable-length sequences: the unpack fused into the scoring kernel's stores ---------------------------------
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("bsz,prompt,resp,vocab,pad,temperature", [(6, 5, 9, 1000, 0, 1.0), (3, 4, 12, 151936, 7, 0.7), (9, 0, 6, 333, 0, 1.3),
                                                                 (1, 3, 3, 50, 2, 1.0)])
def test_packed_token_logprobs_vs_oracle(dtype, bsz, prompt, resp, vocab, pad, temperature):
    """rlx_token_logprob_fwd_packed + the gathered backward against the reference's arithmetic for a packed stream
    (oracle.token_oracle.unpack_logprobs / unpack_sequences, pinned to unpack_fsdp_logprobs / unpack_sequences of
    rlinf/hybrid_engines/fsdp/utils.py in tests/test_reference_reasoning_loop.py): log-probs land shifted right by one inside every
    row's window, the entropy unshifted, zeros where the reference pads, and the gradient w.r.t. the packed logits is the
    reference's autograd -- zero rows (prompt tokens that feed nothing, the padding of a fixed-length pack) included.  prompt 0: a
    sequence's first response log-prob comes from the LAST row of the sequence packed in front of it (or is the prepended zero),
    as the reference has it."""
    g = torch.Generator().manual_seed(bsz * 1000 + resp)
    plen = torch.randint(0, prompt + 1, (bsz,), generator=g)
    rlen = torch.randint(1, resp + 1, (bsz,), generator=g)
    S = prompt + resp
    idx_starts, idx_ends = (prompt - plen).tolist(), (prompt + rlen).tolist()
    L = sum(idx_ends) - sum(idx_starts) + pad
    ids = torch.randint(0, vocab, (1, L), generator=g)
    logits = (torch.randn(1, L, vocab, generator=g) * 3).to(dtype)
    eos = 2
    # oracle on the same stored logits: f32 arithmetic like the reference -- at the Qwen-size vocabulary in f64 (the f32 sums over
    # 1.5e5 terms carry ~1e-4 of rounding of their own: test_large_vocab_properties measures it), the kernel must match the f64 value
    x = (logits.float() / temperature if dtype == torch.float32 else (logits / temperature).float())
    x = (x.double() if vocab > 10000 else x).requires_grad_(True)
    want_lp = TO.unpack_logprobs(x, ids, idx_starts, idx_ends, S, eos)[:, -resp:]
    want_ent = TO.unpack_sequences(TO.entropy_from_logits(x), idx_starts, idx_ends, S, 0)[:, -resp:]
    d_lp, d_ent = torch.randn(bsz, resp, generator=g), torch.randn(bsz, resp, generator=g) * 0.1
    (want_lp * d_lp + want_ent * d_ent).sum().backward()
    want_dx = (x.grad / temperature).float()
    want_lp, want_ent = want_lp.float(), want_ent.float()
    dl = logits.to(DEV).requires_grad_(True)
    lp, ent = token_ops.packed_token_logprobs(dl, ids.to(DEV), idx_starts, idx_ends, max_seq_len_unpack=S, response_len=resp,
                                              eos_token_id=eos, temperature=temperature, with_entropy=True, round_outputs=False)
    assert lp.shape == (bsz, resp) and ent.shape == (bsz, resp)
    close(lp, want_lp, 2e-5, what="packed logprob")
    close(ent, want_ent, 2e-5, what="packed entropy")
    assert torch.equal(lp.cpu() == 0, want_lp.detach() == 0) and torch.equal(ent.cpu() == 0, want_ent.detach() == 0)  # the padding pattern
    (lp * d_lp.to(DEV) + ent * d_ent.to(DEV)).sum().backward()
    got_dx = dl.grad.float().cpu()
    if dtype == torch.float32:
        close(got_dx, want_dx, 1e-6, 2e-5, "packed d_logits")
    else:
        ulp = want_dx.abs().amax(dim=-1, keepdim=True) * 2.0 ** -7 + 1e-6
        assert ((got_dx - want_dx).abs() <= ulp).all()
    zero_rows = want_dx.abs().amax(dim=-1) == 0
    assert bool(zero_rows.any()) or prompt == 0
    assert not got_dx[zero_rows].any()                      # rows that feed nothing: exact zeros (written without being read)
    # log-probs only (no entropy requested): the same values
    lp2, none = token_ops.packed_token_logprobs(logits.to(DEV), ids.to(DEV), idx_starts, idx_ends, max_seq_len_unpack=S, response_len=resp,
                                                eos_token_id=eos, temperature=temperature)
    assert none is None and torch.equal(lp2, lp.detach())


# ---- K2: categorical action sampling ------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("K", [256, 100, 1000])
@pytest.mark.parametrize("temperature,top_k", [(1.0, -1), (0.7, 50), (1.6, 1), (1.0, 5000)])
def test_categorical_sample_vs_oracle(dtype, K, temperature, top_k):
    g = torch.Generator().manual_seed(K)
    B, A, V, pad = 96, 14, K + 200, 64   # 1344 draws per case
    full = (torch.randn(B, A, V, generator=g) * 2).to(dtype)
    window = full[..., V - pad - K:V - pad]
    q = torch.empty(B, A, K, dtype=dtype).exponential_(1, generator=g)
    centers = torch.linspace(-1, 1, K - 1)
    wtok, wlp, processed, wact = TO.categorical_sample(window, q, temperature, top_k, bin_centers=centers)
    dwin = full.to(DEV)[..., V - pad - K:V - pad]
    # (softmax_lanes: the oracle runs torch.softmax on THIS host, so the kernel replays this host's summation order; the
    # product default is a fixed 16 -- test_categorical_default_lane_count_is_fixed)
    tok, lp, act = token_ops.categorical_sample(dwin, q.to(DEV), temperature=temperature, top_k=top_k,
                                                bin_centers=centers.to(DEV), softmax_lanes=token_ops.reference_softmax_lanes())
    # north_star: "bit-exact for action indices".  The kernel replays the reference's CPU softmax operation for operation (Sleef's
    # expf, the per-SIMD-lane row sum of this host's torch build, e * (1 / sum), p / q, first-index argmax): no tie allowance.
    assert torch.equal(tok.cpu(), wtok), (int((tok.cpu() != wtok).sum()), tok.numel())
    same = tok.cpu() == wtok
    if dtype == torch.float32:
        close(lp[same.to(DEV)], wlp[same], 2e-5, what="logprob")
    else:
        exact = TO.logprobs_from_logits(processed.float(), wtok).bfloat16().float()
        assert ((lp.cpu() - exact).abs()[same] <= exact.abs()[same] * 2.0 ** -7 + 1e-6).all()
    assert torch.equal(act.cpu()[same], wact[same])
    # argmax mode: bit-exact including the first-index tie rule (bf16 logits tie often)
    tok0, lp0, _ = token_ops.categorical_sample(dwin, None, temperature=temperature, top_k=top_k)
    assert torch.equal(tok0.cpu(), window.argmax(-1))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_categorical_sample_bit_exact_over_many_rows(dtype):
    """200 000 draws over peaked and flat 256-bin rows (the OpenVLA-OFT action head's shape), temperature and top-k on: every
    sampled index equals torch.multinomial's race on the host (argmax(softmax(x) / q)), i.e. the near ties a 1-ulp difference in
    one exp or one addition would flip all come out the reference's way."""
    g = torch.Generator().manual_seed(77)
    n, K = 200_000, 256
    x = (torch.randn(n, K, generator=g) * torch.rand(n, 1, generator=g) * 4).to(dtype)
    q = torch.empty(n, K, dtype=dtype).exponential_(1, generator=g)
    wtok, _, _, _ = TO.categorical_sample(x, q, 0.8, 40)
    tok, _, _ = token_ops.categorical_sample(x.to(DEV), q.to(DEV), temperature=0.8, top_k=40,
                                             softmax_lanes=token_ops.reference_softmax_lanes())
    assert torch.equal(tok.cpu(), wtok), int((tok.cpu() != wtok).sum())


def test_categorical_default_lane_count_is_fixed(monkeypatch):
    """Without an explicit lane count the sampler uses 16 on every host (same seed -> same tokens on every node of a job);
    RLX_SOFTMAX_LANES=host / 8 / 16 is the opt-in to a particular host's order."""
    g = torch.Generator().manual_seed(5)
    x = (torch.randn(4096, 256, generator=g) * 3).to(DEV)
    q = torch.empty(4096, 256).exponential_(1, generator=g).to(DEV)
    monkeypatch.delenv("RLX_SOFTMAX_LANES", raising=False)
    assert token_ops.default_softmax_lanes() == 16
    base = token_ops.categorical_sample(x, q)[0]
    assert torch.equal(base, token_ops.categorical_sample(x, q, softmax_lanes=16)[0])
    monkeypatch.setenv("RLX_SOFTMAX_LANES", "8")
    assert torch.equal(token_ops.categorical_sample(x, q)[0], token_ops.categorical_sample(x, q, softmax_lanes=8)[0])
    monkeypatch.setenv("RLX_SOFTMAX_LANES", "host")
    assert token_ops.default_softmax_lanes() == token_ops.reference_softmax_lanes()
    monkeypatch.setenv("RLX_SOFTMAX_LANES", "12")
    with pytest.raises(token_ops.RlxError):
        token_ops.default_softmax_lanes()


def test_categorical_sample_follows_the_distribution():
    """10^5 draws from one 256-bin distribution: empirical frequencies within 5 sigma of softmax(x / T) under top-k."""
    from rlinf_amd.models.embodiment.openvla_oft import DiscreteActionHead

    g = torch.Generator(device=DEV).manual_seed(3)
    head = DiscreteActionHead(n_action_bins=256, pad_to_multiple_of=64, action_dim=7, num_action_chunks=8)
    row = torch.randn(32064, device=DEV, generator=g) * 1.5
    n = 7 * 8 * 2000
    logits = row.expand(2000, 56, 32064)
    actions, tokens, logprobs = head.predict(logits, do_sample=True, temperature=0.8, top_k=40, generator=g)
    assert actions.shape == (2000 * 8, 7) and tokens.shape == (2000, 56)
    x = head.action_logits(row[None, None])[0, 0].cpu() / 0.8
    kth = torch.topk(x, 40)[0][-1]
    p = torch.softmax(x.masked_fill(x < kth, float("-inf")), -1)
    freq = torch.bincount(tokens.reshape(-1).cpu(), minlength=256).float() / n
    sigma = (p * (1 - p) / n).sqrt()
    assert ((freq - p).abs() <= 5 * sigma + 1e-6).all()
    assert (freq[p == 0] == 0).all()
    close(logprobs.reshape(-1), torch.log(p)[tokens.reshape(-1).cpu()], 2e-5)
    centers = head.bin_centers
    assert torch.equal(actions.reshape(-1).cpu(), centers[(256 - tokens.reshape(-1).cpu() - 1).clamp(0, 254)])
    # training-time recomputation over the same window is differentiable and agrees with the rollout log-probs
    lg = logits[:4].clone().requires_grad_(True)
    out = head.logprobs_and_entropy(lg, tokens[:4], compute_entropy=True, temperature=1.0)
    out["logprobs"].sum().backward()
    assert lg.grad is not None and float(lg.grad[..., :32064 - 64 - 256].abs().max()) == 0.0


@pytest.mark.parametrize("seq", [1, 5, 300, 1024, 1500, 2048, 8192, 8193, 20000])
@pytest.mark.parametrize("gl", [(1.0, 0.95), (0.99, 0.9), (1.0, 1.0)])
def test_gae_seq_vs_oracle(seq, gl):
    """Reasoning GAE along the contiguous axis against the reference's shaping + its sequential loop (the oracle's
    gae_tb on [seq, bsz]); one / several LDS segments, chunked lanes, the reward on the row's last position."""
    from oracle import ppo_oracle as PO
    gamma, lam = gl
    bsz = 6 if seq > 1000 else 32
    g = torch.Generator().manual_seed(seq)
    values = torch.randn(bsz, seq, generator=g)
    rewards = torch.randn(bsz, generator=g)
    mask = torch.rand(bsz, seq, generator=g) < 0.8
    pre = TO.preprocess_reasoning(rewards, mask, "gae", values=values)
    wadv, wret = PO.gae_tb(pre["rewards"], pre["dones"], values=pre["values"], gamma=gamma, gae_lambda=lam,
                           normalize_advantages=False)
    adv, ret = token_ops.gae_seq(values.to(DEV), rewards.to(DEV), gamma, lam)
    # both sides run the recurrence in f32; the reference's chain of seq dependent additions accumulates ~sqrt(seq)
    # roundings (with gamma*lambda = 1 nothing decays), the kernel's chunked replay accumulates them in another order
    scale = float(wret.abs().max()) + 1.0
    atol = (4e-6 + 2e-7 * seq ** 0.5) * scale
    close(ret, wret.transpose(0, 1), atol, 1e-5, "returns")
    close(adv, wadv.transpose(0, 1), atol, 1e-5, "advantages")
    out = registry.calculate_adv_and_returns(task_type="reasoning", adv_type="gae", rewards=rewards.to(DEV),
                                             loss_mask=mask.to(DEV), values=values.to(DEV), gamma=gamma, gae_lambda=lam,
                                             normalize_advantages=True)
    nadv, _ = PO.gae_tb(pre["rewards"], pre["dones"], values=pre["values"], gamma=gamma, gae_lambda=lam,
                        normalize_advantages=True, loss_mask=pre["loss_mask"])
    if int(mask.sum()) > 1:
        close(out[0], nadv.transpose(0, 1), 2e-5 + 10 * atol, 2e-5, "normalised advantages")
    assert out[0].is_contiguous() and out[1].is_contiguous()


@pytest.mark.parametrize("variant", ["0", "1", "2"])
def test_gae_seq_lookback_many_rows_and_nan_rows(variant, monkeypatch):
    """Long rows, all kernels: the default walk (one 256-lane workgroup per row, 4096-token segments), the round-1 shape of it
    (RLX_GAESEQ_VARIANT=1: 512 lanes, 8192-token segments) and RLX_GAESEQ_VARIANT=2, one segment per workgroup with
    a decoupled look-back for the carry (thousands of workgroups in flight, neighbours racing) -- against the sequential
    oracle; a NaN row must come back NaN, and come back (the look-back never waits on a payload's value, only on its flag)."""
    from oracle import ppo_oracle as PO
    if variant != "0":
        from conftest import need_dev_variants
        need_dev_variants(f"gae_seq variant {variant}")
    monkeypatch.setenv("RLX_GAESEQ_VARIANT", variant)
    bsz, seq = 1500, 6144 + 7
    g = torch.Generator().manual_seed(9)
    values = torch.randn(bsz, seq, generator=g)
    values[3, 4000] = float("nan")
    rewards = torch.randn(bsz, generator=g)
    mask = torch.ones(bsz, seq, dtype=torch.bool)
    pre = TO.preprocess_reasoning(rewards, mask, "gae", values=values)
    wadv, wret = PO.gae_tb(pre["rewards"], pre["dones"], values=pre["values"], gamma=1.0, gae_lambda=0.95, normalize_advantages=False)
    for _ in range(3):  # repeated launches reuse the workspace: the slots must be re-armed every time
        adv, ret = token_ops.gae_seq(values.to(DEV), rewards.to(DEV), 1.0, 0.95)
    ok = torch.ones(bsz, dtype=torch.bool)
    ok[3] = False
    scale = float(wret.transpose(0, 1)[ok].abs().max()) + 1.0
    atol = (4e-6 + 2e-7 * seq ** 0.5) * scale
    close(ret[ok.to(DEV)], wret.transpose(0, 1)[ok], atol, 1e-5, "returns")
    close(adv[ok.to(DEV)], wadv.transpose(0, 1)[ok], atol, 1e-5, "advantages")
    assert torch.isnan(ret[3, :4001]).all() and torch.isfinite(ret[3, 4002 + 200:]).all()  # NaN flows towards t = 0 only


@pytest.mark.parametrize("bsz,seq", [(7, 3000), (5, 4100), (3, 32768), (9, 260), (1500, 6144 + 8), (2, 32772)])
def test_gae_seq_register_kernel_against_the_lds_kernels_and_the_oracle(bsz, seq, monkeypatch):
    """Rows of 16-byte aligned length take gae_seq_reg_kernel (one wave per 2048 tokens, DPP suffix composition, one barrier):
    two / four / sixteen waves per row, an odd row count with two rows per workgroup (a dead wave), a group that only the row's
    first lanes fill, a row longer than sixteen waves cover (falls back to the LDS walk), thousands of rows with a NaN in one.
    Against the LDS kernels (RLX_GAESEQ_REG=0) to rounding, and against the sequential oracle."""
    from oracle import ppo_oracle as PO
    g = torch.Generator().manual_seed(seq + bsz)
    values = torch.randn(bsz, seq, generator=g)
    rewards = torch.randn(bsz, generator=g)
    nan_row = bsz > 1000
    if nan_row:
        values[3, 4000] = float("nan")
    for forced in ("1", "2", None):  # ordinary / non-temporal accesses / the size rule
        if forced is None:
            monkeypatch.delenv("RLX_GAESEQ_REG", raising=False)
        else:
            monkeypatch.setenv("RLX_GAESEQ_REG", forced)
        adv, ret = token_ops.gae_seq(values.to(DEV), rewards.to(DEV), 0.99, 0.95)
        monkeypatch.setenv("RLX_GAESEQ_REG", "0")
        adv0, ret0 = token_ops.gae_seq(values.to(DEV), rewards.to(DEV), 0.99, 0.95)
        ok = torch.ones(bsz, dtype=torch.bool, device=DEV)
        if nan_row:
            ok[3] = False
            assert torch.isnan(ret[3, :4001]).all() and torch.isfinite(ret[3, 4001:]).all()  # NaN flows towards t = 0 only
        close(ret[ok], ret0[ok], 2e-5, 1e-5, "returns vs the LDS kernels")
        close(adv[ok], adv0[ok], 2e-5, 1e-5, "advantages vs the LDS kernels")
        assert torch.equal(adv[ok], ret[ok] - values.to(DEV)[ok])  # the reference's own last operation (advantages.py:79)
    if bsz * seq <= 200000:
        pre = TO.preprocess_reasoning(rewards, torch.ones(bsz, seq, dtype=torch.bool), "gae", values=values)
        wadv, wret = PO.gae_tb(pre["rewards"], pre["dones"], values=pre["values"], gamma=0.99, gae_lambda=0.95, normalize_advantages=False)
        atol = (4e-6 + 2e-7 * seq ** 0.5) * (float(wret.abs().max()) + 1.0)
        close(ret, wret.transpose(0, 1), atol, 1e-5, "returns")
        close(adv, wadv.transpose(0, 1), atol, 1e-5, "advantages")


def test_empty_and_degenerate_inputs():
    """Zero tokens / sequences, a one-entry vocabulary, a one-bin categorical head, an empty tensor inside a synced state
    dict: defined results, no launches with zero-sized grids."""
    from rlinf_amd.hybrid_engines.weight_syncer import EmptyWeightPatch, PatchWeightSyncer

    x = torch.empty(0, 100, device=DEV)
    lp, ent, lse = token_ops.token_logprob_fwd(x, torch.empty(0, dtype=torch.int64, device=DEV), with_entropy=True)
    assert lp.shape == ent.shape == lse.shape == (0,)
    d = token_ops.token_logprob_bwd(x, torch.empty(0, dtype=torch.int64, device=DEV), lse, None, lp, None)
    assert d.shape == (0, 100)
    one = torch.randn(5, 1, device=DEV)
    lp, ent, _ = token_ops.token_logprob_fwd(one, torch.zeros(5, dtype=torch.int64, device=DEV), with_entropy=True)
    assert float(lp.abs().max()) < 1e-6 and float(ent.abs().max()) < 1e-6  # a certain outcome (exp2/log2 round trip)
    tok, clp, _ = token_ops.categorical_sample(one, torch.ones(5, 1, device=DEV))
    assert int(tok.abs().max()) == 0 and float(clp.abs().max()) < 1e-6
    adv = token_ops.grpo_seq_adv(torch.empty(0, device=DEV), torch.empty(0, 7, dtype=torch.bool, device=DEV), 4)
    assert adv.shape == (0, 7)
    a, r = token_ops.gae_seq(torch.empty(0, 9, device=DEV), torch.empty(0, device=DEV))
    assert a.shape == r.shape == (0, 9)
    a, r = token_ops.gae_seq(torch.randn(3, 1, device=DEV), torch.ones(3, device=DEV), 0.9, 0.9)
    assert a.shape == (3, 1) and torch.allclose(r, torch.ones(3, 1, device=DEV))  # one token: return = reward
    state = {"w": torch.randn(4, 4, device=DEV), "empty": torch.empty(0, 3, device=DEV)}
    q = []
    rx, tx = PatchWeightSyncer(), PatchWeightSyncer()
    rx.init_receiver({k: v.clone() for k, v in state.items()}, q.pop, q.append)
    tx.init_sender(state, ["w", "empty"], q.append, lambda: q.pop(0))
    assert isinstance(tx.create_patch(state, 1), EmptyWeightPatch)
    state["w"][1, 2] += 1
    p = tx.create_patch(state, 2)
    assert p.nnz_per_tensor.tolist() == [1] and p.ordinals.tolist() == [0]


# ---- Reinforce++ on reasoning batches (rlx_reinpp_seq_adv) ----------------------------------------------------------------
def _reinpp_close(got, want, seq, what=""):
    # the reference sums the per-token rewards sequentially in f32 (cumsum), the kernel in f64 tiles: ~sqrt(seq) roundings
    scale = float(want.abs().max()) + 1.0
    close(got, want, (4e-6 + 3e-7 * seq ** 0.5) * scale, 2e-5, what)


def test_reinpp_matches_reference_fixture():
    from oracle.make_golden import reinpp_batch
    for case in torch.load(os.path.join(GOLDEN_DIR, "reinpp.pt"), weights_only=False):
        p = case["params"]
        rewards, mask, lp, rlp = reinpp_batch(**p)
        adv, ret = registry.calculate_adv_and_returns(
            task_type="reasoning", adv_type="reinpp", rewards=rewards.to(DEV), loss_mask=mask.to(DEV), group_size=2,
            kl_beta=p["kl_beta"], logprob=lp.to(DEV), ref_logprob=rlp.to(DEV), kl_penalty_type=p["kl"], use_reinpp_baseline=False)
        assert ret is None and adv.is_cuda and adv.is_contiguous()
        _reinpp_close(adv, case["advantages"], p["seq"], str(p))


@pytest.mark.parametrize("seq", [1, 3, 64, 512, 1023, 1024, 1025, 1028, 2048, 3000, 4100])
@pytest.mark.parametrize("kl,beta", [("", 0.0), ("kl", 0.02), ("abs", 0.1), ("mse", 0.1), ("low_var_kl", 0.001)])
@pytest.mark.parametrize("masks", ["prefix", "ragged"])
def test_reinpp_vs_oracle(seq, kl, beta, masks):
    from oracle.make_golden import reinpp_batch
    bsz = 9
    rewards, mask, lp, rlp = reinpp_batch(1000 + seq, bsz, seq, masks)
    want = TO.reinpp_reasoning_advantages(rewards.clone(), mask, 3, False, beta, lp, rlp, kl)
    got = token_ops.reinpp_seq_adv(rewards.to(DEV), mask.to(DEV), lp.to(DEV), rlp.to(DEV), beta, kl or None)
    _reinpp_close(got, want, seq)
    # the reference's own signature ([1, B] rewards, [L, B] tensors) through the registered function
    fn = registry.get_adv_and_returns("reinpp")
    adv, ret = fn(rewards=rewards.unsqueeze(0).to(DEV), loss_mask=mask.t().to(DEV), group_size=3, kl_beta=beta,
                  logprob=lp.t().to(DEV), ref_logprob=rlp.t().to(DEV), kl_penalty_type=kl)
    assert ret is None and tuple(adv.shape) == (seq, bsz)
    _reinpp_close(adv.t(), want, seq)


@pytest.mark.parametrize("bsz,seq", [(7, 3000), (5, 4100), (3, 20000), (9, 260), (600, 2048), (2, 32772)])
@pytest.mark.parametrize("masks", ["prefix", "ragged"])
def test_reinpp_register_kernel_against_the_tile_walk_and_the_oracle(bsz, seq, masks, monkeypatch):
    """Rows of 16-byte aligned length take reinpp_returns_reg_kernel (one wave per 1024 / 2048 tokens, f64 suffix sums on DPP row
    shifts, one barrier): one to sixteen waves per row, dead waves of the last workgroup, a group only the row's first lanes fill,
    the mirrored-mask search, a row longer than sixteen waves cover (tile walk).  Both group counts per wave against the tile walk
    (RLX_REINPP_REG=0) -- the f64 sums round to the same f32 returns, so the normalised output is compared tightly -- and the oracle."""
    from oracle.make_golden import reinpp_batch
    rewards, mask, lp, rlp = reinpp_batch(77 + seq, bsz, seq, masks)
    args = (rewards.to(DEV), mask.to(DEV), lp.to(DEV), rlp.to(DEV), 0.02, "low_var_kl")
    monkeypatch.setenv("RLX_REINPP_REG", "0")
    walk = token_ops.reinpp_seq_adv(*args)
    for forced in ("4", "8", None):
        if forced is None:
            monkeypatch.delenv("RLX_REINPP_REG", raising=False)
        else:
            monkeypatch.setenv("RLX_REINPP_REG", forced)
        got = token_ops.reinpp_seq_adv(*args)
        close(got, walk, 2e-6, 2e-6, f"register kernel ({forced}) vs the tile walk")
    if bsz * seq <= 100000:
        want = TO.reinpp_reasoning_advantages(rewards.clone(), mask, 3, False, 0.02, lp, rlp, "low_var_kl")
        _reinpp_close(got, want, seq)


def test_reinpp_reward_position_follows_the_mirrored_mask():
    """The reference reads the 'last valid token' off the batch-flipped mask: sequence b's reward goes to seq-1 minus the
    first-True index of sequence bsz-1-b.  With kl_beta = 0 the return-to-go is the reward up to that position and 0 after."""
    bsz, seq = 4, 10
    mask = torch.ones(bsz, seq, dtype=torch.bool)
    mask[3, :4] = False   # mirrored partner of sequence 0 starts at 4 -> sequence 0's reward at 9 - 4 = 5
    mask[1, :] = False    # mirrored partner of sequence 2 is empty -> argmax 0 -> position 9
    rewards = torch.tensor([1.0, 2.0, 3.0, 4.0])
    want = TO.reinpp_reasoning_advantages(rewards.clone(), mask, 2)
    got = token_ops.reinpp_seq_adv(rewards.to(DEV), mask.to(DEV))
    _reinpp_close(got, want, seq)
    row0 = got[0].cpu()
    assert torch.allclose(row0[:6], row0[0].expand(6)) and torch.allclose(row0[6:], row0[6].expand(4)) and row0[0] != row0[6]


def test_reinpp_baseline_and_bad_kl_fail_like_the_reference():
    mask = torch.ones(4, 6, dtype=torch.bool, device=DEV)
    r = torch.randn(4, device=DEV)
    with pytest.raises(IndexError, match="Dimension out of range"):
        registry.calculate_adv_and_returns(task_type="reasoning", adv_type="reinpp", rewards=r, loss_mask=mask, group_size=2,
                                           use_reinpp_baseline=True)
    with pytest.raises(NotImplementedError):
        registry.calculate_adv_and_returns(task_type="reasoning", adv_type="reinpp", rewards=r, loss_mask=mask, group_size=2,
                                           kl_beta=0.1, logprob=torch.zeros(4, 6, device=DEV), ref_logprob=torch.zeros(4, 6, device=DEV),
                                           kl_penalty_type="full")


def test_reinpp_llm_size_properties():
    """512 sequences x 8192 tokens: without a KL term every position of a row up to the reward carries the same value --
    the normalised reward -- and the masked moments of the output are 0 / 1; with one, the kernel agrees with the f64
    evaluation of the same formulas better than the reference's f32 cumsum does."""
    bsz, seq = 512, 8192
    g = torch.Generator().manual_seed(5)
    rewards = torch.randn(bsz, generator=g)
    lens = torch.randint(1, seq + 1, (bsz,), generator=g)
    mask = torch.arange(seq)[None, :] < lens[:, None]
    adv = token_ops.reinpp_seq_adv(rewards.to(DEV), mask.to(DEV)).cpu()
    assert torch.equal(adv, adv[:, :1].expand(-1, seq))  # reward at seq-1: constant rows
    sel = adv[mask].double()
    assert abs(float(sel.mean())) < 1e-5 and abs(float(sel.var(unbiased=False)) - 1.0) < 1e-4
    w = mask.sum(1).double()
    mu = float((rewards.double() * w).sum() / w.sum())
    sd = float(((rewards.double() - mu) ** 2 * w).sum() / w.sum()) ** 0.5
    close(adv[:, 0], ((rewards.double() - mu) / sd).float(), 2e-5, 2e-5)
    lp = -torch.rand(bsz, seq, generator=g) * 3
    rlp = lp + 0.3 * torch.randn(bsz, seq, generator=g)
    got = token_ops.reinpp_seq_adv(rewards.to(DEV), mask.to(DEV), lp.to(DEV), rlp.to(DEV), 0.001, "low_var_kl").cpu().double()
    kl = torch.clamp(torch.exp(torch.clamp((rlp - lp), -20, 20)) - torch.clamp((rlp - lp), -20, 20) - 1, -10, 10).double()
    r = -(torch.tensor(0.001, dtype=torch.float32).double()) * kl
    r[:, -1] += rewards.double()
    ret = torch.flip(torch.cumsum(torch.flip(r, [1]), 1), [1])
    m = mask.double()
    mean = (ret * m).sum() / m.sum()
    var = (((ret - mean) ** 2) * m).sum() / m.sum()
    exact = (ret - mean) / var.sqrt()
    ref32 = TO.reinpp_reasoning_advantages(rewards.clone(), mask, 2, False, 0.001, lp, rlp, "low_var_kl").double()
    err_kernel, err_ref = float((got - exact).abs().max()), float((ref32 - exact).abs().max())
    assert err_kernel < 2e-5 and err_kernel <= err_ref + 1e-6, (err_kernel, err_ref)


@pytest.mark.parametrize("adv_type", ["grpo", "reinpp"])
def test_reasoning_advantage_stage(adv_type):
    """FSDPActor.compute_advantages_and_returns + the normalize_advantages step of run_training (fsdp_actor_worker.py:
    900-907,941-978) as TokenLearnerStep exposes them: mask = the response window, log-probs = recomputed | rollout."""
    from oracle import ppo_oracle as PO
    resp, bsz = 21, 8
    b = token_batch(95, bsz, resp, 5)
    pad = torch.zeros(bsz, 4, dtype=torch.bool)
    batch = {"response_mask": torch.cat([pad, b["loss_mask"]], dim=1).to(DEV), "rewards": b["rewards"].to(DEV),
             "rollout_logprobs": b["old_logprobs"].to(DEV), "ref_logprobs": b["ref_logprobs"].to(DEV)}
    step = TokenLearnerStep(response_len=resp, adv_type=adv_type, group_size=4, reinpp_kl_beta=0.01, kl_penalty_type="low_var_kl",
                            normalize_advantages=True)
    step.compute_advantages_and_returns(batch)
    if adv_type == "grpo":
        want = TO.grpo_reasoning_advantages(b["rewards"], b["loss_mask"], 4)
    else:
        want = TO.reinpp_reasoning_advantages(b["rewards"].clone(), b["loss_mask"], 4, False, 0.01, b["old_logprobs"],
                                              b["ref_logprobs"], "low_var_kl")
    close(batch["advantages"], want, 2e-5, 2e-5, "advantages")
    given = batch["advantages"]
    assert step.compute_advantages_and_returns(batch)["advantages"] is given  # already there: left alone
    step.normalize_batch_advantages(batch)
    close(batch["advantages"], PO.masked_normalization(want, b["loss_mask"]), 5e-5, 5e-5, "normalised")

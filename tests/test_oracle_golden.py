"""The CPU oracle against the committed reference outputs (tests/golden/*.pt).  Runs everywhere.

The fixtures were produced by oracle/make_golden.py from the REAL reference; the restatement uses
the same torch CPU ops in the same order, so everything is compared bit-for-bit.
"""

import os

import pytest
import torch

from conftest import GOLDEN_DIR
from oracle import ppo_oracle as O
from oracle.make_golden import perturbation


def _load(name):
    return torch.load(os.path.join(GOLDEN_DIR, name), weights_only=False)


def _eq(a, b):
    assert a.shape == b.shape and a.dtype == b.dtype, (a.shape, b.shape, a.dtype, b.dtype)
    assert torch.equal(a, b)


def test_advantages_golden():
    for case in _load("advantages.pt"):
        p = case["params"]
        lms = case["loss_mask_sum"]
        if lms is not None:
            lms = lms.expand_as(case["loss_mask"])
        out = O.embodied_adv_and_returns(
            adv_type=p["adv_type"], rewards=case["rewards"], dones=case["dones"],
            values=case["values"] if p["adv_type"] == "gae" else None, loss_mask=case["loss_mask"],
            loss_mask_sum=lms, gamma=p["gamma"], gae_lambda=p["gae_lambda"], group_size=p["group_size"],
            reward_type=p["reward_type"], normalize_advantages=p["normalize_advantages"])
        _eq(out["advantages"].contiguous(), case["advantages"])
        if case["returns"] is not None:
            _eq(out["returns"].contiguous(), case["returns"])


def test_loss_mask_golden():
    for case in _load("loss_mask.pt"):
        m, s = O.loss_mask_from_dones(case["dones"])
        _eq(m.contiguous(), case["loss_mask"])
        _eq(s[0].contiguous(), case["loss_mask_sum_row"])


def test_losses_golden():
    for case in _load("losses.pt"):
        p = dict(case["params"])
        lp = case["logprobs"].clone().requires_grad_(True)
        v = case["values"].clone().requires_grad_(True)
        shaped = O.shape_loss_inputs(lp, case["old_logprobs"], case["advantages"], p.pop("logprob_type"),
                                     p.pop("action_dim"), loss_mask=case["loss_mask"],
                                     loss_mask_sum=case["loss_mask_sum"], values=v,
                                     prev_values=case["prev_values"], returns=case["returns"])
        p.pop("masked"), p.pop("variant")
        loss, metrics = O.ppo_actor_critic_loss(**p, **shaped)
        g_lp, g_v = torch.autograd.grad(loss, [lp, v], allow_unused=True)
        _eq(loss.detach(), case["loss"])
        if case["grad_logprobs"] is None:
            assert g_lp is None
        else:
            _eq(g_lp, case["grad_logprobs"])
        _eq(g_v, case["grad_values"])
        for k in ("actor/policy_loss", "actor/approx_kl", "actor/clip_fraction", "actor/ratio",
                  "critic/value_loss", "critic/value_clip_ratio"):
            assert float(metrics[k]) == case["metrics"][k], k


def _policy_from(sd):
    pol = O.OracleMLPPolicy(42, 8, 1)
    pol.load_state_dict(sd, strict=True)
    return pol


def test_policy_golden():
    G = _load("policy.pt")
    pol = _policy_from(G["state_dict"])
    assert [n for n, _ in pol.named_parameters()] == G["param_names"]
    a, lp, v = pol.act(G["states"], eps=G["eps"], mode="train")
    _eq(a, G["action"])
    _eq(lp, G["prev_logprobs"])
    _eq(v, G["prev_values"])
    a, lp, _ = pol.act(G["states"], mode="eval")
    _eq(a, G["eval_action"])
    _eq(lp, G["eval_logprobs"])
    with torch.no_grad():
        for p in pol.parameters():
            p.add_(perturbation(p.shape))
    out = pol.evaluate(G["states"], G["action"])
    _eq(out["logprobs"].detach(), G["train_logprobs"])
    _eq(out["entropy"].detach(), G["train_entropy"])
    _eq(out["values"].detach(), G["train_values"])
    opt = O.build_adamw(pol)
    mb = dict(states=G["states"], action=G["action"], prev_logprobs=G["prev_logprobs"],
              advantages=G["advantages"], prev_values=G["prev_values"], returns=G["returns"])
    # replicate the step manually to be able to inspect gradients before the update
    m = O.ppo_minibatch_step(pol, opt, mb)
    assert float(m["actor/grad_norm"]) == float(G["grad_norm"])
    for n, p in pol.named_parameters():
        _eq(p.detach().reshape(-1)[::16], G["params_after_step_stride16"][n])


def test_shuffle_golden():
    G = _load("shuffle.pt")
    out = O.flatten_and_shuffle(G["batch"], G["perm"])
    for k in ("rewards", "dones", "prev_values", "prev_logprobs"):
        _eq(out[k], G["out"][k])
    for k in ("states", "action"):
        _eq(out["forward_inputs"][k], G["out"]["forward_inputs"][k])


def test_token_path_golden():
    """oracle/token_oracle.py against the reference's outputs for the reasoning micro-batch (t1-t6)."""
    from oracle import token_oracle as TO
    from oracle.make_golden import token_batch

    for case in _load("token_path.pt"):
        p = case["params"]
        b = token_batch(p["seed"], p["bsz"], p["seq"], p["vocab"], zero_first=p["zero_first"])
        for tag, dt in (("f32", torch.float32), ("bf16", torch.bfloat16)):
            want = case["out"][tag]
            logits = b["logits"].to(dt).clone().requires_grad_(True)
            loss, metrics, logprobs, entropy = TO.reasoning_micro_batch_loss(
                logits, b["labels"], b["old_logprobs"], b["advantages"], b["loss_mask"], temperature=p["temperature"],
                loss_agg=p["loss_agg"], clip_ratio_low=p["clip_ratio_low"], clip_ratio_high=p["clip_ratio_high"],
                clip_ratio_c=p["clip_ratio_c"], clip_log_ratio_min=p["clip_log_ratio_min"],
                clip_log_ratio_max=p["clip_log_ratio_max"], calculate_entropy=True, entropy_bonus=p["entropy_bonus"],
                ref_logprobs=b["ref_logprobs"], kl_beta=p["kl_beta"], kl_penalty_type=p["kl_penalty_type"],
                gradient_accumulation=p["gradient_accumulation"])
            loss.backward()
            _eq(logprobs.detach(), want["logprobs"])
            _eq(entropy.detach(), want["entropy"])
            _eq(metrics["actor/final_loss"], want["final_loss"])
            _eq(metrics["actor/entropy_loss"], want["entropy_loss"])
            _eq(metrics["actor/kl_loss"], want["kl_loss"])
            _eq(logits.grad, want["d_logits"])
            assert set(k for k in metrics if k not in ("actor/final_loss", "actor/entropy_loss", "actor/kl_loss")) == \
                set(want["metrics"])
            for k, v in want["metrics"].items():
                _eq(metrics[k], v)
        _eq(TO.grpo_reasoning_advantages(b["rewards"], b["loss_mask"], p["group_size"]), case["grpo_advantages"])
        for k, v in case["kl_terms"].items():
            _eq(TO.kl_penalty(b["ref_logprobs"], b["old_logprobs"], k), v)


def test_reinpp_golden():
    """Reinforce++ on reasoning batches (reward placement as the reference computes it, KL penalty, flip-cumsum-flip,
    masked normalisation) against the reference's outputs, bit for bit."""
    from oracle import token_oracle as TO
    from oracle.make_golden import REINPP_GRID, reinpp_batch

    cases = _load("reinpp.pt")
    assert [c["params"] for c in cases] == [dict(p) for p in REINPP_GRID]
    for case in cases:
        p = case["params"]
        rewards, mask, lp, rlp = reinpp_batch(**p)
        _eq(TO.reinpp_reasoning_advantages(rewards, mask, 2, False, p["kl_beta"], lp, rlp, p["kl"]), case["advantages"])
    with pytest.raises(IndexError):
        TO.reinpp_reasoning_advantages(rewards, mask, 2, True)

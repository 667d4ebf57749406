"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): the reference's MLP policy arithmetic with the OPERAND ROUNDINGS of the bf16
launches written out -- what `rlx_ppo_step(..., bf16)` / `rlx_mlp_rollout_step` (bf16 tiles) compute, as real-number arithmetic.

The reference runs `MLPPolicy.default_forward` (rlinf/models/embodiment/mlp_policy.py:202-236) in whatever precision the model is
given; BASELINE.json configs[1] names bf16.  "The oracle under torch.autocast" (tests/test_gpu_fused_step.py) bounds the bf16
launches from outside: any two bf16 implementations differ by percents because their log-probs land on different sides of PPO's
clip boundaries.  This module removes that freedom: it states WHERE the launches round --

  * operands of the three hidden layers: x, h1, h2 and W1..W3 as bf16 (round to nearest even), products exact, sums wide;
    z + bias, tanh wide; the activation rounded to bf16 ONCE -- that value feeds the next layer, the head, and 1 - h^2 of the
    backward sweep (csrc/ppo_step_bf16.hip epilogue_tanh_b, dtanh_b);
  * the heads keep f32 weights (three exact bf16 planes on the matrix pipe): h3(bf16) . W4 exact, the loss element math in f32;
  * backward: dZ_l = (dH_l) * (1 - h_l^2) rounded to bf16; dH_{l-1} = dZ_l(bf16) . W_l(bf16); weight gradients
    dW_l = dZ_l(bf16)^T . H_{l-1}(bf16), bias gradients the column sums of dZ_l(bf16); head gradients from the unrounded dOut
    and h3(bf16)

-- and evaluates exactly that in float64, so a launch can differ from it only by f32 summation order, the hardware
exp2 / rcp inside tanh (~6e-8 absolute before the rounding) and samples within ~1e-6 of a clip boundary.  With `rounding=False`
every rounding is the identity and the model IS the oracle's `evaluate` (checked against autograd in
tests/test_bf16_operand_model.py), which is pinned to the reference itself (tests/test_oracle_vs_reference.py).
"""
from __future__ import annotations

import math

import torch

LOG_SQRT_2PI = math.log(math.sqrt(2 * math.pi))
HALF_LOG_2PI = 0.5 * math.log(2 * math.pi)


def bf16r(x: torch.Tensor) -> torch.Tensor:
    """Round to the nearest bf16 (ties to even), keep the dtype."""
    return x.to(torch.bfloat16).to(x.dtype)


class _DenseTanh(torch.autograd.Function):
    """h = r(tanh(r(x) . r(W)^T + b)) with the backward sweep's roundings (module docstring); r = bf16r or the identity."""

    @staticmethod
    def forward(ctx, x, w, b, rounding):
        r = bf16r if rounding else (lambda t: t)
        xb, wb = r(x), r(w)
        h = r(torch.tanh(xb @ wb.t() + b))
        ctx.save_for_backward(xb, wb, h)
        ctx.rounding = rounding
        return h

    @staticmethod
    def backward(ctx, g):
        xb, wb, h = ctx.saved_tensors
        r = bf16r if ctx.rounding else (lambda t: t)
        dz = r(g * (1.0 - h * h))
        return dz @ wb, dz.t() @ xb, dz.sum(0), None


def _mlp3(x, layers, rounding):
    h = x
    for lin in layers:
        h = _DenseTanh.apply(h, lin.weight.double(), lin.bias.double(), rounding)
    return h


def evaluate(policy, states: torch.Tensor, action: torch.Tensor, rounding: bool = True) -> dict:
    """`OracleMLPPolicy.evaluate` (oracle/ppo_oracle.py, mlp_policy.py:202-236) through the rounded layers: per-dimension
    log-prob of the stored action, entropy, values; differentiable with respect to the policy's parameters."""
    x = states.double()
    h3 = _mlp3(x, [policy.backbone[0], policy.backbone[2], policy.backbone[4]], rounding)
    mean = (h3 @ policy.actor_mean.weight.double().t() + policy.actor_mean.bias.double()).float()  # f32 head output
    logstd = policy.actor_logstd.expand_as(mean)
    std = torch.exp(logstd)
    logp = -((action - mean) ** 2) / (2 * std ** 2) - std.log() - LOG_SQRT_2PI
    out = dict(logprobs=logp, entropy=0.5 + HALF_LOG_2PI + torch.log(std))
    if getattr(policy, "add_value_head", True):
        mlp = policy.value_head.mlp
        v3 = _mlp3(x, [mlp[0], mlp[2], mlp[4]], rounding)
        out["values"] = (v3 @ mlp[6].weight.double().t()).float()
    return out


@torch.no_grad()
def act(policy, states: torch.Tensor, eps: torch.Tensor, rounding: bool = True):
    """`OracleMLPPolicy.act` (mlp_policy.py:256-320, train mode, injected N(0,1) draw) through the rounded layers."""
    x = states.double()
    h3 = _mlp3(x, [policy.backbone[0], policy.backbone[2], policy.backbone[4]], rounding)
    mean = (h3 @ policy.actor_mean.weight.double().t() + policy.actor_mean.bias.double()).float()
    std = torch.exp(policy.actor_logstd.expand_as(mean))
    action = eps * std + mean
    logp = -((action - mean) ** 2) / (2 * std ** 2) - std.log() - LOG_SQRT_2PI
    mlp = policy.value_head.mlp
    value = (_mlp3(x, [mlp[0], mlp[2], mlp[4]], rounding) @ mlp[6].weight.double().t()).float()
    return action, logp, value

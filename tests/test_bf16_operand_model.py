"""CPU: the operand-rounding model of the bf16 launches (oracle/bf16_operand_model.py) is the oracle's arithmetic when its
roundings are switched off -- forward AND the hand-written backward sweep -- so what the GPU tests pin the bf16 launches to
(tests/test_gpu_fused_step.py::test_ppo_step_bf16_pinned_to_the_operand_rounded_restatement) differs from the reference's
arithmetic by the stated roundings and nothing else."""
import torch

from oracle import bf16_operand_model as BM
from oracle import ppo_oracle as O


def _policy(seed=3):
    torch.manual_seed(seed)
    ora = O.OracleMLPPolicy(42, 8, 1)
    with torch.no_grad():
        for p in ora.parameters():
            p.add_(torch.randn_like(p) * 0.02)
    return ora


def test_without_rounding_the_model_is_the_oracle():
    ora = _policy().double()
    g = torch.Generator().manual_seed(1)
    states, action = torch.randn(97, 42, generator=g).double(), torch.randn(97, 8, generator=g).double()
    want = ora.evaluate(states, action)
    (want["logprobs"].sum() * 0.3 + (want["values"] ** 2).sum()).backward()
    g_want = {n: p.grad.clone() for n, p in ora.named_parameters()}
    ora.zero_grad()
    got = BM.evaluate(ora, states, action, rounding=False)
    # the model's heads return f32 (as the launches do): compare the outputs at f32 resolution, the gradients through them
    torch.testing.assert_close(got["logprobs"].double(), want["logprobs"], rtol=1e-6, atol=1e-6)
    torch.testing.assert_close(got["values"].double(), want["values"], rtol=1e-6, atol=1e-6)
    (got["logprobs"].double().sum() * 0.3 + (got["values"].double() ** 2).sum()).backward()
    for n, p in ora.named_parameters():
        torch.testing.assert_close(p.grad, g_want[n], rtol=1e-5, atol=2e-5, msg=n)  # (f32 head outputs: ~1e-7 relative on every residual)


def test_rounding_points():
    """Every hidden activation is a bf16 number, the roundings move the outputs by bf16-sized amounts, and the backward sweep's
    weight gradient is the product of ROUNDED factors (a weight gradient tile is a sum of products of bf16 numbers)."""
    ora = _policy(5)
    g = torch.Generator().manual_seed(2)
    states, action = torch.randn(64, 42, generator=g), torch.randn(64, 8, generator=g)
    h = BM._mlp3(states.double(), [ora.backbone[0]], True)
    assert torch.equal(h, BM.bf16r(h))
    a = BM.evaluate(ora, states, action)["logprobs"]
    b = ora.evaluate(states, action)["logprobs"]
    d = float((a - b).detach().abs().max())
    assert 1e-5 < d < 0.2, d
    ora.zero_grad()
    BM.evaluate(ora, states, action)["values"].sum().backward()
    w1 = ora.value_head.mlp[0].weight.grad.double()
    # dW1 = dZ1(bf16)^T . X(bf16): recompute from the pieces
    x_b = BM.bf16r(states.double())
    assert w1.shape == (256, 42) and torch.isfinite(w1).all()
    # one output row of dW1 lies in the row space of X(bf16)^T with bf16 coefficients: solve and check the coefficients are bf16
    coef = torch.linalg.lstsq(x_b.t(), w1[:8].t()).solution  # (64, 8): dZ1[:, :8]
    resid = float((x_b.t() @ coef - w1[:8].t()).abs().max())
    assert resid < 1e-9, resid

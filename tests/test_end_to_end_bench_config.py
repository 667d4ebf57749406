"""End-to-end parity AT THE CONFIGURATION bench.py MEASURES (BASELINE.json configs[1]): 1024 envs x 128 steps, global batch
8192, 8 epochs x 16 minibatches = 128 optimizer steps per iteration, hipGraph replay on, f32 and bf16 -- the runner is built by
bench.py's own build_cfg / build_runner, so what is compared is literally what is timed.

Against oracle.ppo_loop.iteration (the CPU restatement of embodied_fsdp_actor_worker.py:186-321,483-700 + env_worker.py:1058-1306,
pinned bit-for-bit against the reference's own run_training in tests/test_reference_learner_loop.py) on the same weights, the same
synthetic env tensors, the same injected N(0,1) draws and the same shuffle.

What can and cannot be bounded tightly:
  * iteration 0's rollout, returns, advantages and the FIRST optimizer step (loss, metrics, gradient) see no optimizer drift:
    f32 holds them to 1e-4 relative (summation order + the 2e-7 tanh), bf16 to the distance of the reference arithmetic's own
    bf16-autocast run from its f32 run (the yardstick tests/test_gpu_fused_step.py uses per kernel).
  * after k AdamW steps two correct implementations differ: Adam divides by sqrt(v), so an element whose gradient is ~0 turns a
    rounding-level sign difference into a full +/- lr step.  Universal bound: |delta theta| <= 2 * lr * k per element; the BULK
    is held much tighter, and every per-step scalar (loss, grad norm, approx-KL, clip fraction, value loss) of all 128 steps is
    compared step by step.
"""

import copy
import json
import os
import time

import pytest
import torch

from oracle import ppo_loop as L
from oracle import ppo_oracle as O

pytestmark = pytest.mark.gpu

T, B, D, A = 128, 1024, 42, 8
GB, EPOCHS, LR = 8192, 8, 3e-4
N_STEPS = (T * B // GB) * EPOCHS
KEYS = ("actor/policy_loss", "actor/approx_kl", "actor/clip_fraction", "actor/ratio", "critic/value_loss")


@pytest.fixture(autouse=True)
def _cpu_threads():
    """torch's default of one thread per logical core is pathologically slow for these small GEMMs on a 256-core host
    (bench.py's cpu_baseline picks its thread count by a probe for the same reason)."""
    old = torch.get_num_threads()
    torch.set_num_threads(max(1, min(32, len(os.sched_getaffinity(0)))))
    yield
    torch.set_num_threads(old)


def _oracle(seed=11):
    torch.manual_seed(seed)
    ora = O.OracleMLPPolicy(D, A, 1)
    return ora, copy.deepcopy(ora.state_dict()), O.build_adamw(ora)


def _runner(precision: str, graph: bool, state_dict):
    import bench
    from rlinf_amd.scheduler import init_distributed
    assert (bench.ENVS, bench.HORIZON, bench.GLOBAL_BATCH, bench.UPDATE_EPOCH) == (B, T, GB, EPOCHS)
    runner = bench.build_runner(bench.build_cfg(1, graph, precision), init_distributed())
    runner.actor.worker.model.load_reference_state_dict(state_dict)
    return runner


def _env():
    env = L.synthetic_env_tensors(1234, T, B, D, max_episode_steps=50)  # cfg.env.train.seed / max_episode_steps of build_cfg
    from rlinf_amd.envs.synthetic_env import generate_tensors
    mine = generate_tensors(1234, T, B, D, 50)
    assert all(torch.equal(env[k], mine[k]) for k in env), "the product's synthetic env tensors are the oracle's"
    return env


def _per_step(worker):
    from rlinf_amd._lib import PPO_OUT_NAMES
    metrics_dev, norms_dev = worker._ws[("metrics", N_STEPS, 1)]
    m = metrics_dev.cpu()
    out = {k: m[:, PPO_OUT_NAMES[k]] for k in KEYS}
    out["actor/total_loss"] = m[:, PPO_OUT_NAMES["loss"]]
    out["actor/grad_norm"] = norms_dev[:, 0].cpu()
    return out


def _oracle_per_step(om):
    return {k: torch.tensor([float(m[k]) for m in om]) for k in KEYS + ("actor/total_loss", "actor/grad_norm")}


def _flat(ora):
    return torch.cat([p.detach().reshape(-1) for p in ora.parameters()])


def _report(name, payload):
    """Measured distances go to gpurun_out/ (merged back from the GPU box) so the thresholds below stay justified."""
    d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    os.makedirs(d, exist_ok=True)
    with open(os.path.join(d, "e2e_bench_config_parity.jsonl"), "a") as f:
        f.write(json.dumps({"test": name, **payload}) + "\n")
    print(f"[{name}] {payload}")


@pytest.mark.parametrize("graph", [True, False])
def test_bench_configuration_f32_matches_oracle(graph):
    """exact-f32 mode.  graph=True is the benchmarked mode: iteration 0 runs eagerly and captures (rollout loop and update
    phase), iteration 1 is replayed; graph=False launches everything eagerly -- both against the oracle: every tensor of the
    rollout batch, all 128 per-step scalars, the parameters."""
    env = _env()
    ora, sd, opt = _oracle()
    runner = _runner("32", graph, sd)
    w = runner.actor.worker
    for it in range(2):
        eps = torch.randn(T, B, A, generator=torch.Generator().manual_seed(100 + it))
        t0 = time.perf_counter()
        batch, om = L.iteration(ora, opt, env, eps, gamma=0.8, gae_lambda=0.9, seed=1234, global_batch=GB, update_epoch=EPOCHS)
        t_oracle = time.perf_counter() - t0
        metrics = runner.run_step(eps.cuda())
        rb = w.rollout_batch
        got, want = _per_step(w), _oracle_per_step(om)
        dev = {}
        for k in want:
            scale = float(want[k].abs().max()) + 1e-6
            err = (got[k] - want[k]).abs() / scale
            dev[k] = [float(err[0]), float(err.max())]
        diff = (w.model.flat.detach().cpu() - _flat(ora)).abs()
        steps_taken = N_STEPS * (it + 1)
        moved = (_flat(ora) - torch.cat([v.reshape(-1) for v in sd.values()])).abs()
        frac_far = float((diff > 0.1 * LR * steps_taken).float().mean())
        rollout_dev = {k: float((rb[k].cpu() - batch[k]).abs().max()) for k in ("prev_logprobs", "prev_values", "rewards", "returns", "advantages")}
        rollout_dev["action"] = float((rb["forward_inputs"]["action"].cpu() - batch["forward_inputs"]["action"]).abs().max())
        _report("f32_graph" if graph else "f32_eager",
                dict(iteration=it, oracle_s=round(t_oracle, 1), rollout_max_abs_dev=rollout_dev, per_step_rel_err_first_max=dev,
                     param_max=float(diff.max()), param_rms=float(diff.pow(2).mean().sqrt()),
                     update_rms=float(moved.pow(2).mean().sqrt()), frac_beyond_10pct_budget=frac_far))
        # iteration 0 starts from identical weights: f32-tight.  Iteration 1 starts from weights that differ by Adam's
        # amplification of rounding (measured: <= 3e-6 per element), so its rollout is held to 1e-4 and its later optimizer
        # steps to a few per cent of each scalar's scale.
        tol = dict(rtol=2e-4, atol=2e-5) if it == 0 else dict(rtol=1e-3, atol=2e-4)
        torch.testing.assert_close(rb["forward_inputs"]["action"].cpu(), batch["forward_inputs"]["action"], **tol)
        torch.testing.assert_close(rb["prev_logprobs"].cpu(), batch["prev_logprobs"], **tol)
        torch.testing.assert_close(rb["prev_values"].cpu(), batch["prev_values"], **tol)
        torch.testing.assert_close(rb["rewards"].cpu(), batch["rewards"], **tol)
        assert torch.equal(rb["dones"].cpu(), batch["dones"])
        torch.testing.assert_close(rb["returns"].cpu(), batch["returns"], **tol)
        torch.testing.assert_close(rb["advantages"].cpu(), batch["advantages"], rtol=tol["rtol"] * 5, atol=tol["atol"] * 5)
        for k, (first, worst) in dev.items():
            # the first optimizer step of an iteration sees no drift within the iteration
            assert first <= (1e-4 if it == 0 else 2e-3), (k, it, first)
            assert worst <= (1e-3 if it == 0 else 0.08), (k, it, worst)
        assert metrics["train/actor/approx_kl"] == pytest.approx(float(want["actor/approx_kl"].mean()), rel=2e-2 * (1 + it), abs=2e-5)
        assert metrics["train/actor/grad_norm"] == pytest.approx(float(want["actor/grad_norm"].mean()), rel=1e-2 * (1 + it))
        assert metrics["rollout/rewards"] == pytest.approx(float(batch["rewards"].mean()), rel=1e-4)
        assert float(diff.max()) <= 2 * LR * steps_taken + 1e-6
        # the bulk: the RMS difference is a small fraction of the RMS distance the parameters travelled
        assert float(diff.pow(2).mean().sqrt()) <= 0.05 * float(moved.pow(2).mean().sqrt()), (it, float(diff.pow(2).mean().sqrt()))
        assert frac_far < 0.01, frac_far
    assert int(w.step_state.sum()) == w.optimizer_steps == 2 * N_STEPS


@pytest.mark.parametrize("precision", ["32", "bf16"])
def test_first_optimizer_step_gradient_at_bench_configuration(precision):
    """The gradient of the FIRST optimizer step of an iteration (minibatch 0 of the shuffled 131072-row buffer, 8192 rows),
    where no optimizer drift applies and -- new policy == behaviour policy -- no sample sits on PPO's clip boundary: the
    product's rollout -> GAE -> normalisation -> shuffle -> fused forward / loss / backward against the oracle's autograd."""
    from rlinf_amd import ops
    from rlinf_amd._lib import PPO_OUT_FLOATS, PPO_OUT_NAMES
    env = _env()
    ora, sd, opt = _oracle()
    bf16 = precision == "bf16"
    runner = _runner(precision, False, sd)
    w = runner.actor.worker
    eps = torch.randn(T, B, A, generator=torch.Generator().manual_seed(100))
    # the runner's own call sequence up to the update (embodied_runner.py:478-563)
    runner.update_rollout_weights()
    runner.env.interact(eps.cuda())
    w.recv_rollout_trajectories(runner.env.send_rollout_trajectories(1).wait()[0])
    w.compute_advantages_and_returns()
    flat, N = w._flatten_and_shuffle()
    assert N == T * B
    mb = {k: v[:GB] for k, v in flat.items()}
    lay = w.model.layout
    grads = torch.full((ops.ppo_step_slabs(lay, GB, bf16=bf16), lay.n_params), float("nan"), device="cuda")
    ws = torch.empty(ops.ppo_step_workspace_bytes(lay, GB), dtype=torch.uint8, device="cuda")
    row = torch.zeros(PPO_OUT_FLOATS, device="cuda")
    ops.ppo_step(w.model.flat.data, lay, w._loss_params(False), mb, grads, row, ws, grad_out=1.0, bf16=bf16)
    got = grads.sum(dim=0).cpu()
    assert torch.isfinite(got).all()

    def oracle_grad(autocast):
        pol = O.OracleMLPPolicy(D, A, 1)
        pol.load_state_dict(sd)
        batch = L.advantages(L.rollout(pol, env, eps, 0.8, True, autocast=autocast), 0.8, 0.9, True)
        fl = O.flatten_and_shuffle(batch, torch.randperm(T * B, generator=torch.Generator().manual_seed(1234)))
        m0 = O.chunk_batch(fl, T * B // GB)[0]
        with O.amp(autocast):  # amp_context around the model forward only (embodied_fsdp_actor_worker.py:624-632)
            out = pol.evaluate(m0["forward_inputs"]["states"], m0["forward_inputs"]["action"])
        out = {k: v.float() for k, v in out.items()}
        shaped = O.shape_loss_inputs(out["logprobs"], m0["prev_logprobs"], m0["advantages"], "action_level", A,
                                     values=out["values"], prev_values=m0["prev_values"], returns=m0["returns"])
        loss, metrics = O.ppo_actor_critic_loss(clip_ratio_low=0.2, clip_ratio_high=0.2, value_clip=1.0, huber_delta=10.0,
                                                **shaped)
        loss.backward()
        return _flat_grad(pol), float(loss), {k: float(v) for k, v in metrics.items()}

    g32, loss32, m32 = oracle_grad(False)
    rel = float((got - g32).norm() / g32.norm())
    cos = float(torch.dot(got, g32) / (got.norm() * g32.norm()))
    host = row.cpu()
    if not bf16:
        _report("first_step_gradient_f32", dict(rel_l2=rel, cos=cos, max_abs=float((got - g32).abs().max()), gmax=float(g32.abs().max())))
        assert rel <= 2e-4 and cos > 1 - 1e-6, (rel, cos)
        assert float((got - g32).abs().max()) <= 2e-4 * float(g32.abs().max())
        assert float(host[PPO_OUT_NAMES["loss"]]) == pytest.approx(loss32, rel=1e-4)
        for k in KEYS:
            assert float(host[PPO_OUT_NAMES[k]]) == pytest.approx(m32[k], rel=1e-4, abs=1e-6), k
        return
    # bf16 operands: the yardstick is the reference arithmetic's own autocast run -- DERIVED BOUND: the product may be at most
    # twice as far from the f32 gradient as torch's bf16 autocast of the same loop is (both round the same operands to 8 bits;
    # the product keeps heads, log-probs, losses and accumulation in f32, so it is normally the closer of the two)
    g16, loss16, _ = oracle_grad(True)
    rel_auto = float((g16 - g32).norm() / g32.norm())
    _report("first_step_gradient_bf16", dict(rel_l2_product=rel, rel_l2_autocast_oracle=rel_auto, cos=cos,
                                             loss=[float(host[PPO_OUT_NAMES["loss"]]), loss32, loss16]))
    assert rel <= 2.0 * rel_auto + 1e-3 and cos > 0.995, (rel, rel_auto, cos)
    assert float(host[PPO_OUT_NAMES["loss"]]) == pytest.approx(loss32, rel=2e-2, abs=2e-3)
    # ... and the PIN: the same minibatch (the product's own rows, so the upstream rollout's rounding is not in the comparison)
    # through oracle/bf16_operand_model.py -- the oracle's arithmetic with the launches' operand roundings written out, float64.
    # First step of an iteration: ratio = 1 and value gap = 0 up to the roundings, no sample near a clip edge.
    from oracle import bf16_operand_model as BM
    polm = O.OracleMLPPolicy(D, A, 1)
    polm.load_state_dict(sd)
    c = {k: v.detach().cpu() for k, v in mb.items() if isinstance(v, torch.Tensor)}
    outm = BM.evaluate(polm, c["states"], c["action"])
    shaped = O.shape_loss_inputs(outm["logprobs"], c["prev_logprobs"], c["advantages"], "action_level", A, values=outm["values"],
                                 prev_values=c["prev_values"], returns=c["returns"])
    lossm, _ = O.ppo_actor_critic_loss(clip_ratio_low=0.2, clip_ratio_high=0.2, value_clip=1.0, huber_delta=10.0, **shaped)
    lossm.backward()
    gm = _flat_grad(polm)
    rel_pin = float((got - gm).norm() / gm.norm())
    _report("first_step_gradient_bf16_pinned", dict(rel_l2_vs_operand_rounded_model=rel_pin, loss=[float(host[PPO_OUT_NAMES["loss"]]), float(lossm)]))
    assert rel_pin <= 1e-3, rel_pin
    assert float(host[PPO_OUT_NAMES["loss"]]) == pytest.approx(float(lossm), rel=1e-4, abs=1e-6)


def _flat_grad(pol):
    return torch.cat([p.grad.reshape(-1) for p in pol.parameters()])


def test_bench_configuration_bf16_matches_autocast_oracle_with_graph_replay():
    """bf16 mode as benchmarked (hipGraph on), two iterations.  Yardsticks: the oracle in f32 and the oracle under
    torch.autocast(bf16) (what `precision: bf16` means in the reference, fsdp_model_manager.py:122-142).  DERIVED BOUNDS:
      rollout / returns / advantages of iteration 0 -- rtol = atol = 2e-2 against the autocast oracle;
      per-step scalars -- within 2x the autocast oracle's own distance from the f32 oracle (+ a floor of 1 % of scale);
      parameters -- RMS distance to the f32 oracle at most 2x the autocast oracle's RMS distance to it, worst element within
      the universal 2 * lr * steps budget."""
    env = _env()
    ora32, sd, opt32 = _oracle()
    ora16 = O.OracleMLPPolicy(D, A, 1)
    ora16.load_state_dict(sd)
    opt16 = O.build_adamw(ora16)
    runner = _runner("bf16", True, sd)
    w = runner.actor.worker
    kw = dict(gamma=0.8, gae_lambda=0.9, seed=1234, global_batch=GB, update_epoch=EPOCHS)
    for it in range(2):
        eps = torch.randn(T, B, A, generator=torch.Generator().manual_seed(100 + it))
        b32, om32 = L.iteration(ora32, opt32, env, eps, **kw)
        t0 = time.perf_counter()
        b16, om16 = L.iteration(ora16, opt16, env, eps, autocast=True, **kw)
        t_auto = time.perf_counter() - t0
        metrics = runner.run_step(eps.cuda())
        rb = w.rollout_batch
        if it == 0:
            tol = dict(rtol=2e-2, atol=2e-2)
            torch.testing.assert_close(rb["forward_inputs"]["action"].cpu(), b16["forward_inputs"]["action"], **tol)
            torch.testing.assert_close(rb["prev_values"].cpu(), b16["prev_values"], **tol)
            torch.testing.assert_close(rb["returns"].cpu(), b16["returns"], rtol=2e-2, atol=4e-2)
            torch.testing.assert_close(rb["advantages"].cpu(), b16["advantages"], rtol=2e-2, atol=6e-2)
            # log-prob of the sampled action: (a - mean) / sigma is eps itself whatever the mean's rounding -> f32-tight
            torch.testing.assert_close(rb["prev_logprobs"].cpu(), b32["prev_logprobs"], rtol=1e-4, atol=1e-4)
        assert torch.equal(rb["dones"].cpu(), b32["dones"])
        got, w32, w16 = _per_step(w), _oracle_per_step(om32), _oracle_per_step(om16)
        dev = {}
        for k in w32:
            scale = float(w32[k].abs().max()) + 1e-6
            ours = float((got[k] - w32[k]).abs().max()) / scale
            auto = float((w16[k] - w32[k]).abs().max()) / scale
            dev[k] = [ours, auto]
            assert ours <= 2.0 * auto + 0.01 * (1 + it), (k, it, ours, auto)
        p32 = _flat(ora32)
        d_ours, d_auto = w.model.flat.detach().cpu() - p32, _flat(ora16) - p32
        rms = lambda x: float(x.pow(2).mean().sqrt())  # noqa: E731
        steps_taken = N_STEPS * (it + 1)
        _report("bf16_graph", dict(iteration=it, autocast_oracle_s=round(t_auto, 1), per_step_rel_err_ours_vs_autocast=dev,
                                   param_rms_ours=rms(d_ours), param_rms_autocast=rms(d_auto), param_max_ours=float(d_ours.abs().max()),
                                   approx_kl=[metrics["train/actor/approx_kl"], float(w32["actor/approx_kl"].mean()),
                                              float(w16["actor/approx_kl"].mean())],
                                   grad_norm=[metrics["train/actor/grad_norm"], float(w32["actor/grad_norm"].mean()),
                                              float(w16["actor/grad_norm"].mean())]))
        assert rms(d_ours) <= 2.0 * rms(d_auto) + 1e-6, (it, rms(d_ours), rms(d_auto))
        assert float(d_ours.abs().max()) <= 2 * LR * steps_taken + 1e-6
        assert metrics["train/actor/grad_norm"] == pytest.approx(float(w32["actor/grad_norm"].mean()), rel=5e-2)
        assert metrics["train/actor/approx_kl"] == pytest.approx(float(w32["actor/approx_kl"].mean()), rel=0.1, abs=5e-4)


@pytest.mark.parametrize("precision", ["32", "bf16"])
def test_ten_iterations_reseeded_oracle_at_bench_configuration(precision):
    """TEN iterations of the benchmarked loop (hipGraph replay from iteration 1 on), drift-free by construction: at the start of
    every iteration the oracle is RE-SEEDED with the product's current weights, so iteration k's rollout tensors, returns,
    advantages and FIRST optimizer step (scalars from the replayed graph's own metric rows, and the gradient vector recomputed
    by the fused step at the iteration-start weights on the iteration's own shuffled minibatch 0) are compared as tightly as
    iteration 0's -- a slow bias (stale weight tiles after graph replay, a mis-stepped device-side Adam counter, a stale
    behaviour-policy row) surfaces as a growing distance, which two iterations cannot show.  Also checked every iteration: the
    device step counter, and that the replayed update actually moved the weights by a plausible AdamW amount.
    Bounds: f32 rel-L2 of the gradient <= 2e-4; bf16 no farther from the f32 gradient than twice the autocast oracle."""
    from rlinf_amd import ops
    from rlinf_amd._lib import PPO_OUT_FLOATS, PPO_OUT_NAMES
    env = _env()
    _, sd, _ = _oracle()
    bf16 = precision == "bf16"
    runner = _runner(precision, True, sd)
    w = runner.actor.worker
    lay = w.model.layout
    grads = torch.empty((ops.ppo_step_slabs(lay, GB, bf16=bf16), lay.n_params), device="cuda")
    ws = torch.empty(ops.ppo_step_workspace_bytes(lay, GB), dtype=torch.uint8, device="cuda")
    row = torch.zeros(PPO_OUT_FLOATS, device="cuda")
    perm = torch.randperm(T * B, generator=torch.Generator().manual_seed(1234))
    from rlinf_amd.models.embodiment.mlp_policy import MLPPolicy
    probe = MLPPolicy(D, A, 1, True, False, compute_dtype=torch.bfloat16 if bf16 else torch.float32).to("cuda")

    def oracle_first_step(state_dict, eps, autocast):
        pol = O.OracleMLPPolicy(D, A, 1)
        pol.load_state_dict(state_dict)
        batch = L.advantages(L.rollout(pol, env, eps, 0.8, True, autocast=autocast), 0.8, 0.9, True)
        m0 = O.chunk_batch(O.flatten_and_shuffle(batch, perm), T * B // GB)[0]
        with O.amp(autocast):
            out = pol.evaluate(m0["forward_inputs"]["states"], m0["forward_inputs"]["action"])
        out = {k: v.float() for k, v in out.items()}
        shaped = O.shape_loss_inputs(out["logprobs"], m0["prev_logprobs"], m0["advantages"], "action_level", A,
                                     values=out["values"], prev_values=m0["prev_values"], returns=m0["returns"])
        loss, metrics = O.ppo_actor_critic_loss(clip_ratio_low=0.2, clip_ratio_high=0.2, value_clip=1.0, huber_delta=10.0, **shaped)
        loss.backward()
        gn = float(_flat_grad(pol).norm())
        return batch, _flat_grad(pol), float(loss), {k: float(v) for k, v in metrics.items()}, gn

    worst = {}
    for it in range(10):
        start = {k: v.detach().cpu().clone() for k, v in w.model.reference_state_dict().items()}
        start_flat = w.model.flat.detach().clone()
        eps = torch.randn(T, B, A, generator=torch.Generator().manual_seed(100 + it))
        b32, g32, loss32, m32, gn32 = oracle_first_step(start, eps, False)
        metrics = runner.run_step(eps.cuda())
        assert (w._graph is not None) and (it == 0 or runner.env.worker._graph is not None), "the benchmarked mode replays hipGraphs"
        rb = w.rollout_batch
        ps = _per_step(w)
        # (1) this iteration's rollout tensors against the oracle re-seeded with this iteration's start weights
        if bf16:
            b16, g16, loss16, _, _ = oracle_first_step(start, eps, True)
            tol = dict(rtol=2e-2, atol=2e-2)
            torch.testing.assert_close(rb["forward_inputs"]["action"].cpu(), b16["forward_inputs"]["action"], **tol)
            torch.testing.assert_close(rb["prev_values"].cpu(), b16["prev_values"], **tol)
            torch.testing.assert_close(rb["returns"].cpu(), b16["returns"], rtol=2e-2, atol=4e-2)
            torch.testing.assert_close(rb["advantages"].cpu(), b16["advantages"], rtol=2e-2, atol=6e-2)
            torch.testing.assert_close(rb["prev_logprobs"].cpu(), b32["prev_logprobs"], rtol=1e-4, atol=1e-4)
        else:
            tol = dict(rtol=2e-4, atol=2e-5)
            torch.testing.assert_close(rb["forward_inputs"]["action"].cpu(), b32["forward_inputs"]["action"], **tol)
            torch.testing.assert_close(rb["prev_logprobs"].cpu(), b32["prev_logprobs"], **tol)
            torch.testing.assert_close(rb["prev_values"].cpu(), b32["prev_values"], **tol)
            torch.testing.assert_close(rb["rewards"].cpu(), b32["rewards"], **tol)
            torch.testing.assert_close(rb["returns"].cpu(), b32["returns"], **tol)
            torch.testing.assert_close(rb["advantages"].cpu(), b32["advantages"], rtol=1e-3, atol=1e-4)
        assert torch.equal(rb["dones"].cpu(), b32["dones"])
        # (2) the first optimizer step's scalars, as the (replayed) update graph itself wrote them
        first = {"loss": float(ps["actor/total_loss"][0]), "grad_norm": float(ps["actor/grad_norm"][0]),
                 **{k: float(ps[k][0]) for k in KEYS}}
        rel_s = 2e-2 if bf16 else 2e-4
        assert first["loss"] == pytest.approx(loss32, rel=rel_s, abs=rel_s * 0.1), (it, first["loss"], loss32)
        assert first["grad_norm"] == pytest.approx(gn32, rel=5e-2 if bf16 else 5e-4), (it, first["grad_norm"], gn32)
        for k in KEYS:
            assert first[k] == pytest.approx(m32[k], rel=rel_s, abs=(2e-3 if bf16 else 2e-6)), (it, k, first[k], m32[k])
        # (3) the gradient vector at the iteration-start weights on this iteration's shuffled minibatch 0
        flat = {k: v for k, v in zip(("states", "action", "prev_logprobs", "advantages", "prev_values", "returns"),
                                     w._ws[[k for k in w._ws if isinstance(k, tuple) and k[0] == "shuf"][0]])}
        mb = {k: v[:GB] for k, v in flat.items()}
        probe.flat.data.copy_(start_flat)
        probe.mark_updated()
        ops.ppo_step(probe.flat.data, lay, w._loss_params(False), mb, grads, row, ws, grad_out=1.0, bf16=bf16)
        got = grads.sum(dim=0).cpu()
        rel = float((got - g32).norm() / g32.norm())
        if bf16:
            rel_auto = float((g16 - g32).norm() / g32.norm())
            assert rel <= 2.0 * rel_auto + 1e-3, (it, rel, rel_auto)
            worst[it] = [rel, rel_auto]
        else:
            assert rel <= 2e-4, (it, rel)
            worst[it] = rel
        # (4) bookkeeping that graph replay must keep right
        assert int(w.step_state.sum()) == w.optimizer_steps == N_STEPS * (it + 1)
        moved = (w.model.flat.detach() - start_flat).abs()
        assert 0 < float(moved.max()) <= 2 * LR * N_STEPS + 1e-6 and float(moved.mean()) > 0.05 * LR
        assert metrics["train/actor/grad_norm"] > 0 and metrics["rollout/rewards"] == pytest.approx(float(b32["rewards"].mean()), rel=2e-2 if bf16 else 1e-4)
    _report(f"ten_iterations_{'bf16' if bf16 else 'f32'}", dict(first_step_gradient_rel_l2_by_iteration=worst))

"""One full actor-learner iteration (rollout -> GAE -> shuffled minibatch PPO updates) through the reference-shaped
workers/runner API, against the CPU oracle loop on identical seeds, weights, injected noise and shuffle order."""

import copy
import os

import pytest
import torch

from oracle import ppo_loop as L
from oracle import ppo_oracle as O


def make_cfg(total_envs=8, steps=16, global_batch=32, micro_batch=None, update_epoch=2, gamma=0.8, lam=0.9,
             auto_reset=True, hip_graph=False, rollout_epoch=1, entropy_bonus=0, stage_num=1, pipeline=False,
             critic_warmup_steps=0, lr_scheduler=None, total_training_steps=0, done_mode=None, entropy_type="action_level",
             pipeline_overlap=True):
    from rlinf_amd.config import DictConfig
    return DictConfig(dict(
        runner=dict(task_type="embodied", max_epochs=1, max_steps=-1, use_training_pipeline=pipeline, pipeline_overlap=pipeline_overlap),
        algorithm=dict(update_epoch=update_epoch, normalize_advantages=True, group_size=1, reward_type="action_level",
                       logprob_type="action_level", entropy_type=entropy_type, adv_type="gae", loss_type="actor_critic",
                       bootstrap_type="always", entropy_bonus=entropy_bonus, clip_ratio_high=0.2, clip_ratio_low=0.2, value_clip=1.0,
                       huber_delta=10.0, gamma=gamma, gae_lambda=lam),
        env=dict(train=dict(rollout_epoch=rollout_epoch, total_num_envs=total_envs, auto_reset=auto_reset, ignore_terminations=False,
                            max_episode_steps=5, max_steps_per_rollout_epoch=steps, seed=0, group_size=1)),
        rollout=dict(pipeline_stage_num=stage_num, enable_cuda_graph=hip_graph),
        actor=dict(training_backend="fsdp", micro_batch_size=micro_batch or global_batch, global_batch_size=global_batch,
                   seed=1234, enable_hip_graph=hip_graph,
                   model=dict(model_type="mlp_policy", obs_dim=42, action_dim=8, num_action_chunks=1, precision="32",
                              add_value_head=True),
                   optim=dict(lr=3e-4, value_lr=3e-4, adam_beta1=0.9, adam_beta2=0.999, adam_eps=1e-8,
                              weight_decay=0.01, clip_grad=0.5, critic_warmup_steps=critic_warmup_steps,
                              **({"lr_scheduler": lr_scheduler, "total_training_steps": total_training_steps} if lr_scheduler else {})),
                   fsdp_config=dict(strategy="fsdp", sharding_strategy="no_shard"))))


def test_config_validation_rules():
    from rlinf_amd.config import validate_cfg
    cfg = validate_cfg(make_cfg())
    assert cfg.runner.weight_sync_interval == 1 and cfg.actor.optim.critic_warmup_steps == 0
    bad = make_cfg()
    bad.actor.model.add_value_head = False
    with pytest.raises(AssertionError, match="add_value_head must be True"):
        validate_cfg(bad)
    bad = make_cfg()
    bad.algorithm.adv_type, bad.algorithm.group_size = "grpo", 1
    with pytest.raises(AssertionError, match="group_size must be greater than 1"):
        validate_cfg(bad)
    bad = make_cfg(total_envs=8)
    bad.env.train.group_size = 3
    with pytest.raises(AssertionError, match="divisible by the group size"):
        validate_cfg(bad)
    bad = make_cfg()
    bad.actor.fsdp_config.sharding_strategy = "full_shard"
    with pytest.raises(AssertionError, match="no_shard"):
        validate_cfg(bad)


def test_config_rejects_shapes_the_kernels_cannot_take():
    """validate_cfg names the limit instead of leaving it to the first launch's errno string (csrc/ppo_step_common.h: HID = 256,
    Tiles::K1P = 64, MAX_OUT = 16)."""
    from rlinf_amd.config import MLP_KERNEL_LIMITS, validate_cfg
    assert MLP_KERNEL_LIMITS == {"hidden_dim": 256, "obs_dim_max": 64, "action_outputs_max": 16}
    ok = make_cfg()
    ok.actor.model.num_action_chunks, ok.actor.model.obs_dim = 2, 64
    validate_cfg(ok)  # 2 x 8 = 16 head outputs and 64 inputs are the edges that still fit
    bad = make_cfg()
    bad.actor.model.obs_dim = 65
    with pytest.raises(AssertionError, match="obs_dim=65.*<= 64"):
        validate_cfg(bad)
    bad = make_cfg(steps=18)
    bad.actor.model.num_action_chunks = 3
    with pytest.raises(AssertionError, match=r"8 \* 3 = 24.*at most 16"):
        validate_cfg(bad)
    bad = make_cfg()
    bad.actor.model.hidden_dim = 512
    with pytest.raises(AssertionError, match="hidden_dim=512"):
        validate_cfg(bad)


def test_synthetic_env_chunk_steps_match_the_oracle_rollouts_indexing():
    """The product's synthetic env and the oracle loop cut the same ENV-step tensors into the same chunk steps (C = 1, 2, 4):
    rewards side by side, done flags raised in the last column for any sub-step, obs after / final obs of the chunk."""
    from rlinf_amd.envs.synthetic_env import SyntheticManiSkillEnv, generate_tensors
    for C in (1, 2, 4):
        T_env, B = 24, 6
        t = generate_tensors(3, T_env, B, 5, 50, mode="bernoulli", p_done=0.2)
        env = SyntheticManiSkillEnv(t, "cpu", C, auto_reset=True)
        obs, _ = env.reset(0)
        assert torch.equal(obs["states"], t["obs"][0])
        for k in range(T_env // C):
            obs, r, term, trunc, infos = env.chunk_step(None)
            assert torch.equal(r, t["rewards"][k * C:(k + 1) * C].transpose(0, 1)) and r.is_contiguous()
            want = torch.zeros(B, C, dtype=torch.bool)
            want[:, -1] = t["dones"][k * C + 1:(k + 1) * C + 1].any(dim=0)
            assert torch.equal(trunc, want) and not term.any()
            assert torch.equal(obs["states"], t["obs"][(k + 1) * C])
            assert torch.equal(infos["final_obs"]["states"], t["final_obs"][(k + 1) * C - 1])


def test_single_thread_randperm_is_torchs_randperm():
    """The learner's shuffles come from torch.randperm with the reference's generators; run on one thread (the multi-threaded
    arange fill in front of the serial shuffle costs ~60 ms per call above 32768 elements) they are the same permutations, from
    the same generator state, and leave the generator where torch leaves it."""
    from rlinf_amd.workers.actor.embodied_fsdp_actor_worker import _host_randperm
    for n in (10, 32768, 32769, 131072):
        g1, g2 = torch.Generator().manual_seed(1234), torch.Generator().manual_seed(1234)
        threads = torch.get_num_threads()
        a, a2 = _host_randperm(n, g1), _host_randperm(n, g1)
        b, b2 = torch.randperm(n, generator=g2), torch.randperm(n, generator=g2)
        assert torch.equal(a, b) and torch.equal(a2, b2) and torch.equal(g1.get_state(), g2.get_state())
        assert torch.get_num_threads() == threads


def test_worker_peers_are_scoped_to_their_configuration():
    """A worker launched by another runner / tool with a different configuration is not a peer (its model must never be adopted
    silently), and a runner that is closed leaves nothing behind."""
    from rlinf_amd.workers import common
    from rlinf_amd.workers.common import Worker, clear_peers, peer

    class A(Worker):
        ROLE = "actor"

    class R(Worker):
        ROLE = "rollout"

    cfg1, cfg2 = make_cfg(), make_cfg(total_envs=16)
    clear_peers()
    a1 = A.create_group(cfg1).launch(None).worker
    assert peer("actor") is a1 and peer("actor", cfg1) is a1 and peer("actor", make_cfg()) is a1  # equal configurations are one job
    assert peer("actor", cfg2) is None
    r2 = R.create_group(cfg2).launch(None).worker
    assert peer("rollout", cfg1) is None and peer("rollout", cfg2) is r2
    clear_peers(a1)
    assert peer("actor") is None and peer("rollout") is r2
    clear_peers()
    assert common._PEERS == {}


def _build(cfg, env_tensors, state_dict):
    from rlinf_amd.config import validate_cfg
    from rlinf_amd.runners import EmbodiedRunner
    from rlinf_amd.scheduler import init_distributed
    from rlinf_amd.workers.actor import EmbodiedFSDPActor
    from rlinf_amd.workers.env import EnvWorker
    from rlinf_amd.workers.rollout.hf import MultiStepRolloutWorker
    if cfg.algorithm.loss_type == "decoupled_actor_critic":
        from rlinf_amd.workers.actor.async_ppo_fsdp_worker import AsyncPPOEmbodiedFSDPActor as EmbodiedFSDPActor  # noqa: F811
    cfg = validate_cfg(cfg)
    ctx = init_distributed()
    actor = EmbodiedFSDPActor.create_group(cfg, ctx).launch(None, name="ActorGroup")
    rollout = MultiStepRolloutWorker.create_group(cfg, ctx).launch(None, name="RolloutGroup")
    env = EnvWorker.create_group(cfg, ctx).launch(None, name="EnvGroup")
    runner = EmbodiedRunner(cfg, actor, rollout, env)
    runner.init_workers(env_tensors=env_tensors)
    actor.worker.model.load_reference_state_dict(state_dict)
    return runner


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [dict(total_envs=8, steps=16, global_batch=32),                      # configs[0]: 8 envs
                                   dict(total_envs=64, steps=20, global_batch=320, micro_batch=160),   # grad accumulation
                                   dict(total_envs=32, steps=12, global_batch=96, hip_graph=True),
                                   dict(total_envs=16, steps=10, global_batch=80, rollout_epoch=3),    # a8: epochs fold
                                   dict(total_envs=16, steps=8, global_batch=64, rollout_epoch=2, hip_graph=True),
                                   dict(total_envs=16, steps=12, global_batch=96, micro_batch=48, entropy_bonus=0.02),  # a22
                                   dict(total_envs=16, steps=12, global_batch=96, entropy_bonus=0.02, hip_graph=True),
                                   dict(total_envs=16, steps=10, global_batch=80, rollout_epoch=2, stage_num=2),   # stages
                                   dict(total_envs=16, steps=10, global_batch=80, stage_num=2, hip_graph=True),
                                   dict(total_envs=16, steps=10, global_batch=80, micro_batch=40, stage_num=2, pipeline=True),
                                   dict(total_envs=16, steps=10, global_batch=80, pipeline=True, hip_graph=True),
                                   dict(total_envs=16, steps=10, global_batch=40, critic_warmup_steps=5),  # 8 steps / iteration
                                   # synchronous learner WITHOUT auto-reset: loss mask + loss_mask_sum / max_episode_steps ratio
                                   # aggregation (embodied_fsdp_actor_worker.py:219-233, losses.py:219-227), whole iteration
                                   dict(total_envs=32, steps=12, global_batch=96, auto_reset=False, done_mode="bernoulli"),
                                   dict(total_envs=32, steps=12, global_batch=192, micro_batch=96, auto_reset=False,
                                        done_mode="bernoulli", entropy_bonus=0.02),
                                   dict(total_envs=32, steps=12, global_batch=96, auto_reset=False, done_mode="bernoulli",
                                        hip_graph=True),
                                   # entropy_type chunk_level under a loss mask (C = 1): the reference's masked_mean broadcasts
                                   # [bsz] x [bsz, 1] into an outer product -- the SUM of the row entropies; reproduced as written
                                   # (no gradient accumulation here: a SUM over the micro-batch's rows is not micro-batch
                                   # invariant, and the oracle loop steps whole global batches)
                                   dict(total_envs=32, steps=12, global_batch=192, auto_reset=False,
                                        done_mode="bernoulli", entropy_bonus=0.001, entropy_type="chunk_level"),
                                   # a schedule that moves every iteration: graphs / prepared plans must follow it
                                   dict(total_envs=16, steps=10, global_batch=80, hip_graph=True, lr_scheduler="torch_cosine",
                                        total_training_steps=3)])
def test_iteration_matches_oracle(shape):
    cfg = make_cfg(**shape)
    E = shape.get("rollout_epoch", 1)
    T, B = shape["steps"] * E, shape["total_envs"]   # T: steps over all rollout epochs
    auto_reset = shape.get("auto_reset", True)
    env = L.synthetic_env_tensors(0, T, B, 42, max_episode_steps=5, mode=shape.get("done_mode") or "periodic", p_done=0.08)
    torch.manual_seed(11)
    ora = O.OracleMLPPolicy(42, 8, 1)
    sd = copy.deepcopy(ora.state_dict())
    opt = O.build_adamw(ora)
    runner = _build(cfg, env, sd)
    n_iter = 3 if shape.get("hip_graph") else 2  # graph: eager warm-up, capture+replay, replay
    pipe, steps_done = None, 0
    sched = None
    if shape.get("lr_scheduler") == "torch_cosine":  # what get_lr_scheduler builds for that name (fsdp/utils.py:594-602)
        sched = torch.optim.lr_scheduler.CosineAnnealingLR(opt, T_max=shape["total_training_steps"], eta_min=1e-6)
    if shape.get("pipeline"):  # the rank's stateful shuffle generator, seeded like _init_pipeline_params (rank 0 of 1)
        pipe = dict(stage_num=shape.get("stage_num", 1), generator=torch.Generator().manual_seed(1234))
    for it in range(n_iter):
        eps = torch.randn(T, B, 8, generator=torch.Generator().manual_seed(100 + it))
        batch, om = L.iteration(ora, opt, env, eps, gamma=0.8, gae_lambda=0.9, seed=1234,
                                global_batch=shape["global_batch"], update_epoch=2, rollout_epoch=E,
                                entropy_bonus=shape.get("entropy_bonus", 0.0), pipeline=pipe,
                                critic_warmup_steps=shape.get("critic_warmup_steps", 0), steps_done=steps_done,
                                auto_reset=auto_reset, max_episode_steps=None if auto_reset else 5,
                                entropy_type=shape.get("entropy_type", "action_level"))
        steps_done += len(om)
        metrics = runner.run_step(eps.cuda())
        # lr_list is 0.0 for every step taken while the critic warms up, the ending step included (fsdp_model_manager.py:451-461)
        warm = max(0, min(len(om), shape.get("critic_warmup_steps", 0) - (steps_done - len(om))))
        assert metrics["train/actor/lr"] == pytest.approx(opt.param_groups[0]["lr"] * (len(om) - warm) / len(om), rel=1e-12)
        assert ("train/critic/lr" in metrics) == (warm < len(om))
        if sched is not None:
            sched.step()  # once per run_training
        rb = runner.actor.worker.rollout_batch
        tol = dict(rtol=2e-4, atol=2e-5) if it == 0 else dict(rtol=5e-3, atol=5e-4)  # later iterations inherit Adam's drift
        torch.testing.assert_close(rb["forward_inputs"]["action"].cpu(), batch["forward_inputs"]["action"], **tol)
        torch.testing.assert_close(rb["prev_logprobs"].cpu(), batch["prev_logprobs"], **tol)
        torch.testing.assert_close(rb["prev_values"].cpu(), batch["prev_values"], **tol)
        torch.testing.assert_close(rb["rewards"].cpu(), batch["rewards"], **tol)
        assert torch.equal(rb["dones"].cpu(), batch["dones"])
        torch.testing.assert_close(rb["returns"].cpu(), batch["returns"], **tol)
        torch.testing.assert_close(rb["advantages"].cpu(), batch["advantages"], rtol=tol["rtol"] * 5, atol=tol["atol"] * 5)
        want_loss = sum(float(m["actor/total_loss"]) for m in om) / len(om) / max(cfg.actor.global_batch_size // cfg.actor.micro_batch_size, 1)
        assert metrics["train/actor/total_loss"] == pytest.approx(want_loss, rel=2e-3, abs=2e-4)
        if shape.get("entropy_bonus"):
            want_ent = sum(float(m["actor/entropy_loss"]) for m in om) / len(om)
            assert metrics["train/actor/entropy_loss"] == pytest.approx(want_ent, rel=1e-4)
        want_gn = sum(float(m["actor/grad_norm"]) for m in om) / len(om)
        assert metrics["train/actor/grad_norm"] == pytest.approx(want_gn, rel=2e-3)
        if auto_reset:
            assert metrics["rollout/rewards"] == pytest.approx(float(batch["rewards"].mean()), rel=1e-4)
        else:  # the loss mask and its per-env sums are integer work: bit-exact; metrics are masked (metric_utils.py:447-460)
            assert torch.equal(rb["loss_mask"].cpu(), batch["loss_mask"]) and torch.equal(rb["loss_mask_sum"].cpu(), batch["loss_mask_sum"])
            assert metrics["rollout/rewards"] == pytest.approx(float(batch["rewards"][batch["loss_mask"]].mean()), rel=1e-4)
            assert metrics["rollout/advantages_max"] == pytest.approx(float(batch["advantages"][batch["loss_mask"]].max()), rel=1e-3)
            for k in ("actor/policy_loss", "critic/value_loss", "actor/approx_kl", "actor/clip_fraction"):
                want_k = sum(float(m[k]) for m in om) / len(om)
                assert metrics["train/" + k] == pytest.approx(want_k, rel=5e-3, abs=5e-5), k
        # parameters after the updates: Adam turns a sign flip of a ~0 gradient into a 2*lr difference, so bound the
        # bulk tightly and the worst element by the number of steps taken
        got = runner.actor.worker.model.flat.detach().cpu()
        want = torch.cat([p.detach().reshape(-1) for p in ora.parameters()])
        diff = (got - want).abs()
        steps_taken = len(om) * (it + 1)
        assert float(diff.max()) <= 2 * 3e-4 * steps_taken + 1e-6
        assert float((diff > 2e-5 * (it + 1)).float().mean()) < 0.02, float((diff > 2e-5).float().mean())
    # the device-side Adam step counter restarts when the warm-up ends (a new optimizer in the reference)
    assert int(runner.actor.worker.step_state.sum()) == runner.actor.worker.optimizer_steps - shape.get("critic_warmup_steps", 0)


@pytest.mark.gpu
@pytest.mark.parametrize("case", [
    # adv, loss, reward_type, logprob_type, auto_reset, hip_graph, precision
    dict(adv="gae", loss="actor_critic", reward="action_level", logprob="action_level", auto_reset=True, hip_graph=False),
    dict(adv="gae", loss="actor_critic", reward="action_level", logprob="action_level", auto_reset=True, hip_graph=True),
    dict(adv="gae", loss="actor_critic", reward="action_level", logprob="token_level", auto_reset=True, hip_graph=False),
    dict(adv="gae", loss="actor_critic", reward="action_level", logprob="action_level", auto_reset=False, hip_graph=False),
    dict(adv="gae", loss="actor_critic", reward="action_level", logprob="action_level", auto_reset=False, hip_graph=True),
    # value-free GRPO on chunk-level rewards (the only reward_type = chunk_level combination the reference's own shapes allow
    # with its MLP policy: its value head always has num_action_chunks outputs, so GAE cannot take a chunk-summed reward)
    dict(adv="grpo", loss="actor", reward="chunk_level", logprob="chunk_level", auto_reset=False, hip_graph=False),
    dict(adv="grpo", loss="actor", reward="chunk_level", logprob="chunk_level", auto_reset=False, hip_graph=True),
    dict(adv="grpo", loss="actor", reward="chunk_level", logprob="action_level", auto_reset=False, hip_graph=False),
    dict(adv="gae", loss="actor_critic", reward="action_level", logprob="action_level", auto_reset=True, hip_graph=True, precision="bf16"),
], ids=lambda c: "-".join(str(v) for v in c.values()))
def test_action_chunks_whole_loop_matches_oracle(case):
    """num_action_chunks = 2 through the WHOLE loop (SURVEY.md 8c "C > 1"): the env's chunk steps ([B, 2] rewards, done flags in
    the last column, bootstrap at the chunk's end), 16 head outputs and 2 value columns per policy step, the [n, B, C] <-> [T, B]
    time-chunk layout inside GAE / the loss mask, loss shaping per logprob_type, the fused optimizer step and AdamW -- eager and
    as replayed hipGraphs -- against oracle.ppo_loop.iteration on identical seeds, weights, injected noise and shuffle order."""
    C, A, B, GB = 2, 8, 32, 96
    steps_env, group = 24, 4                      # 12 chunk steps of 2 env steps
    n = steps_env // C
    bf16 = case.get("precision") == "bf16"
    cfg = make_cfg(total_envs=B, steps=steps_env, global_batch=GB, auto_reset=case["auto_reset"], hip_graph=case["hip_graph"],
                   done_mode=None if case["auto_reset"] else "bernoulli")
    cfg.actor.model.num_action_chunks = C
    cfg.actor.model.precision = case.get("precision", "32")
    alg = cfg.algorithm
    alg.adv_type, alg.loss_type, alg.reward_type, alg.logprob_type = case["adv"], case["loss"], case["reward"], case["logprob"]
    if case["adv"] == "grpo":
        alg.group_size = cfg.env.train.group_size = group
    env = L.synthetic_env_tensors(0, steps_env, B, 42, max_episode_steps=5, mode="periodic" if case["auto_reset"] else "bernoulli",
                                  p_done=0.05)
    torch.manual_seed(11)
    ora = O.OracleMLPPolicy(42, A, C)
    sd = copy.deepcopy(ora.state_dict())
    opt = O.build_adamw(ora)
    runner = _build(cfg, env, sd)
    w = runner.actor.worker
    assert w.model.layout.act_dim == C * A and w.model.layout.val_dim == C
    init = w.model.flat.detach().cpu().clone()
    for it in range(3 if case["hip_graph"] else 2):
        eps = torch.randn(n, B, C * A, generator=torch.Generator().manual_seed(100 + it))
        batch, om = L.iteration(ora, opt, env, eps, gamma=0.8, gae_lambda=0.9, seed=1234, global_batch=GB, update_epoch=2,
                                auto_reset=case["auto_reset"], max_episode_steps=None if case["auto_reset"] else 5,
                                adv_type=case["adv"], loss_type=case["loss"], reward_type=case["reward"],
                                logprob_type=case["logprob"], group_size=group if case["adv"] == "grpo" else 1, autocast=bf16)
        metrics = runner.run_step(eps.cuda())
        rb = w.rollout_batch
        if bf16:
            tol = dict(rtol=3e-2, atol=3e-2)
        else:
            tol = dict(rtol=2e-4, atol=2e-5) if it == 0 else dict(rtol=5e-3, atol=5e-4)
        torch.testing.assert_close(rb["forward_inputs"]["action"].cpu(), batch["forward_inputs"]["action"], **tol)
        torch.testing.assert_close(rb["prev_logprobs"].cpu(), batch["prev_logprobs"], **(tol if not bf16 else dict(rtol=5e-2, atol=5e-2)))
        torch.testing.assert_close(rb["prev_values"].cpu(), batch["prev_values"], **tol)
        torch.testing.assert_close(rb["rewards"].cpu(), batch["rewards"], **tol)
        assert torch.equal(rb["dones"].cpu(), batch["dones"])
        assert rb["rewards"].shape == (n, B, C) and rb["dones"].shape == (n + 1, B, C) and rb["prev_values"].shape == (n + 1, B, C)
        if "returns" in batch:
            torch.testing.assert_close(rb["returns"].cpu(), batch["returns"], **tol)
        if not bf16:
            torch.testing.assert_close(rb["advantages"].cpu(), batch["advantages"], rtol=tol["rtol"] * 5, atol=tol["atol"] * 5)
        if not case["auto_reset"]:  # integer work: bit-exact (chunk_level: reduced with any / last column, :228-230)
            assert torch.equal(rb["loss_mask"].cpu(), batch["loss_mask"]) and torch.equal(rb["loss_mask_sum"].cpu(), batch["loss_mask_sum"])
        if not bf16:
            want_loss = sum(float(m["actor/total_loss"]) for m in om) / len(om)
            assert metrics["train/actor/total_loss"] == pytest.approx(want_loss, rel=2e-3, abs=2e-4)
            want_gn = sum(float(m["actor/grad_norm"]) for m in om) / len(om)
            assert metrics["train/actor/grad_norm"] == pytest.approx(want_gn, rel=2e-3)
            for k in ("actor/policy_loss", "actor/approx_kl", "actor/clip_fraction") + (("critic/value_loss",) if case["loss"] == "actor_critic" else ()):
                want_k = sum(float(m[k]) for m in om) / len(om)
                assert metrics["train/" + k] == pytest.approx(want_k, rel=5e-3, abs=5e-5), k
        got = w.model.flat.detach().cpu()
        want = torch.cat([p.detach().reshape(-1) for p in ora.parameters()])
        diff = (got - want).abs()
        steps_taken = len(om) * (it + 1)
        assert float(diff.max()) <= 2 * 3e-4 * steps_taken + 1e-6
        if not bf16:
            assert float((diff > 2e-5 * (it + 1)).float().mean()) < 0.02, float((diff > 2e-5).float().mean())
        if case["loss"] == "actor":  # no critic: the value head is never touched (grad None in the reference: no decay either)
            for nm in w.model.shapes:
                if "value_head" in nm:
                    b, e = w.model.offsets[nm], w.model.offsets[nm] + w.model.view(nm).numel()
                    assert torch.equal(got[b:e], init[b:e]), nm


@pytest.mark.gpu
@pytest.mark.parametrize("overlap,hip_graph,stage_num,epochs", [(True, False, 2, 2), (True, True, 1, 3), (False, False, 2, 2),
                                                                (False, True, 2, 2)])
def test_pipeline_rollout_epochs_overlap_matches_oracle(overlap, hip_graph, stage_num, epochs):
    """runner.use_training_pipeline with rollout_epoch > 1 (embodied_runner.py:565-642, env_worker.py:1324-1330,
    fsdp_actor_worker_pipeline.py:84-160): every epoch is its own learner batch -- own statistics normalisation, own stage
    shuffles from the rank's running generator -- rolled out with the iteration's frozen weights while the learner trains on
    the previous epoch (rollout on its own HIP stream, one event per epoch; the rollout worker holds its own weight copy).
    Overlapped or not, graph-replayed rollouts or not: the same numbers as the oracle's sequential restatement."""
    T, B, GB = 10, 16, 80
    cfg = make_cfg(total_envs=B, steps=T, global_batch=GB, micro_batch=40, rollout_epoch=epochs, stage_num=stage_num, pipeline=True,
                   hip_graph=hip_graph, pipeline_overlap=overlap)
    env = L.synthetic_env_tensors(0, T * epochs, B, 42, max_episode_steps=5)
    torch.manual_seed(11)
    ora = O.OracleMLPPolicy(42, 8, 1)
    sd = copy.deepcopy(ora.state_dict())
    opt = O.build_adamw(ora)
    runner = _build(cfg, env, sd)
    w = runner.actor.worker
    assert runner.env.worker.overlap == overlap
    pipe = dict(stage_num=stage_num, generator=torch.Generator().manual_seed(1234))
    for it in range(3):
        eps = torch.randn(T * epochs, B, 8, generator=torch.Generator().manual_seed(100 + it))
        batches, om = L.iteration(ora, opt, env, eps, gamma=0.8, gae_lambda=0.9, seed=1234, global_batch=GB, update_epoch=2,
                                  rollout_epoch=epochs, pipeline=pipe)
        metrics = runner.run_step(eps.cuda())
        # overlapped: the rollout worker holds its own (frozen) weight copy; on one stream it aliases the learner's policy
        assert (runner.rollout.worker.hf_model is not w.model) == overlap
        tol = dict(rtol=2e-4, atol=2e-5) if it == 0 else dict(rtol=5e-3, atol=5e-4)
        assert len(w.rollout_batches) == epochs
        for rb, batch in zip(w.rollout_batches, batches):
            torch.testing.assert_close(rb["forward_inputs"]["action"].cpu(), batch["forward_inputs"]["action"], **tol)
            torch.testing.assert_close(rb["prev_values"].cpu(), batch["prev_values"], **tol)
            torch.testing.assert_close(rb["rewards"].cpu(), batch["rewards"], **tol)
            assert torch.equal(rb["dones"].cpu(), batch["dones"])
            torch.testing.assert_close(rb["returns"].cpu(), batch["returns"], **tol)
            torch.testing.assert_close(rb["advantages"].cpu(), batch["advantages"], rtol=tol["rtol"] * 5, atol=tol["atol"] * 5)
        want_gn = sum(float(m["actor/grad_norm"]) for m in om) / len(om)
        assert metrics["train/actor/grad_norm"] == pytest.approx(want_gn, rel=2e-3)
        want_loss = sum(float(m["actor/total_loss"]) for m in om) / len(om) / 2  # 2 micro-batches per global batch
        assert metrics["train/actor/total_loss"] == pytest.approx(want_loss, rel=2e-3, abs=2e-4)
        all_r = torch.cat([b["rewards"] for b in batches], dim=1)
        assert metrics["rollout/rewards"] == pytest.approx(float(all_r.mean()), rel=1e-4)
        assert metrics["rollout/advantages_max"] == pytest.approx(max(float(b["advantages"].max()) for b in batches), rel=1e-3)
        diff = (w.model.flat.detach().cpu() - torch.cat([p.detach().reshape(-1) for p in ora.parameters()])).abs()
        steps_taken = len(om) * (it + 1)
        assert len(om) == epochs * (T * B // GB) * 2
        assert float(diff.max()) <= 2 * 3e-4 * steps_taken + 1e-6
        assert float((diff > 2e-5 * (it + 1)).float().mean()) < 0.02, float((diff > 2e-5).float().mean())


@pytest.mark.gpu
@pytest.mark.parametrize("auto_reset,entropy_bonus,path", [
    (True, 0.0, "fused"), (True, 0.01, "fused"), (False, 0.0, "fused"), (True, 0.01, "graph"), (False, 0.0, "graph"),
    (True, 0.01, "staged"), (False, 0.0, "one-micro-batch"), (True, 0.0, "threshold")])
def test_async_ppo_learner_matches_oracle(auto_reset, entropy_bonus, path):
    """AsyncPPOEmbodiedFSDPActor (decoupled actor-critic) behind the ordinary runner: two iterations whose trajectories carry
    the version they were sampled with (proximal policy interpolated from the version distance), then the same buffer trained
    again with half of it marked one version older and the proximal log-probs recomputed from the current weights.
    ``path``: the fused step (rlx_ppo_step with the decoupled loss + deferred actor scale; the default), the same replayed from
    a hipGraph (the policy version is read on the device), the stage-by-stage entry points (actor.fused_step false), one
    micro-batch per optimizer step, and a behaviour-weight threshold tight enough to mask samples."""
    from rlinf_amd.config import validate_cfg
    from rlinf_amd.runners import EmbodiedRunner
    from rlinf_amd.scheduler import init_distributed
    from rlinf_amd.workers.actor.async_ppo_fsdp_worker import AsyncPPOEmbodiedFSDPActor
    from rlinf_amd.workers.env import EnvWorker
    from rlinf_amd.workers.rollout.hf import MultiStepRolloutWorker
    T, B, GB, MB = 10, 16, 80, (80 if path == "one-micro-batch" else 40)
    cfg = make_cfg(total_envs=B, steps=T, global_batch=GB, micro_batch=MB, auto_reset=auto_reset, entropy_bonus=entropy_bonus,
                   hip_graph=path == "graph")
    cfg.algorithm.loss_type = "decoupled_actor_critic"
    cfg.algorithm.normalize_advantages = True
    if path == "staged":
        cfg.actor.fused_step = False
    if path == "threshold":
        cfg.algorithm.behave_weight_threshold = 1.02
    cfg = validate_cfg(cfg)
    env = L.synthetic_env_tensors(0, T, B, 42, max_episode_steps=5)
    torch.manual_seed(11)
    ora = O.OracleMLPPolicy(42, 8, 1)
    sd = copy.deepcopy(ora.state_dict())
    opt = O.build_adamw(ora)
    ctx = init_distributed()
    actor = AsyncPPOEmbodiedFSDPActor.create_group(cfg, ctx).launch(None, name="ActorGroup")
    runner = EmbodiedRunner(cfg, actor, MultiStepRolloutWorker.create_group(cfg, ctx).launch(None, name="RolloutGroup"),
                            EnvWorker.create_group(cfg, ctx).launch(None, name="EnvGroup"))
    runner.init_workers(env_tensors=env)
    w = actor.worker
    w.model.load_reference_state_dict(sd)
    kw = dict(seed=1234, global_batch=GB, micro_batch=MB, update_epoch=2, entropy_bonus=entropy_bonus,
              max_episode_steps=cfg.env.train.get("max_episode_steps"))
    if path == "threshold":
        kw["behave_weight_threshold"] = 1.02

    def check(metrics, om, norms, steps, it):
        mean = lambda k: sum(float(m[k]) for m in om) / len(om)  # noqa: E731
        tol = dict(rel=2e-3, abs=3e-4) if it == 0 else dict(rel=2e-2, abs=3e-3)
        for k in ("actor/policy_loss", "actor/proximal_ratio", "actor/clipped_proximal_ratio", "actor/clip_fraction",
                  "actor/dual_clip_fraction", "actor/proximal_approx_kl", "actor/behav_approx_kl", "critic/value_loss",
                  "critic/value_clip_ratio", "actor/total_loss", "actor/entropy_loss", "actor/average_version",
                  "actor/current_version"):
            assert metrics[k] == pytest.approx(mean(k), **tol), (it, k)
        assert metrics["actor/grad_norm"] == pytest.approx(sum(norms) / len(norms), rel=2e-3 if it == 0 else 2e-2)
        got = w.model.flat.detach().cpu()
        want = torch.cat([p.detach().reshape(-1) for p in ora.parameters()])
        diff = (got - want).abs()
        assert float(diff.max()) <= 2 * 3e-4 * steps + 1e-6
        assert float((diff > 2e-5 * (it + 1)).float().mean()) < 0.02, float((diff > 2e-5).float().mean())

    steps, batch = 0, None
    for it in range(2):
        eps = torch.randn(T, B, 8, generator=torch.Generator().manual_seed(100 + it))
        batch = L.advantages(L.rollout(ora, env, eps, 0.8, auto_reset), 0.8, 0.9, auto_reset)
        batch["versions"] = torch.full_like(batch["prev_logprobs"], float(it))
        om, norms = L.async_update(ora, opt, batch, version=it, **kw)
        metrics = runner.run_step(eps.cuda())
        rb = w.rollout_batch
        assert torch.equal(rb["versions"].cpu(), batch["versions"])
        torch.testing.assert_close(rb["advantages"].cpu(), batch["advantages"], rtol=1e-3 * (1 + 4 * it), atol=1e-4 * (1 + 4 * it))
        steps += len(norms)
        check({k[len("train/"):]: v for k, v in metrics.items() if k.startswith("train/")}, om, norms, steps, it)
    # the same buffer once more, half of it one version staler, proximal policy = the CURRENT weights, recomputed
    w.set_global_step(2)
    w.rollout_batch["versions"][: T // 2] -= 1.0
    batch["versions"][: T // 2] -= 1.0
    w.compute_proximal_logprobs()
    with torch.no_grad():
        flat_lp = ora.evaluate(batch["forward_inputs"]["states"].reshape(T * B, -1),
                               batch["forward_inputs"]["action"].reshape(T * B, -1))["logprobs"]
    batch["proximal_logprobs"] = flat_lp.view(T, B, -1)
    torch.testing.assert_close(w.rollout_batch["proximal_logprobs"].cpu(), batch["proximal_logprobs"], rtol=5e-3, atol=5e-4)
    om, norms = L.async_update(ora, opt, batch, version=2, **kw)
    metrics = w.run_training()
    steps += len(norms)
    check(metrics, om, norms, steps, 2)
    assert ("aplan_key" in w._ws) == (path != "staged"), "which learner loop ran"
    if path == "graph":
        assert "agraph" in w._ws


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [dict(hip_graph=True), dict(hip_graph=False), dict(hip_graph=True, pipeline=True, rollout_epoch=2),
                                   dict(hip_graph=True, learner="async")],
                         ids=["graph", "eager", "pipeline-e2", "async-learner-graph"])
def test_run_ahead_loop_reads_the_same_metrics_one_iteration_late(shape):
    """EmbodiedRunner.run() queues iteration i + 1 before it reads iteration i's metrics (runner.defer_metrics, the default): every
    iteration's metric dict and the final weights are identical, bit for bit, to the loop that reads each step's numbers before it
    queues the next (defer_metrics false), and every iteration is logged exactly once, in order."""
    E = shape.get("rollout_epoch", 1)
    T, B, n_iter = 16 * E, 64, 5
    env = L.synthetic_env_tensors(0, T, B, 42, max_episode_steps=5)
    torch.manual_seed(11)
    sd = copy.deepcopy(O.OracleMLPPolicy(42, 8, 1).state_dict())
    outs = []
    shape = dict(shape)
    learner = shape.pop("learner", "sync")
    for defer in (False, True):
        cfg = make_cfg(total_envs=B, steps=16, global_batch=256, **shape)
        if learner == "async":  # the decoupled learner's metric vector travels the same way (its replayed graph reads the version word)
            cfg.algorithm.loss_type, cfg.algorithm.behave_weight_threshold = "decoupled_actor_critic", 1.5
        cfg.runner.max_epochs = n_iter
        cfg.runner.defer_metrics = defer
        runner = _build(cfg, env, sd)
        runner.set_max_steps()
        eps = [torch.randn(T, B, 8, generator=torch.Generator().manual_seed(100 + it)).cuda() for it in range(n_iter)]
        hist = runner.run(eps_fn=lambda step: eps[step])
        assert len(hist) == n_iter and runner.global_step == n_iter
        outs.append(([{k: v for k, v in m.items() if not k.startswith(("time/", "perf/"))} for m in hist],
                     runner.actor.worker.model.flat.detach().cpu().clone()))
        runner.close()
    (want, w_want), (got, w_got) = outs
    assert torch.equal(w_want, w_got)
    for a, b in zip(want, got):
        assert a.keys() == b.keys()
        for k in a:
            assert a[k] == b[k] or (a[k] != a[k] and b[k] != b[k]), k
    assert len({m["train/actor/total_loss"] for m in got}) == n_iter  # (the iterations really differ: nothing was logged twice)


@pytest.mark.gpu
def test_non_auto_reset_builds_loss_mask_and_trains():
    """auto_reset=False -> loss mask + mask_sum ratio aggregation (embodied_fsdp_actor_worker.py:219-233, losses.py:219-227)."""
    cfg = make_cfg(total_envs=16, steps=12, global_batch=64, auto_reset=False)
    env = L.synthetic_env_tensors(3, 12, 16, 42, mode="bernoulli", p_done=0.08)
    torch.manual_seed(5)
    sd = O.OracleMLPPolicy(42, 8, 1).state_dict()
    runner = _build(cfg, env, sd)
    m = runner.run_step(torch.randn(12, 16, 8).cuda())
    rb = runner.actor.worker.rollout_batch
    want_mask, want_sum = O.loss_mask_from_dones(rb["dones"].cpu())
    assert torch.equal(rb["loss_mask"].cpu(), want_mask)
    assert torch.equal(rb["loss_mask_sum"].cpu(), want_sum)
    assert all(v == v for k, v in m.items() if k != "train/critic/explained_variance"), m  # no NaNs


@pytest.mark.gpu
@pytest.mark.parametrize("precision", ["32", "bf16"])
def test_example_entry_point_runs_the_shipped_config(precision):
    """examples/embodiment/train_embodied_agent.py with the reference's maniskill_ppo_mlp settings (1024 envs x 50 steps,
    8 epochs of one 6400-row... i.e. 51200 / 6400 = 8 minibatches): three iterations, finite metrics, the loss moves."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "examples", "embodiment", "train_embodied_agent.py"),
                          "--config-name", "maniskill_ppo_mlp", "runner.max_epochs=3", f"actor.model.precision={precision}"],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [json.loads(l) for l in out.stdout.splitlines() if l.startswith("{")]
    assert [l["step"] for l in lines] == [0, 1, 2]
    for l in lines:
        assert all(v == v and abs(v) < 1e9 for k, v in l.items() if isinstance(v, float)), l
    assert lines[0]["train/actor/total_loss"] != lines[2]["train/actor/total_loss"]
    assert lines[2]["perf/env_steps_per_sec"] > 1e5


@pytest.mark.gpu
@pytest.mark.parametrize("hip_graph,precision,A", [(False, "32", 8), (True, "32", 8), (True, "bf16", 8), (True, "32", 7), (False, "bf16", 7)])
def test_value_free_grpo_whole_loop_matches_oracle(hip_graph, precision, A):
    """actor.model.add_value_head False (mlp_policy.py:42-66: the value-free MLP policy of the reference's GRPO / actor-only PPO
    configurations): the policy exposes the reference's parameter set without a value head, prev_values are zeros, nothing is
    bootstrapped, and rollout -> GRPO advantages -> actor loss -> clip + AdamW match oracle.ppo_loop.iteration run on the
    reference-shaped value-free oracle policy (pinned to the reference's own MLPPolicy(add_value_head=False) in
    test_oracle_vs_reference.py) on identical seeds, weights, injected noise and shuffle order.
    A = 7 (any odd action_dim): the phantom value net is aligned to whole float4s, which leaves flat-buffer elements NO tensor owns
    between the exposed parameters and the phantom ones; the gradient kernels never write them and the norm / slab sum run over
    all n elements, so the slabs must be zero-filled once (round 4's advisor finding) -- garbage there reaches clip_grad_norm."""
    B, T, GB, group = 32, 12, 96, 4
    bf16 = precision == "bf16"
    cfg = make_cfg(total_envs=B, steps=T, global_batch=GB, auto_reset=False, hip_graph=hip_graph, done_mode="bernoulli")
    cfg.actor.model.add_value_head = False
    cfg.actor.model.action_dim = A
    cfg.actor.model.precision = precision
    alg = cfg.algorithm
    alg.adv_type, alg.loss_type, alg.group_size = "grpo", "actor", group
    cfg.env.train.group_size = group
    env = L.synthetic_env_tensors(0, T, B, 42, max_episode_steps=5, mode="bernoulli", p_done=0.05)
    torch.manual_seed(11)
    ora = O.OracleMLPPolicy(42, A, 1, add_value_head=False)
    sd = copy.deepcopy(ora.state_dict())
    opt = O.build_adamw(ora)
    if A % 2:  # poison the allocator's free blocks: a slab that is not zero-filled would pick the NaNs up in its unowned gap
        junk = [torch.full((1 << 20,), float("nan"), device="cuda") for _ in range(8)]
        del junk
    runner = _build(cfg, env, sd)
    w = runner.actor.worker
    assert not w.model.has_value_head and list(w.model.state_dict()) == list(sd)
    assert (w.model.n_exposed > w.model.exposed_numel) == bool(A % 2)  # the odd case really has unowned (alignment) elements
    for it in range(3 if hip_graph else 2):
        eps = torch.randn(T, B, A, generator=torch.Generator().manual_seed(100 + it))
        batch, om = L.iteration(ora, opt, env, eps, gamma=0.8, gae_lambda=0.9, seed=1234, global_batch=GB, update_epoch=2,
                                auto_reset=False, max_episode_steps=5, adv_type="grpo", loss_type="actor", group_size=group, autocast=bf16)
        metrics = runner.run_step(eps.cuda())
        rb = w.rollout_batch
        tol = dict(rtol=3e-2, atol=3e-2) if bf16 else (dict(rtol=2e-4, atol=2e-5) if it == 0 else dict(rtol=5e-3, atol=5e-4))
        torch.testing.assert_close(rb["forward_inputs"]["action"].cpu(), batch["forward_inputs"]["action"], **tol)
        torch.testing.assert_close(rb["prev_logprobs"].cpu(), batch["prev_logprobs"], **(tol if not bf16 else dict(rtol=5e-2, atol=5e-2)))
        assert not rb["prev_values"].any() and not batch["prev_values"].any()        # zeros on both sides
        assert torch.equal(rb["rewards"].cpu(), batch["rewards"])                       # nothing bootstrapped
        assert torch.equal(rb["loss_mask"].cpu(), batch["loss_mask"])
        if not bf16:
            torch.testing.assert_close(rb["advantages"].cpu(), batch["advantages"], rtol=tol["rtol"] * 5, atol=tol["atol"] * 5)
            for k in ("actor/total_loss", "actor/grad_norm", "actor/policy_loss", "actor/approx_kl"):
                want_k = sum(float(m[k]) for m in om) / len(om)
                assert metrics["train/" + k] == pytest.approx(want_k, rel=5e-3, abs=5e-5), k
        assert "train/critic/value_loss" not in metrics
        got = w.model.flat.detach().cpu()
        want = torch.cat([p.detach().reshape(-1) for p in ora.parameters()])
        diff = (w.model.exposed_flat().cpu() - want).abs()
        assert float(diff.max()) <= 2 * 3e-4 * len(om) * (it + 1) + 1e-6
        assert not got[w.model.n_exposed:].any()                                        # the phantom value net stays zero
        owned = torch.zeros(w.model.n_params, dtype=torch.bool)
        for nm in w.model.shapes:
            owned[w.model.offsets[nm]:w.model.offsets[nm] + w.model.view(nm).numel()] = True
        assert not got[~owned].any()                                                    # ... and so does every alignment gap
    runner.close() if hasattr(runner, "close") else None


@pytest.mark.gpu
def test_embodied_grpo_iteration():
    """adv_type grpo + loss_type actor on the MLP policy (the reference's libero_spatial_0_grpo_mlp.yaml pairing): the
    advantages match the oracle's GRPO on the rollout the runner produced, the actor trains, and the value head -- whose
    gradients are None in the reference (values are only computed for adv_type gae, :623) -- is left bit-for-bit alone."""
    cfg = make_cfg(total_envs=32, steps=10, global_batch=160, auto_reset=False)
    cfg.algorithm.adv_type, cfg.algorithm.loss_type, cfg.algorithm.group_size = "grpo", "actor", 4
    cfg.env.train.group_size = 4
    env = L.synthetic_env_tensors(3, 10, 32, 42, mode="bernoulli", p_done=0.06)
    torch.manual_seed(5)
    sd = O.OracleMLPPolicy(42, 8, 1).state_dict()
    runner = _build(cfg, env, sd)
    model = runner.actor.worker.model
    before = model.flat.detach().clone()
    vh = [(model.offsets[n], model.offsets[n] + model.view(n).numel()) for n in model.shapes if "value_head" in n]
    m = runner.run_step(torch.randn(10, 32, 8).cuda())
    rb = runner.actor.worker.rollout_batch
    lm, lms = O.loss_mask_from_dones(rb["dones"].cpu())
    want = O.embodied_adv_and_returns(adv_type="grpo", rewards=rb["rewards"].cpu(), dones=rb["dones"].cpu(), loss_mask=lm,
                                      loss_mask_sum=lms, group_size=4)
    torch.testing.assert_close(rb["advantages"].cpu(), want["advantages"].contiguous(), rtol=1e-5, atol=1e-6)
    after = model.flat.detach()
    for b, e in vh:
        assert torch.equal(after[b:e], before[b:e])
    assert float((after - before).abs().max()) > 0
    assert all(v == v for k, v in m.items() if isinstance(v, float) and "explained_variance" not in k), m
    assert "train/critic/value_loss" not in m

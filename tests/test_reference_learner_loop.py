"""The oracle's UPDATE LOOP against the reference's own learner methods, executed on CPU.

rlinf/workers/actor/embodied_fsdp_actor_worker.py cannot be imported here (Ray, FSDP, hydra), so `run_training` (:483-589),
`train_micro_batch` (:591-700) and `FSDPModelManager.optimizer_step` / `build_optimizer`
(rlinf/hybrid_engines/fsdp/fsdp_model_manager.py:429-463,501-590) are compiled from their source on their own and run as
plain functions over a stand-in ``self`` that supplies what FSDP would: the REAL reference MLPPolicy as ``model``, a
pass-through gradient scaler, ``clip_grad_norm_`` = torch's (the no_shard branch, strategy/fsdp.py:363-369), no-op device
moves.  Everything that computes -- shuffling, chunking, policy_loss, entropy bonus, accumulation, clipping, AdamW, critic
warm-up with its optimizer rebuild, metric averaging -- is the reference's code.  oracle.ppo_loop.update must land on the same
parameters and the same metrics: the GPU end-to-end tests compare the HIP learner with that oracle loop.
"""

import contextlib
import copy
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from conftest import free_port
import torch.distributed as dist

from oracle import ppo_loop as L
from oracle import ppo_oracle as O

pytestmark = pytest.mark.reference


class Cfg(dict):
    __getattr__ = dict.__getitem__


def _learner(ref, policy, batch, *, global_batch, micro_batch, update_epoch, entropy_bonus, critic_warmup_steps, auto_reset,
             device="cpu"):
    """``device="cuda"`` (tests/test_gpu_ext_model_registry.py): the same stand-in learner with the reference's own
    put_tensor_device moving every micro-batch to the accelerator, as the real worker does (embodied_fsdp_actor_worker.py:598)."""
    from oracle import reference_loader as R
    mu, utils, nested = ref.metric_utils, ref.utils, ref.nested
    worker_stub = SimpleNamespace(torch_device_type=device, timer=lambda *_a, **_k: (lambda f: f),
                                  torch_platform=SimpleNamespace(current_device=lambda: torch.device(device), is_available=lambda: False))
    models = SimpleNamespace(OPENVLA="openvla", OPENVLA_OFT="openvla_oft", GR00T="gr00t", GR00T_N1D6="gr00t_n1d6",
                             GR00T_N1D7="gr00t_n1d7", ABOT_M0="abot_m0")
    supported = type("SupportedModel", (), {"__new__": staticmethod(lambda cls, name: name), **vars(models)})
    actor_py = "rlinf/workers/actor/embodied_fsdp_actor_worker.py"
    manager_py = "rlinf/hybrid_engines/fsdp/fsdp_model_manager.py"
    run_training = R.load_function(
        actor_py, "EmbodiedFSDPActor.run_training", torch=torch, np=np, process_nested_dict_for_train=nested.process_nested_dict_for_train,
        split_dict_to_chunk=nested.split_dict_to_chunk, append_to_dict=mu.append_to_dict, clear_memory=lambda: None,
        pop_critic_explained_variance_stats=mu.pop_critic_explained_variance_stats, all_reduce_dict=lambda d, op=None: d,
        compute_critic_explained_variance_from_stats=mu.compute_critic_explained_variance_from_stats,
        CRITIC_EXPLAINED_VARIANCE_KEY=mu.CRITIC_EXPLAINED_VARIANCE_KEY)
    train_micro_batch = R.load_function(
        actor_py, "EmbodiedFSDPActor.train_micro_batch", torch=torch,
        put_tensor_device=(lambda d, _dev: d) if device == "cpu" else nested.put_tensor_device, SupportedModel=supported,
        policy_loss=ref.registry.policy_loss, Worker=worker_stub, reshape_entropy=utils.reshape_entropy, masked_mean=utils.masked_mean,
        append_to_dict=mu.append_to_dict)
    optimizer_step = R.load_function(manager_py, "FSDPModelManager.optimizer_step", torch=torch)
    build_optimizer = R.load_function(manager_py, "FSDPModelManager.build_optimizer", torch=torch, Worker=worker_stub,
                                      warmup_optimizer_state=utils.warmup_optimizer_state)
    optim = Cfg(lr=3e-4, value_lr=3e-4, adam_beta1=0.9, adam_beta2=0.999, adam_eps=1e-8, weight_decay=0.01, clip_grad=0.5,
                critic_warmup_steps=critic_warmup_steps)
    alg = Cfg(loss_type="actor_critic", logprob_type="action_level", reward_type="action_level", entropy_type="action_level",
              adv_type="gae", entropy_bonus=entropy_bonus, clip_ratio_high=0.2, clip_ratio_low=0.2, value_clip=1.0, huber_delta=10.0,
              update_epoch=update_epoch)
    cfg = Cfg(algorithm=alg, runner=Cfg(task_type="embodied"), env=Cfg(train=Cfg(max_episode_steps=5, auto_reset=auto_reset)),
              rollout=Cfg(), actor=Cfg(seed=1234, global_batch_size=global_batch, micro_batch_size=micro_batch, optim=optim,
                                       model=Cfg(model_type="mlp_policy", action_dim=8)))
    me = SimpleNamespace(
        cfg=cfg, _cfg=Cfg(optim=optim, fsdp_config={"sharding_strategy": "no_shard"}), model=policy, rollout_batch=batch, _rank=0,
        _world_size=1, device=device, is_weight_offloaded=False, is_optimizer_offloaded=False, enable_sft_co_train=False,
        gradient_accumulation=global_batch // micro_batch, optimizer_steps=0, critic_warmup_steps=critic_warmup_steps,
        amp_context=contextlib.nullcontext(), before_micro_batch=lambda *_a, **_k: contextlib.nullcontext(),
        torch_platform=SimpleNamespace(empty_cache=lambda: None), store_requires_grad_param_name=[],
        _logger=SimpleNamespace(info=lambda *a: None, warning=lambda *a: None),
        grad_scaler=SimpleNamespace(scale=lambda loss: loss, unscale_=lambda _opt: None, update=lambda: None,
                                    step=lambda optimizer: optimizer.step()),
        _strategy=SimpleNamespace(clip_grad_norm_=lambda model: float(torch.nn.utils.clip_grad_norm_(model.parameters(), 0.5, 2.0))),
        lr_scheduler=SimpleNamespace(step=lambda: None), build_lr_scheduler=lambda **_k: SimpleNamespace(step=lambda: None))
    me.build_optimizer = lambda model, enable_critic_warmup=False: build_optimizer(me, model, enable_critic_warmup)
    me.optimizer = me.build_optimizer(policy, enable_critic_warmup=critic_warmup_steps > 0)  # fsdp_model_manager.py:304-306
    me.optimizer_step = lambda: optimizer_step(me)
    me.train_micro_batch = lambda micro_batch, metrics, is_last: train_micro_batch(me, micro_batch, metrics, is_last=is_last)
    me.run_training = lambda: run_training(me)
    return me


@pytest.fixture(scope="module")
def one_rank_group():
    started = not dist.is_initialized()
    if started:
        dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{free_port()}", rank=0, world_size=1)
    yield
    if started:
        dist.destroy_process_group()


@pytest.mark.parametrize("shape", [dict(global_batch=40, micro_batch=40), dict(global_batch=80, micro_batch=20),
                                   dict(global_batch=40, micro_batch=40, entropy_bonus=0.01),
                                   dict(global_batch=40, micro_batch=40, auto_reset=False),
                                   dict(global_batch=40, micro_batch=20, critic_warmup_steps=5)])
def test_update_loop_matches_the_reference_learner(ref, one_rank_group, shape):
    T, B, epochs = 10, 16, 2
    auto_reset = shape.get("auto_reset", True)
    env = L.synthetic_env_tensors(0, T, B, 42, max_episode_steps=5)
    torch.manual_seed(11)
    theirs = ref.mlp_policy.MLPPolicy(42, 8, 1, True, False)
    ours = O.OracleMLPPolicy(42, 8, 1)
    ours.load_state_dict(copy.deepcopy(theirs.state_dict()), strict=True)
    eps = torch.randn(T, B, 8, generator=torch.Generator().manual_seed(100))
    batch = L.advantages(L.rollout(ours, env, eps, 0.8, auto_reset), 0.8, 0.9, auto_reset)
    cw = shape.get("critic_warmup_steps", 0)
    me = _learner(ref, theirs, copy.deepcopy(batch), global_batch=shape["global_batch"], micro_batch=shape["micro_batch"],
                  update_epoch=epochs, entropy_bonus=shape.get("entropy_bonus", 0.0), critic_warmup_steps=cw, auto_reset=auto_reset)
    want = me.run_training()
    opt = O.build_adamw(ours)
    accum = shape["global_batch"] // shape["micro_batch"]
    if accum == 1:
        om = L.update(ours, opt, batch, seed=1234, global_batch=shape["global_batch"], update_epoch=epochs,
                      entropy_bonus=shape.get("entropy_bonus", 0.0), critic_warmup_steps=cw, max_episode_steps=5)
        steps = len(om)
    else:  # gradient accumulation: the oracle's async loop has it; the loss here is the plain actor-critic one
        steps = None
    n_steps = (T * B // shape["global_batch"]) * epochs
    assert me.optimizer_steps == n_steps
    if steps is not None:
        assert steps == n_steps
        for (n, p), (_, q) in zip(theirs.named_parameters(), ours.named_parameters()):
            assert torch.equal(p.detach(), q.detach()), n
        mean = lambda k: float(np.mean([float(m[k]) for m in om]))  # noqa: E731
        for k in ("actor/policy_loss", "actor/approx_kl", "actor/clip_fraction", "critic/value_loss", "actor/grad_norm",
                  "actor/entropy_loss"):
            if k in om[0]:
                assert want[k] == pytest.approx(mean(k), rel=1e-6, abs=1e-9), k
        assert want["actor/total_loss"] == pytest.approx(mean("actor/total_loss"), rel=1e-6, abs=1e-9)
    else:
        # accumulation: the same global batches cut into micro-batches must give (to rounding) the parameters of the
        # un-accumulated run -- mean-of-means over equal micro-batches equals the mean over the global batch when no loss mask
        # re-weights them
        fresh = ref.mlp_policy.MLPPolicy(42, 8, 1, True, False)
        fresh.load_state_dict(copy.deepcopy(ours.state_dict()), strict=True)  # `ours` still holds the initial weights
        single = _learner(ref, fresh, copy.deepcopy(batch), global_batch=shape["global_batch"],
                          micro_batch=shape["global_batch"], update_epoch=epochs, entropy_bonus=0.0, critic_warmup_steps=cw,
                          auto_reset=auto_reset)
        single.run_training()
        for (n, p), (_, q) in zip(theirs.named_parameters(), single.model.named_parameters()):
            torch.testing.assert_close(p.detach(), q.detach(), rtol=2e-4, atol=2e-6, msg=n)


class _TorchOnCpu:
    """``torch`` as the async learner's module sees it, minus the two CUDA calls it makes (current_device / empty_cache)."""

    cuda = SimpleNamespace(current_device=lambda: "cpu", empty_cache=lambda: None)

    def __getattr__(self, name):
        return getattr(torch, name)


@pytest.mark.parametrize("shape", [dict(global_batch=40, micro_batch=40), dict(global_batch=80, micro_batch=20, entropy_bonus=0.01),
                                   dict(global_batch=40, micro_batch=20, auto_reset=False),
                                   dict(global_batch=80, micro_batch=40, stale=True, threshold=1.02)])
def test_async_update_loop_matches_the_reference_learner(ref, one_rank_group, shape, monkeypatch):
    """AsyncPPOEmbodiedFSDPActor.run_training (async_ppo_fsdp_worker.py:274-497) compiled on its own over the stand-in learner
    above (real MLPPolicy, real policy_loss / masked_normalization / optimizer_step): oracle.ppo_loop.async_update -- what the
    HIP async learner is compared with on the GPU -- must reproduce its parameters and metrics, with gradient accumulation,
    entropy bonus, a loss mask (ratio aggregation), mixed policy versions and a behaviour-weight threshold."""
    import os
    from typing import Any

    from oracle import reference_loader as R
    monkeypatch.setattr(torch.Tensor, "cuda", lambda self, *a, **k: self)  # masked_normalization moves its inputs
    monkeypatch.setenv("LOCAL_RANK", "0")
    du = R.load_distributed_utils()
    T, B, epochs = 10, 16, 2
    auto_reset = shape.get("auto_reset", True)
    env = L.synthetic_env_tensors(0, T, B, 42, max_episode_steps=5)
    torch.manual_seed(11)
    theirs = ref.mlp_policy.MLPPolicy(42, 8, 1, True, False)
    ours = O.OracleMLPPolicy(42, 8, 1)
    ours.load_state_dict(copy.deepcopy(theirs.state_dict()), strict=True)
    eps = torch.randn(T, B, 8, generator=torch.Generator().manual_seed(100))
    batch = L.advantages(L.rollout(ours, env, eps, 0.8, auto_reset), 0.8, 0.9, auto_reset)
    batch["versions"] = torch.full_like(batch["prev_logprobs"], 3.0)
    if shape.get("stale"):
        batch["versions"][: T // 2] = 1.0
        batch["prev_logprobs"] = batch["prev_logprobs"] + 0.05 * torch.randn(T, B, 8, generator=torch.Generator().manual_seed(7))
    thr = shape.get("threshold")
    me = _learner(ref, theirs, copy.deepcopy(batch), global_batch=shape["global_batch"], micro_batch=shape["micro_batch"],
                  update_epoch=epochs, entropy_bonus=shape.get("entropy_bonus", 0.0), critic_warmup_steps=0, auto_reset=auto_reset)
    me.cfg.algorithm.update(loss_type="decoupled_actor_critic", normalize_advantages=True, **({"behave_weight_threshold": thr} if thr else {}))
    me.version = 3
    flatten = R.load_function("rlinf/workers/actor/async_ppo_fsdp_worker.py", "flatten_rollout_batch_for_train", torch=torch,
                              Optional=__import__("typing").Optional)
    models = SimpleNamespace(OPENVLA="openvla", OPENVLA_OFT="openvla_oft", GR00T="gr00t", ABOT_M0="abot_m0")
    supported = type("SupportedModel", (), {"__new__": staticmethod(lambda cls, name: name), **vars(models)})
    mu, utils = ref.metric_utils, ref.utils
    run_training = R.load_function(
        "rlinf/workers/actor/async_ppo_fsdp_worker.py", "AsyncPPOEmbodiedFSDPActor.run_training", torch=_TorchOnCpu(), np=np, os=os,
        Any=Any, flatten_rollout_batch_for_train=flatten, masked_normalization=du.masked_normalization,
        split_dict_to_chunk=ref.nested.split_dict_to_chunk, put_tensor_device=lambda d, _dev: d, SupportedModel=supported,
        policy_loss=ref.registry.policy_loss, reshape_entropy=utils.reshape_entropy, masked_mean=utils.masked_mean,
        append_to_dict=mu.append_to_dict, clear_memory=lambda: None, all_reduce_dict=lambda d, op=None: d,
        pop_critic_explained_variance_stats=mu.pop_critic_explained_variance_stats,
        compute_critic_explained_variance_from_stats=mu.compute_critic_explained_variance_from_stats,
        CRITIC_EXPLAINED_VARIANCE_KEY=mu.CRITIC_EXPLAINED_VARIANCE_KEY)
    want = run_training(me)
    opt = O.build_adamw(ours)
    om, norms = L.async_update(ours, opt, batch, seed=1234, global_batch=shape["global_batch"], micro_batch=shape["micro_batch"],
                               update_epoch=epochs, version=3, entropy_bonus=shape.get("entropy_bonus", 0.0),
                               behave_weight_threshold=thr, max_episode_steps=5)
    assert me.optimizer_steps == len(norms) == (T * B // shape["global_batch"]) * epochs
    for (n, p), (_, q) in zip(theirs.named_parameters(), ours.named_parameters()):
        assert torch.equal(p.detach(), q.detach()), n
    mean = lambda k: float(np.mean([float(m[k]) for m in om if k in m]))  # noqa: E731
    for k in ("actor/policy_loss", "actor/proximal_ratio", "actor/clipped_proximal_ratio", "actor/clip_fraction",
              "actor/dual_clip_fraction", "actor/behav_clip_fraction", "actor/proximal_approx_kl", "actor/behav_approx_kl",
              "actor/average_version", "actor/current_version", "critic/value_loss", "critic/value_clip_ratio",
              "actor/entropy_loss", "actor/total_loss"):
        assert want[k] == pytest.approx(mean(k), rel=1e-6, abs=1e-9), k
    assert want["actor/grad_norm"] == pytest.approx(float(np.mean(norms)), rel=1e-6)


@pytest.mark.parametrize("auto_reset", [True, False])
def test_pipeline_data_path_matches_the_reference_env_worker(ref, auto_reset):
    """runner.use_training_pipeline: EnvWorker.send_rollout_trajectories_pipeline with prepare_pipeline_batch,
    compute_advantages_and_returns and pack_pipeline_micro_batches (env_worker.py:1469-1600), compiled on their own and run
    over REAL EmbodiedTrajectoryBuilders (one per stage) -- un-normalised GAE per stage batch, (count, sum, sumsq) summed
    over the stages of the rank, normalisation from those, per-stage stateful shuffles, micro-batches in channel order --
    against oracle.ppo_loop.pipeline_advantages + pipeline_permutation, which the HIP pipeline path is compared with."""
    import asyncio
    from collections import defaultdict

    from oracle import reference_loader as R
    m = R.load_trajectory_builder()
    du = R.load_distributed_utils()
    env_py = "rlinf/workers/env/env_worker.py"
    T, B, stages, micro = 10, 16, 2, 20
    n = B // stages
    env = L.synthetic_env_tensors(0, T, B, 42, max_episode_steps=5)
    torch.manual_seed(11)
    pol = O.OracleMLPPolicy(42, 8, 1)
    eps = torch.randn(T, B, 8, generator=torch.Generator().manual_seed(100))
    raw = L.rollout(pol, env, eps, 0.8, auto_reset)
    # the reference's per-stage builders, filled with the rows of the rank's buffer that belong to the stage (SURVEY A.1 order)
    builders = []
    for st in range(stages):
        sl = slice(st * n, (st + 1) * n)
        b = m.builder.EmbodiedTrajectoryBuilder(max_episode_length=5)
        for t in range(T + 1):
            fields = dict(dones=raw["dones"][t, sl].clone(), terminations=raw["dones"][t, sl].clone(),
                          truncations=torch.zeros(n, 1, dtype=torch.bool), rewards=None if t == 0 else raw["rewards"][t - 1, sl].clone(),
                          prev_values=raw["prev_values"][t, sl].clone())
            if t < T:
                fields.update(actions=raw["forward_inputs"]["action"][t, sl].clone(), prev_logprobs=raw["prev_logprobs"][t, sl].clone(),
                              forward_inputs={"states": raw["forward_inputs"]["states"][t, sl].clone(),
                                              "action": raw["forward_inputs"]["action"][t, sl].clone()})
            b.append_step_result(m.types.ChunkStepResult(**fields))
        builders.append(b)
    alg = Cfg(adv_type="gae", gamma=0.8, gae_lambda=0.9, group_size=1, reward_type="action_level", normalize_advantages=True)
    cfg = Cfg(algorithm=alg, runner=Cfg(task_type="embodied"), env=Cfg(train=Cfg(auto_reset=auto_reset, ignore_terminations=False)),
              actor=Cfg(micro_batch_size=micro, model=Cfg(num_action_chunks=1)))
    sent = []
    me = SimpleNamespace(cfg=cfg, use_training_pipeline=True, shuffle_rollout=True, _rank=0, _group_name="env",
                         shuffle_generators={0: torch.Generator().manual_seed(1234)}, pipeline_actor_env_ranks={0: [0]},
                         pipeline_actor_keys={0: "actor0"}, pipeline_stage_actor_splits={st: [(0, n)] for st in range(stages)},
                         worker_timer=lambda *_a: contextlib.nullcontext(), broadcast=lambda obj, groups=None, src=None: obj)
    bind = lambda name, **g: (lambda f: (lambda *a, **k: f(me, *a, **k)))(R.load_function(env_py, f"EnvWorker.{name}", torch=torch, **g))  # noqa: E731
    me.compute_advantages_and_returns = bind("compute_advantages_and_returns", calculate_adv_and_returns=ref.registry.calculate_adv_and_returns)
    me.prepare_pipeline_batch = bind("prepare_pipeline_batch", convert_trajectories_to_batch=m.types.convert_trajectories_to_batch,
                                     preprocess_embodied_batch=ref.utils.preprocess_embodied_batch)
    me.pack_pipeline_micro_batches = bind("pack_pipeline_micro_batches", flatten_embodied_batch=ref.utils.flatten_embodied_batch,
                                          pack_batch=ref.utils.pack_batch, split_dict_to_chunk=ref.nested.split_dict_to_chunk)
    send = bind("send_rollout_trajectories_pipeline", defaultdict=defaultdict, masked_stats=du.masked_stats,
                normalize_from_stats=du.normalize_from_stats)
    channel = SimpleNamespace(put=lambda item, key=None, async_op=False: sent.append((key, item)))
    asyncio.run(send(builders, channel))
    assert [k for k, _ in sent] == ["actor0"] * (T * B // micro)
    # the oracle: one pass over the rank's whole [T, B] buffer, then the per-stage permutation
    batch = L.pipeline_advantages(raw, 0.8, 0.9, auto_reset)
    perm = L.pipeline_permutation(T, B, stages, torch.Generator().manual_seed(1234))
    flat = O.flatten_and_shuffle(batch, perm)
    keys = ["advantages", "returns", "prev_logprobs", "prev_values", "rewards", "dones"] + ([] if auto_reset else ["loss_mask", "loss_mask_sum"])
    for k in keys:
        got = torch.cat([item[k] for _, item in sent])
        assert got.shape == flat[k].shape and torch.equal(got, flat[k]), k
    assert torch.equal(torch.cat([item["forward_inputs::states"] for _, item in sent]), flat["forward_inputs"]["states"])


@pytest.mark.parametrize("auto_reset,rollout_epoch", [(True, 1), (False, 1), (True, 2)])
def test_whole_iteration_matches_the_reference_learner(ref, one_rank_group, auto_reset, rollout_epoch, monkeypatch):
    """One whole learner iteration by the reference's own methods: _process_received_rollout_batch (epoch fold, loss mask) ->
    compute_advantages_and_returns (+ compute_rollout_metrics) -> run_training, fed from real trajectory builders; against
    oracle.ppo_loop.iteration on the same rollout -- batch tensors, rollout metrics and final parameters."""
    import sys
    import types

    from oracle import reference_loader as R
    m = R.load_trajectory_builder()
    wmod = types.ModuleType("rlinf.scheduler.worker.worker")
    wmod.Worker = SimpleNamespace(torch_platform=SimpleNamespace(current_device=lambda: torch.device("cpu")))
    monkeypatch.setitem(sys.modules, "rlinf.scheduler.worker", types.ModuleType("rlinf.scheduler.worker"))
    monkeypatch.setitem(sys.modules, "rlinf.scheduler.worker.worker", wmod)
    T, B, E, GB = 10, 8, rollout_epoch, 40
    env = L.synthetic_env_tensors(0, T * E, B, 42, max_episode_steps=5)
    torch.manual_seed(11)
    theirs = ref.mlp_policy.MLPPolicy(42, 8, 1, True, False)
    ours = O.OracleMLPPolicy(42, 8, 1)
    ours.load_state_dict(copy.deepcopy(theirs.state_dict()), strict=True)
    eps = torch.randn(T * E, B, 8, generator=torch.Generator().manual_seed(100))
    # what the env worker appends: E epochs of (bootstrap row, T step rows, closing row) stacked on the time axis
    builder = m.builder.EmbodiedTrajectoryBuilder(max_episode_length=5)
    for e in range(E):
        sl = dict(obs=env["obs"][e * T:(e + 1) * T + 1], final_obs=env["final_obs"][e * T:(e + 1) * T],
                  rewards=env["rewards"][e * T:(e + 1) * T], dones=env["dones"][e * T:(e + 1) * T + 1])
        raw = L.rollout(ours, sl, eps[e * T:(e + 1) * T], 0.8, auto_reset)
        for t in range(T + 1):
            fields = dict(dones=raw["dones"][t].clone(), terminations=raw["dones"][t].clone(), truncations=torch.zeros(B, 1, dtype=torch.bool),
                          rewards=None if t == 0 else raw["rewards"][t - 1].clone(), prev_values=raw["prev_values"][t].clone())
            if t < T:
                fields.update(actions=raw["forward_inputs"]["action"][t].clone(), prev_logprobs=raw["prev_logprobs"][t].clone(),
                              forward_inputs={k: v[t].clone() for k, v in raw["forward_inputs"].items()})
            builder.append_step_result(m.types.ChunkStepResult(**fields))
    me = _learner(ref, theirs, None, global_batch=GB, micro_batch=GB, update_epoch=2, entropy_bonus=0.0, critic_warmup_steps=0,
                  auto_reset=auto_reset)
    me.cfg.env.train.update(rollout_epoch=E, ignore_terminations=False)
    me.cfg.algorithm.update(gamma=0.8, gae_lambda=0.9, group_size=1)
    me.cfg.actor.model.update(num_action_chunks=1)
    actor_py = "rlinf/workers/actor/embodied_fsdp_actor_worker.py"
    process = R.load_function(actor_py, "EmbodiedFSDPActor._process_received_rollout_batch", torch=torch,
                              process_nested_dict_for_adv=ref.nested.process_nested_dict_for_adv,
                              compute_loss_mask=ref.metric_utils.compute_loss_mask)
    compute_adv = R.load_function(actor_py, "EmbodiedFSDPActor.compute_advantages_and_returns", torch=torch,
                                  calculate_adv_and_returns=ref.registry.calculate_adv_and_returns,
                                  compute_rollout_metrics=ref.metric_utils.compute_rollout_metrics)
    me.rollout_batch = process(me, m.types.convert_trajectories_to_batch([builder.to_trajectory()]))
    want_rollout_metrics = compute_adv(me)
    want_batch = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in me.rollout_batch.items()}
    want_train = me.run_training()
    opt = O.build_adamw(ours)
    batch, om = L.iteration(ours, opt, env, eps, gamma=0.8, gae_lambda=0.9, seed=1234, global_batch=GB, update_epoch=2,
                            auto_reset=auto_reset, rollout_epoch=E, max_episode_steps=5)
    for k in ("rewards", "dones", "prev_values", "prev_logprobs", "advantages", "returns") + (() if auto_reset else ("loss_mask", "loss_mask_sum")):
        assert want_batch[k].shape == batch[k].shape and torch.equal(want_batch[k], batch[k]), k
    assert want_rollout_metrics["rewards"] == pytest.approx(float(batch["rewards"][batch["loss_mask"]].mean() if not auto_reset
                                                                  else batch["rewards"].mean()), rel=1e-6)
    for (n, p), (_, q) in zip(theirs.named_parameters(), ours.named_parameters()):
        assert torch.equal(p.detach(), q.detach()), n
    assert want_train["actor/total_loss"] == pytest.approx(float(np.mean([float(x["actor/total_loss"]) for x in om])), rel=1e-6)


@pytest.mark.parametrize("case", [dict(loss_agg="token-mean", temperature=1.0), dict(loss_agg="seq-mean-token-sum", temperature=0.7, entropy_bonus=0.01),
                                  dict(loss_agg="seq-mean-token-mean", temperature=1.3, kl_beta=0.05, kl="low_var_kl"),
                                  dict(loss_agg="token-mean", temperature=1.0, dtype=torch.bfloat16, entropy_bonus=0.02, kl_beta=0.1, kl="k2")])
def test_reasoning_training_step_matches_the_reference_learner(ref, one_rank_group, case):
    """FSDPActor.training_step + forward_batch (fsdp_actor_worker.py:434-505,659-813), compiled on their own; the model is a
    stand-in that returns given logits (the transformer is out of this path), two micro-batches of gradient accumulation:
    d(loss)/d(logits) and the step's metrics against oracle.token_oracle.reasoning_micro_batch_loss, which the HIP token
    kernels are compared with."""
    from oracle import reference_loader as R
    from oracle import token_oracle as TO
    from oracle.make_golden import token_batch
    bsz, resp, prompt, vocab = 8, 12, 5, 37
    dt = case.get("dtype", torch.float32)
    b = token_batch(301, bsz, resp, vocab)
    g = torch.Generator().manual_seed(9)
    full_logits = (torch.randn(bsz, prompt + resp, vocab, generator=g) * 2).to(dt)
    full_logits[:, -resp - 1:-1] = b["logits"].to(dt)  # the response window the learner slices out
    input_ids = torch.randint(0, vocab, (bsz, prompt + resp), generator=g)
    input_ids[:, -resp:] = b["labels"]
    fwd_py = "rlinf/workers/actor/fsdp_actor_worker.py"
    worker_stub = SimpleNamespace(torch_device_type="cpu", torch_platform=SimpleNamespace(current_device=lambda: torch.device("cpu")))

    class Model(torch.nn.Module):
        def __init__(self, logits):
            super().__init__()
            self.logits = torch.nn.Parameter(logits.clone())
            self.cursor = 0

        def forward(self, input_ids=None, **_kw):
            n = input_ids.shape[0]
            out = self.logits[self.cursor:self.cursor + n] * 1  # a non-leaf: the learner divides it by the temperature in place
            self.cursor += n
            return SimpleNamespace(logits=out)

    model = Model(full_logits)
    bonus, beta = case.get("entropy_bonus", 0.0), case.get("kl_beta", 0.0)
    alg = Cfg(sampling_params=Cfg(temperature=case["temperature"]), ratio_clip_eps=0.2, clip_ratio_high=0.28, loss_type="actor",
              entropy_bonus=bonus)
    me = SimpleNamespace(
        cfg=Cfg(algorithm=alg), model=model, response_len=resp, enable_dynamic_batch_size=False, variable_seq_lengths=False,
        amp_context=contextlib.nullcontext(), before_micro_batch=lambda *_a, **_k: contextlib.nullcontext(), task_type="reasoning",
        loss_agg_func=ref.utils.get_loss_agg_func(case["loss_agg"]), calculate_entropy=bonus > 0, calculate_entropy_loss=bonus > 0,
        kl_beta=beta, kl_penalty_type=case.get("kl", "low_var_kl"), entropy_op_type="torch", micro_batch_size=4,
        total_batch_size_per_dp=bsz, n_mini_batches=1, gradient_accumulation=None, lr_sched_sync_with_optim=False,
        optimizer=SimpleNamespace(zero_grad=lambda: None), optimizer_step=lambda: (1.5, [1e-6]),
        grad_scaler=SimpleNamespace(scale=lambda loss: loss))
    me.compute_logprobs = lambda logits, target: ref.utils.compute_logprobs_from_logits(logits, target, op_type="torch")
    forward_batch = R.load_function(fwd_py, "FSDPActor.forward_batch", torch=torch, Worker=worker_stub,
                                    compute_entropy_from_logits=ref.utils.compute_entropy_from_logits)
    me.forward_batch = lambda m_batch, calculate_entropy=False: forward_batch(me, m_batch, calculate_entropy)
    mu = ref.metric_utils
    training_step = R.load_function(
        fwd_py, "FSDPActor.training_step", torch=torch, Worker=worker_stub, policy_loss=ref.registry.policy_loss,
        kl_penalty=ref.algo_utils.kl_penalty, append_to_dict=mu.append_to_dict, compute_rollout_train_kl=lambda *_a: None,
        pop_critic_explained_variance_stats=mu.pop_critic_explained_variance_stats, all_reduce_dict=lambda d, op=None: d,
        compute_critic_explained_variance_from_stats=mu.compute_critic_explained_variance_from_stats,
        CRITIC_EXPLAINED_VARIANCE_KEY=mu.CRITIC_EXPLAINED_VARIANCE_KEY, BatchResizingIterator=type("BatchResizingIterator", (), {}))
    rows = lambda lo, hi: {"input_ids": input_ids[lo:hi], "attention_mask": torch.ones(hi - lo, prompt + resp, dtype=torch.bool),  # noqa: E731
                           "position_ids": torch.arange(prompt + resp).expand(hi - lo, -1), "rollout_logprobs": b["old_logprobs"][lo:hi],
                           "advantages": b["advantages"][lo:hi], "ref_logprobs": b["ref_logprobs"][lo:hi],
                           "response_mask": torch.cat([torch.zeros(hi - lo, prompt, dtype=torch.bool), b["loss_mask"][lo:hi]], dim=1)}
    want = training_step(me, iter([rows(0, 4), rows(4, 8)]))
    want_grad = model.logits.grad[:, -resp - 1:-1].float()
    assert not model.logits.grad[:, :-resp - 1].any() and not model.logits.grad[:, -1].any()
    # the oracle, micro-batch by micro-batch on the response window
    window = full_logits[:, -resp - 1:-1].clone().requires_grad_(True)
    metrics = []
    for lo, hi in ((0, 4), (4, 8)):
        loss, mt, _, _ = TO.reasoning_micro_batch_loss(
            window[lo:hi], b["labels"][lo:hi], b["old_logprobs"][lo:hi], b["advantages"][lo:hi], b["loss_mask"][lo:hi],
            temperature=case["temperature"], loss_agg=case["loss_agg"], clip_ratio_low=0.2, clip_ratio_high=0.28, clip_ratio_c=3.0,
            calculate_entropy=bonus > 0, entropy_bonus=bonus, ref_logprobs=b["ref_logprobs"][lo:hi], kl_beta=beta,
            kl_penalty_type=case.get("kl", "low_var_kl"), gradient_accumulation=2)
        loss.backward()
        metrics.append(mt)
    assert torch.equal(window.grad.float(), want_grad)
    for k in ("actor/final_loss", "actor/entropy_loss", "actor/kl_loss", "actor/policy_loss", "actor/approx_kl", "actor/clip_fraction"):
        got = torch.mean(torch.stack([torch.as_tensor(mt[k], dtype=torch.float32) for mt in metrics]))
        assert float(want[k]) == pytest.approx(float(got), rel=1e-6, abs=1e-9), k
    assert want["actor/grad_norm"] == 1.5 and want["actor/lr"] == 1e-6


@pytest.mark.parametrize("do_sample,temperature,top_k", [(True, 1.0, -1), (True, 0.7, 50), (True, 1.6, 1), (False, 1.0, -1)])
def test_discrete_action_head_matches_the_reference_model_method(ref, do_sample, temperature, top_k):
    """OpenVLA-OFT's ``_discrete_prediction`` (openvla_oft_action_model.py:315-414) compiled on its own; the language model is
    a stand-in that returns given logits.  Its own gather of the action positions, vocabulary window, temperature, top-k
    warper, softmax, torch.multinomial and bin-centre lookup against oracle.token_oracle.categorical_sample (the K2 oracle the
    HIP sampler is compared with), the multinomial draw reproduced from the same generator state."""
    pytest.importorskip("transformers")
    from transformers import TopKLogitsWarper

    from oracle import reference_loader as R
    from oracle import token_oracle as TO
    fn = R.load_function("rlinf/models/embodiment/openvla_oft/official/openvla_oft_action_model.py",
                         "OpenVLAOFTForRLActionPrediction._discrete_prediction", torch=torch, np=np, TopKLogitsWarper=TopKLogitsWarper)
    B, A, C, bins, pad, vocab = 6, 7, 2, 256, 64, 1024
    g = torch.Generator().manual_seed(3)
    n_prefix = torch.tensor([3, 5, 4, 3, 6, 5])
    seq = int(n_prefix.max()) + 2 + A * C + 4
    logits = torch.randn(B, seq, vocab, generator=g) * 2
    centers = np.linspace(-1, 1, bins - 1).astype(np.float32)
    lm = lambda **_kw: SimpleNamespace(logits=logits, hidden_states=[torch.zeros(B, seq, 4)])  # noqa: E731
    me = SimpleNamespace(language_model=lm, action_dim=A, num_action_chunks=C, vocab_size=vocab, bin_centers=centers,
                         config=SimpleNamespace(n_action_bins=bins, pad_to_multiple_of=pad))
    torch.manual_seed(7)
    actions, processed, tokens, _ = fn(me, None, None, None, n_prefix, torch.full((B,), 2), do_sample=do_sample,
                                       temperature=temperature, top_k=top_k)
    # what the head is handed on this repo's path: the [B, A*C, bins] window of the same positions
    pos = (n_prefix + 2).unsqueeze(1) + torch.arange(A * C).unsqueeze(0)
    window = logits[torch.arange(B).unsqueeze(-1), pos][..., -bins - pad:-pad]
    q = None
    if do_sample:
        torch.manual_seed(7)
        q = torch.empty(B * A * C, bins).exponential_(1).view(B, A * C, bins)  # the draw torch.multinomial makes internally
    tok, _, proc, act = TO.categorical_sample(window, q, temperature, top_k, bin_centers=torch.from_numpy(centers))
    assert torch.equal(tok, tokens)
    if do_sample:
        assert torch.equal(proc, processed)
    assert np.array_equal(act.reshape(-1, A).numpy(), actions)

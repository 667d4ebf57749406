"""The drop-in boundary, taken literally: the reference's OWN examples/embodiment/train_embodied_agent.py is
executed, unmodified and from where it lies, against this package through the ``rlinf`` import alias -- its imports
(rlinf.config / runners / scheduler.Cluster / utils.placement / workers.*), ``create_group(cfg).launch(cluster, name=...,
placement_strategy=...)``, ``EmbodiedRunner(cfg=..., actor=..., rollout=..., env=..., reward=...)``, ``runner.init_workers()``
and ``runner.run()`` with the runner's channel-carrying worker calls inside.

Without a GPU (this file's CPU test) the sequence runs up to the first kernel launch of ``run()``, which must fail LOUDLY
(RlxError: no CPU fallback) -- not silently fall back; on the GPU box the reference tree is absent, so the `-m gpu` test runs
this package's own entry point, which mirrors the same call sequence line by line, for real (eval + checkpoint + resume)."""

import json
import os
import subprocess
import sys

import pytest
import torch

from conftest import ROOT

REF_ENTRY = "/root/reference/examples/embodiment/train_embodied_agent.py"
CONFIG_DIR = os.path.join(ROOT, "examples", "embodiment", "config")


def _small_cfg(tmp_path, **runner):
    from rlinf_amd.config import load_config
    return load_config(os.path.join(CONFIG_DIR, "maniskill_ppo_mlp.yaml"), search_paths=[CONFIG_DIR], overrides=[
        "env.train.total_num_envs=16", "env.train.max_steps_per_rollout_epoch=8", "actor.global_batch_size=64",
        "actor.micro_batch_size=64", "algorithm.update_epoch=1", f"runner.logger.log_path={tmp_path}",
        "runner.logger.experiment_name=t", *[f"runner.{k}={v}" for k, v in runner.items()]])


@pytest.mark.reference
@pytest.mark.skipif(not os.path.exists(REF_ENTRY), reason="reference tree not present on this machine")
def test_reference_entry_point_runs_unchanged_against_the_alias(tmp_path):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "helpers", "reference_entry_runner.py"), REF_ENTRY,
                          str(tmp_path)], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    res = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert res["alias_is_same_module"] and res["hydra_config_name"] == "maniskill_ppo_openvlaoft"
    # everything in front of run() happened: config dump, Cluster, placement, the three groups, runner.init_workers()
    assert res["config_dumped"] and res["workers_initialised"]
    if res["cuda"]:
        assert res["outcome"] == "ran" and res["actor_steps"] == 2 * 2  # 2 iterations x (16*8/64) minibatches x 1 epoch
    else:  # first kernel entry of runner.run(): loud, no fallback
        assert res["outcome"] == "rlx_error" and res["actor_steps"] == 0, res


def test_runner_surface_matches_the_reference_signatures():
    """Constructor / method names and arguments the reference's entry point and runner use (embodied_runner.py:52-66,163-206,
    478-563,644-660; env_worker.py:999-1009,1058,1374; huggingface_worker.py; embodied_fsdp_actor_worker.py:186)."""
    import inspect

    from rlinf_amd.runners.embodied_runner import EmbodiedRunner
    from rlinf_amd.workers.actor.embodied_fsdp_actor_worker import EmbodiedFSDPActor
    from rlinf_amd.workers.actor.fsdp_actor_worker_pipeline import PipelineEmbodiedFSDPActor
    from rlinf_amd.workers.env.env_worker import EnvWorker
    from rlinf_amd.workers.rollout.hf.huggingface_worker import MultiStepRolloutWorker
    sig = lambda f: list(inspect.signature(f).parameters)  # noqa: E731
    assert sig(EmbodiedRunner.__init__)[:7] == ["self", "cfg", "actor", "rollout", "env", "reward", "critic"]
    for name in ("init_workers", "update_rollout_weights", "evaluate", "run", "run_pipeline", "_save_checkpoint", "set_max_steps",
                 "_maybe_eval_and_checkpoint"):
        assert callable(getattr(EmbodiedRunner, name)), name
    assert sig(EnvWorker.interact)[1:5] == ["input_channel", "rollout_channel", "reward_channel", "actor_channel"]
    assert sig(EnvWorker.evaluate)[1:] == ["input_channel", "rollout_channel"]
    assert sig(EnvWorker.prefetch_train_bootstrap)[1:] == ["rollout_channel"]
    assert sig(MultiStepRolloutWorker.generate)[1:3] == ["input_channel", "output_channel"]
    assert sig(MultiStepRolloutWorker.evaluate)[1:3] == ["input_channel", "output_channel"]
    assert sig(EmbodiedFSDPActor.recv_rollout_trajectories)[1:] == ["input_channel"]
    assert sig(EmbodiedFSDPActor.save_checkpoint)[1:] == ["save_path", "step"] and sig(EmbodiedFSDPActor.load_checkpoint)[1:] == ["load_path"]
    assert sig(EmbodiedFSDPActor.run_training)[1:] == ["input_channel"]
    assert issubclass(PipelineEmbodiedFSDPActor, EmbodiedFSDPActor)
    assert list(inspect.signature(EmbodiedFSDPActor.create_group).parameters)[0] == "cfg"


def test_checkpoint_round_trip_on_host_buffers(tmp_path):
    """save_checkpoint / load_checkpoint (fsdp_model_manager.py:342-389): every piece of training state returns bit for bit,
    in place (buffer addresses unchanged), and model_state_dict/full_weights.pt carries the reference's parameter names."""
    from oracle import ppo_oracle as O
    from rlinf_amd.config import validate_cfg
    from rlinf_amd.scheduler import DistContext
    from rlinf_amd.workers.actor.embodied_fsdp_actor_worker import EmbodiedFSDPActor
    cfg = validate_cfg(_small_cfg(tmp_path))
    ctx = DistContext(device=torch.device("cpu"))
    a = EmbodiedFSDPActor(cfg, ctx)
    a.init_worker()
    g = torch.Generator().manual_seed(3)
    a.exp_avg.copy_(torch.randn(a.model.n_params, generator=g)), a.exp_avg_sq.copy_(torch.rand(a.model.n_params, generator=g))
    a.step_state.copy_(torch.tensor([37, 1], dtype=torch.int32))
    a.optimizer_steps, a.version = 37, 5
    a.lr_scheduler.step()
    want = dict(flat=a.model.flat.detach().clone(), m=a.exp_avg.clone(), v=a.exp_avg_sq.clone(), st=a.step_state.clone(),
                lr=a.lr_scheduler.get_last_lr(), rng=torch.get_rng_state())
    a.save_checkpoint(str(tmp_path / "actor"), 7)
    full = torch.load(tmp_path / "actor" / "model_state_dict" / "full_weights.pt", weights_only=False)
    ora = O.OracleMLPPolicy(42, 8, 1)
    ora.load_state_dict(full)  # the reference-shaped module accepts the file as it is
    b = EmbodiedFSDPActor(cfg, ctx)
    b.init_worker()
    ptrs = (b.model.flat.data_ptr(), b.exp_avg.data_ptr(), b.exp_avg_sq.data_ptr(), b.step_state.data_ptr())
    torch.manual_seed(99)
    b.load_checkpoint(str(tmp_path / "actor"))
    assert ptrs == (b.model.flat.data_ptr(), b.exp_avg.data_ptr(), b.exp_avg_sq.data_ptr(), b.step_state.data_ptr())
    assert torch.equal(b.model.flat, want["flat"]) and torch.equal(b.exp_avg, want["m"]) and torch.equal(b.exp_avg_sq, want["v"])
    assert torch.equal(b.step_state, want["st"]) and b.optimizer_steps == 37 and b.version == 5
    assert b.lr_scheduler.get_last_lr() == want["lr"] and torch.equal(torch.get_rng_state(), want["rng"])
    # a weights-only directory (what the reference's save_full_model_weights leaves) loads too
    only = tmp_path / "weights_only" / "model_state_dict"
    only.mkdir(parents=True)
    torch.save({k: v + 1.0 for k, v in full.items()}, only / "full_weights.pt")
    b.load_checkpoint(str(tmp_path / "weights_only"))
    assert torch.equal(b.model.flat, want["flat"] + 1.0)


@pytest.mark.gpu
def test_entry_point_trains_evaluates_checkpoints_and_resumes(tmp_path):
    """This package's entry point (the reference's call sequence) on the GPU: 4 iterations with validation every 2 and a
    checkpoint every 2; a second process resumes from global_step_2 and reproduces iterations 2-3 bit for bit."""
    entry = os.path.join(ROOT, "examples", "embodiment", "train_embodied_agent.py")
    common = ["--config-name", "maniskill_ppo_mlp", "env.train.total_num_envs=64", "env.train.max_steps_per_rollout_epoch=16",
              "actor.global_batch_size=256", "actor.micro_batch_size=256", "algorithm.update_epoch=2", "runner.max_epochs=4",
              "runner.val_check_interval=2", "runner.save_interval=2", f"runner.logger.log_path={tmp_path}",
              "runner.logger.experiment_name=t", "env.eval.total_num_envs=32", "env.eval.max_steps_per_rollout_epoch=60",
              "env.eval.max_episode_steps=20", "env.eval.auto_reset=True"]

    def run(*extra):
        out = subprocess.run([sys.executable, entry, *common, *extra], capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stderr[-3000:]
        return [json.loads(l) for l in out.stdout.splitlines() if l.startswith("{")]

    first = run()
    assert [l["step"] for l in first] == [0, 1, 2, 3]
    assert "eval/return" in first[1] and first[1]["eval/num_trajectories"] > 0 and "eval/return" not in first[0]
    ckpt = tmp_path / "t" / "checkpoints" / "global_step_2"
    assert (ckpt / "actor" / "model_state_dict" / "full_weights.pt").exists()
    assert (ckpt / "actor" / "local_shard_checkpoint" / "checkpoint_rank_0.pt").exists()
    resumed = run(f"runner.resume_dir={ckpt}")
    assert [l["step"] for l in resumed] == [2, 3]
    for a, b in zip(first[2:], resumed):
        for k in ("rollout/rewards", "train/actor/total_loss", "train/actor/approx_kl", "train/critic/value_loss", "train/actor/grad_norm"):
            assert a[k] == b[k], (k, a[k], b[k])

#!/usr/bin/env python
"""Entry point with the reference's name and selection logic (examples/embodiment/train_async.py:36-120): the learner class
follows ``algorithm.loss_type``.  Of the reference's asynchronous learners the decoupled-PPO one is on this repo's path
(SURVEY.md 8f-4); SAC / DAgger / RLT learners are other algorithms and are refused by name.

    python examples/embodiment/train_async.py --config-name maniskill_async_ppo_mlp [runner.max_epochs=20 ...]

Rollout and learner share the resident trajectory buffer of one process here, so an iteration is rollout -> advantages ->
run_training like the synchronous runner; what is asynchronous PPO about it is the data: every trajectory carries the
policy version it was sampled with and the loss corrects for the distance to the version being trained.
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

_OTHER_ALGORITHMS = ("embodied_sac", "rlt_ac", "embodied_dagger")


def select_actor_cls(cfg):
    loss_type = cfg.algorithm.loss_type
    if loss_type == "decoupled_actor_critic":
        from rlinf_amd.workers.actor import AsyncPPOEmbodiedFSDPActor
        return AsyncPPOEmbodiedFSDPActor
    if loss_type in _OTHER_ALGORITHMS:
        raise NotImplementedError(f"loss type {loss_type}: only the decoupled-PPO learner of the async runner is built here")
    raise ValueError(f"Unsupported loss type {loss_type} for async embodied runner")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config-path", default=os.path.join(os.path.dirname(os.path.abspath(__file__)), "config"))
    ap.add_argument("--config-name", default="maniskill_async_ppo_mlp")
    ap.add_argument("overrides", nargs="*")
    args = ap.parse_args()

    from rlinf_amd.config import load_config, validate_cfg
    from rlinf_amd.runners import EmbodiedRunner
    from rlinf_amd.scheduler import init_distributed
    from rlinf_amd.workers.env import EnvWorker
    from rlinf_amd.workers.rollout.hf import MultiStepRolloutWorker

    cfg = validate_cfg(load_config(os.path.join(args.config_path, args.config_name + ".yaml"), overrides=args.overrides,
                                   search_paths=[args.config_path]))
    ctx = init_distributed()
    actor = select_actor_cls(cfg).create_group(cfg, ctx).launch(None, name=cfg.actor.get("group_name", "ActorGroup"))
    rollout = MultiStepRolloutWorker.create_group(cfg, ctx).launch(None, name=cfg.rollout.get("group_name", "RolloutGroup"))
    env = EnvWorker.create_group(cfg, ctx).launch(None, name="EnvGroup")
    runner = EmbodiedRunner(cfg, actor, rollout, env)
    runner.init_workers()
    for step in range(runner.max_steps):
        m = runner.run_step()
        if ctx.rank == 0:
            keep = ("rollout/rewards", "train/actor/total_loss", "train/actor/proximal_approx_kl", "train/actor/behav_approx_kl",
                    "train/actor/average_version", "train/critic/value_loss", "train/actor/grad_norm", "perf/env_steps_per_sec",
                    "time/step")
            print(json.dumps({"step": step, **{k: (round(v, 6) if isinstance(v, float) else v) for k, v in m.items() if k in keep}}),
                  flush=True)


if __name__ == "__main__":
    main()

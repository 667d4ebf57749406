#!/usr/bin/env python
"""Entry point with the reference's name and call sequence (examples/embodiment/train_embodied_agent.py:35-172 ->
rlinf/runners/embodied_runner.py): validate the config, Cluster, component placement, the three worker groups launched with
``create_group(cfg).launch(cluster, name=..., placement_strategy=...)``, EmbodiedRunner, init_workers, run.

    python examples/embodiment/train_embodied_agent.py --config-name maniskill_ppo_mlp [runner.max_epochs=20 ...]
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 examples/embodiment/train_embodied_agent.py ...

hydra is not available here: ``--config-path`` / ``--config-name`` and trailing ``key=value`` overrides are handled by
rlinf_amd.config.load_config, which reads the reference's own YAML files (defaults lists, interpolation) as they are.  The
reference's own script runs against this package unchanged through the ``rlinf`` import alias (rlinf_amd.compat.install_alias,
tests/test_reference_entry_point.py).
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main(cfg) -> None:
    from rlinf_amd.config import validate_cfg
    from rlinf_amd.runners.embodied_runner import EmbodiedRunner
    from rlinf_amd.scheduler import Cluster
    from rlinf_amd.utils.placement import HybridComponentPlacement
    from rlinf_amd.workers.env.env_worker import EnvWorker
    from rlinf_amd.workers.rollout.hf.huggingface_worker import MultiStepRolloutWorker

    cfg = validate_cfg(cfg)
    cluster = Cluster(cluster_cfg=cfg.get("cluster", None), distributed_log_dir=cfg.runner.get("per_worker_log_path", None))
    component_placement = HybridComponentPlacement(cfg, cluster)

    actor_placement = component_placement.get_strategy("actor")
    if bool(cfg.runner.get("use_training_pipeline", False)):
        from rlinf_amd.workers.actor.fsdp_actor_worker_pipeline import PipelineEmbodiedFSDPActor as actor_worker_cls
    elif cfg.algorithm.loss_type == "decoupled_actor_critic":
        from rlinf_amd.workers.actor.async_ppo_fsdp_worker import AsyncPPOEmbodiedFSDPActor as actor_worker_cls
    else:
        from rlinf_amd.workers.actor.embodied_fsdp_actor_worker import EmbodiedFSDPActor as actor_worker_cls
    actor_group = actor_worker_cls.create_group(cfg).launch(cluster, name=cfg.actor.get("group_name", "ActorGroup"),
                                                           placement_strategy=actor_placement)
    rollout_group = MultiStepRolloutWorker.create_group(cfg).launch(
        cluster, name=cfg.rollout.get("group_name", "RolloutGroup"), placement_strategy=component_placement.get_strategy("rollout"))
    env_group = EnvWorker.create_group(cfg).launch(cluster, name=cfg.env.get("group_name", "EnvGroup"),
                                                   placement_strategy=component_placement.get_strategy("env"))

    runner = EmbodiedRunner(cfg=cfg, actor=actor_group, rollout=rollout_group, env=env_group, reward=None)
    runner.init_workers()
    rank = cluster.ctx.rank
    keep = ("rollout/rewards", "train/actor/total_loss", "train/actor/approx_kl", "train/critic/value_loss",
            "train/actor/grad_norm", "perf/env_steps_per_sec", "time/step", "eval/return", "eval/num_trajectories")
    first = runner.global_step
    for step, m in enumerate(runner.iter_steps(), start=first):  # runner.run(), one printed line per iteration (read one iteration late)
        if rank == 0:
            print(json.dumps({"step": step, **{k: (round(v, 6) if isinstance(v, float) else v) for k, v in m.items() if k in keep}}),
                  flush=True)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--config-path", default=os.path.join(os.path.dirname(os.path.abspath(__file__)), "config"))
    ap.add_argument("--config-name", default="maniskill_ppo_mlp")
    ap.add_argument("overrides", nargs="*")
    args = ap.parse_args()
    from rlinf_amd.config import load_config
    main(load_config(os.path.join(args.config_path, args.config_name + ".yaml"), overrides=args.overrides,
                     search_paths=[args.config_path]))

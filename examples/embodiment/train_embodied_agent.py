#!/usr/bin/env python
"""Entry point with the reference's name and call sequence (examples/embodiment/train_embodied_agent.py ->
rlinf/runners/embodied_runner.py:52-66,163,478): load + validate the config, create the three worker groups, run.

    python examples/embodiment/train_embodied_agent.py --config-name maniskill_ppo_mlp [runner.max_epochs=20 ...]
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 examples/embodiment/train_embodied_agent.py ...

hydra is not available here: ``--config-path`` / ``--config-name`` and trailing ``key=value`` overrides are handled by
rlinf_amd.config.load_config, which reads the reference's own YAML files (defaults lists, interpolation) as they are.
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config-path", default=os.path.join(os.path.dirname(os.path.abspath(__file__)), "config"))
    ap.add_argument("--config-name", default="maniskill_ppo_mlp")
    ap.add_argument("overrides", nargs="*")
    args = ap.parse_args()

    from rlinf_amd.config import load_config, validate_cfg
    from rlinf_amd.runners import EmbodiedRunner
    from rlinf_amd.scheduler import init_distributed
    from rlinf_amd.workers.actor import EmbodiedFSDPActor
    from rlinf_amd.workers.env import EnvWorker
    from rlinf_amd.workers.rollout.hf import MultiStepRolloutWorker

    cfg = validate_cfg(load_config(os.path.join(args.config_path, args.config_name + ".yaml"), overrides=args.overrides,
                                   search_paths=[args.config_path]))
    ctx = init_distributed()
    actor = EmbodiedFSDPActor.create_group(cfg, ctx).launch(None, name=cfg.actor.get("group_name", "ActorGroup"))
    rollout = MultiStepRolloutWorker.create_group(cfg, ctx).launch(None, name=cfg.rollout.get("group_name", "RolloutGroup"))
    env = EnvWorker.create_group(cfg, ctx).launch(None, name="EnvGroup")
    runner = EmbodiedRunner(cfg, actor, rollout, env)
    runner.init_workers()
    for step in range(runner.max_steps):
        m = runner.run_step()
        if ctx.rank == 0:
            keep = ("rollout/rewards", "train/actor/total_loss", "train/actor/approx_kl", "train/critic/value_loss",
                    "train/actor/grad_norm", "perf/env_steps_per_sec", "time/step")
            print(json.dumps({"step": step, **{k: (round(v, 6) if isinstance(v, float) else v) for k, v in m.items() if k in keep}}),
                  flush=True)


if __name__ == "__main__":
    main()

"""Dev tool (GPU box): A/B of one development switch inside ONE process -- a fresh runner (fresh graph capture) per value, the
values alternating round after round so that the box's drift shows up as a trend instead of hiding in the comparison.
    python tools/ab_env.py --env RLX_DW_NBUF --values 3,4 [--rounds 3] [--steps 40]"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--env", required=True)
    ap.add_argument("--values", required=True)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--steps", type=int, default=40)
    args = ap.parse_args()
    from rlinf_amd.scheduler import init_distributed
    ctx = init_distributed()
    dev = ctx.device
    for r in range(args.rounds):
        for val in args.values.split(","):
            os.environ[args.env] = val
            runner = bench.build_runner(bench.build_cfg(1, True, "bf16", total_envs=bench.ENVS, global_batch=bench.GLOBAL_BATCH), ctx)
            for _ in range(4):
                runner.run_step()
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            pending = None
            for _ in range(args.steps):
                step = runner.run_step(defer=True)
                if pending is not None:
                    pending.result()
                pending = step
            pending.result()
            torch.cuda.synchronize(dev)
            print(f"round {r} {args.env}={val}: {(time.perf_counter() - t0) / args.steps * 1e3:.3f} ms / iteration", flush=True)
            runner.close()
            del runner
            torch.cuda.empty_cache()


if __name__ == "__main__":
    main()

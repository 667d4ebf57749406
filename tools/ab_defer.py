"""Dev tool (GPU box): the benchmark loop with metrics read (a) before the next iteration is queued and (b) one iteration late,
alternating in ONE process on one runner (same box, same clocks), plus the host time one queued iteration costs.
    python tools/ab_defer.py [--steps 40] [--rounds 4]"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--rounds", type=int, default=4)
    ap.add_argument("--pipeline-epochs", type=int, default=0, help="> 0: pipeline mode with this many rollout epochs")
    args = ap.parse_args()
    from rlinf_amd.scheduler import init_distributed
    ctx = init_distributed()
    pipe = args.pipeline_epochs > 0
    runner = bench.build_runner(bench.build_cfg(1, True, "bf16", pipeline=pipe, rollout_epochs=max(args.pipeline_epochs, 1),
                                                total_envs=bench.ENVS, global_batch=bench.GLOBAL_BATCH), ctx)
    for _ in range(5):
        runner.run_step()
    dev = ctx.device
    for r in range(args.rounds):
        for defer in (False, True):
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            pending, host = None, []
            for _ in range(args.steps):
                h0 = time.perf_counter()
                step = runner.run_step(defer=defer)
                host.append(time.perf_counter() - h0)
                if pending is not None:
                    pending.result()
                pending = step if defer else None
            if pending is not None:
                pending.result()
            torch.cuda.synchronize(dev)
            dt = (time.perf_counter() - t0) / args.steps * 1e3
            host.sort()
            print(f"round {r} defer={int(defer)}: {dt:.3f} ms / iteration; run_step() returned after median {host[len(host) // 2] * 1e3:.3f} ms "
                  f"(min {host[0] * 1e3:.3f}, max {host[-1] * 1e3:.3f})", flush=True)
    runner.close()


if __name__ == "__main__":
    main()

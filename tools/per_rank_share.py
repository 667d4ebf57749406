"""Dev tool (GPU box): ONE GPU running the share of the benchmark that a rank of an N-GPU strong-scaling job would run
(1024 / N envs, 8192 / N minibatch rows, no exchange): the compute rows of DESIGN.md section 6's table, measured instead of
modelled.  Run it under `rocprofv3 --kernel-trace --stats` for the per-kernel times.
    python tools/per_rank_share.py --share 8 [--steps 20]"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--share", type=int, required=True)
    ap.add_argument("--steps", type=int, default=20)
    args = ap.parse_args()
    from rlinf_amd.scheduler import init_distributed
    ctx = init_distributed()
    envs, gb = bench.ENVS // args.share, bench.GLOBAL_BATCH // args.share
    runner = bench.build_runner(bench.build_cfg(1, True, "bf16", total_envs=envs, global_batch=gb), ctx)
    for _ in range(4):
        runner.run_step()
    torch.cuda.synchronize(ctx.device)
    t0 = time.perf_counter()
    pending = None
    for _ in range(args.steps):
        step = runner.run_step(defer=True)
        if pending is not None:
            pending.result()
        pending = step
    pending.result()
    torch.cuda.synchronize(ctx.device)
    print(f"share 1/{args.share}: {envs} envs, {gb}-row minibatches: {(time.perf_counter() - t0) / args.steps * 1e3:.3f} ms / iteration (no exchange)", flush=True)
    runner.close()


if __name__ == "__main__":
    main()

"""Does pipeline mode's rollout / training overlap pay on one MI355X?  (rlinf/runners/embodied_runner.py:565-642)

One process, one box, configurations ALTERNATING (the boxes drift by several per cent over minutes): the synchronous loop, pipeline
mode with 4 rollout epochs on ONE stream (no overlap), the same with the rollout of epoch e + 1 on its own stream (overlap), and
the overlap with the rollout stream restricted to n compute units (hipExtStreamCreateWithCUMask: RLX_ROLLOUT_CUS) -- alone, and
with the training stream restricted to the complementary CUs.  Prints one table; `--trace-hint` prints the rocprofv3 command whose
per-queue timeline (tools/rocpd_stats.py --streams) shows what the two streams do to each other.
    python tools/pipeline_overlap_probe.py [--steps 60] [--rounds 2]"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def run(ctx, name, steps, *, pipeline, epochs=1, overlap=True, rollout_cus=0, train_complement=False, precision="bf16"):
    from rlinf_amd.utils.streams import MaskedStream
    os.environ["RLX_ROLLOUT_CUS"] = str(rollout_cus)
    owner = None
    try:
        if train_complement and rollout_cus:
            owner = MaskedStream(ctx.device, rollout_cus, complement=True)
            with torch.cuda.stream(owner.stream):
                r = bench.measure(ctx, precision=precision, steps=steps, warmup=3, pipeline=pipeline, rollout_epochs=epochs, overlap=overlap)
        else:
            r = bench.measure(ctx, precision=precision, steps=steps, warmup=3, pipeline=pipeline, rollout_epochs=epochs, overlap=overlap)
        return {"config": name, "ms_per_step": r["ms_per_step"], "windows": r["ms_per_step_windows"]}
    except Exception as e:  # noqa: BLE001
        return {"config": name, "error": f"{type(e).__name__}: {e}"[:300]}
    finally:
        os.environ["RLX_ROLLOUT_CUS"] = "0"
        pass  # (the masked training stream is left alive: torch's allocator may still hold blocks tagged with it)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--rounds", type=int, default=2)
    ap.add_argument("--cus", default="16,32,64")
    ap.add_argument("--precision", default="bf16")
    args = ap.parse_args()
    from rlinf_amd.scheduler import init_distributed
    ctx = init_distributed()
    cus = [int(c) for c in args.cus.split(",") if c]
    configs = [("sync (no pipeline)", dict(pipeline=False)),
               ("pipeline E=4, one stream (no overlap)", dict(pipeline=True, epochs=4, overlap=False)),
               ("pipeline E=4, rollout on its own stream (overlap)", dict(pipeline=True, epochs=4, overlap=True))]
    for c in cus:
        configs.append((f"pipeline E=4, overlap, rollout stream on {c} CUs", dict(pipeline=True, epochs=4, overlap=True, rollout_cus=c)))
    for c in cus[:2]:
        configs.append((f"pipeline E=4, overlap, rollout on {c} CUs + training on the other CUs",
                        dict(pipeline=True, epochs=4, overlap=True, rollout_cus=c, train_complement=True)))
    rows = []
    for rnd in range(args.rounds):
        for name, kw in configs:
            r = run(ctx, name, args.steps, precision=args.precision, **kw)
            r["round"] = rnd
            rows.append(r)
            print(json.dumps(r), flush=True)
    print("\n== per configuration (ms per iteration, every round) ==")
    for name, _ in configs:
        v = [r.get("ms_per_step") for r in rows if r["config"] == name]
        print(f"{name:78s} " + "  ".join("error" if x is None else f"{x:7.3f}" for x in v))


if __name__ == "__main__":
    main()

"""Measured per-optimizer-step cost of the gradient exchange's OWN launches, on one device (bench.py ``scaling_model``).

For W in (2, 4, 8): the chain a rank of a W-GPU job runs per optimizer step -- stage this rank's slabs, hand-shake, reduce (W <= 3)
or reduce-scatter + hand-shake + gather (W >= 4), clip + AdamW -- through ``rlx_xgmi_clip_adamw_step`` on a communicator whose
peers are all this rank's own buffer (``rlx_xgmi_connect_self``), against the single-GPU chain (``rlx_clip_adamw_step``: slab sum +
norm, clip + AdamW) on the same buffers.  The difference is what the exchange adds per step in launches and LOCAL memory traffic;
what it cannot contain is the link: peer reads go to this device's HBM instead of over xGMI and every hand-shake is satisfied at
once.  A lower bound of the exchange cost, measured instead of modelled.  Parameters / moments are scratch copies (the reduced
values are not a valid all-reduce: every shard is this rank's).
    python tools/exchange_self.py [--steps 400]"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def measure(device, n_params: int, slabs: int, worlds=(2, 4, 8), steps: int = 400, groups=None) -> dict:
    from rlinf_amd import ops
    from rlinf_amd.scheduler.xgmi import SelfAliasedXgmi
    lib = ops._lib.load()
    dev = torch.device(device)
    g = torch.Generator(device=dev).manual_seed(7)
    grads = torch.randn(slabs, n_params, device=dev, generator=g) * 1e-3
    groups = groups or [(0, n_params, 3e-4)]
    kw = dict(betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01, max_grad_norm=0.5)
    ws = torch.empty(lib.rlx_adamw_workspace_bytes(n_params), dtype=torch.uint8, device=dev)

    def fresh():
        return (torch.randn(n_params, device=dev, generator=g) * 0.05, torch.zeros(n_params, device=dev), torch.zeros(n_params, device=dev),
                torch.zeros(2, device=dev), torch.zeros(2, dtype=torch.int32, device=dev), torch.zeros(n_params, device=dev))

    def time_chain(call) -> float:
        """us per optimizer step: the learner replays its update phase as ONE hipGraph, so the chain is timed the same way -- 50
        steps captured (each behind a small launch that rewrites a slab element, like the weight-gradient launch in front of the
        real step), replayed until ``steps`` have run.  (Eager launches from Python cost more host time per launch than these
        kernels run: round 5's eager numbers measured the host.)"""
        stream = torch.cuda.current_stream(dev).cuda_stream
        for _ in range(5):
            call(stream)
        torch.cuda.synchronize(dev)
        side = torch.cuda.Stream(dev)
        per_graph = 50
        with torch.cuda.stream(side):
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=side):
                for _ in range(per_graph):
                    grads[0, :64].add_(0.0)
                    call(side.cuda_stream)
            graph.replay()
            side.synchronize()
            reps = max(1, steps // per_graph)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(side)
            for _ in range(reps):
                graph.replay()
            e1.record(side)
            side.synchronize()
        return e0.elapsed_time(e1) * 1e3 / (reps * per_graph)

    out = {"n_params": n_params, "slabs": slabs, "steps": steps}
    with torch.cuda.device(dev):
        p, m, v, st, ss, gf = fresh()
        # what one GPU runs: the optimizer step as one launch (rlx_adamw_params.sync_words); the exchange's chains keep their launches
        single = ops.PreparedAdamw(p, grads, m, v, groups, grad_scale=1.0, stats=st, step_state=ss, workspace=ws,
                                   sync=ops.adamw_sync_words(n_params, dev), **kw)
        out["single_gpu_chain_us"] = round(time_chain(single), 3)
        for W in worlds:
            row = {}
            for form in ("one_launch", "chain"):
                comm = SelfAliasedXgmi(dev, W, n_params, timing=True)
                p, m, v, st, ss, gf = fresh()
                step = ops.PreparedAdamw(p, grads, m, v, groups, grad_scale=1.0 / W, stats=st, step_state=ss, workspace=ws, xgmi=comm,
                                         grad_flat=gf, sync=ops.adamw_sync_words(n_params, dev) if form == "one_launch" else None, **kw)
                us = time_chain(step)
                row[form] = {"us": round(us, 3), "extra_us_per_step": round(us - out["single_gpu_chain_us"], 3),
                             "no_wait_timed_out": bool(comm.status_ok())}
                if form == "chain":
                    row[form]["form"] = comm.algo
                comm.close()
            # the learners run the one-launch exchange (start-up validation permitting): that is the row the model uses
            out[f"w{W}"] = {"form": "one launch (pushed self-validating words)", "chain_us": row["one_launch"]["us"],
                            "extra_us_per_step": row["one_launch"]["extra_us_per_step"],
                            "no_wait_timed_out": row["one_launch"]["no_wait_timed_out"], "launch_chain_fallback": row["chain"]}
    out["what"] = ("per optimizer step of one rank, replayed from a hipGraph like the learner's update phase, every peer aliased to "
                   "this device: `one launch` = the exchange inside the optimizer launch in its timing emulation (the rank's blocks b, b + nblk / W, ... "
                   "play ranks 0, 1, ...'s copies of owned block b: an owner's contributions come from, and its answers go to, other "
                   "workgroups running concurrently -- a rank's pushes, polls and local memory traffic); `launch_chain_fallback` = "
                   "stage, hand-shake, reduce / reduce-scatter + gather, clip + AdamW.  Link latency and peer skew are NOT in either "
                   "(a lower bound)")
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--n", type=int, default=287504)
    ap.add_argument("--slabs", type=int, default=10)
    a = ap.parse_args()
    print(json.dumps(measure("cuda:0", a.n, a.slabs, steps=a.steps)))

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
        stream = torch.cuda.current_stream(dev).cuda_stream
        for _ in range(20):
            call(stream)
        torch.cuda.synchronize(dev)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(steps):
            call(stream)
        e1.record()
        torch.cuda.synchronize(dev)
        return e0.elapsed_time(e1) * 1e3 / steps  # us per optimizer step's chain

    out = {"n_params": n_params, "slabs": slabs, "steps": steps}
    with torch.cuda.device(dev):
        p, m, v, st, ss, gf = fresh()
        # what one GPU runs: the optimizer step as one launch (rlx_adamw_params.sync_words); the exchange's chains keep their launches
        single = ops.PreparedAdamw(p, grads, m, v, groups, grad_scale=1.0, stats=st, step_state=ss, workspace=ws,
                                   sync=ops.adamw_sync_words(n_params, dev), **kw)
        out["single_gpu_chain_us"] = round(time_chain(single), 3)
        for W in worlds:
            comm = SelfAliasedXgmi(dev, W, n_params)
            p, m, v, st, ss, gf = fresh()
            chain = ops.PreparedAdamw(p, grads, m, v, groups, grad_scale=1.0 / W, stats=st, step_state=ss, workspace=ws, xgmi=comm,
                                      grad_flat=gf, **kw)
            us = time_chain(chain)
            ok = comm.status_ok()
            out[f"w{W}"] = {"form": comm.algo, "chain_us": round(us, 3), "extra_us_per_step": round(us - out["single_gpu_chain_us"], 3),
                            "no_wait_timed_out": bool(ok)}
            comm.close()
    out["what"] = ("per-step launch chain of one rank (stage, hand-shake, reduce / reduce-scatter + gather, clip + AdamW) with every peer "
                   "aliased to this device: launches + local memory; link latency and peer skew NOT included (a lower bound)")
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--n", type=int, default=287504)
    ap.add_argument("--slabs", type=int, default=10)
    a = ap.parse_args()
    print(json.dumps(measure("cuda:0", a.n, a.slabs, steps=a.steps)))

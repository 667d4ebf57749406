"""Dev tool (GPU box): the bf16 fused step launch in its forms -- row-split (ppo_step_bf16_rows.hip) against column-split
(ppo_step_bf16.hip) -- at the minibatch sizes of a 1 / 2 / 4 / 8-GPU strong-scaling rank: phase stamps of the row-split kernel
(workgroup (0, 1), wave 0) and HIP-event time of fused + weight-gradient launches back to back.  Run it under
`rocprofv3 --kernel-trace` and read the per-kernel times with `tools/rocpd_stats.py <db> rlx --by-grid`.
    python tools/fused_rows_probe.py [--iters 200]"""
import argparse
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rlinf_amd import _lib, ops
from rlinf_amd._lib import PPO_OUT_FLOATS
from rlinf_amd.models.embodiment.mlp_policy import MLPPolicy

ROWS_NAMES = ["start", "inputs staged", "gemm L1", "epi1+image", "gemm L2", "epi2+image", "gemm L3", "epi3", "head (mfma)",
              "loss pass + metric sums + park", "head grads", "dz3 + slab + image", "bwd W3 + epi + image", "bwd W2 + epi + image",
              "stores drained"]


def minibatch(M, g):
    mb = dict(states=torch.randn(M, 42, generator=g), action=torch.randn(M, 8, generator=g) * 0.6,
              prev_logprobs=torch.randn(M, 8, generator=g) * 0.1 - 1.0, advantages=torch.randn(M, 1, generator=g),
              prev_values=torch.randn(M, 1, generator=g), returns=torch.randn(M, 1, generator=g))
    return {k: v.cuda() for k, v in mb.items()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=200)
    ap.add_argument("--sizes", default="8192,4096,2048,1024")
    ap.add_argument("--dev", action="store_true", help="add the development variants (RLX_ROWS_DEV: results are garbage, timing only)")
    args = ap.parse_args()
    raw = ctypes.CDLL(_lib.LIB_PATH)
    raw.rlx_dev_set_timing_buffer.argtypes = [ctypes.c_void_p]
    torch.manual_seed(0)
    pol = MLPPolicy(42, 8, 1, True, False, compute_dtype=torch.bfloat16).to("cuda")
    lay = pol.layout
    lp = ops.make_ppo_params(logprob_type="action_level", action_dim=8, chunks=1, clip_ratio_low=0.2, clip_ratio_high=0.2,
                             value_clip=1.0, huber_delta=10.0, max_episode_steps=50, has_critic=True)
    row = torch.zeros(PPO_OUT_FLOATS, device="cuda")
    buf = torch.zeros(64, dtype=torch.int64, device="cuda")
    variants = [("cols32", dict(RLX_FUSED_ROWS="0", RLX_FUSED_RT="2", RLX_ROWS_DEV="0")), ("rows", dict(RLX_FUSED_ROWS="1", RLX_ROWS_NSLOT="3")),
                ("rows-2slot", dict(RLX_ROWS_NSLOT="2"))]
    if args.dev:  # timing experiments that break the results: no weight copies / no fragment reads / neither
        variants += [("rows-nodma", dict(RLX_ROWS_NSLOT="3", RLX_ROWS_DEV="1")), ("rows-noread", dict(RLX_ROWS_DEV="2")),
                     ("rows-neither", dict(RLX_ROWS_DEV="3"))]
    for M in [int(x) for x in args.sizes.split(",")]:
        mb = minibatch(M, torch.Generator().manual_seed(1))
        ws = torch.empty(ops.ppo_step_workspace_bytes(lay, M), dtype=torch.uint8, device="cuda")
        ref = None
        for name, env in variants:
            os.environ.update(env)
            grads = torch.empty((ops.ppo_step_slabs(lay, M, bf16=True), lay.n_params), device="cuda")
            step = lambda: ops.ppo_step(pol.flat.data, lay, lp, mb, grads, row, ws, grad_out=1.0, bf16=True, tiles=pol.tiles())  # noqa: E731
            for _ in range(5):
                step()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.iters):
                step()
            e1.record()
            torch.cuda.synchronize()
            g = grads.sum(dim=0)
            if ref is None:
                ref = g.clone()
            rel = float((g - ref).norm() / ref.norm())
            print(f"M={M:5d} {name:7s}: fused + dW {e0.elapsed_time(e1) * 1e3 / args.iters:7.2f} us eager   |grad - cols32| / |cols32| = {rel:.2e}"
                  f"   loss {float(row[0]):.6f}", flush=True)
            if name.startswith("rows"):
                raw.rlx_dev_set_timing_buffer(buf.data_ptr())
                for rep in range(2 if name == "rows" else 1):
                    buf.zero_()
                    step()
                    step()
                    buf.zero_()
                    step()
                    torch.cuda.synchronize()
                    t = buf.cpu().tolist()
                    n = len(ROWS_NAMES)
                    print(f"   rows stamps M={M}: total {t[n - 1] - t[0]} ticks: " + "  ".join(f"{ROWS_NAMES[i]}: {t[i] - t[i - 1]}" for i in range(1, n)))
                raw.rlx_dev_set_timing_buffer(None)


if __name__ == "__main__":
    main()

"""Roofline probe of the token tier: token_logprob fwd / bwd at LLM sizes.

    python tools/bench_token.py [--tokens 8192] [--vocab 151936] [--dtype bf16] [--iters 20]

Prints one JSON line per kernel: algorithmic bytes (fwd: tokens*vocab*sizeof; bwd: twice that), the average launch
duration from HIP events on the launch stream, GB/s and the fraction of the 8 TB/s HBM peak.  (The CPU baseline of
this tier is timed by bench.py, the only place outside tests/ and smoke() that runs the oracle.)
"""

import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rlinf_amd import token_ops  # noqa: E402

HBM_PEAK = 8.0e12


def timed(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in evs:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) * 1e-3 for a, b in evs)
    return sum(ts) / len(ts), ts[0], ts[len(ts) // 2]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tokens", type=int, default=8192)
    ap.add_argument("--vocab", type=int, default=151936)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--mask-frac", type=float, default=1.0, help="fraction of tokens with a non-zero gradient")
    ap.add_argument("--temperature", type=float, default=1.0)
    args = ap.parse_args()
    dt = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(1)
    N, V = args.tokens, args.vocab
    x = torch.empty(N, V, dtype=dt, device=dev)
    for i in range(0, N, 1024):  # chunked so the f32 staging buffer stays small
        x[i:i + 1024] = (torch.randn(min(1024, N - i), V, device=dev, generator=g) * 4).to(dt)
    labels = torch.randint(0, V, (N,), device=dev, generator=g)
    dlp = torch.randn(N, device=dev, generator=g)
    dlp = dlp * (torch.rand(N, device=dev, generator=g) < args.mask_frac)
    dent = torch.full((N,), -0.01, device=dev) * (dlp != 0)
    out = torch.empty_like(x)
    esz = x.element_size()
    res = []
    for ent in (False, True):
        lp, e, lse = token_ops.token_logprob_fwd(x, labels, args.temperature, with_entropy=ent)
        mean, best, med = timed(lambda: token_ops.token_logprob_fwd(x, labels, args.temperature, with_entropy=ent), args.iters)
        by = N * V * esz
        res.append(dict(kernel="token_logprob_fwd", entropy=ent, dtype=args.dtype, tokens=N, vocab=V, bytes=by,
                        us=mean * 1e6, us_min=best * 1e6, us_median=med * 1e6, GBps=by / mean / 1e9,
                        frac=by / mean / HBM_PEAK))
        live = float((dlp != 0).float().mean())
        mean, best, med = timed(lambda: token_ops.token_logprob_bwd(x, labels, lse, e, dlp, dent if ent else None,
                                                                    args.temperature, out=out), args.iters)
        by = int(N * V * esz * (1 + live))  # masked rows are written but never read
        res.append(dict(kernel="token_logprob_bwd", entropy=ent, dtype=args.dtype, tokens=N, vocab=V, bytes=by,
                        live_rows=live, us=mean * 1e6, us_min=best * 1e6, us_median=med * 1e6, GBps=by / mean / 1e9,
                        frac=by / mean / HBM_PEAK))
    for r in res:
        print(json.dumps(r))


if __name__ == "__main__":
    main()

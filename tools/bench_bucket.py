"""Roofline probe of rlx_copy_segments (bucket_copy.hip): one bucket of f32 masters -> bf16 transport buffer.

    python tools/bench_bucket.py [--tensors 16] [--rows 8192] [--cols 8192]

Algorithmic bytes = n * (4 read + 2 written).  Also times the reference's chain on the same device: one
`tensor.to(dtype=bf16)` per parameter into separately allocated outputs (bucket_syncer.py:110-121), and the receiver's
per-parameter copy_ (load_state_dict) against one launch of the same kernel in the other direction.
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rlinf_amd.hybrid_engines.weight_syncer.bucket_syncer import WeightBucket, _run_segments, load_bucket, pack_bucket  # noqa: E402


def ev_time(fn, iters=10):
    fn()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in evs:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    return sum(a.elapsed_time(b) for a, b in evs) / iters * 1e-3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tensors", type=int, default=16)
    ap.add_argument("--rows", type=int, default=8192)
    ap.add_argument("--cols", type=int, default=8192)
    a = ap.parse_args()
    dev = torch.device("cuda")
    state = {f"layers.{i}.w": torch.randn(a.rows, a.cols, device=dev) for i in range(a.tensors)}
    n = a.tensors * a.rows * a.cols
    items = [(k, v, torch.bfloat16) for k, v in state.items()]
    bucket = pack_bucket(items, dev)
    segs = [(v, bucket[k].view(-1)) for k, v in state.items()]
    t_kernel = ev_time(lambda: _run_segments([(s.view(-1), d) for s, d in segs], dev))  # table upload + one launch
    t_pack = ev_time(lambda: pack_bucket(items, dev))                                      # + allocation of the flat buffer
    t_torch = ev_time(lambda: {k: v.to(dtype=torch.bfloat16) for k, v in state.items()})
    for k, v in state.items():
        assert torch.equal(bucket[k], v.bfloat16())
    by = n * 6
    print(json.dumps({"kernel": "copy_segments f32->bf16", "tensors": a.tensors, "elems": n, "bytes": by, "us": t_kernel * 1e6,
                      "GBps": by / t_kernel / 1e9, "frac": by / t_kernel / 8e12, "pack_bucket_us": t_pack * 1e6,
                      "torch_per_tensor_to_us": t_torch * 1e6, "speedup_vs_torch": t_torch / t_pack}))
    replica = {k: torch.zeros(a.rows, a.cols, dtype=torch.bfloat16, device=dev) for k in state}
    t_load = ev_time(lambda: load_bucket(replica, bucket))

    def torch_load():
        for k, v in bucket.items():
            replica[k].copy_(v)
    t_tload = ev_time(torch_load)
    by2 = n * 4
    print(json.dumps({"kernel": "copy_segments bf16->bf16 (receiver)", "bytes": by2, "us": t_load * 1e6, "GBps": by2 / t_load / 1e9,
                      "frac": by2 / t_load / 8e12, "torch_per_tensor_copy_us": t_tload * 1e6}))
    # many small tensors: the case the per-parameter chain is launch-bound on
    small = {f"p{i}": torch.randn(256, 257, device=dev) for i in range(600)}
    sitems = [(k, v, torch.bfloat16) for k, v in small.items()]
    t_sp = ev_time(lambda: pack_bucket(sitems, dev))
    t_st = ev_time(lambda: {k: v.to(dtype=torch.bfloat16) for k, v in small.items()})
    from rlinf_amd.hybrid_engines.weight_syncer import BucketWeightSyncer
    sync_us = {}
    for persistent in (False, True):
        syncer = BucketWeightSyncer(1 << 30, "bf16", dev, persistent_buckets=persistent)
        syncer.init_sender(small, list(small))
        sync_us[persistent] = ev_time(lambda: syncer.sync(small, lambda b: None, 1)) * 1e6
    print(json.dumps({"kernel": "BucketWeightSyncer.sync 600 small tensors", "fresh_buffers_us": sync_us[False],
                      "persistent_buffers_us": sync_us[True], "torch_per_tensor_to_us": t_st * 1e6}))
    print(json.dumps({"kernel": "copy_segments 600 small tensors", "elems": 600 * 256 * 257, "pack_bucket_us": t_sp * 1e6,
                      "torch_per_tensor_to_us": t_st * 1e6, "speedup_vs_torch": t_st / t_sp}))


if __name__ == "__main__":
    main()

"""Roofline probe of the patch-stream codec (csrc/zplane_codec.hip): compress / decompress of the three stream kinds a weight patch
carries, at sizes that stream from HBM.

    python tools/bench_zplane.py [--nnz 2e8]

Streams (what PatchBuilder emits for `nnz` changed elements of a [rows, 8192] bf16 tensor):
  rows    delta-encoded row indices, uint8: 0 inside a row, 1 at a row start
  cols    delta-encoded column indices: sparse update -> int32 gaps (high bytes zero); dense update -> uint8, all 1
  values  the changed bf16 values' bytes (uint8): incompressible, falls back to raw planes
Algorithmic bytes: compress reads the input twice (measure, pack) and writes the stream; decompress reads the stream and writes
the output.  Reported against the 8 TB/s HBM peak, next to the ratio."""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rlinf_amd import _lib  # noqa: E402
from rlinf_amd.ops import _stream_ptr  # noqa: E402


def ev_time(fn, iters=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nnz", type=float, default=2e8)
    args = ap.parse_args()
    n = int(args.nnz)
    dev = torch.device("cuda")
    g = torch.Generator(device=dev).manual_seed(0)
    lib = _lib.load()
    streams = {
        "rows (uint8 deltas: 0, 1 at row starts)": (torch.rand(n, device=dev, generator=g) < 1 / 800).to(torch.uint8),
        "cols sparse (int32 gaps, mean 800)": torch.randint(1, 1600, (n // 4,), device=dev, generator=g, dtype=torch.int32),
        "cols dense (uint8, all 1)": torch.ones(n, dtype=torch.uint8, device=dev),
        "values (bf16 bytes as uint8)": (torch.randn(n // 2, device=dev, generator=g) * 0.02).to(torch.bfloat16).view(torch.uint8),
    }
    for name, t in streams.items():
        es, ne = t.element_size(), t.numel()
        out = torch.empty(lib.rlx_zplane_bound_bytes(ne, es), dtype=torch.uint8, device=dev)
        ws = torch.empty(lib.rlx_zplane_workspace_bytes(ne, es), dtype=torch.uint8, device=dev)
        length = torch.zeros(1, dtype=torch.int64, device=dev)
        st = _stream_ptr(dev)
        comp = lambda: _lib.check(lib.rlx_zplane_compress(t.data_ptr(), ne, es, out.data_ptr(), out.numel(), length.data_ptr(),  # noqa: E731
                                                          ws.data_ptr(), ws.numel(), st), "compress")
        tc = ev_time(comp)
        clen = int(length.item())
        back = torch.empty_like(t)
        status = torch.zeros(1, dtype=torch.int32, device=dev)
        dec = lambda: _lib.check(lib.rlx_zplane_decompress(out.data_ptr(), clen, back.data_ptr(), ne, es, status.data_ptr(), st),  # noqa: E731
                                 "decompress")
        td = ev_time(dec)
        ok = bool(torch.equal(back, t)) and int(status.item()) == 0
        raw = ne * es
        print(json.dumps({"stream": name, "raw_bytes": raw, "compressed_bytes": clen, "ratio": round(raw / clen, 2), "round_trip_ok": ok,
                          "compress_us": round(tc * 1e6, 1), "compress_input_GBps": round(raw / tc / 1e9, 1),
                          "compress_algorithmic_GBps": round((2 * raw + clen) / tc / 1e9, 1),
                          "compress_frac_of_8TBps": round((2 * raw + clen) / tc / 8e12, 3),
                          "decompress_us": round(td * 1e6, 1), "decompress_output_GBps": round(raw / td / 1e9, 1),
                          "decompress_algorithmic_GBps": round((raw + clen) / td / 1e9, 1),
                          "decompress_frac_of_8TBps": round((raw + clen) / td / 8e12, 3)}), flush=True)
        del out, ws, back


if __name__ == "__main__":
    main()

"""Roofline probe of the weight-patch kernels: patch_scan (+offsets) and patch_emit on one big bf16 tensor.

    python tools/bench_patch.py [--elems 2e9] [--density 1e-3] [--dtype bf16]

patch_scan's algorithmic bytes = 2 * n * sizeof(dtype) (tensor + snapshot, read once); emit touches n/8 mask bytes plus
the changed elements.  Also times the reference's tensor-op chain on the same device (ne / nonzero / gather / scatter /
delta_encode through torch) for the sender side.
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rlinf_amd import _lib  # noqa: E402
from rlinf_amd.hybrid_engines.weight_syncer import PatchBuilder  # noqa: E402
from rlinf_amd.hybrid_engines.weight_syncer.patch_syncer import _dtype_code  # noqa: E402


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
    ap.add_argument("--elems", type=float, default=2e9)
    ap.add_argument("--cols", type=int, default=8192)
    ap.add_argument("--density", type=float, default=1e-3)
    ap.add_argument("--dtype", default="bf16")
    args = ap.parse_args()
    dt = {"bf16": torch.bfloat16, "f32": torch.float32}[args.dtype]
    rows = int(args.elems) // args.cols
    n = rows * args.cols
    dev = torch.device("cuda")
    g = torch.Generator(device=dev).manual_seed(0)
    snap = torch.empty(rows, args.cols, dtype=dt, device=dev)
    for i in range(0, rows, 8192):
        snap[i:i + 8192] = torch.randn(min(8192, rows - i), args.cols, device=dev, generator=g).to(dt)
    new = snap.clone()
    k = int(n * args.density)
    idx = torch.randint(0, n, (k,), device=dev, generator=g)
    new.view(-1)[idx] += 1
    lib = _lib.load()
    code = _dtype_code(dt)
    wsb = lib.rlx_patch_workspace_bytes(n)
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    nnz = torch.zeros(1, dtype=torch.int64, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    scan = lambda: _lib.check(lib.rlx_patch_scan(new.data_ptr(), code, snap.data_ptr(), code, n, ws.data_ptr(), wsb,  # noqa: E731
                                                 nnz.data_ptr(), st), "scan")
    t_scan = ev_time(scan)
    count = int(nnz.item())
    r = torch.empty(count, dtype=torch.int64, device=dev)
    c = torch.empty(count, dtype=torch.int64, device=dev)
    v = torch.empty(count, dtype=dt, device=dev)
    mx = torch.zeros(2, dtype=torch.int64, device=dev)
    snap2 = snap.clone()
    emit = lambda: _lib.check(lib.rlx_patch_emit(new.data_ptr(), code, snap2.data_ptr(), code, n, args.cols, 1, ws.data_ptr(),  # noqa: E731
                                                 count, r.data_ptr(), c.data_ptr(), v.data_ptr(), mx.data_ptr(), st), "emit")
    t_emit = ev_time(emit)
    es = snap.element_size()
    out = {"kernel": "patch_scan+offsets", "dtype": args.dtype, "elems": n, "nnz": count, "bytes": 2 * n * es,
           "us": t_scan * 1e6, "GBps": 2 * n * es / t_scan / 1e9, "frac": 2 * n * es / t_scan / 8e12}
    print(json.dumps(out))
    eb = n // 8 + count * (es * 3 + 16)
    print(json.dumps({"kernel": "patch_emit", "nnz": count, "bytes": eb, "us": t_emit * 1e6, "GBps": eb / t_emit / 1e9}))

    def torch_chain():  # what GPUSnapshotPatchBuilder.create_patch runs for one tensor (patch_syncer.py:712-730)
        changed = new.ne(snap)
        rr, cc = changed.nonzero(as_tuple=True)
        vals = new[rr, cc]
        snap2[rr, cc] = vals
        return PatchBuilder.delta_encode(rr, cc)

    t_ref = ev_time(torch_chain, iters=3)
    print(json.dumps({"kernel": "torch tensor-op chain (reference's sender, same GPU)", "us": t_ref * 1e6,
                      "speedup_scan_plus_emit": t_ref / (t_scan + t_emit)}))



if __name__ == "__main__":
    main()

"""Dev tool: time gae_scan variants (HIP events on the launch stream) at the contract shape and the
scaled shape.  Run on the GPU box: python tools/sweep_gae.py [--out gpurun_out/sweep_gae.json]
Under `rocprofv3 --kernel-trace --stats` the per-variant kernel names give exact durations."""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rlinf_amd import ops
from rlinf_amd._lib import RlxError


def V(vec, nseg, rows=0, pf=0, nt=0):
    return vec | (nseg << 8) | (rows << 16) | (pf << 24) | (nt << 25)


def time_variant(bufs, variant, normalize, iters=20, warm=3):
    for i in range(warm):
        r, v, d = bufs[i % len(bufs)]
        ops.gae_scan(r, v, d, None, 0.99, 0.95, normalize_advantages=normalize, variant=variant)
    torch.cuda.synchronize()
    evs = []
    for i in range(iters):
        r, v, d = bufs[i % len(bufs)]
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ops.gae_scan(r, v, d, None, 0.99, 0.95, normalize_advantages=normalize, variant=variant)
        e1.record()
        evs.append((e0, e1))
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) * 1e3 for a, b in evs)  # us
    return ts[len(ts) // 2], ts[0]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="gpurun_out/sweep_gae.json")
    ap.add_argument("--quick", action="store_true")
    a = ap.parse_args()
    res = []
    shapes = [(128, 65536, 5), (128, 1024, 1)] if a.quick else [(128, 65536, 5), (128, 1024, 1), (128, 16384, 8), (50, 1024, 1)]
    for (T, B, nbuf) in shapes:
        g = torch.Generator().manual_seed(0)
        bufs = []
        for _ in range(nbuf):
            r = torch.rand(T, B, 1, generator=g).cuda()
            v = torch.randn(T + 1, B, 1, generator=g).cuda()
            d = (torch.rand(T + 1, B, 1, generator=g) < 0.02).cuda()
            bufs.append((r, v, d))
        n = T * B
        combos = []
        for vec in (1, 2, 4):
            for nseg in (1, 2, 4, 8, 16):
                base = 4 if vec == 4 else 8
                for rows, pf, nt in ((0, 0, 0), (0, 0, 1), (2 * base, 0, 0), (2 * base, 0, 1)):
                    combos.append((vec, nseg, rows, pf, nt))
        for vec, nseg, rows, pf, nt in combos:
            try:
                med, best = time_variant(bufs, V(vec, nseg, rows, pf, nt), False)
            except RlxError:
                continue
            row = dict(T=T, B=B, vec=vec, nseg=nseg, rows=rows, pf=pf, nt=nt, med_us=round(med, 2), min_us=round(best, 2),
                       GBps=round(n * 17 / med / 1e3, 1))
            res.append(row)
            print(row, flush=True)
        for normalize in (False, True):
            med, best = time_variant(bufs, 0, normalize)
            print(dict(T=T, B=B, variant="auto", normalize=normalize, med_us=round(med, 2),
                       GBps=round(n * (25 if normalize else 17) / med / 1e3, 1)), flush=True)
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    json.dump(res, open(a.out, "w"), indent=1)
    best = {}
    for r in res:
        k = (r["T"], r["B"])
        if k not in best or r["med_us"] < best[k]["med_us"]:
            best[k] = r
    print("BEST", best)


if __name__ == "__main__":
    main()

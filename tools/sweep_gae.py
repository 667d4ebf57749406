"""Dev tool: time every gae_scan variant (HIP events) at the contract shape and the scaled shape.
Run on the GPU box: python tools/sweep_gae.py [--out gpurun_out/sweep_gae.json]"""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rlinf_amd import ops
from rlinf_amd._lib import RlxError


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
    a = ap.parse_args()
    res = []
    for (T, B, nbuf) in [(128, 1024, 1), (128, 65536, 5), (128, 16384, 8), (50, 1024, 1)]:
        g = torch.Generator().manual_seed(0)
        bufs = []
        for _ in range(nbuf):
            r = torch.rand(T, B, 1, generator=g).cuda()
            v = torch.randn(T + 1, B, 1, generator=g).cuda()
            d = (torch.rand(T + 1, B, 1, generator=g) < 0.02).cuda()
            bufs.append((r, v, d))
        n = T * B
        for vec in (1, 2, 4):
            for nseg in (1, 2, 4, 8):
                variant = vec | (nseg << 8)
                for normalize in (False, True):
                    try:
                        med, best = time_variant(bufs, variant, normalize)
                    except RlxError as e:
                        continue
                    bytes_ = n * (25 if normalize else 17)
                    row = dict(T=T, B=B, vec=vec, nseg=nseg, normalize=normalize, med_us=round(med, 2),
                               min_us=round(best, 2), GBps=round(bytes_ / med / 1e3, 1))
                    res.append(row)
                    print(row, flush=True)
        med, best = time_variant(bufs, 0, True)
        print(dict(T=T, B=B, variant="auto", normalize=True, med_us=round(med, 2), GBps=round(n * 25 / med / 1e3, 1)), flush=True)
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    json.dump(res, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()

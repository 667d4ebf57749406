"""Dev tool: gae_scan variants in the HBM regime (inputs AND outputs rotate over buffer sets larger than the 256 MiB
Infinity Cache), HIP events on the launch stream.  python tools/sweep_gae_hbm.py [B] [T]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rlinf_amd import ops
from rlinf_amd._lib import RlxError


def V(vec, nseg, rows=0, nt=0, regseg=0, seg16=0, handoff_seg=0):
    """handoff_seg: 16 / 32 / 64 selects gae_scan_handoff (register-resident segments, early hand-off) with that segment length"""
    v = vec | (nseg << 8) | (rows << 16) | (nt << 25) | (regseg << 26) | (seg16 << 27)
    if handoff_seg:
        v |= (1 << 28) | ({16: 0, 32: 1, 64: 2}[handoff_seg] << 29)
    return v


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
    T = int(sys.argv[2]) if len(sys.argv) > 2 else 128
    nbuf = 5
    g = torch.Generator().manual_seed(0)
    bufs = []
    for _ in range(nbuf):
        r = torch.rand(T, B, 1, generator=g).cuda()
        v = torch.randn(T + 1, B, 1, generator=g).cuda()
        d = (torch.rand(T + 1, B, 1, generator=g) < 0.02).cuda()
        bufs.append((r, v, d, torch.empty_like(r), torch.empty_like(r)))
    combos = [(V(1, 1, rows, nt), f"stream rows={rows} nt={nt}") for rows in (8, 64) for nt in (0, 1)]
    combos += [(V(vec, 1, 0, nt, 1, s16), f"regseg vec={vec} seg={16 if s16 else 32} nt={nt}")
               for vec, s16 in ((1, 0), (1, 1)) for nt in (0, 1)]
    combos += [(V(1, 1, 0, nt, handoff_seg=seg), f"handoff seg={seg} nt={nt}") for seg in (64, 32, 16) for nt in (0, 1)]
    # correctness: every variant must reproduce the streaming scan bit for bit
    r, v, d, a0, q0 = bufs[0]
    ops.gae_scan(r, v, d, None, 0.99, 0.95, normalize_advantages=False, variant=V(1, 1), out=(a0, q0))
    ref_a, ref_q = a0.clone(), q0.clone()
    for var, name in combos:
        a1, q1 = torch.full_like(a0, float("nan")), torch.full_like(q0, float("nan"))
        ops.gae_scan(r, v, d, None, 0.99, 0.95, normalize_advantages=False, variant=var, out=(a1, q1))
        ok = torch.equal(a1, ref_a) and torch.equal(q1, ref_q)
        print("bit-exact" if ok else "MISMATCH", name, flush=True)
    for rep in range(2):
        for var, name in combos:
            try:
                for i in range(3):
                    r, v, d, a, q = bufs[i % nbuf]
                    ops.gae_scan(r, v, d, None, 0.99, 0.95, normalize_advantages=False, variant=var, out=(a, q))
            except RlxError as e:
                print("skip", name, e)
                continue
            torch.cuda.synchronize()
            evs = []
            for i in range(30):
                r, v, d, a, q = bufs[i % nbuf]
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                ops.gae_scan(r, v, d, None, 0.99, 0.95, normalize_advantages=False, variant=var, out=(a, q))
                e1.record()
                evs.append((e0, e1))
            torch.cuda.synchronize()
            ts = sorted(x.elapsed_time(y) * 1e3 for x, y in evs)
            avg = sum(ts) / len(ts)
            print(f"rep{rep} {name:34s} avg {avg:7.2f} us  med {ts[len(ts)//2]:7.2f}  min {ts[0]:7.2f}"
                  f"  -> {17 * T * B / avg / 1e3:7.1f} GB/s ({17 * T * B / avg / 1e3 / 80:.1f}% of 8 TB/s)", flush=True)


if __name__ == "__main__":
    main()

"""Summarise a rocprofv3 rocpd .db (kernel trace): per-kernel count / median / min / mean duration (us).
usage: python tools/rocpd_stats.py results.db [name-substring] [--by-grid]
       python tools/rocpd_stats.py results.db --gaps     (device idle time between consecutive kernels: totals, histogram, largest)
       python tools/rocpd_stats.py results.db --streams  (two-stream runs: per-queue busy time, the time both queues have a kernel
                                                          in flight, and every kernel's duration alone / while the other queue runs)"""
import re, sqlite3, sys
from collections import defaultdict


def gaps(db):
    cols = [r[1] for r in db.execute("pragma table_info(kernels)").fetchall()] or [d[0] for d in db.execute("select * from kernels limit 1").description]
    s_col = "start" if "start" in cols else next(c for c in cols if "start" in c.lower())
    e_col = "end" if "end" in cols else next(c for c in cols if "end" in c.lower())
    rows = db.execute(f'select name, "{s_col}", "{e_col}" from kernels order by "{s_col}"').fetchall()
    if not rows:
        print("no kernels")
        return
    short = lambda n: re.sub(r"\(.*\)$", "", re.sub(r"\(anonymous namespace\)::", "", n))[-48:]
    span = (max(r[2] for r in rows) - rows[0][1]) / 1e3
    busy = sum(r[2] - r[1] for r in rows) / 1e3
    g = []
    reach = rows[0][2]
    for (n0, _s0, _e0), (n1, s1, e1) in zip(rows, rows[1:]):
        g.append(((s1 - reach) / 1e3, short(n0), short(n1)))   # negative: overlap with something still running
        reach = max(reach, e1)
    idle = sum(x[0] for x in g if x[0] > 0)
    print(f"kernels {len(rows)}  span {span / 1e3:.3f} ms  sum of durations {busy / 1e3:.3f} ms  idle between kernels {idle / 1e3:.3f} ms ({100 * idle / span:.1f} % of the span)")
    edges = [0, 0.5, 1, 2, 3, 5, 10, 20, 50, 100, 1000, 1e9]
    for lo, hi in zip(edges, edges[1:]):
        sel = [x[0] for x in g if lo < x[0] <= hi]
        if sel:
            print(f"  gap {lo:>6g} .. {hi:<6g} us: {len(sel):6d} gaps, {sum(sel) / 1e3:8.3f} ms")
    by_pair = defaultdict(list)
    for d, a, b in g:
        if d > 0:
            by_pair[(a, b)].append(d)
    print("  idle by (kernel -> next kernel):")
    for (a, b), v in sorted(by_pair.items(), key=lambda kv: -sum(kv[1]))[:14]:
        v = sorted(v)
        print(f"    {a:48s} -> {b:48s} n {len(v):5d}  med {v[len(v) // 2]:7.2f} us  total {sum(v) / 1e3:7.3f} ms")


def streams(db):
    """Per-queue timeline of a multi-stream run: who is busy when, and what co-running does to each kernel's duration."""
    rows = db.execute('select name, queue_id, stream_id, "start", "end" from kernels order by "start"').fetchall()
    if not rows:
        print("no kernels")
        return
    short = lambda n: re.sub(r"<.*", "", re.sub(r"\(.*\)$", "", re.sub(r"\(anonymous namespace\)::", "", n)))[-44:]
    by_q = defaultdict(list)
    for n, q, st, s, e in rows:
        by_q[(q, st)].append((s, e, short(n)))
    t0, t1 = rows[0][3], max(r[4] for r in rows)
    print(f"span {(t1 - t0) / 1e6:.3f} ms, {len(rows)} kernels on {len(by_q)} (queue, stream) pairs")
    for key, v in sorted(by_q.items(), key=lambda kv: -len(kv[1])):
        busy = sum(e - s for s, e, _ in v)
        names = defaultdict(int)
        for _, _, n in v:
            names[n] += 1
        top = ", ".join(f"{n} x{c}" for n, c in sorted(names.items(), key=lambda kv: -kv[1])[:4])
        print(f"  queue {key[0]} stream {key[1]}: {len(v):6d} kernels, busy {busy / 1e6:8.3f} ms ({100 * busy / (t1 - t0):5.1f} % of the span)  [{top}]")
    if len(by_q) < 2:
        return
    # interval sweep: time with kernels of >= 2 different queues in flight
    ev = []
    for key, v in by_q.items():
        for s, e, _ in v:
            ev.append((s, 1, key)), ev.append((e, -1, key))
    ev.sort(key=lambda x: (x[0], x[1]))
    live, both, last = defaultdict(int), 0, ev[0][0]
    for t, d, key in ev:
        if sum(1 for c in live.values() if c > 0) >= 2:
            both += t - last
        last = t
        live[key] += d
    print(f"  >= 2 queues with a kernel in flight: {both / 1e6:.3f} ms ({100 * both / (t1 - t0):.1f} % of the span)")
    # each kernel: alone vs overlapped by a kernel of another queue (any intersection)
    allk = sorted(((s, e, n, key) for key, v in by_q.items() for s, e, n in v), key=lambda x: x[0])
    import bisect
    starts = [k[0] for k in allk]
    longest = max(e - s for s, e, _, _ in allk)
    stat = defaultdict(lambda: ([], []))
    for s, e, n, key in allk:
        lo = bisect.bisect_left(starts, s - longest)
        hi = bisect.bisect_right(starts, e)
        co = any(k[3] != key and k[1] > s and k[0] < e for k in allk[lo:hi])
        stat[n][1 if co else 0].append((e - s) / 1e3)
    print(f"  {'kernel':46s} {'alone: n':>9s} {'med_us':>8s} | {'co-running: n':>13s} {'med_us':>8s} {'stretch':>8s}")
    for n, (alone, co) in sorted(stat.items(), key=lambda kv: -(sum(kv[1][0]) + sum(kv[1][1])))[:12]:
        ma = sorted(alone)[len(alone) // 2] if alone else float("nan")
        mc = sorted(co)[len(co) // 2] if co else float("nan")
        print(f"  {n:46s} {len(alone):9d} {ma:8.2f} | {len(co):13d} {mc:8.2f} {(mc / ma if alone and co else float('nan')):8.2f}")


def main():
    path = sys.argv[1]
    if "--gaps" in sys.argv:
        gaps(sqlite3.connect(path))
        return
    if "--streams" in sys.argv:
        streams(sqlite3.connect(path))
        return
    sub = sys.argv[2] if len(sys.argv) > 2 and not sys.argv[2].startswith("--") else ""
    by_grid = "--by-grid" in sys.argv
    db = sqlite3.connect(path)
    rows = db.execute("select name, grid_x, workgroup_x, duration, vgpr_count, lds_size from kernels").fetchall()
    agg = defaultdict(list)
    meta = {}
    for name, gx, wx, dur, vgpr, lds in rows:
        if sub and sub not in name:
            continue
        short = re.sub(r"\(anonymous namespace\)::", "", name)
        short = re.sub(r"\(.*\)$", "", short)
        key = (short, gx // max(wx, 1)) if by_grid else (short,)
        agg[key].append(dur / 1000.0)
        meta[key] = (wx, vgpr, lds)
    tot = sum(sum(v) for v in agg.values())
    print(f"{'kernel':70s} {'blocks':>7s} {'calls':>6s} {'med_us':>9s} {'min_us':>9s} {'mean_us':>9s} {'total_ms':>9s} {'%':>6s} wg vgpr lds")
    for key, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        v = sorted(v)
        wx, vgpr, lds = meta[key]
        blocks = key[1] if by_grid else "-"
        print(f"{key[0][-70:]:70s} {str(blocks):>7s} {len(v):6d} {v[len(v)//2]:9.2f} {v[0]:9.2f} {sum(v)/len(v):9.2f} {sum(v)/1e3:9.3f} {100*sum(v)/tot:6.2f} {wx} {vgpr} {lds}")


if __name__ == "__main__":
    main()

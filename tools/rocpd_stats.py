"""Summarise a rocprofv3 rocpd .db (kernel trace): per-kernel count / median / min / mean duration (us).
usage: python tools/rocpd_stats.py results.db [name-substring] [--by-grid]
       python tools/rocpd_stats.py results.db --gaps     (device idle time between consecutive kernels: totals, histogram, largest)"""
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


def main():
    path = sys.argv[1]
    if "--gaps" in sys.argv:
        gaps(sqlite3.connect(path))
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

"""Summarise a rocprofv3 rocpd .db (kernel trace): per-kernel count / median / min / mean duration (us).
usage: python tools/rocpd_stats.py results.db [name-substring] [--by-grid]"""
import re, sqlite3, sys
from collections import defaultdict


def main():
    path = sys.argv[1]
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

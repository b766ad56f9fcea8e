"""Per-kernel statistics (calls, total / average / min / max ms, share) from a rocprofv3 rocpd database, as CSV on stdout:
the summary kept under profiles/ when rocprofv3 writes .db files instead of *_kernel_stats.csv.

    python tools/rocpd_stats.py <results.db> [min_share_percent]
"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"^void ", "", name)
    return name if len(name) <= 150 else name[:147] + "..."


def main():
    c = sqlite3.connect(sys.argv[1])
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    namecol = "name" if "name" in cols else [x for x in cols if "name" in x][0]
    rows = c.execute("select %s, count(*), sum(end-start), min(end-start), max(end-start) from kernels group by %s" % (namecol, namecol)).fetchall()
    total = sum(r[2] for r in rows) or 1
    floor = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
    print("Name,Calls,TotalDurationMs,AverageMs,MinMs,MaxMs,Percentage")
    for name, calls, tot, mn, mx in sorted(rows, key=lambda r: -r[2]):
        if 100.0 * tot / total < floor:
            continue
        print('"%s",%d,%.4f,%.4f,%.4f,%.4f,%.2f' % (short(name), calls, tot / 1e6, tot / 1e6 / calls, mn / 1e6, mx / 1e6, 100.0 * tot / total))


if __name__ == "__main__":
    main()

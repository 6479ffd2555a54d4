"""Per-kernel summary (calls, total / average / min / max duration) of a rocprofv3 rocpd database (ROCm 7 writes `*_results.db` by default),
in the column layout of `rocprofv3 --stats` CSV files.  Usage: python tools/rocpd_kernel_stats.py results.db > kernel_stats.csv"""
import sqlite3
import sys


def main(path):
    db = sqlite3.connect(path)
    rows = db.execute("select name, count(*), sum(end - start), avg(end - start), min(end - start), max(end - start) from kernels group by name order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows) or 1
    print('"Name","Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs"')
    for n, c, t, a, mn, mx in rows:
        print('"%s",%d,%d,%.1f,%.4f,%d,%d' % (n.replace('"', "'"), c, t, a, 100.0 * t / tot, mn, mx))


if __name__ == "__main__":
    main(sys.argv[1])

#!/usr/bin/env python3
"""Summarises a rocprofv3 rocpd SQLite database (ROCm 7.2 default output) into the per-kernel table that
`rocprofv3 --kernel-trace --stats` prints as CSV in older releases, and (for --pmc runs) per-kernel counter sums.

usage: tools/rocpd_summary.py <results.db> [--pmc]   -> markdown on stdout"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    if "--pmc" in sys.argv:
        cols = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
        kcol = "kernel_name" if "kernel_name" in cols else "name"
        rows = cur.execute("select %s, counter_name, count(*), sum(value), avg(value) from counters_collection group by 1,2 order by 4 desc" % kcol).fetchall()
        print("| kernel | counter | dispatches | sum | avg/dispatch |\n|---|---|---|---|---|")
        for k, c, n, s, a in rows:
            print("| %s | %s | %d | %.6g | %.6g |" % (k[:70], c, n, s, a))
        return
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name = "name" if "name" in cols else "kernel_name"
    rows = cur.execute("select %s, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by 1 order by 3 desc" % name).fetchall()
    tot = sum(r[2] for r in rows) or 1
    print("| kernel | calls | total (us) | avg (us) | min (us) | max (us) | % |\n|---|---|---|---|---|---|---|")
    for k, n, s, a, mn, mx in rows:
        print("| %s | %d | %.1f | %.2f | %.2f | %.2f | %.1f |" % (k[:90], n, s / 1e3, a / 1e3, mn / 1e3, mx / 1e3, 100.0 * s / tot))


if __name__ == "__main__":
    main()

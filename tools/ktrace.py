#!/usr/bin/env python3
"""rocprofv3 --kernel-trace --stats of any command, summarised per kernel (rocpd database): python tools/ktrace.py [--top N] -- <command ...>"""
import os
import shutil
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import profile_round as P  # noqa: E402

a = sys.argv[1:]
top = 20
listk = None
while a and a[0] in ("--top", "--list"):
    if a[0] == "--top":
        top = int(a[1])
    else:
        listk = a[1]          # also print the individual dispatch durations (us, in start order) of kernels whose name contains this
    a = a[2:]
if a and a[0] == "--":
    a = a[1:]
out = "/tmp/ktrace_%d" % os.getpid()
shutil.rmtree(out, ignore_errors=True)
r = subprocess.run(["rocprofv3", "--kernel-trace", "--stats", "-d", out, "--"] + a, env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
print(r.stdout[-1500:])
rows = P.kernel_table(P.find_db(out))
tot = sum(x[2] for x in rows) or 1
print("| kernel | calls | total (us) | avg (us) | min (us) | max (us) | % |\n|---|---|---|---|---|---|---|")
for k, n, s, av, mn, mx in rows[:top]:
    print("| %s | %d | %.1f | %.2f | %.2f | %.2f | %.1f |" % (k.replace("(anonymous namespace)::", "")[:80], n, s / 1e3, av / 1e3, mn / 1e3, mx / 1e3, 100.0 * s / tot))
if listk:
    import sqlite3
    cur = sqlite3.connect(P.find_db(out)).cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name = "name" if "name" in cols else "kernel_name"
    rows = cur.execute("select %s, start, end-start from kernels order by start" % name).fetchall()
    sel = [(n, d) for n, s_, d in rows if listk in n]
    print("dispatches of *%s* in start order (us):" % listk, " ".join("%.1f" % (d / 1e3) for n, d in sel[-15:]))
shutil.rmtree(out, ignore_errors=True)

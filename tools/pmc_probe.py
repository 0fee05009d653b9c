#!/usr/bin/env python3
"""hardware counters of any command, one rocprofv3 --pmc pass per counter group (never together with a tracing domain), averaged per dispatch and kernel.
usage: python tools/pmc_probe.py [--kernel SUBSTR] --pmc A,B,C [--pmc D,E ...] -- <command ...>      (run on the GPU box through gpurun)"""
import glob
import os
import sqlite3
import subprocess
import sys


def main():
    a = sys.argv[1:]
    groups, ksub, limit = [], None, 60
    while a and a[0] != "--":
        x = a.pop(0)
        if x == "--pmc": groups.append(a.pop(0).split(","))
        elif x == "--kernel": ksub = a.pop(0)
        elif x == "--limit": limit = int(a.pop(0))
    cmd = a[1:]
    res = {}
    for gi, g in enumerate(groups):
        out = "/tmp/pmcprobe_%d_%d" % (os.getpid(), gi)
        try:      # (a counter group the hardware cannot schedule in one pass may hang the run: every pass has its own time limit)
            r = subprocess.run(["timeout", "-s", "KILL", str(limit), "rocprofv3", "--pmc"] + g + ["-d", out, "--"] + cmd, env=dict(os.environ, TMPDIR="/tmp"), cwd="/tmp", stdout=subprocess.PIPE,
                               stderr=subprocess.STDOUT, text=True)
        except Exception as e:
            print("pass failed:", g, e, flush=True); continue
        if r.returncode:
            print(r.stdout[-600:]); print("pass failed (rc %d):" % r.returncode, g, flush=True); continue
        print("pass ok:", g, flush=True)
        dbs = sorted(glob.glob(os.path.join(out, "**", "*.db"), recursive=True), key=os.path.getmtime)
        cur = sqlite3.connect(dbs[-1]).cursor()
        cols = [c[1] for c in cur.execute("pragma table_info(counters_collection)")]
        kcol = "kernel_name" if "kernel_name" in cols else "name"
        for k, c, n, av in cur.execute("select %s, counter_name, count(*), avg(value) from counters_collection group by 1,2" % kcol):
            if ksub and ksub not in k: continue
            res.setdefault(k, {})[c] = (n, av)
    for k, d in res.items():
        print(k[:110])
        for c, (n, av) in sorted(d.items()):
            print("    %-44s %16.1f   (%d dispatches)" % (c, av, n))


if __name__ == "__main__":
    main()
